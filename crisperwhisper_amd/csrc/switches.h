// A/B switches of the kernel launchers and the engine.  TWO LIFETIMES, by group (see the struct): the "engine" group is read when a
// context is created (cw_create -> read_switches()): setting the environment before Engine() selects it per context, which is how
// the differential tests build two engines in one process.  Every other group is read by the kernel launchers ONCE per process
// (cw_switches(), first use) and a later change of the environment is ignored -- the tests flip those in-process through
// cw_test_set_option instead.  No launch path calls getenv.  Every switch defaults to the fast path; DESIGN.md "A/B switches" lists what each
// one selects and the profile that measured it.
#pragma once

namespace cw_sw {

struct Switches {
    // engine (cw_create)
    bool no_graph, no_ln_fold, no_fuse6, rows_ln, no_rows_hilo, no_stack_center, no_mid16, dtw_block, fuse_mlp, no_wpack, mlp_pair, mlp_pair_fence, declayer, no_qkv_self, mlp_chain, no_fuse_rows, no_fuse_rows8, no_fuse_beam, no_own_cols, no_short_hist;
    int skinny, prefetch, prefetch_wide, prefetch_what, stack_nt3, stack_nt5;
    // attention launchers
    bool attn_v1, anc_attn_v1, cross_per_row, cross_valu, cross_no_tr, cross8_valu, cross_mfma1;
    int cross_lds_pad, cross8_nsb, dl_depth, dl_kvwait, mlp_chain_delay, qkv_self_dbg;
    // GEMM / GEMV launchers
    bool no_glds, no_gemm256, no_gemm_pp, no_gemm_8ph, gemm_w128, no_gemv_loop, comb_nt2, comb_no_rowgroups, mt_no_prea, comb_g4;
    int gemv_loop_cap, fc2_ksplit, mt_variant, own_nt;
    // sampling, mel
    bool beam_topk_1block, mel_valu;
    int mel_dbg;
    // timing loops of the cw_test_* entry points (0 = off)
    int test_gemm_reps, test_attn_reps;
};

const Switches& cw_switches();   // parsed on first use, then fixed for the process (the launchers)
Switches read_switches();            // parsed now (cw_create: a context's own switches are fixed when it is created)

}  // namespace cw_sw
