// Environment switches, parsed once (see switches.h).
#include "switches.h"
#include <cstdlib>

namespace cw_sw {

static bool flag(const char* name) { return getenv(name) != nullptr; }
static int num(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

Switches read_switches() {
    Switches s{};
    s.no_graph = flag("CW_NO_GRAPH");
    s.no_ln_fold = flag("CW_NO_LN_FOLD");
    s.no_fuse6 = flag("CW_NO_FUSE6");
    s.rows_ln = flag("CW_ROWS_LN");
    s.no_rows_hilo = flag("CW_NO_ROWS_HILO");
    s.no_stack_center = flag("CW_NO_STACK_CENTER");
    s.no_mid16 = flag("CW_NO_MID16");
    s.dtw_block = flag("CW_DTW_BLOCK");
    s.fuse_mlp = flag("CW_FUSE_MLP");
    s.no_wpack = flag("CW_NO_WPACK");
    s.mlp_pair = flag("CW_MLP_PAIR");
    s.mlp_pair_fence = flag("CW_MLP_PAIR_FENCE");
    s.declayer = flag("CW_DECLAYER");
    s.no_qkv_self = flag("CW_NO_QKV_SELF");
    s.mlp_chain = flag("CW_MLP_CHAIN");
    s.no_fuse_rows = flag("CW_NO_FUSE_ROWS");
    s.no_fuse_rows8 = flag("CW_NO_FUSE_ROWS8");
    s.no_fuse_beam = flag("CW_NO_FUSE_BEAM");
    s.no_own_cols = flag("CW_NO_OWN_COLS");
    s.no_short_hist = flag("CW_NO_SHORT_HIST");
    s.skinny = num("CW_SKINNY", 0);
    s.prefetch = num("CW_PREFETCH", 0);
    s.prefetch_wide = num("CW_PREFETCH_WIDE", 0);
    s.prefetch_what = num("CW_PREFETCH_WHAT", 3);
    s.stack_nt3 = num("CW_STACK_NT3", 0);
    s.stack_nt5 = num("CW_STACK_NT5", 0);
    s.attn_v1 = flag("CW_ATTN_V1");
    s.anc_attn_v1 = flag("CW_ANC_ATTN_V1");
    s.cross_per_row = flag("CW_CROSS_PER_ROW");
    s.cross_valu = flag("CW_CROSS_VALU");
    s.cross_no_tr = flag("CW_CROSS_NO_TR");
    s.cross8_valu = flag("CW_CROSS8_VALU");
    s.cross_mfma1 = flag("CW_CROSS_MFMA1");
    s.cross_lds_pad = num("CW_CROSS_LDS_PAD", 0);
    s.cross8_nsb = num("CW_CROSS8_NSB", 0);
    s.dl_depth = num("CW_DL_DEPTH", 16);
    s.dl_kvwait = num("CW_DL_KVWAIT", 0);
    s.mlp_chain_delay = num("CW_MLP_CHAIN_DELAY", 0);
    s.qkv_self_dbg = num("CW_QKV_SELF_DBG", 0);
    s.no_glds = flag("CW_NO_GLDS");
    s.no_gemm256 = flag("CW_NO_GEMM256");
    s.no_gemm_pp = flag("CW_NO_GEMM_PP");
    s.no_gemm_8ph = flag("CW_NO_GEMM_8PH");
    s.gemm_w128 = flag("CW_GEMM_W128");
    s.no_gemv_loop = flag("CW_NO_GEMV_LOOP");
    s.comb_nt2 = flag("CW_COMB_NT2");
    s.comb_no_rowgroups = flag("CW_COMB_NO_ROWGROUPS");
    s.comb_g4 = flag("CW_COMB_G4");
    s.mt_no_prea = flag("CW_MT_NO_PREA");
    s.gemv_loop_cap = num("CW_GEMV_LOOP_CAP", 512);
    s.fc2_ksplit = num("CW_FC2_KSPLIT", 0);
    s.mt_variant = num("CW_MT_VARIANT", -1);
    s.own_nt = num("CW_OWN_NT", 1);
    if (s.mt_variant < -1 || s.mt_variant > 2) s.mt_variant = -1;
    s.beam_topk_1block = flag("CW_BEAM_TOPK_1BLOCK");
    s.mel_valu = flag("CW_MEL_VALU");
    s.mel_dbg = num("CW_MEL_DBG", 0);
    s.test_gemm_reps = num("CW_TEST_GEMM_REPS", 0);
    s.test_attn_reps = num("CW_TEST_ATTN_REPS", 0);
    // numeric switches are divisors / sizes in the launchers: keep them in the range the launch code is written for
    if (s.gemv_loop_cap < 1) s.gemv_loop_cap = 1;
    if (s.fc2_ksplit < 0) s.fc2_ksplit = 0;
    if (s.cross_lds_pad < 0) s.cross_lds_pad = 0;
    if (s.cross_lds_pad > 150 * 1024) s.cross_lds_pad = 150 * 1024;
    if (s.stack_nt3 < 0 || s.stack_nt3 > 3) s.stack_nt3 = 0;
    if (s.stack_nt5 < 0 || s.stack_nt5 > 3) s.stack_nt5 = 0;
    if (s.prefetch < 0) s.prefetch = 0;
    if (s.mlp_chain_delay < 0 || s.mlp_chain_delay > 4096) s.mlp_chain_delay = 0;
    if (s.test_gemm_reps < 0) s.test_gemm_reps = 0;
    if (s.test_attn_reps < 0) s.test_attn_reps = 0;
    return s;
}

const Switches& cw_switches() {
    static const Switches s = read_switches();   // thread-safe one-time initialisation
    return s;
}

}  // namespace cw_sw
