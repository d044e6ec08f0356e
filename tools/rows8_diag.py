"""Where do attn_cross_mfma8_rows_kernel and one-row blocks of attn_cross_mfma8_kernel differ?  Raw partial planes, (m, l) pairs and
alignment rows per hypothesis row, for several rows-per-item counts (GPU; run through gpurun)."""
import ctypes as C
import sys

import numpy as np

from crisperwhisper_amd import _native
from crisperwhisper_amd.engine import Engine, _ptr
from tests import helpers as Hh


def raw(eng, lib, q, k, v, kv_div, align_head):
    q, k, v = (np.ascontiguousarray(t, np.float32) for t in (q, k, v))
    B, H, _ = q.shape
    S = k.shape[2]
    NS = 6
    po = np.zeros((NS, B, H * 64), np.float32); ml = np.zeros((B, H, NS, 2), np.float32)
    al = np.zeros((B, S), np.float32); aml = np.zeros((B, NS, 2), np.float32)
    rc = lib.cw_test_cross_attention(eng.ctx, B, H, S, int(kv_div), _ptr(q), _ptr(k), _ptr(v), int(align_head), _ptr(po), _ptr(ml), _ptr(al), _ptr(aml))
    assert rc == 0, rc
    return po, ml, al, aml


def main():
    g, v, W, spec = Hh.tiny_setup()
    eng = Engine(spec, dtype="bf16", max_batch=4)
    lib = _native.load()
    lib.cw_test_set_option(b"cross_test_fp8", 1)
    H, S, items = 2, 1500, 2
    for nq in (2, 4, 5, 8):
        B = items * nq
        rng = np.random.default_rng(nq)
        q = (rng.standard_normal((B, H, 64)) * 0.35).astype(np.float32)
        k = rng.standard_normal((items, H, S, 64)).astype(np.float32)
        vv = rng.standard_normal((items, H, S, 64)).astype(np.float32)
        k = (k.view(np.uint32) & 0xFFFF0000).view(np.float32); vv = (vv.view(np.uint32) & 0xFFFF0000).view(np.float32)
        a = raw(eng, lib, q, k, vv, nq, H - 1)
        a2 = raw(eng, lib, q, k, vv, nq, H - 1)
        print(f"nq={nq}: rows kernel run twice: po {np.array_equal(a[0], a2[0])} ml {np.array_equal(a[1], a2[1])} al {np.array_equal(a[2], a2[2])}")
        b = raw(eng, lib, q, np.repeat(k, nq, axis=0), np.repeat(vv, nq, axis=0), 1, H - 1)
        lib.cw_test_set_option(b"cross_per_row", 1)
        c = raw(eng, lib, q, k, vv, nq, H - 1)               # one-row blocks over the SHARED cache (b / kv_div indexing)
        lib.cw_test_set_option(b"cross_per_row", 0)
        print(f"nq={nq}: per_row(shared cache) == kv_div=1(replicated cache): po {np.array_equal(b[0], c[0])} ml {np.array_equal(b[1], c[1])} al {np.array_equal(b[2], c[2])}")
        for row in range(B):
            dpo = np.abs(a[0][:, row] - b[0][:, row]); dm = np.abs(a[1][row, :, :, 0] - b[1][row, :, :, 0]); dl = np.abs(a[1][row, :, :, 1] - b[1][row, :, :, 1])
            dal = np.abs(a[2][row] - b[2][row])
            nz = int((a[0][:, row] != b[0][:, row]).sum())
            print(f"  row {row} (item {row // nq}, query {row % nq}: tile {(row % nq) // 4}, g {(row % nq) % 4}): part_o max|d| {dpo.max():.3e} ({nz} of {dpo.size} differ) "
                  f"m {dm.max():.3e} l {dl.max():.3e} (rel {(dl / np.abs(b[1][row, :, :, 1])).max():.2e}) align {dal.max():.3e} ({int((a[2][row] != b[2][row]).sum())} differ)")
            keys = np.nonzero(a[2][row] != b[2][row])[0]
            if len(keys):
                per = 250
                w = (keys % per) // 32; t = ((keys % per) % 32) // 16; r = (keys % per) % 16
                print(f"      align keys that differ: splits {np.bincount(keys // per, minlength=6).tolist()} waves {np.bincount(w, minlength=8).tolist()} tile t {np.bincount(t, minlength=2).tolist()} r {np.bincount(r, minlength=16).tolist()}")
                kk = keys[:6]
                print(f"      first: keys {kk.tolist()} rows-kernel {a[2][row][kk].tolist()} one-row {b[2][row][kk].tolist()}")
            if nz and row % nq in (1, 4):
                sp, col = np.unravel_index(np.argmax(dpo), dpo.shape)
                print(f"      worst: split {sp} column {col} (head {col // 64}, dim {col % 64}): rows {a[0][sp, row, col]!r} one {b[0][sp, row, col]!r}")
                hh = col // 64
                print(f"      per split of that head: differing dims {[int((a[0][s, row, hh * 64:(hh + 1) * 64] != b[0][s, row, hh * 64:(hh + 1) * 64]).sum()) for s in range(6)]}")
    lib.cw_test_set_option(b"cross_test_fp8", 0)
    eng.close()


if __name__ == "__main__":
    sys.exit(main())
