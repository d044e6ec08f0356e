"""Per-kernel HBM traffic from two rocprofv3 PMC passes (rocpd databases).
usage: python profiles/summarize_pmc.py gpurun_out/pmc/fetch_results.db gpurun_out/pmc/write_results.db [gpurun_out/pmc/mfma_results.db] > profiles/<name>.json
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE counts half the bytes of a 16 B/lane
coalesced stream (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence hbm_read_bytes = FETCH_SIZE * 1024 * 2."""
import json
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    q = (f"select s.kernel_name, count(*), avg(p.value) from {pe} p join {kd} d on p.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id group by s.kernel_name")
    return {r[0]: (r[1], r[2]) for r in db.execute(q)}


def mfma_util(path):
    """{kernel: (busy quad-cycles per XCD instance, GPU-active cycles, utilisation)} from a pass with
    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE.  The SQ counter comes back per XCD instance in quad-cycles on this stack:
    x4 reproduces exactly (kernel MFMA count) x 16 cycles for the 16x16x32 bf16 MFMAs of the GEMMs (checked against the
    algorithmic FLOPs of the QKV / cross-KV / fc1 GEMMs), so utilisation = 4 * busy / (active * 32 CUs * 4 SIMDs)."""
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    pi = [t for t in tabs if "info_pmc" in t][0]
    q = (f"select s.kernel_name, i.name, avg(p.value) from {pe} p join {pi} i on p.pmc_id = i.id join {kd} d on p.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name")
    acc = {}
    for k, n, v in db.execute(q):
        acc.setdefault(k, {})[n] = v
    out = {}
    for k, d in acc.items():
        busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
        if busy > 0 and act > 0:
            out[k] = (busy, act, 4.0 * busy / (act * 128.0))
    return out


def main(fetch_db, write_db, cmd, mfma_db=None):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    mu = mfma_util(mfma_db) if mfma_db else {}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + cmd,
           "units": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch; hbm_read_bytes = FETCH_SIZE*1024*2 (gfx950 correction for "
                    "16 B/lane coalesced streams, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated; mfma_util = 4 * "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128 SIMDs per XCD), third pass (see summarize_pmc.py)",
           "batch": 8, "kernels": {}}
    for k in sorted(f, key=lambda k: -f[k][1]):
        out["kernels"][k] = {"dispatches": f[k][0], "FETCH_SIZE_KiB_avg": round(f[k][1], 1),
                             "WRITE_SIZE_KiB_avg": round(w.get(k, (0, 0.0))[1], 1),
                             "hbm_read_bytes_per_launch": int(round(f[k][1] * 1024 * 2))}
        if k in mu:
            out["kernels"][k].update({"mfma_busy_quadcycles_per_xcd": round(mu[k][0]), "gpu_active_cycles": round(mu[k][1]),
                                      "mfma_util": round(mu[k][2], 4)})
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    mfma = [a for a in sys.argv[3:] if a.endswith(".db")]
    rest = [a for a in sys.argv[3:] if not a.endswith(".db")]
    main(sys.argv[1], sys.argv[2], rest[0] if rest else
         "python bench.py --batch 8 --tokens 4 --steps 1 --warmup 0 --no-cpu-baseline --kernel-iters 3", mfma[0] if mfma else None)
