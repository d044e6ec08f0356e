"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table, like `--stats`.
usage: python profiles/summarize.py gpurun_out/prof/run_results.db > profiles/<name>.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc")
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"# {path}\n# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows:
        print(f"{r[0][:100]:100s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100 * r[2] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
