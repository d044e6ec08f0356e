"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database.  usage: python tools/pmc_table.py run_results.db [filter] [--each]
(--each: one line per dispatch of the kernels matching the filter, in dispatch order, instead of averages)"""
import sqlite3, sys
db = sqlite3.connect([a for a in sys.argv[1:] if a != "--each"][0])
each = "--each" in sys.argv
args = [a for a in sys.argv[1:] if a != "--each"]
flt = args[1] if len(args) > 1 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
pe = [t for t in tabs if "pmc_event" in t][0]
pi = [t for t in tabs if "info_pmc" in t][0]
q = (f"select s.kernel_name, i.name, avg(p.value), count(*) from {pe} p join {pi} i on p.pmc_id = i.id join {kd} d on p.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name")
if each:
    qe = (f"select d.event_id, s.kernel_name, i.name, sum(p.value) from {pe} p join {pi} i on p.pmc_id = i.id join {kd} d on p.event_id = d.event_id "
          f"join {ks} s on d.kernel_id = s.id group by d.event_id, i.name order by d.event_id")
    for ev, k, n, v in db.execute(qe):
        if flt and flt not in k:
            continue
        print(f"{ev:8d} {k[:60]:60s} {n:20s} {v:16.1f}")
    sys.exit(0)
acc = {}
for k, n, v, c in db.execute(q):
    acc.setdefault(k, {})[n] = (v, c)
for k in sorted(acc):
    if flt and flt not in k:
        continue
    print(k[:110])
    for n, (v, c) in sorted(acc[k].items()):
        print(f"    {n:32s} {v:16.1f}   ({c} samples)")
