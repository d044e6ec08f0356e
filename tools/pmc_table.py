"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database.  usage: python tools/pmc_table.py run_results.db [filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
pe = [t for t in tabs if "pmc_event" in t][0]
pi = [t for t in tabs if "info_pmc" in t][0]
q = (f"select s.kernel_name, i.name, avg(p.value), count(*) from {pe} p join {pi} i on p.pmc_id = i.id join {kd} d on p.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name")
acc = {}
for k, n, v, c in db.execute(q):
    acc.setdefault(k, {})[n] = (v, c)
for k in sorted(acc):
    if flt and flt not in k:
        continue
    print(k[:110])
    for n, (v, c) in sorted(acc[k].items()):
        print(f"    {n:32s} {v:16.1f}   ({c} samples)")
