#!/usr/bin/env python3
"""Kernel time table from a rocprofv3 results database:  python tools/profdb.py gpurun_out/prof_x/insp_results.db [filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
for r in rows:
    if flt in r[0]:
        print(f"{r[0][:100]:100s} n={r[1]:5d} total_ms={r[2]:9.3f} avg_us={r[3]:10.1f} min_us={r[4]:10.1f}")
