#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) as text: per-kernel totals and, for the E-step
kernel, per-launch-shape (document-length bucket) durations.  Usage: prof_summary.py results.db"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("## per kernel (top_kernels view)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {name[:110]}")
    print("## per launch shape")
    print(f"{'grid_wg':>9} {'wg':>4} {'lds_B':>7} {'vgpr':>5} {'sgpr':>5} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  name")
    q = ("select name, grid_x/workgroup_x, workgroup_x, lds_size, vgpr_count, sgpr_count, count(*), avg(duration)/1e3, "
         "min(duration)/1e3, max(duration)/1e3 from kernels group by name, grid_x, lds_size order by 8 desc")
    for r in c.execute(q):
        print(f"{r[1]:9d} {r[2]:4d} {r[3]:7d} {r[4]:5d} {r[5]:5d} {r[6]:6d} {r[7]:10.2f} {r[8]:10.2f} {r[9]:10.2f}  {r[0][:70]}")


if __name__ == "__main__":
    main(sys.argv[1])
