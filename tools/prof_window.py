#!/usr/bin/env python
"""Kernel timeline between two occurrences of a marker kernel, from a rocprofv3 --kernel-trace rocpd database.
  prof_window.py <results.db> <marker substring> [k]   dispatches after the k-th last marker up to the next one (us)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.display_name, d.queue_id, d.grid_size_x from {kd} d "
                       f"join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(ends) < back + 2:
        print("not enough markers", len(ends)); return
    lo, hi = ends[-back - 2] + 1, ends[-back - 1]
    t0 = rows[lo][0]
    print(f"# dispatches between marker {len(ends) - back - 2} and {len(ends) - back - 1} of {len(ends)} ('{marker}'): start_us end_us dur_us queue grid kernel")
    for st, en, name, q, g in rows[lo:hi + 1]:
        print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q:<3d} g{g:<8d} {name[:80]}")
    print(f"# period: {(rows[ends[-back - 1] + 1][0] - t0) / 1e3:.1f} us" if ends[-back - 1] + 1 < len(rows) else "")


if __name__ == "__main__":
    main()
