#!/usr/bin/env python
"""Per-kernel durations over the LAST n iterations of a rocprofv3 --kernel-trace rocpd database (an iteration ends with the marker kernel):
calls per iteration, mean duration, busy time per iteration, and the iterations' mean period and first-to-last span.
  prof_lastn.py <results.db> <marker substring> [n]"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1]); marker = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.display_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(ends) < n + 1:
        print("not enough markers", len(ends)); return
    lo, hi = ends[-n - 1] + 1, ends[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for st, en, name in rows[lo:hi + 1]:
        a = agg[name[:70]]; a[0] += 1; a[1] += (en - st) / 1e3
    period = (rows[ends[-1]][1] - rows[ends[-n - 1]][1]) / 1e3 / n
    spans = []
    for k in range(n):
        a, b = ends[-n - 1 + k] + 1, ends[-n + k]
        docs = [r for r in rows[a:b + 1] if "estep" in r[2]]
        stats = [r for r in rows[a:b + 1] if "termstats" in r[2] or "colsum" in r[2]]
        if docs and stats:
            spans.append(((max(r[1] for r in docs) - docs[0][0]) / 1e3, (max(r[1] for r in stats) - docs[0][0]) / 1e3))
    print(f"# last {n} iterations of {len(ends)} (marker '{marker}'): mean period {period:.1f} us"
          + (f"; document kernels first start -> last end {sum(s[0] for s in spans) / len(spans):.1f} us, -> last statistics / column-sum kernel end {sum(s[1] for s in spans) / len(spans):.1f} us" if spans else ""))
    print("#  calls/it   mean_us   busy_us/it  kernel")
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {c / n:7.2f}  {t / c:8.1f}  {t / n:10.1f}  {name}")


if __name__ == "__main__":
    main()
