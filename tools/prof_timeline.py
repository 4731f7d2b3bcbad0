#!/usr/bin/env python
"""Kernel timeline of an LDA run from a rocprofv3 --kernel-trace rocpd database.
  prof_timeline.py <results.db> [k]    dispatches of the k-th iteration from the end (default 3), times in us
                                       relative to its first document kernel, plus the E-step span statistics
An iteration ends with beta_norm_kernel.  E-step span = first lda_estep* start .. last termstats* end: the
interval bench.py times with events on the context stream (roofline.estep_ms)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select d.start, d.end, s.display_name, d.queue_id, d.grid_size_x from {kd} d "
                       f"join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [i for i, r in enumerate(rows) if "beta_norm" in r[2]]
    # per-iteration E-step spans
    spans, iters = [], []
    prev = 0
    for e in ends:
        it = rows[prev:e + 1]
        docs = [r for r in it if "lda_estep" in r[2]]
        stats = [r for r in it if "termstats" in r[2]]
        if docs and stats:
            spans.append((max(r[1] for r in stats) - min(r[0] for r in docs)) / 1e3)
            iters.append((min(r[0] for r in docs), e))
        prev = e + 1
    if len(iters) > back + 1:
        first, e = iters[-back - 1]
        lo = next(i for i, r in enumerate(rows) if r[0] == first)
        nxt = iters[-back][0]
        print(f"# iteration {len(iters) - back - 1} of {len(iters)}: start_us end_us dur_us queue grid kernel")
        for st, en, name, q, g in rows[lo:]:
            if st >= nxt:
                break
            print(f"{(st - first) / 1e3:9.1f} {(en - first) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q:<3d} g{g:<8d} {name[:72]}")
        print(f"# iteration period (first document kernel to the next iteration's): {(nxt - first) / 1e3:.1f} us")
    if spans:
        steady = spans[3:] if len(spans) > 6 else spans
        print(f"# E-step span over {len(steady)} steady iterations (us): mean {sum(steady) / len(steady):.1f} "
              f"min {min(steady):.1f} max {max(steady):.1f}")


if __name__ == "__main__":
    main()
