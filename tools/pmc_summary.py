#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (rocpd databases) per kernel.
Usage: pmc_summary.py fetch.db write.db --iters N [--json PATH] [--model lda|ctm|ctpf]
N = outer iterations the profiled command ran (warm-up + timed): the E-step issues several dispatches of
the same kernel per iteration (document pieces), so the figure that matters is KB per ITERATION."""
import json
import sqlite3
import sys


def load(path):
    c = sqlite3.connect(path)
    out = {}
    for name, cnt, n, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                       "group by kernel_name, counter_name"):
        out[name] = (cnt, n, tot)
    return out


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    js = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    iters = int(sys.argv[sys.argv.index("--iters") + 1])
    print("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), KB per outer iteration")
    print(f"# (sum over all dispatches of the kernel / {iters} iterations of the profiled command)")
    print("# gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads")
    print("#   -> corrected_read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is used uncorrected")
    print(f"{'calls':>6} {'FETCH_KB/it':>12} {'WRITE_KB/it':>12}  kernel")
    rows = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0, 0))[2])):
        f = fetch.get(k, (None, 0, 0.0)); w = write.get(k, (None, 0, 0.0))
        print(f"{max(f[1], w[1]):6d} {f[2] / iters:12.1f} {w[2] / iters:12.1f}  {k[:100]}")
        rows[k] = {"calls": max(f[1], w[1]), "iterations": iters, "fetch_kb_per_iteration": f[2] / iters,
                   "write_kb_per_iteration": w[2] / iters}
    if js:
        # stamp the kernel sources the counters were collected on: bench.py reports roofline.traffic only when
        # the stamp matches the sources of the running build (a stale summary yields null)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_hash
        fam = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "lda"      # lda | ctm | ctpf: whose sources to stamp
        rows["_meta"] = {"kernel_source_hash": kernel_source_hash(fam), "model_family": fam, "iterations": iters}
        json.dump(rows, open(js, "w"), indent=1)


if __name__ == "__main__":
    main()
