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


def load_all(path):
    c = sqlite3.connect(path)
    out = {}
    for name, cnt, n, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                       "group by kernel_name, counter_name"):
        out.setdefault(name, {})[cnt] = (n, tot)
    return out


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    valu = load_all(sys.argv[sys.argv.index("--valu") + 1]) if "--valu" in sys.argv else {}
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
    if valu:
        # third pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE): VALU instructions issued and the cycles a SIMD's VALU was
        # issuing (4 cycles per wave64 instruction; SQ_ACTIVE_INST_VALU is summed over the 1024 SIMDs), per kernel and iteration
        print("# VALU pass: instructions per iteration / per wave, and VALU-issue cycles per SIMD per iteration = 4 * SQ_ACTIVE_INST_VALU / 1024")
        print(f"{'insts/it':>14} {'insts/wave':>11} {'issue_cyc/SIMD/it':>18} {'busy%':>6}  kernel")
        for k, r in sorted(valu.items(), key=lambda kv: -kv[1].get("SQ_ACTIVE_INST_VALU", (0, 0))[1]):
            iv = r.get("SQ_INSTS_VALU", (0, 0.0))[1]; av = r.get("SQ_ACTIVE_INST_VALU", (0, 0.0))[1]
            wv = r.get("SQ_WAVES", (0, 0.0))[1]; ga = r.get("GRBM_GUI_ACTIVE", (0, 0.0))[1]
            busy = 100.0 * 4.0 * av / (ga * 128.0) if ga else float("nan")
            print(f"{iv / iters:14.0f} {iv / wv if wv else float('nan'):11.1f} {4.0 * av / 1024.0 / iters:18.0f} {busy:6.1f}  {k[:100]}")
            rows.setdefault(k, {"iterations": iters})
            rows[k].update({"insts_valu_per_iteration": iv / iters, "waves_per_iteration": wv / iters,
                            "valu_issue_cycles_per_simd_per_iteration": 4.0 * av / 1024.0 / iters, "valu_busy_frac_alone": busy / 100.0})
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
