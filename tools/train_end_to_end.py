#!/usr/bin/env python
"""The PCIe-inclusive rate of train!: what a caller of the drop-in boundary sees when it hands over HOST arrays.

    python tools/train_end_to_end.py [lda ctm ctpf]

The boundary (src/gpuLDA.jl:347-376: update_buffer!, the loop, update_host!) takes the model from host memory and returns it there.  bench.py's
`value` is the rate of the loop with everything resident in HBM; this tool times the three parts of one train!(iter=150, checkelbo=Inf) call on
the bench workloads:
    create      corpus upload (CSR, 8 B per posting) + the inverted index / document buckets built on the host + the handle's allocations
    train       update_buffer! (state host -> device), 150 iterations, update_host! (device -> host, fp32 -> fp64 conversion on the host)
    loop        the same 150 iterations timed inside (tmvb_*_train alone, from a second call on the warm handle: no state traffic)
and prints iterations per second with and without the transfers."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tmvb_amd

tm = tmvb_amd.pkg
ITER = int(os.environ.get("ITER", 150))


def run(name):
    K = 50
    pc = tm.syn_citeu() if name == "ctpf" else tm.syn_nsf()
    cls = {"lda": tm.gpuLDA, "ctm": tm.gpuCTM, "ctpf": tm.gpuCTPF}[name]
    kw = {"recs": False} if name == "ctpf" else {}
    warm = cls(pc, K)                                   # first touch of the device, code objects, clocks: not part of the figure
    warm.train(iter=20, checkelbo=np.inf, printelbo=False, **kw)
    warm.close()
    # the library calls of one train! timed inside it: the loop (tmvb_<model>_train returns with the stream drained), the state's way down
    # (tmvb_<model>_set_state[_old]) and back up (tmvb_<model>_get_state), the topics' sort (tmvb_topic_order)
    L = tm.lib()
    spent = {}

    def wrap(sym, key):
        orig = getattr(L, sym)

        def timed(*a):
            ts = time.perf_counter()
            rc = orig(*a)
            spent[key] = spent.get(key, 0.0) + time.perf_counter() - ts
            return rc
        setattr(L, sym, timed)
        return sym, orig
    saved = [wrap(f"tmvb_{name}_train", "loop"), wrap(f"tmvb_{name}_set_state", "up"), wrap(f"tmvb_{name}_get_state", "down"), wrap("tmvb_topic_order", "topics")]
    if name == "ctpf":
        saved.append(wrap("tmvb_ctpf_set_state_old", "up"))
    try:
        t0 = time.perf_counter()
        g = cls(pc, K)
        g.ctx.synchronize()
        t1 = time.perf_counter()
        created_up = spent.pop("up", 0.0)                  # the constructor's own update_buffer!: part of `create`
        g.train(iter=ITER, checkelbo=np.inf, printelbo=False, **kw)
        t2 = time.perf_counter()
    finally:
        for sym, orig in saved:
            setattr(L, sym, orig)
    loop, up, down, topics = spent.get("loop", 0.0), spent.get("up", 0.0), spent.get("down", 0.0), spent.get("topics", 0.0)
    host = (t2 - t1) - loop - up - down - topics            # the Python mirror: check_model on the host arrays, fresh numpy arrays for update_host!
    state_mb = sum(np.asarray(getattr(g, f)).nbytes for f in vars(g) if isinstance(getattr(g, f), np.ndarray)) / 1e6
    print(f"{name} K={K} M={pc.M} nnz={pc.nnz}: create {1e3 * (t1 - t0):.1f} ms (its update_buffer! {1e3 * created_up:.1f}); train!(iter={ITER}) {1e3 * (t2 - t1):.1f} ms = "
          f"update_buffer! {1e3 * up:.1f} + loop {1e3 * loop:.1f} + update_host! {1e3 * down:.1f} + topics {1e3 * topics:.1f} + Python-side checks and allocations {1e3 * host:.1f} ms "
          f"(host state {state_mb:.0f} MB as fp64); {ITER / loop:.0f} it/s in the loop (from the cold start), {ITER / (loop + up + down):.0f} it/s with the state "
          f"transfers, {ITER / (t2 - t1):.0f} it/s for the whole call, {ITER / (t2 - t0):.0f} it/s with corpus upload and index build as well",
          flush=True)
    g.close()


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["lda", "ctm", "ctpf"]):
        run(n)
