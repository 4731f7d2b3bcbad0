#!/usr/bin/env python
"""Schedule of the lane-per-document CTM launch from a TMVB_CTM_WAVE_LOG dump (TMVB_CTM_PROF=1 TMVB_CTM_WAVE_LOG=<file>):
per-item start / end on the 100 MHz wall clock -> span, mean busy time per SIMD slot, the tail, list-scheduling what-ifs."""
import heapq
import sys
import numpy as np

w = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
st = w[:, 0].astype(np.int64); en = w[:, 1].astype(np.int64)
nmax = (w[:, 3] & 0xffffffff).astype(np.int64); ntrip = (w[:, 3] >> 32).astype(np.int64)
t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0; dur = en - st
print(f"items {len(w)}  span {en.max():.0f} us  sum of item times / {slots} slots {dur.sum() / slots:.0f} us  longest item {dur.max():.0f} us")
print("item time us: first %d items min/median/max %.0f/%.0f/%.0f, the rest %.0f/%.0f/%.0f" % (
    slots, *np.percentile(dur[:slots], [0, 50, 100]), *(np.percentile(dur[slots:], [0, 50, 100]) if len(dur) > slots else (0, 0, 0))))
A = np.stack([nmax, ntrip, np.ones_like(nmax)], 1).astype(float)
coef = np.linalg.lstsq(A, dur, rcond=None)[0]
print("fit: item time = %.2f us x longest document + %.1f us x Newton trips + %.0f us  (rms %.0f us)" % (*coef, np.sqrt(((A @ coef - dur) ** 2).mean())))
if len(dur) > slots:
    e1 = np.sort(en[:slots]); s2 = np.sort(st[slots:]); gap = s2 - e1[:len(s2)]
    print("k-th start after the first round minus k-th end of the first round, us: min/median/max %.1f/%.1f/%.1f" % tuple(np.percentile(gap, [0, 50, 100])))
print("items running at t (us):", ", ".join(f"{t}: {int(((st <= t) & (en > t)).sum())}" for t in range(500, int(en.max()) + 500, 500)))


def sched(d):
    h = [0.0] * slots; heapq.heapify(h)
    for x in d:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)


print(f"list scheduling of the measured item times on {slots} slots: queue order {sched(dur):.0f} us, longest first {sched(np.sort(dur)[::-1]):.0f} us")
