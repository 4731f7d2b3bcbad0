#!/usr/bin/env python
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database, plus the VALU-busy figure when
SQ_ACTIVE_INST_VALU and GRBM_GUI_ACTIVE are present and the MFMA-busy figure when SQ_VALU_MFMA_BUSY_CYCLES /
SQ_BUSY_CYCLES are present.  Usage: counter_summary.py results.db [results2.db ...]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        for name, cnt, n, tot in c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                           "group by kernel_name, counter_name"):
            acc[name][cnt][0] += n; acc[name][cnt][1] += tot
    counters = sorted({c for k in acc for c in acc[k]})
    print("# per-dispatch averages; counters:", " ".join(counters))
    print("# VALU busy = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE * 128): the counter ticks once per 4-cycle issue of a wave64 VALU")
    print("#   instruction per SIMD, GRBM_GUI_ACTIVE is summed over the 8 XCDs (128 SIMDs each)")
    hdr = f"{'kernel':<72} {'calls':>6} " + " ".join(f"{c[:22]:>22}" for c in counters) + "  derived"
    print(hdr)
    order = sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", acc[k].get(counters[0], [0, 0]))[1])
    for k in order:
        row = acc[k]
        calls = max(v[0] for v in row.values())
        avg = {c: (row[c][1] / row[c][0] if c in row and row[c][0] else float("nan")) for c in counters}
        der = []
        if "SQ_ACTIVE_INST_VALU" in row and "GRBM_GUI_ACTIVE" in row:
            der.append(f"VALU busy {100 * 4 * avg['SQ_ACTIVE_INST_VALU'] / (avg['GRBM_GUI_ACTIVE'] * 128):.1f}%")
        if "SQ_INSTS_VALU" in row and "SQ_WAVES" in row and avg["SQ_WAVES"]:
            der.append(f"VALU/wave {avg['SQ_INSTS_VALU'] / avg['SQ_WAVES']:.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "SQ_BUSY_CYCLES" in row and avg["SQ_BUSY_CYCLES"]:
            der.append(f"MFMA busy {100 * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CYCLES']:.2f}% of SQ busy cycles")
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in row and "SQ_INSTS_VALU" in row and avg["SQ_INSTS_VALU"]:
            der.append(f"MFMA f32 mops {avg['SQ_INSTS_VALU_MFMA_MOPS_F32']:.0f}")
        print(f"{k[:72]:<72} {calls:6d} " + " ".join(f"{avg[c]:22.1f}" for c in counters) + "  " + "; ".join(der))


if __name__ == "__main__":
    main()
