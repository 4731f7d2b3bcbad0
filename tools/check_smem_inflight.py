#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -save-temps .s) for reads of SGPRs that are the destination of a scalar load still in
flight (between the s_load and the next s_waitcnt lgkmcnt(0)).  The batched CTM kernel streams invsigma through SGPRs with
hand-placed loads and waits (csrc/tmvb_ctm_batch.h); a compiler-inserted copy of an in-flight group would read garbage.
Second check, for the fixed-register streaming (16-register blocks of s[36:99] named literally in the asm): between the
"; CBFX_BEGIN" and "; CBFX_END" markers a block belongs to the stream from its s_load_dwordx16 to the last v_pk_fma_f32
that reads one of its registers before the block is loaded again (or the region ends); no other instruction may WRITE a
register of the block in that interval.  (Outside those intervals the compiler may use the registers: the blocks are
physical-register operands of the asm statements, so it knows when they are dead.)
Usage: check_smem_inflight.py file.s [kernel-name-substring]    exit code 1 when a hazard is found."""
import re
import sys


WRITER = re.compile(r"(s_\w+|v_readlane_b32|v_readfirstlane_b32|v_cmp\w*_e64|v_cmpx?\w*)\s+(s\[(\d+):(\d+)\]|s(\d+))")
NOT_WRITER = re.compile(r"s_(load|waitcnt|nop|cbranch|branch|barrier|sleep|setprio|endpgm|cmp|bitcmp)")


def check_region(path, lines, already):
    """lines: [(line number, text, kernel)] of one streaming region.  Returns the number of hazards found."""
    loads = []                                     # (index, lo, hi)
    for i, (_, t, _) in enumerate(lines):
        m = re.match(r"s_load_dwordx16\s+s\[(\d+):(\d+)\]", t)
        if m and 36 <= int(m.group(1)) <= 99:
            loads.append((i, int(m.group(1)), int(m.group(2))))
    bad = 0
    for k, (i0, lo, hi) in enumerate(loads):
        nxt = next((j for j, l2, _ in loads[k + 1:] if l2 == lo), len(lines))
        last = i0
        for j in range(i0 + 1, nxt):
            t = lines[j][1]
            if t.startswith("v_pk_fma_f32"):
                for mm in re.finditer(r"s\[(\d+):(\d+)\]", t):
                    if lo <= int(mm.group(1)) <= hi:
                        last = j
        for j in range(i0 + 1, last):
            ln, t, kern = lines[j]
            m0 = WRITER.match(t)
            if m0 and not NOT_WRITER.match(t):
                wlo = int(m0.group(3) or m0.group(5)); whi = int(m0.group(4) or m0.group(5))
                if whi >= lo and wlo <= hi:
                    bad += 1
                    if already + bad <= 10:
                        print(f"{path}:{ln}: [{kern}] writes a reserved SGPR inside a streaming region: {t[:90]}")
    return bad


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    inflight, bad, kern, on = set(), 0, None, want is None
    region, regions, region_lines = False, 0, []
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if "CBFX_BEGIN" in t:
            region = True; regions += 1; region_lines = []
        elif "CBFX_END" in t:
            region = False
            if on:
                bad += check_region(path, region_lines, bad)
        lab = re.match(r"^([A-Za-z_][\w$.]*):", t)
        if lab and not t.startswith("."):
            kern = lab.group(1); on = want is None or want in kern; inflight = set()
        if not on or not t or t.startswith(";") or t.startswith("."):
            continue
        if region and on:
            region_lines.append((ln, t, kern))
        if "s_waitcnt" in t and ("lgkmcnt(0)" in t or re.search(r"s_waitcnt\s+0x?0*\b", t)):
            inflight = set(); continue
        m = re.match(r"s_(?:buffer_)?load_dword(?:x\d+)?\s+s\[?(\d+)(?::(\d+))?\]?", t)
        if m:
            a = int(m.group(1)); b = int(m.group(2)) if m.group(2) else a
            inflight |= set(range(a, b + 1)); continue
        if inflight:
            regs = set()
            for mm in re.finditer(r"s\[(\d+):(\d+)\]", t):
                regs |= set(range(int(mm.group(1)), int(mm.group(2)) + 1))
            for mm in re.finditer(r"\bs(\d+)\b", t):
                regs.add(int(mm.group(1)))
            if regs & inflight:
                bad += 1
                if bad <= 10:
                    print(f"{path}:{ln}: [{kern}] reads in-flight SGPR(s) {sorted(regs & inflight)[:4]}: {t[:90]}")
    print(f"{bad} hazards ({regions} fixed-register streaming regions checked)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
