#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -save-temps .s) for reads of SGPRs that are the destination of a scalar load still in
flight (between the s_load and the next s_waitcnt lgkmcnt(0)).  The batched CTM kernel streams invsigma through SGPRs with
hand-placed loads and waits (csrc/tmvb_ctm_batch.h); a compiler-inserted copy of an in-flight group would read garbage.
Second check, for the fixed-register streaming (s[34:99] named literally in the asm): between the "; CBFX_BEGIN" and
"; CBFX_END" markers no compiler-generated instruction may WRITE an SGPR in 34..99 (the only writers are the hand-placed
s_mov_b64 s[34:35] and s_load_dwordx16).
Usage: check_smem_inflight.py file.s [kernel-name-substring]    exit code 1 when a hazard is found."""
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    inflight, bad, kern, on = set(), 0, None, want is None
    region, regions = False, 0
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if "CBFX_BEGIN" in t:
            region = True; regions += 1
        elif "CBFX_END" in t:
            region = False
        lab = re.match(r"^([A-Za-z_][\w$.]*):", t)
        if lab and not t.startswith("."):
            kern = lab.group(1); on = want is None or want in kern; inflight = set()
        if not on or not t or t.startswith(";") or t.startswith("."):
            continue
        if region and on and not t.startswith(";") and not t.startswith("."):
            wr = None
            m0 = re.match(r"(s_\w+|v_readlane_b32|v_readfirstlane_b32|v_cmp\w*_e64|v_cmpx?\w*)\s+(s\[(\d+):(\d+)\]|s(\d+))", t)
            if m0 and not re.match(r"s_(load|waitcnt|nop|cbranch|branch|barrier|sleep|setprio|endpgm|cmp|bitcmp)", t):
                lo = int(m0.group(3) or m0.group(5)); hi = int(m0.group(4) or m0.group(5))
                if hi >= 34 and lo <= 99 and not t.startswith("s_mov_b64 s[34:35]"):
                    bad += 1
                    if bad <= 10:
                        print(f"{path}:{ln}: [{kern}] writes a reserved SGPR inside a streaming region: {t[:90]}")
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
