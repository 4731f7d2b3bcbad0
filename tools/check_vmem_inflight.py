#!/usr/bin/env python
"""ISA check of the hand-scheduled token loop of ctm_estep_quad_kernel (csrc/tmvb_ctm_quad.h): its row / id loads are inline asm whose waits are
placed by hand, so the compiler does not know that a destination register is in flight.  Between the CQVM_BEGIN / CQVM_END markers of a kernel, for
every global_load inside the innermost loop that holds >= 4 global_load_dwordx4, walk forward (around the back edge, and out of the loop through its exits up to the next s_waitcnt vmcnt(0)) until the
hand-placed s_waitcnt vmcnt that makes the destination valid, and report any instruction in between that names a destination register.
    python tools/check_vmem_inflight.py <isa.s> [kernel-name-substring]      exit status 1 on a finding"""
import re
import sys

def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

def all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= regs_of(tok)
    return out

def check(body, name):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    beg = [i for i, l in enumerate(body) if "CQVM_BEGIN" in l]
    end = [i for i, l in enumerate(body) if "CQVM_END" in l]
    if not beg or not end:
        print(f"{name}: compiler-managed loop (no CQVM markers), nothing to check"); return 0
    best = None
    for i, l in enumerate(body):
        if not (beg[0] < i < end[-1]):
            continue
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            g = sum(1 for k in range(a, i) if "global_load_dwordx4" in body[k])
            if g >= 4 and (best is None or i - a < best[1] - best[0]):
                best = (a, i)
    if best is None:
        print(f"{name}: no hand-scheduled loop found"); return 0
    a, b = best
    loop = [l.split(";")[0].strip() for l in body[a:b + 1]]
    loads = [(i, l) for i, l in enumerate(loop) if l.startswith("global_load")]
    waits = [(i, int(re.search(r"vmcnt\((\d+)\)", l).group(1))) for i, l in enumerate(loop) if l.startswith("s_waitcnt") and "vmcnt" in l]
    findings = 0
    n = len(loop)
    for li, l in loads:
        dst = regs_of(l.split()[1].rstrip(","))
        # walk forward around the loop: count younger loads; the load is complete at the first wait with vmcnt(N) <= younger
        younger = 0
        k = li + 1
        steps = 0
        while steps < 2 * n:
            kk = k % n
            ins = loop[kk]
            if ins.startswith("global_load"):
                younger += 1
            elif ins.startswith("s_waitcnt") and "vmcnt" in ins:
                N = int(re.search(r"vmcnt\((\d+)\)", ins).group(1))
                if N <= younger:
                    break
            elif ins and not ins.startswith(".") and (all_vregs(ins) & dst):
                print(f"{name}: loop+{kk}: '{ins}' touches {sorted(all_vregs(ins) & dst)} while loop+{li} '{l}' is in flight"); findings += 1
            k += 1; steps += 1
        else:
            print(f"{name}: no wait found for loop+{li} '{l}'"); findings += 1
    # the prologue: loads between CQVM_BEGIN and the loop, walked forward into the loop
    pro = [l.split(";")[0].strip() for l in body[beg[0]:a]]
    seq = pro + loop + loop
    for li, l in enumerate(pro):
        if not l.startswith("global_load"):
            continue
        dst = regs_of(l.split()[1].rstrip(","))
        younger = 0
        for kk in range(li + 1, len(seq)):
            ins = seq[kk]
            if ins.startswith("global_load"):
                younger += 1
            elif ins.startswith("s_waitcnt") and "vmcnt" in ins:
                if int(re.search(r"vmcnt\((\d+)\)", ins).group(1)) <= younger:
                    break
            elif ins and not ins.startswith(".") and (all_vregs(ins) & dst):
                print(f"{name}: prologue+{kk}: '{ins}' touches {sorted(all_vregs(ins) & dst)} while prologue+{li} '{l}' is in flight"); findings += 1
        else:
            print(f"{name}: no wait found for prologue+{li} '{l}'"); findings += 1
    # the exit path: from the loop's end to the first vmcnt(0), nothing may touch any load destination
    alld = set()
    for li, l in loads:
        alld |= regs_of(l.split()[1].rstrip(","))
    k = b + 1
    while k < len(body):
        ins = body[k].split(";")[0].strip()
        if ins.startswith("s_waitcnt") and "vmcnt(0)" in ins:
            break
        if ins.startswith("s_endpgm"):
            print(f"{name}: no s_waitcnt vmcnt(0) behind the loop"); findings += 1; break
        if ins and not ins.startswith(".") and (all_vregs(ins) & alld):
            print(f"{name}: exit+{k - b}: '{ins}' touches in-flight registers {sorted(all_vregs(ins) & alld)}"); findings += 1
        k += 1
    print(f"{name}: loop of {n} instructions, {len(loads)} loads, {len(waits)} waits, {findings} finding(s)")
    return findings

def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else "ctm_estep_quad_kernel"
    total = 0
    for m in re.finditer(r"\n(_Z\w*%s\w*):[^\n]*\n" % pat, s):
        i0 = m.end()
        i1 = s.index(".end_amdhsa_kernel", i0)
        total += check(s[i0:i1].split("\n"), m.group(1)[:48])
    sys.exit(1 if total else 0)

main()
