#!/usr/bin/env python
"""Register / scratch / LDS figures of every kernel in libtmvb_hip.so, from the code objects' own metadata.

    python tools/kernel_resources.py [lib.so]             table of every kernel (sorted by scratch, then registers)
    python tools/kernel_resources.py --check [lib.so]     the build-time check: every kernel named in BENCHED must exist and keep its
                                                          private segment (scratch) within its ceiling; build() fails otherwise

Round-4 review: the benched CTM kernel owned 512 registers and still spilled to 428 B / lane of scratch, and nothing in the repository said
so.  The .so carries one clang offload bundle per translation unit (section .hip_fatbin, magic __CLANG_OFFLOAD_BUNDLE__); each gfx950 entry
is an ELF code object whose NT_AMDGPU_METADATA note (llvm-readelf --notes) lists, per kernel, .vgpr_count / .agpr_count / .sgpr_count /
.private_segment_fixed_size / .vgpr_spill_count / .sgpr_spill_count / .group_segment_fixed_size.  No GPU needed.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = os.environ.get("LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# The kernels behind the numbers of bench.py's line (demangled-name prefixes) and the scratch bytes per lane each may use.  Everything that is
# timed runs without scratch -- except the lane-per-document CTM kernel, whose 388 B (428 in round 4) round 5 could not remove
# without losing time (profiles/r5_ctm_token_experiments.txt); its ceiling is that figure, so that it cannot silently grow.
BENCHED = {
    # config 2 / 3: LDA K = 50 (13 chunks per row) and K = 100 (25)
    "lda_estep_grid_kernel<13,": 0, "lda_estep_grid_kernel<25,": 0, "lda_estep_grid_long_kernel<13,": 0, "lda_estep_grid_long_kernel<25,": 0,
    "termstats_recompute_kernel<13,": 0, "termstats_recompute_kernel<25,": 0, "termstats_multi_kernel": 0, "beta_norm_kernel": 0, "lda_alpha_kernel<": 0,
    # config 4: CTM K = 50
    "ctm_estep_batch_kernel<52, false, false>": 388, "ctm_estep_quad_kernel<52, false>": 96, "ctm_scatter_mfma_kernel": 0, "ctm_sigma_mu_kernel": 0,
    # config 5: CTPF K = 50
    "ctpf_estep_grid_narrow_kernel<13>": 0, "ctpf_estep_grid_wide_kernel<13>": 0, "ctpf_estep_grid_long2_kernel<13>": 0, "termstats_recompute2_kernel<13": 0,
    "ctpf_mstep_kernel": 0,
}


def code_objects(path):
    """the gfx950 ELF images inside the clang offload bundles of `path`"""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += 24
    return out


def demangle(names):
    try:
        exe = next(e for e in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt", "c++filt") if e == "c++filt" or os.path.exists(e))
        r = subprocess.run([exe], input="\n".join(names) + "\n", capture_output=True, text=True, check=True)
        out = r.stdout.splitlines()
        return out if len(out) == len(names) else names
    except Exception:
        return names


def kernels(path):
    rows = []
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.(?=agpr_count|args)", txt):
            m = re.search(r"\.name:\s+(\S+)", blk)
            if not m or ".vgpr_count" not in blk:
                continue
            g = lambda k, d=0: int((re.search(r"\." + k + r":\s+(\d+)", blk) or [0, d])[1])
            rows.append({"name": m.group(1), "vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"),
                         "scratch": g("private_segment_fixed_size"), "vgpr_spills": g("vgpr_spill_count"), "sgpr_spills": g("sgpr_spill_count"),
                         "lds": g("group_segment_fixed_size"), "wg": g("max_flat_workgroup_size")})
    dm = demangle([r["name"] for r in rows])
    for r, d in zip(rows, dm):
        r["demangled"] = re.sub(r"^void ", "", d)
    return rows


def check(rows):
    bad = []
    for pref, ceiling in BENCHED.items():
        hit = [r for r in rows if r["demangled"].startswith(pref)]
        if not hit:
            bad.append(f"benched kernel {pref!r} not found in the library")
        for r in hit:
            if r["scratch"] > ceiling:
                bad.append(f"{r['demangled'][:90]}: {r['scratch']} B of scratch per lane (ceiling {ceiling}); {r['vgpr_spills']} VGPR + {r['sgpr_spills']} SGPR spills")
    return bad


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = argv[0] if argv else os.path.join(ROOT, "topicmodelsvb.jl_amd", "libtmvb_hip.so")
    rows = kernels(lib)
    if "--check" in sys.argv:
        bad = check(rows)
        for b in bad:
            print("KERNEL RESOURCE CHECK:", b)
        print(f"kernel resource check: {len(rows)} kernels, {len(BENCHED)} benched prefixes, {'FAILED' if bad else 'ok'}")
        sys.exit(1 if bad else 0)
    if "--json" in sys.argv:
        print(json.dumps(rows)); return
    if "--benched" in sys.argv:
        rows = [r for r in rows if any(r["demangled"].startswith(p) for p in BENCHED)]
    rows.sort(key=lambda r: (-r["scratch"], -(r["vgpr"] + r["agpr"]), r["demangled"]))
    print(f"# {os.path.basename(lib)}: {len(rows)} gfx950 kernels (llvm-readelf --notes of the embedded code objects; tools/kernel_resources.py)")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch_B':>9} {'v_spill':>7} {'s_spill':>7} {'lds_B':>7}  benched  kernel")
    for r in rows:
        b = "*" if any(r["demangled"].startswith(p) for p in BENCHED) else " "
        print(f"{r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['scratch']:9d} {r['vgpr_spills']:7d} {r['sgpr_spills']:7d} {r['lds']:7d}     {b}     {r['demangled'][:130]}")


if __name__ == "__main__":
    main()
