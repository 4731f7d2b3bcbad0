"""
ctypes binding of libtmvb_hip.so (the C ABI declared in include/tmvb.h).

There is NO CPU fallback: if the shared library is missing, or no gfx950 device is visible, every
compute entry point raises.  `build()` cross-compiles the library with hipcc for gfx950 (works
without a GPU) -- the built .so lives in-tree next to this file.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libtmvb_hip.so")
# experiments only (tools/build_variant.sh): TMVB_LIB_VARIANT=<name> loads libtmvb_hip_<name>.so, the same sources built with extra -D
# flags, so that a compile-time variant can be timed against the shipped library inside ONE gpurun call (boxes differ by up to 8 %)
_VARIANT = os.environ.get("TMVB_LIB_VARIANT", "")
SOURCES = ["tmvb_core.hip", "tmvb_comm.hip", "tmvb_lda.hip", "tmvb_flda.hip", "tmvb_ctm.hip", "tmvb_ctpf.hip", "tmvb_ctpf_recs.hip", "tmvb_topics.hip"]

OK, EINVAL, ESHAPE, ECORPUS, ENOMEM, EHIP, ENONFINITE, ENODEVICE, ERCCL = range(9)


class TopicModelError(Exception):
    """src/modelutils.jl:1-5"""


class CorpusError(Exception):
    """src/Corpus.jl:85-89"""


class DocumentError(Exception):
    """src/Corpus.jl:30-34"""


class EngineError(RuntimeError):
    """HIP runtime / device failure inside libtmvb_hip.so"""


class CorpusInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("M", "V", "U", "nnz", "nR", "sum_counts", "sum_ratings", "max_doc_len",
                                         "max_readers", "n_empty_docs", "n_docs_with_duplicate_terms",
                                         "n_docs_with_duplicate_readers")]


def _sources():
    out = []
    for s in SOURCES:
        p = os.path.join(_HERE, "csrc", s)
        if os.path.exists(p):
            out.append(p)
    return out


LAST_BUILD = {"mode": "not run", "compiled": [], "linked": False}


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of libtmvb_hip.so (in-tree).  LAST_BUILD records what this call did: "reused" (the
    library is newer than every source), or "compiled" with the translation units that went through hipcc."""
    global LAST_BUILD
    srcs = _sources()
    deps = srcs + [os.path.join(_HERE, "csrc", h) for h in sorted(os.listdir(os.path.join(_HERE, "csrc"))) if h.endswith(".h")] \
        + [os.path.join(_ROOT, "include", "tmvb.h")]
    if not force and os.path.exists(LIB_PATH):
        if all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
            LAST_BUILD = {"mode": "reused", "compiled": [], "linked": False}
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-pass-failed",
             "-I", os.path.join(_ROOT, "include"), "-I", os.path.join(_HERE, "csrc")]
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs, compiled = [], [], []
    for src in srcs:                      # one hipcc per translation unit, in parallel
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdrs = [d for d in deps if d.endswith(".h")]
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs):
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if os.path.basename(src) == "tmvb_ctm.hip":
            cmd.insert(-4, "-save-temps=obj")      # keeps build/tmvb_ctm-hip-amdgcn-amd-amdhsa-gfx950.s for the ISA check below
        if verbose:
            print(" ".join(cmd))
        compiled.append(os.path.basename(src))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    # The batched CTM kernel streams invsigma through SGPRs with hand-placed scalar loads and waits (csrc/tmvb_ctm_batch.h).
    # A compiler-inserted read of a destination that is still in flight, or a write to the reserved registers, would be a
    # silent wrong-answer bug, so the generated ISA is checked at build time and the build fails on a finding.
    isa = os.path.join(objdir, "tmvb_ctm-hip-amdgcn-amd-amdhsa-gfx950.s")
    checker = os.path.join(_ROOT, "tools", "check_smem_inflight.py")
    if os.path.exists(isa) and os.path.exists(checker):
        res = subprocess.run([sys.executable, checker, isa, "ctm_estep"], capture_output=True, text=True)
        if res.returncode != 0:
            for o in objs:
                if os.path.basename(o).startswith("tmvb_ctm.hip") and os.path.exists(o):
                    os.remove(o)
            raise EngineError("ISA check of the batched CTM kernel failed:\n" + res.stdout[-2000:])
        if verbose:
            print(res.stdout.strip().splitlines()[-1])
    # The four-waves-per-item CTM kernel (csrc/tmvb_ctm_quad.h) issues the row gathers of its token loop as inline asm with hand-counted
    # s_waitcnt vmcnt: the same kind of check for VGPRs -- nothing may touch a destination register between its load and the wait that lands it.
    vchecker = os.path.join(_ROOT, "tools", "check_vmem_inflight.py")
    if os.path.exists(isa) and os.path.exists(vchecker):
        res = subprocess.run([sys.executable, vchecker, isa, "ctm_estep_quad_kernel"], capture_output=True, text=True)
        if res.returncode != 0:
            for o in objs:
                if os.path.basename(o).startswith("tmvb_ctm.hip") and os.path.exists(o):
                    os.remove(o)
            raise EngineError("ISA check of the hand-scheduled token loop of ctm_estep_quad_kernel failed (build with -DTMVB_CTM_QASM=0 for the "
                              "compiler-managed loop, or take the instantiation out of cq_asm_loop):\n" + res.stdout[-2000:])
        if verbose:
            print(res.stdout.strip().splitlines()[-1])
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(hipcc)), "lib")
    # RCCL carries the document-sharded all-reduce (tmvb_comm.hip); it is bound with dlopen at the first communicator call
    # (the rpath lets that dlopen find /opt/rocm/lib/librccl.so.1), so single-GPU use needs no RCCL on the machine
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs + ["-ldl", "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # Register / scratch budget of the kernels behind the benched numbers (tools/kernel_resources.py: the code objects' own metadata).  A performance
    # DIAGNOSTIC, not a functional gate (round-5 advice): its ceilings are exact figures of one hipcc release, and an already linked, correct library
    # must stay usable on another.  By default a finding is a warning on stderr; TMVB_STRICT_KERNEL_RESOURCES=1 (this repository's own rounds and CI:
    # tests/test_kernel_resources.py asserts the same on the shipped library) turns it into a build failure.  Skipped without llvm-readelf or a demangler.
    kres = os.path.join(_ROOT, "tools", "kernel_resources.py")
    if os.path.exists(kres) and not _VARIANT:
        LAST_BUILD_KRES = _kernel_resource_check(kres, verbose)
    else:
        LAST_BUILD_KRES = "skipped"
    LAST_BUILD = {"mode": "compiled", "compiled": compiled, "linked": True, "kernel_resources": LAST_BUILD_KRES}
    return LIB_PATH


def _kernel_resource_check(kres: str, verbose: bool = False) -> str:
    """Runs tools/kernel_resources.py --check on the linked library.  Returns "ok", "skipped (...)" or "warned"; raises only under
    TMVB_STRICT_KERNEL_RESOURCES=1."""
    import shutil
    readelf = os.environ.get("LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")
    filt = any(os.path.exists(e) for e in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt")) or shutil.which("c++filt")
    if not (os.path.exists(readelf) or shutil.which(readelf)) or not filt:
        return "skipped (no llvm-readelf / c++filt on this machine)"
    try:
        res = subprocess.run([sys.executable, kres, "--check", LIB_PATH], capture_output=True, text=True)
    except OSError as e:
        return f"skipped ({e})"
    if res.returncode == 0:
        if verbose:
            print(res.stdout.strip().splitlines()[-1])
        return "ok"
    msg = "kernel resource check (tools/kernel_resources.py --check):\n" + (res.stdout[-2000:] + res.stderr[-2000:]).strip()
    if os.environ.get("TMVB_STRICT_KERNEL_RESOURCES", "") not in ("", "0"):
        raise EngineError(msg)
    print("WARNING: " + msg + "\n(performance diagnostic only: the library is built and usable; TMVB_STRICT_KERNEL_RESOURCES=1 makes this fatal)", file=sys.stderr)
    return "warned"


_lib = None

P_i64 = C.POINTER(C.c_int64)
P_i32 = C.POINTER(C.c_int32)
P_dbl = C.POINTER(C.c_double)
VP = C.c_void_p


def exported_symbols():
    """Every function include/tmvb.h declares (used by the loader test)."""
    import re
    hdr = open(os.path.join(_ROOT, "include", "tmvb.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tmvb_[a-z0-9_]+)\s*\(", hdr)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950). "
                              "The HIP engine has no CPU fallback.")
        # eight hardware queues for the process's HIP streams (tmvb_core.hip: tmvb_env_defaults); only effective before HIP initialises
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        path = LIB_PATH
        if _VARIANT:
            path = os.path.join(_HERE, f"libtmvb_hip_{_VARIANT}.so")
            if not os.path.exists(path):
                raise EngineError(f"TMVB_LIB_VARIANT={_VARIANT}: {path} is missing (tools/build_variant.sh)")
        L = C.CDLL(path)
        L.tmvb_last_error.restype = C.c_char_p
        L.tmvb_abi_version.restype = C.c_int
        L.tmvb_device_count.restype = C.c_int
        _lib = L
    return _lib


def check(rc: int):
    if rc == OK:
        return
    msg = lib().tmvb_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError(msg)              # the reference throws ArgumentError
    if rc in (ESHAPE, ENONFINITE):
        raise TopicModelError(msg)
    if rc == ECORPUS:
        raise CorpusError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise EngineError(msg)
