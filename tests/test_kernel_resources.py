"""
tools/kernel_resources.py reads the register / scratch / LDS figures of every kernel out of libtmvb_hip.so's embedded gfx950 code objects (no GPU
needed) and build() fails when a kernel behind a benched number uses scratch beyond its recorded ceiling (round-4 review: the CTM lane kernel's
428 B per lane had been invisible).  Here: the shipped library passes, the parser finds what hipcc wrote, and the check has teeth.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

LIB = os.path.join(ROOT, "topicmodelsvb.jl_amd", "libtmvb_hip.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(kr.READELF), reason="needs the built library and llvm-readelf")


@pytest.fixture(scope="module")
def rows():
    return kr.kernels(LIB)


def test_every_benched_kernel_is_found_and_within_its_scratch_ceiling(rows):
    assert len(rows) > 300
    assert kr.check(rows) == []
    ctm = [r for r in rows if r["demangled"].startswith("ctm_estep_batch_kernel<52, false, false>")]
    assert len(ctm) == 1 and ctm[0]["vgpr"] == 512 and 0 < ctm[0]["scratch"] <= 388       # the one benched kernel that spills: recorded, bounded
    lda = [r for r in rows if r["demangled"].startswith("lda_estep_grid_kernel<13,")]
    assert lda and all(r["scratch"] == 0 and r["vgpr_spills"] == 0 and r["vgpr"] <= 256 for r in lda)   # two waves per SIMD, no scratch


def test_the_check_fails_on_a_spilling_or_missing_kernel(rows):
    worse = [dict(r) for r in rows]
    for r in worse:
        if r["demangled"].startswith("termstats_recompute_kernel<13,"):
            r["scratch"] = 16
    assert any("termstats_recompute_kernel<13" in b for b in kr.check(worse))
    gone = [r for r in rows if not r["demangled"].startswith("ctpf_mstep_kernel")]
    assert any("not found" in b for b in kr.check(gone))
