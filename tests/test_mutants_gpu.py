"""
NEGATIVE CONTROLS of the parity suite (round-4 review: "there is no deliberately wrong variant that must fail").

Five mutant libraries (three of round 5, two of round 6 for the SURVEY.md section 8(f) rows) -- the shipped objects with ONE translation unit recompiled under a -DTMVB_MUTANT_* flag (csrc/tmvb_internal.h, tools/build_mutants.sh) -- each
wrong in one operator, in a way a careless port of the reference would be:
  mut_lda_eps    epsilon dropped from LDA's update_phi! / update_gamma!            src/LDA.jl:152, :145
  mut_ctpf_bet   `log bet` where update_xi! needs `log vav`                         src/CTPF.jl:336 -- the reference's own OpenCL path has this bug, src/gpuCTPF.jl:624
  mut_ctm_mu     update_sigma! centred on the NEW mu (update_mu! first)            src/CTM.jl:207-208, quirk Q2
  mut_flda_eps   log(beta) for log(beta + eps) in the filtered models' table        src/fLDA.jl:184, :191 (@boink)
  mut_fctm_order fCTM's sweep in CTM's order (update_vsq! before update_lambda!)    src/fCTM.jl:239-240 against src/CTM.jl:198-199
For each, a NAMED parity test is run in a fresh pytest process with TMVB_LIB_VARIANT=<mutant> and must FAIL with an assertion of that test (not an import
error, not a crash), while the shipped library passes the same test in the ordinary suite.  A parity suite that stays green on these would have no teeth.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "topicmodelsvb.jl_amd")

MUTANTS = {
    "mut_lda_eps": ("tmvb_lda.hip", "-DTMVB_MUTANT_LDA_NO_EPS=1",
                    ["tests/test_lda_gpu.py::test_epsilon_keeps_phi_defined_where_a_beta_column_is_zero"]),
    "mut_ctpf_bet": ("tmvb_ctpf.hip", "-DTMVB_MUTANT_CTPF_LOG_BET=1",
                     ["tests/test_ctpf_gpu.py::test_teacher_forced_step[syn_k12]", "tests/test_ctpf_gpu.py::test_teacher_forced_step[syn_k50]"]),
    "mut_ctm_mu": ("tmvb_ctm.hip", "-DTMVB_MUTANT_CTM_SIGMA_NEW_MU=1",
                   ["tests/test_ctm_gpu.py::test_sigma_uses_previous_mu_quirk_q2", "tests/test_ctm_gpu.py::test_teacher_forced_step[syn_k12]"]),
    "mut_flda_eps": ("tmvb_flda.hip", "-DTMVB_MUTANT_FLDA_NO_EPS=1",
                     ["tests/test_flda_gpu.py::test_epsilon_keeps_phi_defined_where_a_beta_column_is_zero"]),
    "mut_fctm_order": ("tmvb_ctm.hip", "-DTMVB_MUTANT_FCTM_VSQ_FIRST=1",
                       ["tests/test_fctm_gpu.py::test_teacher_forced_fixed_sweeps[5]", "tests/test_fctm_gpu.py::test_teacher_forced_step[syn_k12]"]),
}


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):                       # the order of the shell's `cat csrc/*` (C locale: plain byte order of these ASCII names)
        h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "tmvb.h"), "rb").read())
    return h.hexdigest()[:16]


def _ensure(name):
    """the mutant library, built here if the tree does not carry a current one (tools/build_variant.sh links it from the shipped objects)"""
    unit, flag, _ = MUTANTS[name]
    lib = os.path.join(PKG, f"libtmvb_hip_{name}.so")
    stamp = os.path.join(PKG, f"libtmvb_hip_{name}.stamp")      # written by tools/build_variant.sh: content hash of the sources + the flags
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().split() == [_source_hash(), unit, flag]:
        return lib
    if not os.path.exists(os.path.join(PKG, "build", unit + ".o")):
        import tmvb_amd
        tmvb_amd.pkg.build(force=True)
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), name, unit, flag], check=True, timeout=1500, capture_output=True)
    return lib


def _run(test_id, variant):
    env = dict(os.environ, TMVB_LIB_VARIANT=variant)
    env.pop("TMVB_TOL_RECORD", None)                      # the record mode never fails a comparison
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", test_id], capture_output=True, text=True,
                          env=env, cwd=ROOT, timeout=900)


@pytest.mark.parametrize("name", sorted(MUTANTS))
def test_the_parity_suite_fails_on_the_mutant(name):
    _ensure(name)
    for test_id in MUTANTS[name][2]:
        r = _run(test_id, name)
        out = r.stdout[-3000:]
        assert r.returncode == 1, f"{test_id} did NOT fail on {name} (rc {r.returncode}):\n{out}\n{r.stderr[-1500:]}"
        assert "AssertionError" in out or "assert " in out, out        # a comparison failed -- not a loader error or a crash
        assert "1 failed" in out and "error" not in out.splitlines()[-1], out


def test_the_same_tests_pass_on_the_shipped_library():
    """the control of the control: the subprocess harness itself is sound (one test of each family, shipped library)"""
    for test_id in ("tests/test_lda_gpu.py::test_epsilon_keeps_phi_defined_where_a_beta_column_is_zero",
                    "tests/test_ctpf_gpu.py::test_teacher_forced_step[syn_k12]", "tests/test_ctm_gpu.py::test_sigma_uses_previous_mu_quirk_q2",
                    "tests/test_flda_gpu.py::test_epsilon_keeps_phi_defined_where_a_beta_column_is_zero", "tests/test_fctm_gpu.py::test_teacher_forced_fixed_sweeps[5]"):
        r = _run(test_id, "")
        assert r.returncode == 0, (test_id, r.stdout[-2000:], r.stderr[-1000:])
