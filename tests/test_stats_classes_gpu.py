"""
The statistics pass cuts frequent ids' posting lists at document-id class boundaries and orders the chunks class by class
(tmvb_build_inv_index in csrc/tmvb_core.hip: L2 locality on the 8 XCDs).  The default only does so for corpora of more than
32 768 documents, which no oracle-sized parity test reaches; here the same parity tests (LDA, CTM, CTPF: term and reader
indices) are run again in a child process that forces 8 classes and cuts every id with two or more postings
(TMVB_STATS_CLASSES / TMVB_CLASS_MIN_POSTINGS are read once per process).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suites_with_forced_document_classes():
    env = dict(os.environ, TMVB_STATS_CLASSES="8", TMVB_CLASS_MIN_POSTINGS="2")
    sel = ["tests/test_lda_gpu.py::test_teacher_forced_fixed_sweeps", "tests/test_lda_gpu.py::test_free_running_train_vs_golden",
           "tests/test_lda_gpu.py::test_long_documents_stream_through_the_tile", "tests/test_lda_gpu.py::test_train_equals_stepwise_pipelined",
           "tests/test_ctm_gpu.py::test_teacher_forced_step", "tests/test_ctpf_gpu.py::test_teacher_forced_step",
           "tests/test_ctpf_gpu.py::test_free_running_train_vs_golden", "tests/test_flda_gpu.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
