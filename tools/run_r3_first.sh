#!/bin/bash
# round 3, first GPU call: the new tests, the comm tests on the lazily bound RCCL, CTM tolerance measurements, bench.py with other_configs
mkdir -p gpurun_out/r3a
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_ctm_gpu.py tests/test_ctpf_gpu.py tests/test_comm_gpu.py tests/test_multigpu_rccl.py tests/test_stats_classes_gpu.py -m gpu -x -q ) > gpurun_out/r3a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a/tests.log
( time python tests/measure_tolerances.py ) > gpurun_out/r3a/tolerances.log 2>&1
( time python bench.py ) > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/tests.log
