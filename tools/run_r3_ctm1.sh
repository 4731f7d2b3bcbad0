#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
( time python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_predict_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
