#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R
( time python -m pytest tests/test_lda_gpu.py tests/test_ctpf_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -30 $O/tests.log
