#!/bin/bash
# CTM K > 60: conjugate-gradient form of the generic kernel against round 1's Gauss-Jordan form
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm9; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_predict_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
for cfg in "64 1" "100 1" "128 1" "100 0"; do
  set -- $cfg
  K=$1 TMVB_CTM_GENERIC_CG=$2 ITERS=8 timeout 600 python tools/ctm_probe.py > $O/probe_K$1_CG$2.txt 2>&1
  echo "K=$1 cg=$2"; tail -2 $O/probe_K$1_CG$2.txt | cut -c1-200
done
