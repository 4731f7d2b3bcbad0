#!/bin/bash
# CTM K = 51 ... 60 on the lane-per-document kernel (KP = 60 instantiation) against the wave-per-document kernel; K = 64 / 100 for the record
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm8; mkdir -p $O; cd $R
( time python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
for cfg in "60 1" "60 0" "56 1" "56 0" "64 1" "100 1"; do
  set -- $cfg
  K=$1 TMVB_CTM_BATCH60=$2 ITERS=12 python tools/ctm_probe.py > $O/probe_K$1_B$2.txt 2>&1
  echo "K=$1 batch60=$2"; tail -2 $O/probe_K$1_B$2.txt | cut -c1-200
done
