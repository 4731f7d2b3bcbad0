#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm12; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_predict_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -2 $O/tests.log
for K in 53 60 64 100 128; do
  K=$K ITERS=30 timeout 600 python tools/ctm_probe.py > $O/probe_K$K.txt 2>&1
  echo "K=$K"; tail -2 $O/probe_K$K.txt | head -1 | cut -c1-150
done
for cfg in "64 8 32" "64 6 128" "64 4 128" "100 8 32" "100 6 128" "100 4 128"; do
  set -- $cfg
  K=$1 TMVB_CTM_CG_WAVES=$2 TMVB_CTM_CG_TILE=$3 ITERS=30 timeout 600 python tools/ctm_probe.py > $O/probe_K$1_W$2_T$3.txt 2>&1
  echo "K=$1 waves=$2 tile<=$3"; tail -2 $O/probe_K$1_W$2_T$3.txt | head -1 | cut -c1-80
done
