#!/bin/bash
# K > 52 CTM kernel: waves per workgroup (register budget of the instantiation x LDS window)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm13; mkdir -p $O; cd $R
for cfg in "53 16 16" "53 16 14" "53 12 12" "64 16 16" "64 16 14" "64 12 12" "100 12 10" "100 8 8" "128 12 6" "128 8 6" "80 12 12" "80 8 8"; do
  set -- $cfg
  K=$1 TMVB_CTM_CG_MAXW=$2 TMVB_CTM_CG_WAVES=$3 ITERS=30 timeout 600 python tools/ctm_probe.py > $O/probe_K$1_M$2_W$3.txt 2>&1
  echo "K=$1 maxw=$2 waves=$3"; tail -2 $O/probe_K$1_M$2_W$3.txt | head -1 | cut -c1-50
done
