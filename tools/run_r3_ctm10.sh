#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm10; mkdir -p $O; cd $R
for K in 50 40 60 30; do
  K=$K TMVB_CTM_FORCE_GENERIC=1 ITERS=40 timeout 600 python tools/ctm_probe.py > $O/probe_K${K}_gen.txt 2>&1
  echo "K=$K generic-cg"; tail -2 $O/probe_K${K}_gen.txt | cut -c1-200
  K=$K ITERS=40 timeout 600 python tools/ctm_probe.py > $O/probe_K${K}_def.txt 2>&1
  echo "K=$K default"; tail -2 $O/probe_K${K}_def.txt | cut -c1-200
done
