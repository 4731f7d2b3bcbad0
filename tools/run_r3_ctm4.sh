#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm4; mkdir -p $O; cd $R
TMVB_CTM_BATCH_MAX_LEN=2048 TMVB_CTM_PROF=1 TMVB_CTM_WAVE_LOG=$O/wl.bin ITERS=40 python tools/ctm_probe.py > $O/probe.txt 2>&1
tail -3 $O/probe.txt
