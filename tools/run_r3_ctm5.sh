#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm5; mkdir -p $O; cd $R
export TMVB_CTM_BATCH_MAX_LEN=2048
for P in 1 0; do
  TMVB_CTM_PERSISTENT=$P ITERS=40 python tools/ctm_probe.py > $O/probe_P$P.txt 2>&1
  echo "persistent=$P"; tail -4 $O/probe_P$P.txt | grep "^iter"
done
TMVB_CTM_PROF=1 TMVB_CTM_WAVE_LOG=$O/wl.bin ITERS=40 python tools/ctm_probe.py > $O/probe_prof.txt 2>&1
tail -3 $O/probe_prof.txt
( time python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
