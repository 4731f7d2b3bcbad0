#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm6; mkdir -p $O; cd $R
for T in default 0 150 180 250 300; do
  if [ $T = default ]; then unset TMVB_CTM_TWIN_LEN; else export TMVB_CTM_TWIN_LEN=$T; fi
  ITERS=40 python tools/ctm_probe.py > $O/probe_T$T.txt 2>&1
  echo "twin_len=$T"; tail -3 $O/probe_T$T.txt | grep "^iter\|elbo"
done
unset TMVB_CTM_TWIN_LEN
TMVB_CTM_PROF=1 TMVB_CTM_WAVE_LOG=$O/wl.bin ITERS=40 python tools/ctm_probe.py > $O/probe_prof.txt 2>&1
tail -3 $O/probe_prof.txt
( time python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
