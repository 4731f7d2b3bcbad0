#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm7; mkdir -p $O; cd $R
for P in 1 0 1 0; do
  TMVB_CTM_PERSISTENT=$P ITERS=40 python tools/ctm_probe.py > $O/probe_P$P.txt 2>&1
  echo "persistent=$P"; tail -3 $O/probe_P$P.txt | grep "^iter"
done
for P in 1 0; do
TMVB_CTM_PERSISTENT=$P TMVB_CTM_PROF=1 TMVB_CTM_WAVE_LOG=$O/wl_P$P.bin ITERS=40 python tools/ctm_probe.py > $O/probe_prof_P$P.txt 2>&1
tail -3 $O/probe_prof_P$P.txt
python tools/ctm_wave_log.py $O/wl_P$P.bin > $O/wave_log_P$P.txt; cat $O/wave_log_P$P.txt
done
( time python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_predict_gpu.py tests/test_comm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
tail -3 $O/tests.log
