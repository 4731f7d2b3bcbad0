set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ctm_gpu.py -q -m gpu -x > gpurun_out/t_ctm_batch.txt 2>&1; echo "rc=$?" >> gpurun_out/t_ctm_batch.txt
timeout 600 python tools/ctm_probe.py > gpurun_out/ctm_probe_batch.txt 2>&1
TMVB_CTM_BATCH=0 timeout 600 python tools/ctm_probe.py > gpurun_out/ctm_probe_wave.txt 2>&1
