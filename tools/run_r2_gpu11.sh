set -x
cd $GRAFT_REPO_ROOT
for f in 0 8; do echo "== TMVB_DEBUG_FLAGS=$f"; TMVB_DEBUG_FLAGS=$f M=32000 ITERS=5 python tools/ctm_probe.py; done > gpurun_out/ctm_mv.txt 2>&1
timeout 1200 python -m pytest tests/test_ctm_gpu.py tests/test_random_shapes_gpu.py tests/test_predict_gpu.py tests/test_dist_gpu.py tests/test_comm_gpu.py -q -m gpu > gpurun_out/t_subset.txt 2>&1; echo "rc=$?" >> gpurun_out/t_subset.txt
