set -x
cd $GRAFT_REPO_ROOT
python tools/ctpf_probe.py > gpurun_out/ctpf_probe_r2.txt 2>&1
K=100 python tools/ctpf_probe.py > gpurun_out/ctpf_probe_r2_k100.txt 2>&1
timeout 1200 python -m pytest tests/test_ctpf_gpu.py tests/test_random_shapes_gpu.py tests/test_comm_gpu.py tests/test_dist_gpu.py -q -m gpu > gpurun_out/t_subset.txt 2>&1; echo "rc=$?" >> gpurun_out/t_subset.txt
