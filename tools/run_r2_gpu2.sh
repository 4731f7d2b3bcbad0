# round-2 GPU session 2: communicator tests, the previously green suites touched by the ABI change, bench line
set -x
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_comm_gpu.py -x -q -m gpu > gpurun_out/t_comm.txt 2>&1; echo "comm rc=$?" >> gpurun_out/t_comm.txt
timeout 900 python -m pytest tests/test_ctpf_gpu.py tests/test_dist_gpu.py tests/test_random_shapes_gpu.py -x -q -m gpu > gpurun_out/t_other.txt 2>&1; echo "other rc=$?" >> gpurun_out/t_other.txt
timeout 600 python bench.py > gpurun_out/bench_r2_a.json 2> gpurun_out/bench_r2_a.err; echo "bench rc=$?" >> gpurun_out/bench_r2_a.err
TMVB_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --burnin 5 --no-cpu-baseline --plateau-cap 100 > gpurun_out/bench_r2_w2.json 2> gpurun_out/bench_r2_w2.err; echo "w2 rc=$?" >> gpurun_out/bench_r2_w2.err
