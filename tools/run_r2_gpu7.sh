set -x
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_ctm_gpu.py tests/test_ctpf_gpu.py tests/test_dist_gpu.py tests/test_lda_gpu.py tests/test_comm_gpu.py -q -m gpu > gpurun_out/t_subset.txt 2>&1; echo "rc=$?" >> gpurun_out/t_subset.txt
python bench.py --docs 16100 --no-cpu-baseline --no-plateau > gpurun_out/small_bench2.json 2> gpurun_out/small_bench2.err
python bench.py --docs 32200 --no-cpu-baseline --no-plateau --no-cold > gpurun_out/small_bench3.json 2> gpurun_out/small_bench3.err
python tools/model_bench.py > gpurun_out/models_bench_r2.jsonl 2> gpurun_out/models_bench_r2.err
