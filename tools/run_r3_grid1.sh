#!/bin/bash
mkdir -p gpurun_out/r3b
cd $GRAFT_REPO_ROOT
./tools/probes/grid_probe > gpurun_out/r3b/grid_probe.txt 2>&1
( time python -m pytest tests/test_lda_gpu.py tests/test_random_shapes_gpu.py tests/test_dist_gpu.py tests/test_predict_gpu.py -m gpu -x -q ) > gpurun_out/r3b/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b/tests.log
( python bench.py --no-cpu-baseline --no-plateau --no-other-configs ) > gpurun_out/r3b/bench_grid.json 2> gpurun_out/r3b/bench_grid.err
( TMVB_LDA_GRID=0 python bench.py --no-cpu-baseline --no-plateau --no-other-configs ) > gpurun_out/r3b/bench_nogrid.json 2> gpurun_out/r3b/bench_nogrid.err
tail -5 gpurun_out/r3b/grid_probe.txt; tail -5 gpurun_out/r3b/tests.log
