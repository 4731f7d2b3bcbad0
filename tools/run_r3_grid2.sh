#!/bin/bash
# grid kernel, second pass: parity, bench, kernel trace timeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
( time python -m pytest tests/test_lda_gpu.py tests/test_random_shapes_gpu.py tests/test_dist_gpu.py tests/test_predict_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
python bench.py --no-cpu-baseline --no-plateau --no-other-configs > $O/bench_grid.json 2> $O/bench_grid.err
python bench.py --no-cpu-baseline --no-plateau --no-other-configs --docs 16100 --steps 50 > $O/bench_grid_16100.json 2> $O/bench_grid_16100.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-plateau --no-other-configs --no-cold > $O/prof.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/prof_summary.txt 2>&1
python tools/prof_timeline.py $(find $O/prof -name "*.db" | head -1) > $O/prof_timeline.txt 2>&1
find $O -name "*.db" -delete
tail -3 $O/tests.log
