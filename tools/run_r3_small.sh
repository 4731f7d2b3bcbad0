#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
( time python -m pytest tests/test_lda_gpu.py tests/test_flda_gpu.py tests/test_dist_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
python bench.py --no-cpu-baseline --no-plateau --no-other-configs --no-cold --docs 16100 --steps 50 > $O/bench_16100.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-plateau --no-other-configs --no-cold --docs 16100 --steps 50 > $O/prof.log 2>&1
cd $R
python tools/prof_timeline.py $(find $O/prof -name "*.db" | head -1) > $O/timeline.txt 2>&1
find $O -name "*.db" -delete
cat $O/timeline.txt; tail -3 $O/tests.log; python -c "
import json; r=json.loads(open('$O/bench_16100.json').readline()); print(r['value'], r['ms_per_step'], r['roofline']['estep_ms'])"
