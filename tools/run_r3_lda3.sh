#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
( time python -m pytest tests/test_lda_gpu.py tests/test_dist_gpu.py tests/test_comm_gpu.py tests/test_stats_classes_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
for v in 1 0 1 0; do TMVB_LDA_SPLIT_LAST=$v python bench.py --no-cpu-baseline --no-plateau --no-other-configs --no-cold 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('split_last=$v', '%.1f it/s  %.4f ms  estep %.4f' % (r['value'], r['ms_per_step'], r['roofline']['estep_ms']))" >> $O/split.txt; done
cat $O/split.txt; tail -3 $O/tests.log
