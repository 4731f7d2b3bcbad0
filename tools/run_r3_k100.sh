#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
( time python -m pytest tests/test_lda_gpu.py tests/test_random_shapes_gpu.py tests/test_dist_gpu.py tests/test_comm_gpu.py tests/test_stats_classes_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
python tools/model_bench.py lda100 > $O/lda100.json 2> $O/lda100.err
TMVB_LDA_GRID=0 python - > $O/lda100_nogrid.json 2> $O/lda100_nogrid.err <<'PY'
import sys, json
sys.path.insert(0, 'tools')
import model_bench
print(json.dumps(model_bench.lda100(cpu=False)))
PY
tail -3 $O/tests.log
