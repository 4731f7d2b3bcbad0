#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
( time python -m pytest tests/test_ctpf_gpu.py tests/test_ctpf_recs_gpu.py tests/test_random_shapes_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
cat > /tmp/ctpf_b.py <<PY
import sys, json
sys.path.insert(0, '$R/tools'); sys.path.insert(0, '$R')
import model_bench
r = model_bench.ctpf(cpu=False)
print(json.dumps({k: r[k] for k in ('value','ms_per_step','estep_ms','ms_per_checked_step','cold_start')}))
PY
for p in 1 0; do TMVB_CTPF_GRID_ANY=$p python /tmp/ctpf_b.py > $O/block$p.json 2>$O/err$p; cat $O/block$p.json; done
tail -3 $O/tests.log
