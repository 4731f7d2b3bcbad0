#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time python -m pytest tests/test_ctpf_gpu.py tests/test_ctpf_recs_gpu.py tests/test_comm_gpu.py tests/test_predict_gpu.py tests/test_random_shapes_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
cat > /tmp/ctpf_b.py <<'PY'
import sys, json
sys.path.insert(0, 'tools')
import model_bench
r = model_bench.ctpf(cpu=False)
print(json.dumps({k: r[k] for k in ('value','ms_per_step','estep_ms','ms_per_checked_step','cold_start')}))
PY
python /tmp/ctpf_b.py > $O/ctpf_grid_any.json 2>$O/err1
TMVB_CTPF_GRID_ANY=0 python /tmp/ctpf_b.py > $O/ctpf_grid_sep.json 2>$O/err2
TMVB_CTPF_GRID=0 python /tmp/ctpf_b.py > $O/ctpf_nogrid.json 2>$O/err3
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python /tmp/ctpf_b.py > $O/prof.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/prof_summary.txt 2>&1
find $O -name "*.db" -delete
tail -3 $O/tests.log; cat $O/ctpf_*.json
