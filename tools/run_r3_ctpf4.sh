#!/bin/bash
# CTPF: update_elbo! with the one-pass psi / lgamma, one synchronisation; tests, checked-iteration time, timeline of a checked iteration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctpf4; mkdir -p $O; cd $R
( time python -m pytest tests/test_ctpf_gpu.py tests/test_ctpf_recs_gpu.py tests/test_random_shapes_gpu.py tests/test_special_gpu.py tests/test_comm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
cat > /tmp/ctpf_b.py <<PY
import sys, json
sys.path.insert(0, '$R/tools'); sys.path.insert(0, '$R')
import model_bench
r = model_bench.ctpf(cpu=False)
print(json.dumps({k: r[k] for k in ('value','ms_per_step','estep_ms','ms_per_checked_step','cold_start')}))
PY
python /tmp/ctpf_b.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
cat > /tmp/ctpf_c.py <<PY
import sys
sys.path.insert(0, '$R')
import numpy as np, tmvb_amd
tm = tmvb_amd.pkg
pc = tm.syn_citeu(); gm = tm.gpuCTPF(pc, 50)
for it in range(12):
    gm.estep(); gm.reduce_docs(); gm.mstep(); e = gm.update_elbo()
print(e)
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_c -- python /tmp/ctpf_c.py > $O/trace_run.txt 2>&1
db=$(find /tmp/prof_c -name "*.db" | head -1)
python $R/tools/prof_window.py $db ctpf_elbo_final 1 > $O/timeline_checked.txt 2>&1
cat $O/timeline_checked.txt | cut -c1-140
