#!/bin/bash
# CTM K=50: the iteration's tail (update_sigma! staged under the statistics pass, regrouping moved behind the E-step, scatter / column sums)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm14; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_predict_gpu.py tests/test_comm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log
for S in 1 0; do
TMVB_CTM_SPECULATE=$S python tools/model_bench.py --gpu-only ctm 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('speculate=$S', round(d['value'], 1), round(d['ms_per_step'], 3), d['estep_ms'])"
done
cat > /tmp/ctm_c.py <<PY
import sys
sys.path.insert(0, '$R')
import numpy as np, tmvb_amd
tm = tmvb_amd.pkg
pc = tm.syn_nsf(); gm = tm.gpuCTM(pc, 50)
gm.beta = np.asfortranarray(tm.dirichlet_rows(50, pc.V, seed=7)); gm.beta_old = gm.beta.copy(order="F"); gm.update_buffer()
for it in range(16):
    gm.estep(); gm.reduce_docs(); gm.update_beta(); gm.update_sigma(); gm.update_mu()
gm.synchronize()
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_c -- python /tmp/ctm_c.py > $O/trace_run.txt 2>&1
db=$(find /tmp/prof_c -name "*.db" | head -1)
python $R/tools/prof_window.py $db ctm_estep_batch_kernel 1 > $O/timeline.txt 2>&1
cat $O/timeline.txt | cut -c1-130
