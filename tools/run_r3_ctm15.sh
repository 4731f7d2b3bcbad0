#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm15; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests/test_ctm_gpu.py tests/test_fctm_gpu.py tests/test_comm_gpu.py -m gpu -x -q ) > $O/tests.log 2>&1
grep -E "passed|failed" $O/tests.log
for S in 1 0 1 0; do
TMVB_CTM_WAVESORT=$S python tools/model_bench.py --gpu-only ctm 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('wavesort=$S', round(d['value'], 1), round(d['ms_per_step'], 3), d['estep_ms'])"
done
