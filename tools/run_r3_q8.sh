#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q8; mkdir -p $O; cd $R
for cfg in "8 1" "8 0" "default 2" "8 1"; do
  set -- $cfg
  if [ $1 = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$1; fi
  TMVB_LDA_SIDE_STREAM=$2 python bench.py --no-cpu-baseline > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[0])
print("queues=$1 side=aux[$2] bench", round(d["value"], 1), "cold", round(d.get("cold_start", {}).get("value", 0), 1), {k: round(v["value"], 1) for k, v in d["other_configs"].items()})
PY
done
for m in flda fctm ctm100; do GPU_MAX_HW_QUEUES=8 TMVB_LDA_SIDE_STREAM=1 python tools/model_bench.py --gpu-only $m 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('q8', d['metric'][:40], round(d['value'], 1))"; done
GPU_MAX_HW_QUEUES=8 TMVB_LDA_SIDE_STREAM=1 python bench.py --docs 16100 --steps 50 --no-cpu-baseline --no-plateau --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[0]); print('q8 16100 docs ms', d['ms_per_step'])"
