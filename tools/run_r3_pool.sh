#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pool; mkdir -p $O; cd $R
for cfg in "1 0" "1 1" "0 0"; do
  set -- $cfg
  TMVB_STREAM_POOL=$1 TMVB_CTPF_LONG_PRIO=$2 python bench.py --no-cpu-baseline > $O/b$1$2.json 2> $O/b$1$2.err
  python - <<PY
import json
d = json.loads(open("$O/b$1$2.json").read().strip().splitlines()[0])
print("pool=$1 ctpf_long_prio=$2 bench", round(d["value"], 1), {k: round(v["value"], 1) for k, v in d["other_configs"].items()})
PY
done
TMVB_STREAM_POOL=1 TMVB_CTPF_LONG_PRIO=0 python tools/model_bench.py ctpf lda100 ctm ctpf 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('standalone pool=1 prio=0', d['metric'][:40], round(d['value'], 1))"
