#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q8; mkdir -p $O; cd $R
for Q in 4 6 8 16; do for S in 1 0; do
  GPU_MAX_HW_QUEUES=$Q TMVB_LDA_SIDE_STREAM=$S python bench.py --no-cpu-baseline --no-plateau --no-other-configs > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[0])
print("queues=$Q side=aux[$S] bench", round(d["value"], 1), "estep_ms", round(d["roofline"].get("estep_ms", 0), 4), "cold", round(d.get("cold_start", {}).get("value", 0), 1))
PY
done; done
