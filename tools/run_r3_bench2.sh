#!/bin/bash
# the bench.py line and the model lines after the CTM / CTPF changes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3bench2; mkdir -p $O; cd $R
( time timeout 1500 python tools/model_bench.py lda100 ctm ctpf flda fctm ctm100 ) > $O/models.jsonl 2> $O/models.err; echo "rc=$?" >> $O/models.err
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
python - <<PY
import json
for l in open("$O/models.jsonl"):
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); print(d["metric"][:50], round(d["value"], 1), round(d["ms_per_step"], 3), d.get("ms_per_checked_step"))
d = json.loads(open("$O/bench.json").read().strip().splitlines()[0])
print("bench", round(d["value"], 1), d["roofline"]["frac"], {k: round(v["value"], 1) for k, v in d["other_configs"].items()})
PY
