#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3benchvar; mkdir -p $O; cd $R
for flags in "--no-cpu-baseline" "--no-cpu-baseline --no-plateau" "--no-cpu-baseline --no-plateau --no-cold"; do
  python bench.py $flags > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[0])
print("$flags", round(d["value"], 1), {k: round(v["value"], 1) for k, v in d["other_configs"].items()})
PY
done
OMP_WAIT_POLICY=passive python bench.py > $O/b2.json 2> $O/b2.err
python - <<PY
import json
d = json.loads(open("$O/b2.json").read().strip().splitlines()[0])
print("passive", round(d["value"], 1), {k: round(v["value"], 1) for k, v in d["other_configs"].items()})
PY
