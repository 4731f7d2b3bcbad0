#!/bin/bash
# LDA K=50: piece count / cut fractions (the statistics passes run one after the other from the end of the first piece on)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3fracs; mkdir -p $O; cd $R
run() {
  python bench.py --no-cpu-baseline --no-plateau --no-other-configs --no-cold > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[0])
print("pieces=${TMVB_LDA_PIECES:-3} fracs=${TMVB_LDA_PIECE_FRACS:-default}", round(d["value"], 1), "estep_ms", round(d["roofline"].get("estep_ms", 0), 4))
PY
}
run
for F in "0.45,0.8" "0.5,0.8" "0.4,0.75" "0.48,0.82" "0.5,0.85" "0.42,0.8" "0.46,0.76" "0.55,0.85"; do TMVB_LDA_PIECES=3 TMVB_LDA_PIECE_FRACS=$F run; done
run
