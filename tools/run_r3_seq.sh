#!/bin/bash
# a sequence of models in ONE process (what a Julia session does) under the candidate defaults
R=$GRAFT_REPO_ROOT; cd $R
for cfg in "unset unset"; do
  set -- $cfg
  unset GPU_MAX_HW_QUEUES TMVB_LDA_SIDE_STREAM
  python - <<PY 2>/dev/null | tail -1
import sys
sys.path.insert(0, "tools")
import model_bench
out = []
for name in ("ctpf", "lda100", "ctm", "ctpf", "lda100", "ctm", "ctpf", "flda"):
    out.append("%s %.0f" % (name, model_bench.ALL[name](cpu=False)["value"]))
print("queues=$1 side=aux[$2]:", ", ".join(out))
PY
done
