#!/bin/bash
# CTM K=50: sweep of the length limit of the lane-per-document kernel (documents beyond it run the wave-per-document kernel beside it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ctm2; mkdir -p $O; cd $R
for L in default 2048 400 320 280 256 224 192 160 128; do
  if [ $L = default ]; then unset TMVB_CTM_BATCH_MAX_LEN; else export TMVB_CTM_BATCH_MAX_LEN=$L; fi
  python - <<PY > $O/len_$L.json 2> $O/len_$L.err
import sys, json
sys.path.insert(0, "tools")
import model_bench
l = model_bench.ctm(cpu=False)
print(json.dumps({"limit": "$L", "it_s": l["value"], "ms": l["ms_per_step"], "estep_ms": l["estep_ms"], "cold": l["cold_start"]["value"]}))
PY
  tail -1 $O/len_$L.json
done
