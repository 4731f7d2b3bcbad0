#!/bin/bash
# does the HIP runtime torch brings (bundled ROCm 7.0 libamdhip64) cost the launch-bound configurations something against /opt/rocm's?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3torchhip; mkdir -p $O; cd $R
for T in 0 1 2; do
python - <<PY > $O/t$T.txt 2>&1
import sys, os, json
sys.path.insert(0, "tools")
if $T >= 1:
    import torch
    if $T == 2:
        torch.cuda.init(); torch.zeros(1, device="cuda")
import model_bench
l = model_bench.lda100(cpu=False)
print("torch=$T lda100", round(l["value"], 1), round(l["ms_per_step"], 4))
l = model_bench.ctpf(cpu=False)
print("torch=$T ctpf", round(l["value"], 1), round(l["ms_per_step"], 4))
print([m.split()[-1] for m in open("/proc/self/maps") if "libamdhip64" in m or "librccl" in m or "libhsa-runtime" in m][::8])
PY
tail -3 $O/t$T.txt
done
