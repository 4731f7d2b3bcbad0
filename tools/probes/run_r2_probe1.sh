set -x
cd $GRAFT_REPO_ROOT
./tools/probes/issue_probe > gpurun_out/issue_probe.txt 2>&1
for f in 0 1 2 3; do echo "== TMVB_DEBUG_FLAGS=$f"; TMVB_DEBUG_FLAGS=$f M=32000 ITERS=4 python tools/ctm_probe.py; done > gpurun_out/ctm_split.txt 2>&1
