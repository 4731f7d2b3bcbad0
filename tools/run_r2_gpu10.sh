set -x
cd $GRAFT_REPO_ROOT
for w in 1 4; do for kb in 16 8; do echo "== TMVB_CTM_WPS=$w TMVB_CTM_MAX_TILE_KB=$kb"; TMVB_CTM_WPS=$w TMVB_CTM_MAX_TILE_KB=$kb M=32000 ITERS=4 python tools/ctm_probe.py; done; done > gpurun_out/ctm_wps.txt 2>&1
