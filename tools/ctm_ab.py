"""
A/B of builds of the library (tools/build_variant.sh) on one model_bench.py configuration inside ONE gpurun call: model_bench.py runs once
per variant and round (TMVB_LIB_VARIANT is read at import), the shipped library first, alternating; prints it/s per run.
    python tools/ctm_ab.py <variant>[+<variant>...] [rounds] [model]      e.g.  python tools/ctm_ab.py ch3+ch4 3 ctm
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = sys.argv[3] if len(sys.argv) > 3 else "ctm"
variants = [""] + variant.split("+")
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ, TMVB_LIB_VARIANT=v)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "model_bench.py"), model, "--gpu-only"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(out.stderr[-2000:]); sys.exit(1)
        j = json.loads(line[-1])
        res[v].append(round(j["value"], 2))
        print(f"round {r} variant '{v or 'shipped'}': {j['value']:.2f} {j['unit']}  ({j['ms_per_step']:.4f} ms/step)", flush=True)
print(json.dumps({(v or "shipped"): res[v] for v in variants}))
