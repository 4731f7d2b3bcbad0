#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R; export TMPDIR=/tmp
cat > /tmp/ctpf_b.py <<PY
import sys, json
sys.path.insert(0, '$R/tools'); sys.path.insert(0, '$R')
import model_bench
r = model_bench.ctpf(cpu=False)
print(json.dumps({k: r[k] for k in ('value','ms_per_step','estep_ms','ms_per_checked_step','cold_start')}))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python /tmp/ctpf_b.py > $O/prof.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/prof_summary.txt 2>&1
python - $(find $O/prof -name "*.db" | head -1) > $O/timeline.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select d.start, d.end, s.display_name, d.queue_id, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# print 60 dispatches from the steady window (skip the checked iterations at the end)
idx = [i for i, r in enumerate(rows) if 'ctpf_estep_grid_any' in r[2]]
i0 = idx[len(idx)//2]
t0 = rows[i0][0]
for st, en, name, q, g in rows[i0:i0+40]:
    print(f"{(st-t0)/1e3:9.1f} {(en-t0)/1e3:9.1f} {(en-st)/1e3:8.1f} q{q} g{g:<8d} {name[:80]}")
print("---- a checked iteration")
idx = [i for i, r in enumerate(rows) if 'elbo' in r[2]]
i0 = idx[len(idx)//2] - 12
t0 = rows[i0][0]
for st, en, name, q, g in rows[i0:i0+40]:
    print(f"{(st-t0)/1e3:9.1f} {(en-t0)/1e3:9.1f} {(en-st)/1e3:8.1f} q{q} g{g:<8d} {name[:80]}")
PY
find $O -name "*.db" -delete
head -30 $O/prof_summary.txt
