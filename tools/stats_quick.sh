#!/bin/bash
# per-kernel times of the default step (GPU box): bash tools/stats_quick.sh [bench args]
O=gpurun_out/stats_quick; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o s -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 "$@" > $O/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/raw/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("busy ms/step", round(tot / 8 / 1e6, 2))
for r in rows[:24]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:70]:70s} {r['Calls']:>5s} {int(r['TotalDurationNs'])/8e6:8.3f} ms/step  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
