#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log
python bench.py --steps 20 --warmup 4 --no-exact-f32 > $O/default.json 2> $O/default.err; tail -2 $O/default.err
python bench.py --workload cfg5 --steps 4 --warmup 2 > $O/cfg5.json 2> $O/cfg5.err; tail -2 $O/cfg5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline_mask_head'], (d.get('cpu_baseline') or {}).get('parity_vs_hip'))
    except Exception as e: print(f, 'ERR', e)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 > $O/stats.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2f/stats/s_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total busy ms per step', tot/8/1e6)
for r in rows[:22]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
