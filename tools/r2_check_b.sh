#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "data_parallel or graphed or device_loader or toy_exp or direct_grad" > $O/tests_new.log 2>&1; tail -5 $O/tests_new.log
for g in off on; do
python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --graph $g > $O/cfg4_$g.json 2> $O/cfg4_$g.err; tail -c 300 $O/cfg4_$g.json | head -c 300; echo; tail -2 $O/cfg4_$g.err
done
python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32 --graph on > $O/b32_on.json 2> $O/b32_on.err; tail -2 $O/b32_on.err
python bench.py --batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32 --graph off > $O/b32_off.json 2> $O/b32_off.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2b/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['ms_per_step_median'])
    except Exception as e: print(f, 'ERR', e)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o s -- python3 bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --graph off > $O/stats_cfg4.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2b/stats_cfg4/s_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total busy ms per step', tot/13/1e6)
for r in rows[:25]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
