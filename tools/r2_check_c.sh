#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
python tools/bench_recurrence.py 8 32 128 768 1536 3072 > $O/recurrence.jsonl 2>$O/recurrence.err; cut -c1-400 $O/recurrence.jsonl; tail -2 $O/recurrence.err
python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err
python bench.py --steps 20 --warmup 4 --no-exact-f32 > $O/default.json 2> $O/default.err; tail -2 $O/default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['ms_per_step_median'], (d.get('cpu_baseline') or {}).get('parity_vs_hip'))
    except Exception as e: print(f, 'ERR', e)
PY
