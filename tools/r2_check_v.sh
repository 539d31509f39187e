#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2v; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python tools/stress_recurrence.py 300 2>/dev/null | tail -1
python tools/bench_recurrence.py 8 32 768 3072 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['N'], d['onchip_fwd_ms'], d['onchip_bwd_ms'])"
python bench.py --steps 20 --warmup 4 --no-exact-f32 > $O/default.json 2>/dev/null
python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline > $O/cfg4.json 2>/dev/null
python bench.py --workload cfg5 --steps 4 --warmup 2 --no-cpu-baseline > $O/cfg5.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2v/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['ms_per_step_median'], (d.get('cpu_baseline') or {}).get('parity_vs_hip'))
PY
