#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 3 2 3 2; do
  TSSEP_WGRAD_PRODUCTS=$w TSSEP_GEMM_PRECISION=bf16x3 python tools/bench_gemm.py 768 2>/dev/null | grep -E "wgrad" | python -c "
import sys,json
print('products=$w', ' '.join('%s:%.1f' % (json.loads(l)['name'][6:18], json.loads(l)['tflops']) for l in sys.stdin))"
done
for w in 3 2 3 2; do
  TSSEP_WGRAD_PRODUCTS=$w python bench.py --steps 15 --warmup 3 --no-exact-f32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['cpu_baseline']['parity_vs_hip']; print('products=$w', d['value'], d['ms_per_step'], 'grad err max', p['max_rel_grad_err'], p['worst_gradient'], 'median', p['median_rel_grad_err'])"
done
