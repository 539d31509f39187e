#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm" 2>&1 | tail -2
for w in 1 2 1 2; do
  TSSEP_GEMM_WIDE=$w TSSEP_GEMM_PRECISION=bf16x3 python tools/bench_gemm.py 768 2>/dev/null | grep -E "in\"" | python -c "
import sys,json
print('wide=$w', ' '.join('%s:%.1f' % (json.loads(l)['name'][:12], json.loads(l)['tflops']) for l in sys.stdin))"
done
for w in 1 2 1 2; do
  TSSEP_GEMM_WIDE=$w python bench.py --steps 15 --warmup 3 --no-exact-f32 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
