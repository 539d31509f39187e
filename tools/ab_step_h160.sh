B="--no-cpu-baseline --no-exact-f32"
for w in 0 1 0 1 0 1; do env TSSEP_GEMM_TN_H160=$w python bench.py --steps 15 --warmup 3 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(switch='TSSEP_GEMM_TN_H160', value=$w, frames_per_s=d['value'], ms_per_step=d['ms_per_step'], ms_per_step_median=d['ms_per_step_median'], gemm_tflops=d['roofline']['achieved'])))"; done
