# alternating A/B of the step on one environment switch: bash tools/ab_step_env.sh VAR [pairs]
B="--no-cpu-baseline --no-exact-f32"
var=$1; n=${2:-3}
for i in $(seq $n); do for w in 0 1; do env $var=$w python bench.py --steps 15 --warmup 3 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(switch='$var', value=$w, frames_per_s=d['value'], ms_per_step=d['ms_per_step'], ms_per_step_median=d['ms_per_step_median'], gemm_tflops=d['roofline']['achieved'])))"; done; done
