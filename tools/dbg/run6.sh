export TMPDIR=/tmp; mkdir -p gpurun_out/r4g
python tools/sweep_gemm_shapes.py --batch 768 --reps 5 --out gpurun_out/r4g/gemm_shape_sweep.jsonl 2> gpurun_out/r4g/sweep.err; tail -1 gpurun_out/r4g/sweep.err
python bench.py --steps 20 --warmup 4 --no-exact-f32 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python tools/sweep_gemm_shapes.py --batch 8 --reps 10 --quick --out gpurun_out/r4g/gemm_shape_sweep_b8.jsonl 2> gpurun_out/r4g/sweep_b8.err; tail -1 gpurun_out/r4g/sweep_b8.err
python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', d['value'], d['ms_per_step'])"
