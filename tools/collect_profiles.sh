#!/bin/bash
# Runs on the GPU box (gpurun): every measurement DESIGN.md cites for this round, into gpurun_out/final/.
# usage: bash tools/collect_profiles.sh
set -u
O=gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-exact-f32"
python tools/bench_recurrence.py 8 32 64 128 256 512 768 1024 1536 2048 3072 > $O/recurrence_microbench.jsonl 2>/dev/null
python tools/stress_recurrence.py 400 2>/dev/null | tail -1 > $O/recurrence_stress.json
python tools/bench_onchip16.py 32 160 768 1536 3072 > $O/onchip16_microbench.jsonl 2>/dev/null
TSSEP_GEMM_PRECISION=bf16x3 python tools/bench_gemm.py 768 2>/dev/null | grep name > $O/gemm_microbench_bf16x3.jsonl
TSSEP_GEMM_PRECISION=f32 python tools/bench_gemm.py 768 2>/dev/null | grep name > $O/gemm_microbench_f32.jsonl
# round 3: the streaming and the big-tile kernel against the tiled kernels they replace, shape by shape, alternating
TSSEP_GEMM_BIG=0 python tools/bench_gemm_ab.py TSSEP_GEMM_STREAM 1 0 > $O/ab_gemm_stream_shapes.jsonl 2>/dev/null
TSSEP_GEMM_STREAM=1 python tools/bench_gemm_ab.py TSSEP_GEMM_BIG 1 0 > $O/ab_gemm_big_shapes.jsonl 2>/dev/null
python tools/bench_gemm_ab.py TSSEP_GEMM_TN_BIG 1 0 768 tn > $O/ab_wgrad_big_shapes.jsonl 2>/dev/null
python tools/bench_gemm_ab.py TSSEP_GEMM_TN_XC 1 0 768 tn > $O/ab_wgrad_xc_shapes.jsonl 2>/dev/null
python tools/bench_gemm_ab.py TSSEP_GEMM_TN_W160 1 0 768 shift > $O/ab_wgrad_w160_shapes.jsonl 2>/dev/null
python tools/sweep_wgrad_splits.py TSSEP_GEMM_TN_W160 1 0 > $O/wgrad_w160_split_sweep.jsonl 2>/dev/null
python tools/sweep_wgrad_splits.py TSSEP_GEMM_TN_XC 1 0 > $O/wgrad_xc_split_sweep.jsonl 2>/dev/null
python tools/bench_tail.py > $O/tail_microbench.jsonl 2>/dev/null
python tools/grad_parity.py 4 > $O/parity_full_size.jsonl 2>/dev/null
for b in 8 32 64 128 256 384 512 768 1152 1536; do
  python bench.py --batch $b --steps 20 --warmup 4 $B 2>/dev/null | tail -1
done > $O/batch_sweep.jsonl
# round-3 kernel changes, each alternating off / on in this one job (the switches are read per call)
for var in TSSEP_GEMM_STREAM TSSEP_GEMM_BIG TSSEP_GEMM_NT_W160 TSSEP_GEMM_REMAP_WIDE TSSEP_GEMM_TN_BIG TSSEP_GEMM_TN_W160 TSSEP_GEMM_TN_H160 TSSEP_GEMM_TN_XC TSSEP_ONCHIP16 TSSEP_ONCHIP16_BWD; do
  for w in 0 1 0 1; do env $var=$w python bench.py --steps 15 --warmup 3 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(switch='$var', value=$w, frames_per_s=d['value'], ms_per_step=d['ms_per_step'], ms_per_step_median=d['ms_per_step_median'], gemm_tflops=d['roofline']['achieved'], mask_head_frac=d['roofline_mask_head']['frac'])))"; done
done > $O/ab_gemm_kernels.jsonl
for w in 2 4 2 4; do TSSEP_GEMM_TN_BIG=0 TSSEP_GEMM_TN_TALL=$w python bench.py --steps 15 --warmup 3 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(dict(switch='TSSEP_GEMM_TN_TALL', value=$w, frames_per_s=d['value'], ms_per_step=d['ms_per_step'], gemm_tflops=d['roofline']['achieved'])))"; done > $O/ab_wgrad_tile.jsonl
for w in 3 2 3 2; do TSSEP_WGRAD_PRODUCTS=$w python bench.py --steps 15 --warmup 3 --no-exact-f32 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['cpu_baseline']['parity_vs_hip']; print(json.dumps(dict(wgrad_products=$w, frames_per_s=d['value'], ms_per_step=d['ms_per_step'], max_rel_grad_err=p['max_rel_grad_err'], median_rel_grad_err=p['median_rel_grad_err'], worst=p['worst_gradient'])))"; done > $O/ab_wgrad_products.jsonl
python bench.py --workload cfg4 --steps 40 --warmup 5 > $O/bench_cfg4.json 2>$O/bench_cfg4.err
python bench.py --workload cfg4 --steps 40 --warmup 5 --graph off --no-cpu-baseline > $O/bench_cfg4_nograph.json 2>/dev/null
python bench.py --workload cfg5 --steps 5 --warmup 2 > $O/bench_cfg5.json 2>$O/bench_cfg5.err
python bench.py --gemm f32 > $O/bench_f32.json 2>$O/bench_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 2 $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -o s -- python3 bench.py --gemm f32 --steps 6 --warmup 2 $B > $O/stats_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o s -- python3 bench.py --workload cfg4 --graph off --steps 10 --warmup 3 $B > $O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 bench.py --workload cfg5 --steps 3 --warmup 1 $B > $O/stats_cfg5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmc_write.log 2>&1
# MFMA pipe and wave-state counters of the same command (own pass: SQ counters only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o q -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmc_sq.log 2>&1
bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq/summary.jsonl $O/sq_wave_states.jsonl
python tools/gemm_in_step.py 768 > $O/gemm_in_step_b768.jsonl 2>/dev/null
python tools/step_clock.py 768 > $O/step_clock.json 2>/dev/null
ls -la $O | head -60
