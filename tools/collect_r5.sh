#!/bin/bash
# Runs on the GPU box (gpurun): every measurement DESIGN.md cites for round 5, into gpurun_out/final/.
# usage: bash tools/collect_r5.sh        then, in the build container: bash tools/install_profiles.sh 5
# (round 5: the production library has no run-time switches; kernel-against-kernel comparisons come from
#  tools/sweep_gemm_shapes.py, which names kernels through tssep_gemm_f32_on)
set -u
O=gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-exact-f32 --no-headline-parity"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/bench_recurrence.py 8 32 64 128 256 512 768 1024 1536 2048 3072 > $O/recurrence_microbench.jsonl 2>/dev/null
python tools/stress_recurrence.py 400 2>/dev/null | tail -1 > $O/recurrence_stress.json
python tools/bench_onchip16.py 32 160 768 1536 3072 > $O/onchip16_microbench.jsonl 2>/dev/null
python tools/bench_gemm.py 768 bf16x3 2>/dev/null | grep name > $O/gemm_microbench_bf16x3.jsonl
python tools/bench_gemm.py 768 f32 2>/dev/null | grep name > $O/gemm_microbench_f32.jsonl
python tools/bench_tail.py > $O/tail_microbench.jsonl 2>/dev/null
python tools/grad_parity.py 4 > $O/parity_full_size.jsonl 2>/dev/null
for b in 8 32 64 128 256 384 512 768 1152 1536; do
  python bench.py --batch $b --steps 20 --warmup 4 $B 2>/dev/null | tail -1
done > $O/batch_sweep.jsonl
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
python tools/ab_big_p.py 768 > $O/ab_gemm_big_p.jsonl 2>/dev/null
python tools/ab_tn_p320.py 768 > $O/ab_wgrad_tn_p320.jsonl 2>/dev/null
python tools/step_clock.py 768 > $O/step_clock.json 2>/dev/null
python tools/bench_maskhead.py 64 256 768 > $O/maskhead_microbench.jsonl 2>/dev/null
python tools/abl_onchip16.py 32 768 3072 2>/dev/null | grep lib > $O/onchip16_fwd_variants.jsonl
# raw counter tables are large: keep what install_profiles.sh reads
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O | head -80
