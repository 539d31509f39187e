#!/bin/bash
# Runs on the GPU box (gpurun): every measurement DESIGN.md cites, into gpurun_out/final/.
# usage: bash tools/collect_profiles.sh [tag]
set -u
O=gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/bench_recurrence.py 8 32 64 128 192 256 512 768 800 1024 1536 2048 > $O/recurrence_microbench.jsonl 2>/dev/null
TSSEP_GEMM_PRECISION=bf16x3 python tools/bench_gemm.py 192 2>/dev/null | grep name > $O/gemm_microbench_bf16x3.jsonl
TSSEP_GEMM_PRECISION=f32 python tools/bench_gemm.py 192 2>/dev/null | grep name > $O/gemm_microbench_f32.jsonl
python tools/bench_maskhead.py > $O/maskhead_microbench.txt 2>/dev/null
for b in 8 32 64 128 160 192 256 384 512 768 1152 1536; do
  python bench.py --batch $b --steps 20 --warmup 4 --no-cpu-baseline --no-exact-f32 2>/dev/null | tail -1
done > $O/batch_sweep.jsonl
python bench.py > $O/bench_default.json 2>$O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -20
ls -la $O
# MFMA pipe and wave-state counters of the same command (own pass: SQ counters only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o q -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 > $O/pmc_sq.log 2>&1
# eval-time MVDR beamformer (TorchBF): microbench with the CPU oracle beside it + kernel stats
python tools/bench_mvdr.py > $O/mvdr_microbench.jsonl 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mvdr_stats -o m -- python3 tools/bench_mvdr.py --no-cpu --iters 10 > $O/mvdr_stats.log 2>&1
python tools/gemm_in_step.py 768 > $O/gemm_in_step_b768.jsonl 2>/dev/null
python tools/bench_input_pipeline.py 768 > $O/input_pipeline.jsonl 2>/dev/null; python tools/bench_input_pipeline.py 384 >> $O/input_pipeline.jsonl 2>/dev/null
python tools/step_clock.py 768 > $O/step_clock.json 2>/dev/null
# experimental GEMM probes (pre-split planes + asynchronous copies) and their ablations
python tools/probe_presplit.py > $O/gemm_presplit_probe.jsonl 2>/dev/null
python tools/probe_presplit_ablation.py > $O/gemm_presplit_ablation.jsonl 2>/dev/null
python tools/probe_presplit_tn.py > $O/gemm_presplit_tn_probe.jsonl 2>/dev/null
