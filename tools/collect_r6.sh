#!/bin/bash
# Runs on the GPU box (gpurun): the ONE profile refresh of round 6, into gpurun_out/final/ (no kernel of
# the training step changed this round -- the refresh documents HEAD; the round's experiments have their own records:
# profiles/r6_l2s_probe.jsonl, r6_gemm_noa_probe.jsonl, r6_maskhead_kinner_rejected.jsonl).
# usage: bash tools/collect_r6.sh        then, in the build container: bash tools/install_profiles.sh 6
set -u
O=gpurun_out/final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-exact-f32 --no-headline-parity"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --workload cfg4 --steps 40 --warmup 5 > $O/bench_cfg4.json 2>$O/bench_cfg4.err
python bench.py --workload cfg5 --steps 5 --warmup 2 > $O/bench_cfg5.json 2>$O/bench_cfg5.err
python bench.py --gemm f32 > $O/bench_f32.json 2>$O/bench_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 2 $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o s -- python3 bench.py --workload cfg4 --graph off --steps 10 --warmup 3 $B > $O/stats_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 bench.py --workload cfg5 --steps 3 --warmup 1 $B > $O/stats_cfg5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o q -- python3 bench.py --steps 2 --warmup 1 $B > $O/pmc_sq.log 2>&1
python tools/bench_maskhead.py 64 256 768 > $O/maskhead_microbench.jsonl 2>/dev/null
python tools/step_clock.py 768 > $O/step_clock.json 2>/dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O | head -60
