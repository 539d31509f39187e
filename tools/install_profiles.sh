#!/bin/bash
# Copies what tools/collect_profiles.sh wrote under gpurun_out/final/ into profiles/ (tracked).
# usage: bash tools/install_profiles.sh <batch> <gemm>
set -e
F=gpurun_out/final; B=${1:-768}; G=${2:-bf16x3}
python tools/pmc_traffic.py $F/pmc_fetch/f_counter_collection.csv $F/pmc_write/w_counter_collection.csv $B $G > profiles/r1_traffic_pmc.json
python tools/pmc_mfma.py $F/pmc_sq/q_counter_collection.csv $B $G > profiles/r1_mfma_pmc.json
cp $F/recurrence_microbench.jsonl profiles/r1_recurrence_microbench.jsonl
cp $F/gemm_microbench_bf16x3.jsonl profiles/r1_gemm_microbench_bf16x3.jsonl
cp $F/gemm_microbench_f32.jsonl profiles/r1_gemm_microbench_f32.jsonl
cp $F/maskhead_microbench.txt profiles/r1_maskhead_microbench.jsonl
cp $F/batch_sweep.jsonl profiles/r1_batch_sweep_final.jsonl
cp $F/bench_default.json profiles/r1_bench_default.json
[ -f $F/gemm_in_step_b$B.jsonl ] && cp $F/gemm_in_step_b$B.jsonl profiles/r1_gemm_in_step_b$B.jsonl
[ -f $F/mvdr_microbench.jsonl ] && cp $F/mvdr_microbench.jsonl profiles/r1_mvdr_microbench.jsonl
for f in input_pipeline.jsonl gemm_presplit_probe.jsonl gemm_presplit_ablation.jsonl gemm_presplit_tn_probe.jsonl; do [ -s $F/$f ] && cp $F/$f profiles/r1_$f; done
[ -s $F/step_clock.json ] && cp $F/step_clock.json profiles/r1_step_clock.json
python - <<PY
import csv, os
if os.path.exists('$F/mvdr_stats/m_kernel_stats.csv'):
    rows = list(csv.DictReader(open('$F/mvdr_stats/m_kernel_stats.csv')))
    with open('profiles/r1_kernel_stats_mvdr.txt', 'w') as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_mvdr.py --no-cpu --iters 10   (4 configurations, see profiles/r1_mvdr_microbench.jsonl)\n")
        f.write(f"{'Name':72s} {'Calls':>6s} {'TotalNs':>12s} {'AvgNs':>11s} {'Pct':>6s} {'MinNs':>9s} {'MaxNs':>9s}\n")
        for r in rows:
            n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            f.write(f"{n[:72]:72s} {r['Calls']:>6s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):11.0f} {r['Percentage']:>6s} {r['MinNs']:>9s} {r['MaxNs']:>9s}\n")
rows = list(csv.DictReader(open('$F/stats/s_kernel_stats.csv')))
with open('profiles/r1_kernel_stats_default_b${B}_${G}.txt', 'w') as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-f32   (default config: batch $B, $G; 8 steps in total)\n")
    f.write(f"{'Name':72s} {'Calls':>6s} {'TotalNs':>12s} {'AvgNs':>11s} {'Pct':>6s} {'MinNs':>9s} {'MaxNs':>9s}\n")
    for r in rows:
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        f.write(f"{n[:72]:72s} {r['Calls']:>6s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):11.0f} {r['Percentage']:>6s} {r['MinNs']:>9s} {r['MaxNs']:>9s}\n")
tot = sum(int(r['TotalDurationNs']) for r in rows)
g = sum(int(r['TotalDurationNs']) for r in rows if 'gemm' in r['Name'])
rec = sum(int(r['TotalDurationNs']) for r in rows if 'blstm' in r['Name'])
print("busy ms/step", round(tot / 8 / 1e6, 2), "gemm %", round(100 * g / tot, 1), "recurrence %", round(100 * rec / tot, 1))
PY
