#!/bin/bash
# Copies what tools/collect_r6.sh wrote under gpurun_out/final/ into profiles/ (tracked), names r<round>_*.
# usage: bash tools/install_profiles.sh [round] [batch] [gemm]      then re-run `python bench.py` on the GPU box
# for the default line with the PMC fields filled (profiles/r<round>_bench_default.json).
set -e
R=${1:-6}; B=${2:-768}; G=${3:-bf16x3}; F=gpurun_out/final; P=profiles/r${R}
python tools/pmc_traffic.py $F/pmc_fetch/f_counter_collection.csv $F/pmc_write/w_counter_collection.csv $B $G cfg3 > ${P}_traffic_pmc.json
python tools/pmc_mfma.py $F/pmc_sq/q_counter_collection.csv $B $G cfg3 > ${P}_mfma_pmc.json
for f in recurrence_microbench.jsonl recurrence_stress.json gemm_microbench_bf16x3.jsonl gemm_microbench_bf16x3_wide0.jsonl \
         gemm_microbench_bf16x3_wgrad2.jsonl gemm_microbench_f32.jsonl tail_microbench.jsonl parity_full_size.jsonl \
         batch_sweep.jsonl ab_gemm_wide.jsonl ab_wgrad_products.jsonl bench_cfg4.json bench_cfg4_nograph.json bench_cfg5.json \
         bench_f32.json gemm_in_step_b768.jsonl step_clock.json ab_fusions.jsonl splitk_sweep.jsonl \
         ab_gemm_stream_shapes.jsonl ab_gemm_big_shapes.jsonl ab_wgrad_big_shapes.jsonl ab_gemm_kernels.jsonl ab_wgrad_tile.jsonl \
         ab_wgrad_xc_shapes.jsonl ab_wgrad_w160_shapes.jsonl wgrad_w160_split_sweep.jsonl wgrad_xc_split_sweep.jsonl sq_wave_states.jsonl onchip16_microbench.jsonl \
         bench_default.json store_flavour_probe_16384.json store_flavour_probe_81920.json ab_gemm_big_p.jsonl ab_wgrad_tn_p320.jsonl ab_prepare_derived.jsonl maskhead_microbench.jsonl onchip16_fwd_variants.jsonl; do
  [ -s $F/$f ] && cp $F/$f ${P}_$f
done
python - <<PY
import csv, os
def table(src, dst, header):
    if not os.path.exists(src):
        return None
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w') as f:
        f.write("# " + header + "\n")
        f.write(f"{'Name':78s} {'Calls':>6s} {'TotalNs':>12s} {'AvgNs':>11s} {'Pct':>6s} {'MinNs':>9s} {'MaxNs':>9s}\n")
        for r in rows:
            n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
            f.write(f"{n[:78]:78s} {r['Calls']:>6s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):11.0f} {r['Percentage']:>6s} {r['MinNs']:>9s} {r['MaxNs']:>9s}\n")
    return rows
cmd = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py %s --no-cpu-baseline --no-exact-f32 --no-headline-parity"
rows = table('$F/stats/s_kernel_stats.csv', '${P}_kernel_stats_default_b${B}_${G}.txt', cmd % "--steps 6 --warmup 2" + "   (default config: batch $B, $G; 3 setup + 2 warm-up + 6 timed = 11 steps in total)")
table('$F/stats_f32/s_kernel_stats.csv', '${P}_kernel_stats_b${B}_f32.txt', cmd % "--gemm f32 --steps 6 --warmup 2" + "   (fp32 GEMMs; 11 steps in total)")
table('$F/stats_cfg4/s_kernel_stats.csv', '${P}_kernel_stats_cfg4_b8.txt', cmd % "--workload cfg4 --graph off --steps 10 --warmup 3" + "   (8 utterances per GPU: the configs[3] shard; 16 steps in total)")
table('$F/stats_cfg5/s_kernel_stats.csv', '${P}_kernel_stats_cfg5_b96.txt', cmd % "--workload cfg5 --steps 3 --warmup 1" + "   (8 speakers x 30 s, batch 96; 7 steps in total)")
tot = sum(int(r['TotalDurationNs']) for r in rows)
g = sum(int(r['TotalDurationNs']) for r in rows if 'gemm' in r['Name'])
rec = sum(int(r['TotalDurationNs']) for r in rows if 'blstm' in r['Name'])
print("busy ms/step", round(tot / 11 / 1e6, 2), "gemm %", round(100 * g / tot, 1), "recurrence %", round(100 * rec / tot, 1))
PY
ls profiles | grep "^r${R}_" | head -40
