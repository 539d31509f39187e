"""Summarise a rocprofv3 --pmc pass of SQ counters (tools/collect_profiles.sh) per kernel:
MFMA-pipe busy fraction and wave-state split.  usage: python tools/pmc_mfma.py counter_collection.csv batch gemm"""
import collections
import csv
import json
import re
import sys

path, batch, gemm = sys.argv[1:4]
workload = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
N_SIMD, N_CU = 1024, 256
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                  "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT --output-format csv "
                  "-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32",
       "config": {"batch_per_gpu": int(batch), "gemm": gemm, "workload": workload},
       "definitions": "per launch averages, summed over the chip by the profiler.  mfma_busy_frac = "
                      "(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CU_CYCLES / 256 CUs): the share of the "
                      "kernel's CU-busy time during which a SIMD's matrix pipe is executing (32 cycles per "
                      "v_mfma_f32_32x32x16_bf16).  wave states are shares of SQ_WAVE_CYCLES: wait_any = parked on "
                      "s_waitcnt / barrier, wait_inst = issue stalls (MFMA dependencies, busy pipes), active = issuing.",
       "kernels": {}}
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CU_CYCLES", [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    wc = m["SQ_WAVE_CYCLES"]
    out["kernels"][k] = dict(
        launches=len(cs["SQ_BUSY_CU_CYCLES"]),
        mfma_busy_frac=round((m["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD) / (m["SQ_BUSY_CU_CYCLES"] / N_CU), 4),
        issued_bf16_mfma_flops=m["SQ_VALU_MFMA_BUSY_CYCLES"] / 32 * 32768,
        wait_any=round(m["SQ_WAIT_ANY"] / wc, 3), wait_inst=round(m["SQ_WAIT_INST_ANY"] / wc, 3),
        active=round(m["SQ_ACTIVE_INST_ANY"] / wc, 3),
        lds_bank_conflict_cycles=m["SQ_LDS_BANK_CONFLICT"], raw={c: round(v) for c, v in m.items()})
print(json.dumps(out, indent=1))
