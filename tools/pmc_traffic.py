"""Build profiles/*_traffic_pmc.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
the default bench command.  usage: python tools/pmc_traffic.py fetch.csv write.csv batch gemm"""
import collections
import csv
import json
import re
import sys

fetch_csv, write_csv, batch, gemm = sys.argv[1:5]
workload = sys.argv[5] if len(sys.argv) > 5 else "cfg3"


def per_kernel(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        acc[name].append(float(r["Counter_Value"]))
    return acc


F, W = per_kernel(fetch_csv), per_kernel(write_csv)
kernels = {}
for k in sorted(set(F) | set(W)):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    kernels[k] = {}
    if k in F:
        kernels[k]["FETCH_SIZE"] = dict(launches=len(F[k]), avg_kb_per_launch=round(sum(F[k]) / len(F[k]), 1))
    if k in W:
        kernels[k]["WRITE_SIZE"] = dict(launches=len(W[k]), avg_kb_per_launch=round(sum(W[k]) / len(W[k]), 1))


def avg(d, names):
    v = [x for n in names for x in d.get(n, [])]
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)


gem = [k for k in kernels if k.startswith("gemm_bf16x3")]
gf, n = avg(F, gem)
gw, _ = avg(W, gem)
tail = ["istft_kernel<true>", "rfft_frames_kernel<true>"]     # mask head + iSTFT, iSTFT adjoint + mask-head backward
mf, nm = avg(F, tail)
mw, _ = avg(W, tail)
rec = {k: dict(fetch_kb=round(avg(F, [k])[0], 1), write_kb=round(avg(W, [k])[0], 1), launches=avg(W, [k])[1])
       for k in kernels if k.startswith("blstm_onchip")}
B = int(batch)
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv "
               "-- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32",
    "config": {"batch_per_gpu": B, "gemm": gemm, "workload": workload},
    "units": "KB per launch as reported (x1024 = bytes). gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
             "under-reports wide (16 B/lane) streaming loads by 2x; 8 B/lane accesses are uncalibrated. Raw "
             "values are stored; 'bytes_raw' = (FETCH+WRITE)*1024, 'bytes_fetch_x2' applies the 2x correction "
             "to the fetch side.",
    "dominant": {
        "gemm_bf16x3": {"launches": n, "fetch_kb": round(gf, 1), "write_kb": round(gw, 1),
                        "bytes_raw": int((gf + gw) * 1024), "bytes_fetch_x2": int((2 * gf + gw) * 1024)},
        "maskhead_fwd+bwd": {"launches": nm, "bytes_raw": int((mf + mw) * 1024),
                             "bytes_fetch_x2": int((2 * mf + mw) * 1024),
                             "algorithmic_bytes": B * 253 * (16 * 4 * 513 + 8 * 513),
                             "note": "the FUSED tail kernels (mask head + iSTFT; iSTFT adjoint + mask-head backward), "
                                     "per launch; algorithmic_bytes = the unfused mask head's (SURVEY 8d)"},
    },
    "recurrence": rec,
    "kernels": kernels,
}
print(json.dumps(out, indent=1))
