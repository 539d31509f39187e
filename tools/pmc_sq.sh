#!/bin/bash
# SQ wave-state counters per kernel of the default step (GPU box): where the waves' cycles go
# usage: bash tools/pmc_sq.sh [bench args]
O=gpurun_out/pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES \
  --output-format csv -d $O/raw -o q -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-headline-parity "$@" > $O/run.log 2>&1
python - <<PY
import csv, glob, collections, json
f = glob.glob("$O/raw/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$O/summary.jsonl", "w") as out:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]:
        wc = v.get("SQ_WAVE_CYCLES", 1)
        row = dict(kernel=k, waves=v.get("SQ_WAVES"), **{n[3:].lower() + "_frac": round(x / wc, 3) for n, x in v.items() if n not in ("SQ_WAVE_CYCLES", "SQ_WAVES")})
        print(json.dumps(row)); out.write(json.dumps(row) + "\n")
PY
