export TMPDIR=/tmp; mkdir -p gpurun_out/r4e
for BY in 16384 81920 131072; do
export PROBE_BYTES=$BY PROBE_REPS=100
cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4e/probe$BY -o w -- python3 $GRAFT_REPO_ROOT/tools/probe_rewrite.py > $GRAFT_REPO_ROOT/gpurun_out/r4e/probe$BY.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r4e/probe$BY -name "*counter_collection.csv" | head -1); python tools/probe_rewrite.py --summarise $f > gpurun_out/r4e/store_flavour_probe_$BY.json
python - <<PY
import json
d=json.load(open('gpurun_out/r4e/store_flavour_probe_$BY.json'))
print(d['what'][:120])
for c in d['cases']: print(c['flavour'].ljust(8), c['peer_reads_with'].ljust(13), c['pressure_bytes_between_rewrites'], c['write_size_over_line_bytes'])
PY
done
