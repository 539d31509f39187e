export TMPDIR=/tmp; mkdir -p gpurun_out/r4d
python tools/dbg/gemm_mismatch.py 2>&1 | grep -v amdgpu.ids | grep -E "stream|!=" 
(python -m pytest tests -m gpu -q --timeout 1500 2>&1 | tail -150) > gpurun_out/r4d/pytest.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/r4d/pytest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['achieved'], 'ref_width', d['reference_width']['value'], d['reference_width']['ms_per_step'])
PY
cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d/probe -o w -- python3 $GRAFT_REPO_ROOT/tools/probe_rewrite.py > $GRAFT_REPO_ROOT/gpurun_out/r4d/probe.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r4d/probe -name "*counter_collection.csv" | head -1); echo $f; python tools/probe_rewrite.py --summarise $f > gpurun_out/r4d/store_flavour_probe.json; cat gpurun_out/r4d/store_flavour_probe.json | head -80
