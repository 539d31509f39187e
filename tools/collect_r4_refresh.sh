#!/bin/bash
# Runs on the GPU box (gpurun): the subset of tools/collect_r4.sh that the last changes of round 4 move (weight-gradient
# block order, graphed step with the weight layouts on a branch), into gpurun_out/final/ -- then
# `bash tools/install_profiles.sh 4` in the build container.
set -u
O=gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-exact-f32"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for b in 8 32 64 128 256 384 512 768 1152 1536; do
  python bench.py --batch $b --steps 20 --warmup 4 $B 2>/dev/null | tail -1
done > $O/batch_sweep.jsonl
python bench.py --workload cfg4 --steps 40 --warmup 5 > $O/bench_cfg4.json 2>$O/bench_cfg4.err
python bench.py --workload cfg4 --steps 40 --warmup 5 --graph off --no-cpu-baseline > $O/bench_cfg4_nograph.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 2 $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o s -- python3 bench.py --workload cfg4 --graph off --steps 10 --warmup 3 $B > $O/stats_cfg4.log 2>&1
python tools/gemm_in_step.py 768 > $O/gemm_in_step_b768.jsonl 2>/dev/null
for r in 1 2; do for v in 0 1; do
  python tools/ab_prepare_derived.py $v --workload cfg4 --steps 60 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(dict(prepare_derived=$v, ms_per_step=d['ms_per_step'], median=d['ms_per_step_median'], steps=60)))"
done; done > $O/ab_prepare_derived.jsonl
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
