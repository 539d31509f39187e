#!/bin/bash
# round-2 first GPU pass: parity suite, then the three workloads of bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err; tail -c 600 $O/cfg4.json
python bench.py --workload cfg5 --steps 4 --warmup 2 > $O/cfg5.json 2> $O/cfg5.err; tail -c 1500 $O/cfg5.json; tail -3 $O/cfg5.err
python bench.py --steps 20 --warmup 4 > $O/default.json 2> $O/default.err; tail -c 3000 $O/default.json; tail -3 $O/default.err
python bench.py --gpus 2 > $O/gpus2.json 2> $O/gpus2.err; echo "gpus2 rc=$?"; tail -2 $O/gpus2.err
