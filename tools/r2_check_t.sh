#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; mkdir -p $O
python -m pytest tests -m gpu -q -k "istft or fused or losses or model_end or known_answer or full_size or graphed" > $O/tests.log 2>&1; tail -3 $O/tests.log
for lib in libtssep_hip.so; do
  echo $lib; TSSEP_HIP_LIB=$GRAFT_REPO_ROOT/tssep_amd/$lib python tools/bench_tail.py 2>/dev/null | cut -c1-330
done
