#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "blstm or onchip" 2>&1 | tail -2
for lib in libtssep_hip_base.so libtssep_hip.so libtssep_hip_base.so libtssep_hip.so; do
  echo $lib; TSSEP_HIP_LIB=$GRAFT_REPO_ROOT/tssep_amd/$lib python tools/bench_recurrence.py 8 32 256 768 3072 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['N'], d['onchip_fwd_ms'], d['onchip_bwd_ms'])"
done
