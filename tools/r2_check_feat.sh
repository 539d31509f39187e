#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q -k "features or feature or known_answer or model_end or full_size or toy" 2>&1 | tail -3
python - <<'PY'
import torch, time, sys
sys.path.insert(0,'.')
from tssep_amd import hip_ops as h
from oracle import features as of
B,T=768,253
X=torch.randn(B,T,513,dtype=torch.complex64,device='cuda')
fb,dct=of.mfcc_tables(1024); fb,dct=fb.cuda(),dct.cuda()
for _ in range(3): out,_=h.feat_fwd(X,fb,dct,40)
torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
for _ in range(10): out,_=h.feat_fwd(X,fb,dct,40)
e.record(); torch.cuda.synchronize(); print('feat_fwd ms', s.elapsed_time(e)/10)
PY
