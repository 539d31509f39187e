"""Per-shape GEMM time inside the training step vs standalone (GPU box): shows what the GEMMs lose
to their surroundings (other stream, cold L2, clocks).  usage: python tools/gemm_in_step.py [batch]"""
import json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tssep_amd import hip_ops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
H.GEMM_PRECISION = "bf16x3"
calls = []
orig = H.gemm


def traced(A, lda, Bm, ldb, C, ldc, M, N, K, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    orig(A, lda, Bm, ldb, C, ldc, M, N, K, **kw)
    e.record()
    kind = ("t" if kw.get("a_kmajor") else "n") + ("n" if kw.get("b_kmajor") else "t")
    calls.append(((kind, M, N, K, kw.get("splitk", 1), bool(kw.get("kperiod"))), s, e,
                  (A, lda, Bm, ldb, C, ldc, M, N, K, kw)))


H.gemm = traced
dev = torch.device("cuda", 0)
model = bench.build_model().to(dev)
from tssep_amd.train.optimizer import Adam
opt = Adam(gradient_clipping=10.0, lr=1e-5); opt.set_parameters(model.parameters())
obs, aux, tgt = bench.synth_batch(B, 4, 64000, 0)
ex0 = dict(observation=torch.as_tensor(obs).to(dev), auxInput=torch.as_tensor(aux).to(dev),
           speaker_reverberation_early_ch0=torch.as_tensor(tgt).to(dev), reference_channel=0, dataset=["b"] * B)
np.random.seed(0)
for it in range(4):
    calls.clear()
    opt.zero_grad(); ex = dict(ex0); out = model(ex); model.review(ex, out)["loss"].backward(); opt.step()
torch.cuda.synchronize()
instep = collections.OrderedDict()
for key, s, e, args in calls:
    instep.setdefault(key, []).append((s.elapsed_time(e), args))
H.gemm = orig
tot_in = tot_alone = 0.0
for key, lst in instep.items():
    t_in = sum(t for t, _ in lst) / len(lst)
    A, lda, Bm, ldb, C, ldc, M, N, K, kw = lst[0][1]
    for _ in range(2):
        orig(A, lda, Bm, ldb, C, ldc, M, N, K, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        orig(A, lda, Bm, ldb, C, ldc, M, N, K, **kw)
    e.record(); torch.cuda.synchronize()
    t_al = s.elapsed_time(e) / 5
    tot_in += t_in * len(lst); tot_alone += t_al * len(lst)
    print(json.dumps(dict(kind=key[0], M=key[1], N=key[2], K=key[3], splitk=key[4], shift=key[5], calls=len(lst),
                          in_step_ms=round(t_in, 3), alone_ms=round(t_al, 3),
                          alone_tflops=round(2 * key[1] * key[2] * key[3] / t_al / 1e9, 1))))
print(json.dumps(dict(total_in_step_ms=round(tot_in, 2), total_alone_ms=round(tot_alone, 2))))
