"""How long does the HOST need to enqueue one step vs. how long the GPU needs to run it?"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tssep_amd import hip_ops as H
from tssep_amd.distributed import GradBucket
H.GEMM_PRECISION = "bf16x3"
for B in [int(a) for a in sys.argv[1:]] or [8, 64, 256]:
    model = bench.build_model().cuda()
    bucket = GradBucket(model.parameters())
    obs, aux, tgt = bench.synth_batch(B, 4, 64000, 0)
    ex0 = dict(observation=torch.as_tensor(obs).cuda(), auxInput=torch.as_tensor(aux).cuda(),
               speaker_reverberation_early_ch0=torch.as_tensor(tgt).cuda(), reference_channel=0, dataset=["b"] * B)
    def step():
        ex = dict(ex0); bucket.zero()
        out = model(ex); s = model.review(ex, out); s["loss"].backward(); bucket.all_reduce()
    for _ in range(3): step()
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(5):
        t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    print(json.dumps(dict(B=B, host_enqueue_ms=round(1e3 * min(hs), 2), total_ms=round(1e3 * min(ts), 2))), flush=True)
    del model, bucket
