"""Full-size parity of masks and parameter gradients against the CPU oracle for every combination
of GEMM arithmetic and recurrence kernel (GPU box).  usage: python tools/grad_parity.py [batch]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tssep_amd import hip_ops as H
from oracle import model as omodel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_num_threads(16)
model = bench.build_model().cuda()
obs, aux, tgt = bench.synth_batch(B, 4, 64000, 7)
x = [torch.as_tensor(a) for a in (obs, aux, tgt)]
p = {"mask_estimator." + k: v.detach().cpu().clone().requires_grad_()
     for k, v in model.mask_estimator.state_dict().items()}
np.random.seed(5)
o = omodel.forward_loss(p, *x, cfg=dict(odim=513, combination="mul", ts_vad=4, output_resolution="tf"), fast=True)
o["loss"].sum().backward()
ex = dict(observation=x[0].cuda(), auxInput=x[1].cuda(), speaker_reverberation_early_ch0=x[2].cuda(),
          reference_channel=0, dataset=["p"] * B)
for gemm, rec, products in [(g, r, "3") for g in ("f32", "bf16x3") for r in ("stream", "cluster", "onchip")] + \
        [("bf16x3", "onchip", "2")]:
    if True:
        H.GEMM_PRECISION, H.RECURRENCE = gemm, rec
        os.environ["TSSEP_WGRAD_PRODUCTS"] = products
        model.zero_grad(set_to_none=True)
        np.random.seed(5)
        out = model(ex)
        model.review(ex, out)["loss"].backward()
        torch.cuda.synchronize(); H.check_cluster_errors()
        errs = {k: float((v.grad.cpu() - p["mask_estimator." + k].grad).abs().max()
                         / (p["mask_estimator." + k].grad.abs().max() + 1e-12))
                for k, v in model.mask_estimator.named_parameters()}
        worst = max(errs, key=errs.get)
        gs = sorted(errs.values())
        print(json.dumps(dict(gemm=gemm, recurrence=rec, wgrad_products=int(products), batch=B,
                              max_abs_mask_err=float((out.mask.detach().cpu() - o["mask"]).abs().max()),
                              median_rel_grad_err=gs[len(gs) // 2],
                              max_rel_grad_err=errs[worst], worst_param=worst)), flush=True)
