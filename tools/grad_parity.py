"""Full-size parity of masks and parameter gradients against the CPU oracle for every combination
of GEMM arithmetic and recurrence kernel (GPU box).  usage: python tools/grad_parity.py [batch]

LogMAE's gradient is sign(estimate - target) / (N mean): where the oracle's residual is within one ulp
of zero, a 1e-7 difference in the estimate flips that sample's sign and moves every parameter gradient
by ~1e-4 relative -- a property of the loss, not of the kernels.  Each row therefore counts those flips
(`logmae_sign_flips`); batch 2 of this seed holds such a sample ([0, 0, 290], residual -7.5e-9), batch 4
does not, which is why the committed table is taken at batch 4."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tssep_amd import hip_ops as H
from oracle import model as omodel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
        H.WGRAD_PRODUCTS = int(products)
        model.zero_grad(set_to_none=True)
        np.random.seed(5)
        out = model(ex)
        model.review(ex, out)["loss"].backward()
        torch.cuda.synchronize(); H.check_cluster_errors()
        errs = {k: float((v.grad.cpu() - p["mask_estimator." + k].grad).abs().max()
                         / (p["mask_estimator." + k].grad.abs().max() + 1e-12))
                for k, v in model.mask_estimator.named_parameters()}
        worst = max(errs, key=errs.get)
        resid_o, resid_h = o["time_estimate"].detach() - x[2], out.time_estimate.detach().cpu() - x[2]
        gs = sorted(errs.values())
        print(json.dumps(dict(gemm=gemm, recurrence=rec, wgrad_products=int(products), batch=B,
                              max_abs_mask_err=float((out.mask.detach().cpu() - o["mask"]).abs().max()),
                              logmae_sign_flips=int((torch.sign(resid_o) != torch.sign(resid_h)).sum()),
                              median_rel_grad_err=gs[len(gs) // 2],
                              max_rel_grad_err=errs[worst], worst_param=worst)), flush=True)
