"""Which source line launches each torch (ATen) kernel / copy of one eager training step at 8 utterances: a TorchDispatchMode
that logs every ATen call that is not a view with the innermost frames inside the package.  `python tools/trace_small_kernels.py [batch]`"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tssep_amd.hip_ops as H  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    H.GEMM_PRECISION = "bf16x3"
    model = bench.build_model(bench.K_SPK).to(dev)
    from tssep_amd.train.optimizer import Adam
    opt = Adam(gradient_clipping=10.0, lr=1e-5)
    opt.set_parameters(model.parameters())
    obs, aux, tgt = bench.synth_batch(B, bench.K_SPK, bench.N_SAMPLES, seed=0)
    ex0 = dict(observation=torch.as_tensor(obs).to(dev), auxInput=torch.as_tensor(aux).to(dev),
               speaker_reverberation_early_ch0=torch.as_tensor(tgt).to(dev), reference_channel=0, dataset=["bench"] * B)
    np.random.seed(0)

    def step():
        opt.zero_grad()
        ex = dict(ex0)
        model.review(ex, model(ex))["loss"].backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    seen = collections.OrderedDict()
    skip = ("aten.view", "aten.as_strided", "aten.detach", "aten.empty", "aten.slice", "aten.select", "aten.t.", "aten.transpose",
            "aten.reshape", "aten._unsafe_view", "aten.alias", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.permute",
            "aten.record_stream", "aten.is_pinned", "aten._local_scalar_dense", "aten.narrow", "aten.unbind", "aten.split")

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not name.startswith(skip):
                fr = [f for f in traceback.extract_stack() if "/tssep_amd/" in f.filename or f.filename.endswith("bench.py")]
                where = " <- ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in fr[-2:][::-1]) if fr else "?"
                shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
                key = (name, where, str(shapes))
                seen[key] = seen.get(key, 0) + 1
            return func(*args, **(kwargs or {}))

    with Log():
        step()
    torch.cuda.synchronize()
    for (name, where, shapes), n in seen.items():
        print(f"{n:3d} {name:34s} {shapes:40s} {where}")


if __name__ == "__main__":
    main()
