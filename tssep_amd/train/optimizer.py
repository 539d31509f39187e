"""Optimizer -- drop-in for ``padertorch.train.optimizer.Adam`` as configured by the reference
(tssep/exp/init_cfg_common.yaml:85-94, tssep/train/experiment.py:147-150): global-norm gradient
clipping + Adam, executed as ONE fused HIP launch pair over flat parameter / gradient buffers."""
import ctypes

import torch

from .. import _lib
from ..configurable import Configurable
from ..distributed import GradBucket, flat_offsets


class Adam(Configurable):
    def __init__(self, gradient_clipping=1e10, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad (false in every shipped config)")
        self.gradient_clipping = gradient_clipping
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.amsgrad = amsgrad
        self.bucket = None
        self.step_count = 0
        self.allreduce_events = None      # a list here collects (start, end) HIP events around each all-reduce

    def set_parameters(self, parameters):
        """Flatten parameters (each ``p.data`` becomes a view of one buffer: names, shapes and
        state_dict are unchanged) and attach a flat gradient bucket."""
        params = [p for p in parameters if p.requires_grad]
        dev = params[0].device
        offsets, n = flat_offsets(params)      # the gradient bucket's layout: 256-byte aligned tensors, zero gaps
        self.flat_param = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, off in zip(params, offsets):
            self.flat_param[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + p.numel()].view_as(p)
        self.params = params
        self._offsets = offsets
        from .. import hip_ops
        hip_ops.weights_changed()
        self.bucket = GradBucket(params)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self._ws = torch.empty(int(_lib.lib().tssep_adam_workspace_bytes()) // 4, device=dev,
                               dtype=torch.float32)

    def zero_grad(self):
        self.bucket.zero()

    def step(self):
        """Clip to ``gradient_clipping`` (global L2 norm) and apply one Adam update.  Returns the
        pre-clip gradient norm as a device tensor (no host sync)."""
        from .. import hip_ops
        on_gpu = self.flat_param.is_cuda
        # The guard of the update: this rank's err[0] (a W-stationary recurrence launch gave up on a peer: this step's
        # gradient is garbage) is written into the bucket's guard slot and SUMMED over the ranks with the gradient, so
        # that every rank skips the update when ANY rank's gradient was bad -- the replicas stay identical and nobody
        # trains on the garbage that the all-reduce has already mixed into every peer's bucket.  The host
        # raises at its next flag check (Trainer.agree_on_failure); `step_count` counts attempted updates.
        self.bucket.set_guard(hip_ops._err_flag(self.flat_param.device) if on_gpu else None)
        if self.allreduce_events is not None and self.bucket.flat.is_cuda:
            self.bucket.sync()              # the side stream's weight gradients are not the collective's time
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            self.bucket.all_reduce()
            ev[1].record()
            self.allreduce_events.append(tuple(ev))
        else:
            self.bucket.all_reduce()        # joins the side stream; SUM over ranks when distributed
        self.step_count += 1
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        # (the kernel tests the 32 bits of guard[0] against zero: 0.0f is the only value of the slot that applies the update)
        guard = self.bucket.guard if on_gpu else None
        rc = _lib.lib().tssep_adam_step_guarded(
            p(self.flat_param), p(self.exp_avg), p(self.exp_avg_sq), p(self.bucket.flat),
            self.flat_param.numel(), self.step_count, float(self.gradient_clipping), float(self.lr),
            float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
            p(self.grad_norm), p(self._ws), p(guard) if guard is not None else None,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "adam_step")
        hip_ops.weights_changed()            # derived weight layouts (packs, transposes) are stale now
        return self.grad_norm

    def _per_parameter(self, flat):
        return [flat[off:off + p.numel()] for p, off in zip(self.params, self._offsets)]

    def state_dict(self):
        """Moments per parameter, in parameter order (independent of the padding of the flat buffers)."""
        return dict(step=self.step_count, exp_avg=[t.cpu().clone() for t in self._per_parameter(self.exp_avg)],
                    exp_avg_sq=[t.cpu().clone() for t in self._per_parameter(self.exp_avg_sq)])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        for key, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            src = sd[key]
            if torch.is_tensor(src):      # checkpoints written before the aligned layout: one unpadded flat tensor
                sizes = [p.numel() for p in self.params]
                assert src.numel() == sum(sizes), (src.numel(), sum(sizes))
                src = list(torch.split(src.reshape(-1), sizes))
            assert len(src) == len(self.params), (len(src), len(self.params))
            for dst, t in zip(self._per_parameter(flat), src):
                dst.copy_(t.reshape(-1))
