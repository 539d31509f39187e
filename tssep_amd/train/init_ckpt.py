"""Checkpoint initialisation -- drop-in for tssep/train/init_ckpt.py:18-89."""
import dataclasses
from pathlib import Path

import torch

from ..configurable import Configurable


@dataclasses.dataclass
class InitCheckPoint(Configurable):
    init_ckpt: "str | Path" = None
    strict: bool = True

    def load_model_state_dict(self, eg, ckpt):
        ckpt = Path(ckpt)
        assert ckpt.exists(), ckpt
        state_dict = torch.load(str(ckpt), map_location="cpu")
        return eg.trainer.model.load_state_dict(state_dict["model"], strict=self.strict)

    def __call__(self, eg):
        if self.init_ckpt is not None:
            self.load_model_state_dict(eg, self.init_ckpt)


@dataclasses.dataclass
class InitCheckPointVAD2Sep(InitCheckPoint):
    """Broadcast the TS-VAD head to TS-SEP: repeat_interleave linear2.{weight,bias} over the
    frequency axis (init_ckpt.py:54-89, mode='repeat')."""
    bcast: tuple = ("mask_estimator.post_net.linear2.weight", "mask_estimator.post_net.linear2.bias")
    mode: str = "repeat"

    def load_model_state_dict(self, eg, ckpt):
        ckpt = Path(ckpt)
        assert ckpt.exists(), ckpt
        state_dict = torch.load(str(ckpt), map_location="cpu")
        for k in self.bcast:
            shape = eg.trainer.model.get_parameter(k).shape
            p = state_dict["model"][k]
            assert len(p.shape) == len(shape), (p.shape, shape)
            assert self.mode == "repeat", f"ToDO: Implement {self.mode}"
            for i, (actual, desired) in enumerate(zip(p.shape, shape)):
                if actual == desired:
                    pass
                elif actual < desired:
                    assert desired % actual == 0, (p.shape, shape, actual, desired)
                    p = torch.repeat_interleave(p, desired // actual, dim=i)
                    state_dict["model"][k] = p
                else:
                    raise Exception(p.shape, shape, actual, desired)
        return eg.trainer.model.load_state_dict(state_dict["model"], strict=self.strict)
