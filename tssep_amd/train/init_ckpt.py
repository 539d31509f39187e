"""Checkpoint initialisation -- drop-in for tssep/train/init_ckpt.py:18-89 (same class names, dataclass
fields and call protocol ``init_ckpt(eg)``; written from the behaviour):

* ``InitCheckPoint``: when ``init_ckpt`` is set, read the file on the CPU and load its ``"model"`` entry
  into ``eg.trainer.model`` (``strict`` as configured).
* ``InitCheckPointVAD2Sep``: the same, after growing every tensor named in ``bcast`` to the shape the
  model expects: an axis that is ``r`` times too short has each entry repeated ``r`` times in place
  (``[K,320] -> [513K,320]``: row k becomes rows 513k .. 513k+512, which is the (spk, freq) order of the
  TS-SEP head, net.py:637-641), so the TS-SEP logits start equal to the TS-VAD logits in every bin.
"""
import dataclasses
from pathlib import Path

import torch

from ..configurable import Configurable


def _read_model_entry(path):
    path = Path(path)
    if not path.exists():
        raise AssertionError(path)                      # the reference asserts (init_ckpt.py:26)
    return torch.load(str(path), map_location="cpu")["model"]


def grow_by_repetition(tensor, target_shape, name=""):
    """Repeat entries along every axis that is an integer factor short of ``target_shape``."""
    have, want = tuple(tensor.shape), tuple(target_shape)
    if len(have) != len(want):
        raise AssertionError((name, have, want))
    for axis, (h, w) in enumerate(zip(have, want)):
        if h > w:
            raise Exception(f"{name}: checkpoint axis {axis} has {h} entries, the model only {w}", have, want)
        if w % h:
            raise AssertionError((name, have, want, f"axis {axis}: {w} is not a multiple of {h}"))
        if w != h:
            tensor = tensor.repeat_interleave(w // h, dim=axis)
    return tensor


@dataclasses.dataclass
class InitCheckPoint(Configurable):
    init_ckpt: "str | Path" = None
    strict: bool = True

    def adapt(self, model, entries):
        return entries

    def load_model_state_dict(self, eg, ckpt):
        model = eg.trainer.model
        return model.load_state_dict(self.adapt(model, _read_model_entry(ckpt)), strict=self.strict)

    def __call__(self, eg):
        if self.init_ckpt is None:
            return None
        return self.load_model_state_dict(eg, self.init_ckpt)


@dataclasses.dataclass
class InitCheckPointVAD2Sep(InitCheckPoint):
    bcast: tuple = ("mask_estimator.post_net.linear2.weight", "mask_estimator.post_net.linear2.bias")
    mode: str = "repeat"

    def adapt(self, model, entries):
        if self.mode != "repeat":
            raise AssertionError(f"bcast mode {self.mode!r}: only 'repeat' exists (init_ckpt.py:72)")
        for name in self.bcast:
            entries[name] = grow_by_repetition(entries[name], model.get_parameter(name).shape, name)
        return entries
