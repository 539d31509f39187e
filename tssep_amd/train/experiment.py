"""Experiment -- drop-in for tssep/train/experiment.py:131-320 (orchestration only)."""
import dataclasses
from pathlib import Path

import torch

from ..configurable import Configurable
from .init_ckpt import InitCheckPoint


class Experiment(Configurable):
    @classmethod
    def finalize_dogmatic_config(cls, config):           # experiment.py:132-151
        vmb = 12
        trainer = config.get("trainer") or {}
        defaults = {
            "factory": "tssep_amd.train.trainer.Trainer",
            "model": {"factory": "tssep_amd.train.model.Model"},
            "summary_trigger": [1000 // vmb, "iteration"],
            "checkpoint_trigger": [12000 // vmb, "iteration"],
            "stop_trigger": [5_000_000 // vmb, "iteration"],
            "virtual_minibatch_size": vmb,
            "optimizer": {"factory": "tssep_amd.train.optimizer.Adam", "gradient_clipping": 10},
        }
        for k, v in defaults.items():
            trainer.setdefault(k, v)
        config["trainer"] = trainer
        if config.get("init_ckpt") is None:
            config["init_ckpt"] = {"factory": InitCheckPoint}

    def __init__(self, trainer=None, train_batchsize=None, validation_batchsize=None,
                 init_ckpt=None, init_ckpt_strict=True):
        self.trainer = trainer
        self.train_batchsize = train_batchsize
        self.validation_batchsize = validation_batchsize
        self.init_ckpt = init_ckpt if init_ckpt is not None else InitCheckPoint()
        self.init_ckpt_strict = init_ckpt_strict

    @property
    def device(self):
        """experiment.py:166-191 picks the single visible GPU and refuses more; here every process owns
        the GPU torchrun assigned to it (LOCAL_RANK), one process per GPU."""
        if not torch.cuda.is_available():
            raise RuntimeError("tssep_amd needs an MI355X: there is no CPU path")
        from .. import distributed as _dist
        _, _, local_rank = _dist.init_from_env()
        return local_rank

    def load_model_state_dict(self, ckpt, strict=True):     # experiment.py:199-206
        ckpt = Path(ckpt)
        assert ckpt.exists(), ckpt
        state_dict = torch.load(str(ckpt), map_location="cpu")
        return self.trainer.model.load_state_dict(state_dict["model"], strict=strict)

    def add_log_files(self, **kwargs):                       # experiment.py:208-217
        log_dir = self.trainer.storage_dir / "log"
        log_dir.mkdir(exist_ok=True, parents=True)
        (log_dir / "experiment.txt").write_text(str(self))
        (log_dir / "model.txt").write_text(str(self.trainer.model))
        for k, v in kwargs.items():
            (log_dir / f"{k}.txt").write_text(str(v))

    def train(self):                                         # experiment.py:219-320
        t = self.trainer
        model = t.model
        resume = (t.checkpoint_dir / "ckpt_latest.pth").exists()
        if not resume:
            self.init_ckpt(self)
        dev = self.device                                    # joins the torchrun job when there is one
        model.to(torch.device("cuda", dev))
        val = model.prepare_validate_dataset(device=dev, batch_size=self.validation_batchsize)
        train = model.prepare_train_dataset(device=dev, batch_size=self.train_batchsize)
        t.test_run(train, val)
        t.register_validation_hook(val, max_checkpoints=None)
        return t.train(train, device=dev, resume=resume)
