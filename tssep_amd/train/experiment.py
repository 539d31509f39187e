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
        # (new here, no counterpart in the reference: the host layer's runtime policy -- GEMM arithmetic, recurrence
        # family, folds, graph replay; train/runtime.py -- frozen COMPLETE into config.yaml: a run is reproducible
        # from its storage dir alone, whatever shell it was started from)
        from . import runtime as _runtime
        config["runtime"] = dict(_runtime.defaults(), **(config.get("runtime") or {}))

    def __init__(self, trainer=None, train_batchsize=None, validation_batchsize=None,
                 init_ckpt=None, init_ckpt_strict=True, runtime=None):
        self.trainer = trainer
        self.train_batchsize = train_batchsize
        self.validation_batchsize = validation_batchsize
        self.init_ckpt = init_ckpt if init_ckpt is not None else InitCheckPoint()
        self.init_ckpt_strict = init_ckpt_strict
        self.runtime = dict(runtime or {})

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
        from . import runtime as _runtime
        _runtime.apply(self.runtime)                         # the policy frozen in config.yaml (defaults: the benchmark's arithmetic)
        resume = (t.checkpoint_dir / "ckpt_latest.pth").exists()
        if not resume:
            self.init_ckpt(self)
        dev = self.device                                    # joins the torchrun job when there is one
        model.to(torch.device("cuda", dev))
        val = model.prepare_validate_dataset(device=dev, batch_size=self.validation_batchsize)
        train = model.prepare_train_dataset(device=dev, batch_size=self.train_batchsize)
        from .. import distributed as _dist
        if _dist.get_rank() == 0:                            # what this run computed with, beside its checkpoints
            import json
            log_dir = t.storage_dir / "log"
            log_dir.mkdir(exist_ok=True, parents=True)
            (log_dir / "runtime.json").write_text(json.dumps(_runtime.describe(torch.device("cuda", dev)), indent=1))
        t.test_run(train, val)
        t.register_validation_hook(val, max_checkpoints=None)
        return t.train(train, device=dev, resume=resume)
