"""Minimal trainer -- the part of ``padertorch.train.trainer.Trainer`` the reference relies on
(tssep/train/experiment.py:219-320): iterate the training set, ``model(ex)`` + ``model.review``,
``loss.backward()``, optimizer step every ``virtual_minibatch_size`` examples, validation +
checkpoints every ``checkpoint_trigger`` iterations (``checkpoints/ckpt_<iter>.pth`` with the
``model`` / ``optimizer`` / ``iteration`` / ``epoch`` keys, ``ckpt_latest.pth`` and
``ckpt_best_loss.pth`` links), resume from ``ckpt_latest.pth``."""
import os
from pathlib import Path

import torch

from ..configurable import Configurable


class Trainer(Configurable):
    def __init__(self, model, storage_dir, optimizer, summary_trigger=(1, "epoch"),
                 checkpoint_trigger=(1, "epoch"), stop_trigger=(1, "epoch"),
                 virtual_minibatch_size=1):
        self.model = model
        self.storage_dir = Path(storage_dir).expanduser().resolve()
        self.optimizer = optimizer
        self.summary_trigger = tuple(summary_trigger)
        self.checkpoint_trigger = tuple(checkpoint_trigger)
        self.stop_trigger = tuple(stop_trigger)
        self.virtual_minibatch_size = virtual_minibatch_size
        self.iteration, self.epoch = 0, 0
        self.validation_dataset = None
        self.best_loss = float("inf")
        self.history = []

    @property
    def checkpoint_dir(self):
        return self.storage_dir / "checkpoints"

    def register_validation_hook(self, validation_iterator, max_checkpoints=None, **_):
        self.validation_dataset = validation_iterator

    def test_run(self, train_iterator, validation_iterator, **_):
        """Pre-flight (experiment.py:259-292): one training and one validation example."""
        self.model.train()
        for ex in train_iterator[:1]:
            self.model.review(ex, self.model(ex))["loss"].backward()
        self.model.eval()
        with torch.no_grad():
            for ex in validation_iterator[:1]:
                self.model.review(ex, self.model(ex))
        for p in self.model.parameters():
            p.grad = None

    def _triggered(self, trigger):
        n, unit = trigger
        return unit == "iteration" and self.iteration % n == 0

    def validate(self):
        self.model.eval()
        losses = []
        with torch.no_grad():
            for ex in self.validation_dataset:
                losses.append(self.model.review(ex, self.model(ex))["loss"].detach())
        self.model.train()
        return float(torch.stack(losses).mean()) if losses else float("nan")

    def save_checkpoint(self, val_loss):
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        path = self.checkpoint_dir / f"ckpt_{self.iteration}.pth"
        torch.save({"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                    "optimizer": self.optimizer.state_dict(), "iteration": self.iteration,
                    "epoch": self.epoch}, path)
        links = ["ckpt_latest.pth"]
        if val_loss == val_loss and val_loss <= self.best_loss:
            self.best_loss = val_loss
            links.append("ckpt_best_loss.pth")
        for name in links:
            link = self.checkpoint_dir / name
            if link.is_symlink() or link.exists():
                link.unlink()
            os.symlink(path.name, link)
        return path

    def load_checkpoint(self, path):
        sd = torch.load(str(path), map_location="cpu")
        self.model.load_state_dict(sd["model"])
        if "optimizer" in sd and self.optimizer.bucket is not None:
            self.optimizer.load_state_dict(sd["optimizer"])
        self.iteration, self.epoch = sd.get("iteration", 0), sd.get("epoch", 0)

    def train(self, train_dataset, resume=False, device=0, **_):
        self.model.to(torch.device("cuda", device) if isinstance(device, int) else device)
        self.model.train()
        self.optimizer.set_parameters(self.model.parameters())
        if resume:
            self.load_checkpoint(self.checkpoint_dir / "ckpt_latest.pth")
        stop_n, stop_unit = self.stop_trigger
        assert stop_unit == "iteration", self.stop_trigger
        self.optimizer.zero_grad()
        while self.iteration < stop_n:
            for ex in train_dataset:
                summary = self.model.review(ex, self.model(ex))
                summary["loss"].backward()
                self.iteration += 1
                if self.iteration % self.virtual_minibatch_size == 0:
                    self.optimizer.step()
                    self.optimizer.zero_grad()
                if self._triggered(self.summary_trigger):
                    self.history.append((self.iteration, float(summary["loss"])))
                if self._triggered(self.checkpoint_trigger) and self.validation_dataset is not None:
                    self.save_checkpoint(self.validate())
                if self.iteration >= stop_n:
                    break
            self.epoch += 1
        if self.validation_dataset is not None and not (self.checkpoint_dir / "ckpt_latest.pth").exists():
            self.save_checkpoint(self.validate())
        return self.history
