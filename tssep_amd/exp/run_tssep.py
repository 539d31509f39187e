"""Toy TS-SEP run initialised from the TS-VAD checkpoint -- counterpart of
tssep/exp/run_tssep.py:43-74."""
from pathlib import Path

from ..train import run as _run

_cwd = Path(__file__).parent


def main(configs=(f"{_cwd}/toy_common.yaml", f"{_cwd}/toy_tssep.yaml"), storage_dir=f"{_cwd}/tssep",
         checkpoint=f"{_cwd}/tsvad/checkpoints/ckpt_best_loss.pth", overrides=(), failure="raise"):
    storage_dir = Path(storage_dir).resolve()
    return _run.main(["train", "with", *map(str, configs), f"eg.trainer.storage_dir={storage_dir}",
                      f"eg.init_ckpt.init_ckpt={Path(checkpoint).resolve()}", *overrides])


if __name__ == "__main__":
    main()
