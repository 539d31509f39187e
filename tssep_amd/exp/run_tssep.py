"""Toy TS-SEP run initialised from the TS-VAD checkpoint -- counterpart of tssep/exp/run_tssep.py:43-74
(``init`` is skipped when the storage dir already holds a config.yaml, :66-70)."""
from pathlib import Path

from ._stages import _cwd, two_stages


def main(configs=(f"{_cwd}/toy_common.yaml", f"{_cwd}/toy_tssep.yaml"), storage_dir=f"{_cwd}/tssep",
         checkpoint=f"{_cwd}/tsvad/checkpoints/ckpt_best_loss.pth", overrides=(), failure="raise"):
    storage_dir = Path(storage_dir).resolve()
    two_stages(configs, storage_dir, (f"eg.init_ckpt.init_ckpt={Path(checkpoint).resolve()}", *overrides),
               skip_init=(storage_dir / "config.yaml").exists(), failure=failure)
    return storage_dir


if __name__ == "__main__":
    main(failure="exit")
