"""Toy TS-VAD run -- counterpart of tssep/exp/run_tsvad.py:43-71 (init, then train), in-process."""
from pathlib import Path

from ..train import run as _run

_cwd = Path(__file__).parent


def main(configs=(f"{_cwd}/toy_common.yaml", f"{_cwd}/toy_tsvad.yaml"), storage_dir=f"{_cwd}/tsvad",
         overrides=(), failure="raise"):
    storage_dir = Path(storage_dir).resolve()
    return _run.main(["train", "with", *map(str, configs), f"eg.trainer.storage_dir={storage_dir}",
                      *overrides])


if __name__ == "__main__":
    main()
