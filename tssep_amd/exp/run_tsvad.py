"""Toy TS-VAD run -- counterpart of tssep/exp/run_tsvad.py:43-71: ``init`` freezes the configuration into
the storage dir (skipped when the directory exists), then a second process trains from config.yaml there."""
from pathlib import Path

from ._stages import _cwd, two_stages


def main(configs=(f"{_cwd}/toy_common.yaml", f"{_cwd}/toy_tsvad.yaml"), storage_dir=f"{_cwd}/tsvad",
         overrides=(), failure="raise"):
    storage_dir = Path(storage_dir).resolve()
    two_stages(configs, storage_dir, tuple(overrides), skip_init=storage_dir.exists(), failure=failure)
    return storage_dir


if __name__ == "__main__":
    main(failure="exit")
