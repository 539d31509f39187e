"""Shared by run_tsvad.py / run_tssep.py: the reference's two-process flow
(tssep/exp/run_tsvad.py:19-71): every stage is a fresh child process."""
import os
import shlex
import sys
from pathlib import Path

_cwd = Path(__file__).parent
MODULE = "tssep_amd.train.run"


def run(cmd, failure):
    """One stage as a FRESH child process (os.system, like the reference: it hands Ctrl-C to the child);
    `failure`: 'raise' | 'exit' on a non-zero return code."""
    cmd = cmd if isinstance(cmd, str) else shlex.join(cmd)
    print(f"\033[92m$ {cmd}\033[0m", flush=True)
    returncode = os.system(cmd)
    if returncode == 0:
        return
    print(f"\033[91m$ {cmd}\033[0m failed with return code {returncode}")
    if failure == "exit":
        sys.exit(returncode if returncode < 256 else returncode >> 8)
    if failure == "raise":
        raise RuntimeError(f"Command {cmd} failed with return code {returncode}")
    raise ValueError(f"Unknown failure mode {failure}")


def two_stages(configs, storage_dir, extra, skip_init, failure):
    """Stage 1 (unless `skip_init`): ``python -m tssep_amd.train.run init with <yaml...> k=v`` freezes the
    configuration into storage_dir/config.yaml (+ Makefile, logs).  Stage 2, inside storage_dir:
    ``python -m tssep_amd.train.run with config.yaml`` trains from the frozen file only."""
    env = f"PYTHONPATH={shlex.quote(str(_cwd.parent.parent))}${{PYTHONPATH:+:$PYTHONPATH}}"
    if skip_init:
        print(f"\033[96mStorage dir {storage_dir} already exists. Skipping init.\033[0m")
    else:
        run(f"{env} " + shlex.join([sys.executable, "-m", MODULE, "init", "with",
                                     *[os.fspath(Path(c).resolve()) for c in configs],
                                     f"eg.trainer.storage_dir={storage_dir}", *extra]), failure)
    run(f"cd {shlex.quote(str(storage_dir))} && {env} {shlex.quote(sys.executable)} -m {MODULE} with config.yaml",
        failure)
