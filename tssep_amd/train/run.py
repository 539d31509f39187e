"""Command line -- the sacred commands of tssep/train/run.py:154-201 without sacred:

    python -m tssep_amd.train.run init with a.yaml b.yaml eg.trainer.storage_dir=/path key=value
    python -m tssep_amd.train.run with config.yaml          (== train; run inside storage_dir)
    python -m tssep_amd.train.run print_config with config.yaml
    python -m tssep_amd.train.run makefile with config.yaml  (re-writes storage_dir/Makefile)

YAML files are merged left to right, then ``key=value`` overrides (dotted paths, YAML values)."""
import datetime
import os
import shlex
import sys
from pathlib import Path

import yaml

from .experiment import Experiment


def _deep_update(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = v
    return dst


def build_config(args):
    cfg = {}
    for a in args:
        if "=" in a and not Path(a).exists():
            key, value = a.split("=", 1)
            node = cfg
            parts = key.split(".")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = yaml.safe_load(value)
        else:
            with open(a) as f:
                _deep_update(cfg, yaml.safe_load(f) or {})
    cfg.setdefault("eg", {})
    cfg["eg"] = Experiment.get_config(cfg["eg"])
    return cfg


def write_text_atomic(path, text):
    """The reference writes config.yaml through paderbox's ``write_text_atomic`` (run.py:145-151): a
    reader -- another rank of a torchrun job parsing config.yaml at the same moment -- sees the old or
    the new file, never a truncated one."""
    path = Path(path)
    tmp = path.with_name(f".{path.name}.{os.getpid()}.tmp")
    tmp.write_text(text)
    os.replace(tmp, path)


def _chief():
    """Under torchrun every rank enters this module with the same argv; only rank 0 writes into the
    storage dir (config.yaml, backups, Makefile, python_history.txt, log/), the others only read."""
    return int(os.environ.get("RANK", 0)) == 0


def dump_config(storage_dir, cfg):                      # run.py:138-151 (+ backup :104-135)
    storage_dir = Path(storage_dir)
    path = storage_dir / "config.yaml"
    text = yaml.safe_dump(cfg, sort_keys=False)
    old = path.read_text() if path.exists() else None
    if old == text:
        return
    if old is not None:
        stamp = datetime.datetime.today().strftime("%Y_%m_%d_%H_%M_%S")
        (storage_dir / "backup").mkdir(exist_ok=True)
        write_text_atomic(storage_dir / "backup" / f"config_{stamp}.yaml", old)
    write_text_atomic(path, text)


def makefile(cfg, dump=True):
    """storage_dir/Makefile with the targets of tssep/train/makefile.py:10-32 (help, init, run, makefile),
    each re-entering this module with the frozen config.yaml."""
    mod = "tssep_amd.train.run"
    targets = [("help", ["cat Makefile"]),
               ("init", ["# Update config.yaml and Makefile. Print config.", f"python -m {mod} init with config.yaml"]),
               ("run", [f"python -m {mod} with config.yaml"]),
               ("makefile", ["@# Update this makefile.", f"python -m {mod} makefile with config.yaml"])]
    text = "SHELL := /bin/bash\n"
    for name, recipe in targets:
        text += f"\n.PHONY: {name}\n{name}:\n" + "".join(f"\t{line}\n" for line in recipe)
    if dump and _chief():
        write_text_atomic(Path(cfg["eg"]["trainer"]["storage_dir"]) / "Makefile", text)
    return text


def init(cfg):
    storage_dir = Path(cfg["eg"]["trainer"]["storage_dir"])
    cwd = Path.cwd()                                   # run.py:167-172: a sibling directory of the
    if cwd.parts[:-1] == storage_dir.parts[:-1]:       # storage dir is almost certainly a mistake
        assert cwd == storage_dir, (cwd, storage_dir)
    if not _chief():                                   # ranks > 0 of a torchrun job: read-only
        return Experiment.from_config(cfg["eg"])
    storage_dir.mkdir(exist_ok=True, parents=True)
    with open(storage_dir / "python_history.txt", "a") as fd:      # run.py:159-165
        print(f"{shlex.join(sys.argv)}  # {datetime.datetime.today():%Y.%m.%d %H:%M:%S}  # {Path.cwd()}",
              file=fd)
    dump_config(storage_dir, cfg)
    print(yaml.safe_dump(cfg, sort_keys=False))
    makefile(cfg)
    eg = Experiment.from_config(cfg["eg"])
    eg.add_log_files()
    print(f"Initialized {storage_dir}")
    return eg


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    command = "train"
    if argv and argv[0] in ("init", "train", "print_config", "makefile"):
        command = argv.pop(0)
    if argv and argv[0] == "with":
        argv.pop(0)
    cfg = build_config(argv)
    if cfg["eg"]["trainer"].get("storage_dir") is None:
        cfg["eg"]["trainer"]["storage_dir"] = str(Path.cwd())
    if command == "print_config":
        print(yaml.safe_dump(cfg, sort_keys=False))
        return cfg
    if command == "makefile":
        makefile(cfg)
        return cfg
    eg = init(cfg)
    if command == "train":
        eg.train()
    return eg


if __name__ == "__main__":
    print(shlex.join(sys.argv))
    main()
    # data pipelines left open anywhere: their generators' `finally` stops and joins the loader threads BEFORE the
    # interpreter starts to tear the HIP runtime down (a daemon thread inside a HIP call at that point aborts the process)
    import gc
    gc.collect()
