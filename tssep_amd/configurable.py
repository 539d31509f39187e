"""Minimal re-implementation of padertorch's ``Configurable`` factory protocol -- the
reference's operator/plugin interface (resolved by
``Experiment.from_config``, tssep/train/run.py:188).  A config is a nested dict whose
``factory`` key names a class by import path; the remaining keys are constructor kwargs.

Import paths written for the reference (``tssep.train.net.MaskEstimator_v2`` ...) resolve to the
drop-in classes of this package, so the reference's YAML files load unchanged.
"""
import importlib
import inspect

import yaml

_ALIASES = (
    ("tssep.train.", "tssep_amd.train."),
    ("tssep.data", "tssep_amd.data"),
    ("padertorch.train.optimizer.", "tssep_amd.train.optimizer."),
    ("padertorch.train.trainer.Trainer", "tssep_amd.train.trainer.Trainer"),
)


def resolve(factory):
    if not isinstance(factory, str):
        return factory
    for old, new in _ALIASES:
        if factory.startswith(old):
            factory = new + factory[len(old):]
            break
    module, _, name = factory.rpartition(".")
    return getattr(importlib.import_module(module), name)


def factory_path(cls):
    """Report the reference's import path for our drop-in classes (config round trips)."""
    path = f"{cls.__module__}.{cls.__qualname__}"
    # most specific alias first: tssep_amd.train.trainer.Trainer is padertorch's Trainer in the
    # reference's frozen config (init_cfg_common.yaml:9,86), not a `tssep.train.trainer` module
    for old, new in sorted(_ALIASES, key=lambda a: -len(a[1])):
        if path.startswith(new):
            return old + path[len(new):]
    return path


def _signature_defaults(cls):
    out = {}
    for p in list(inspect.signature(cls.__init__).parameters.values())[1:]:
        if p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
            continue
        if p.default is not inspect.Parameter.empty:
            out[p.name] = p.default
    return out


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            if "factory" in v and resolve(v["factory"]) is not resolve(dst[k].get("factory", v["factory"])):
                dst[k] = dict(v)
            else:
                _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


class Configurable:
    @classmethod
    def finalize_dogmatic_config(cls, config):
        pass

    @classmethod
    def get_config(cls, updates=None):
        config = {"factory": cls}
        if updates and "factory" in updates:
            config["factory"] = resolve(updates["factory"])
        klass = config["factory"]
        config.update(_signature_defaults(klass))
        if hasattr(klass, "finalize_dogmatic_config"):
            klass.finalize_dogmatic_config(config)
        if updates:
            _merge(config, {k: v for k, v in updates.items() if k != "factory"})
            if hasattr(klass, "finalize_dogmatic_config"):
                # dependent defaults see the user's values (padertorch's "dogmatic" update)
                redo = dict(config)
                klass.finalize_dogmatic_config(redo)
                for k, v in redo.items():
                    if k not in updates:
                        config[k] = v if not isinstance(v, dict) else _merge(
                            v, config[k] if isinstance(config.get(k), dict) else {})
        return _normalise(config)

    @classmethod
    def from_config(cls, config):
        return _instantiate(config)

    @classmethod
    def new(cls, updates=None):
        return cls.from_config(cls.get_config(updates))

    @classmethod
    def from_file(cls, path, in_config_path=""):
        with open(path) as f:
            cfg = yaml.safe_load(f)
        for key in [k for k in in_config_path.split(".") if k]:
            cfg = cfg[key]
        return cls.from_config(cfg)


def _normalise(config):
    """Fill nested factories' defaults; factories are reported as reference import paths."""
    out = {}
    for k, v in config.items():
        if k == "factory":
            out[k] = factory_path(resolve(v))
        elif isinstance(v, dict) and "factory" in v:
            klass = resolve(v["factory"])
            if hasattr(klass, "get_config"):
                out[k] = klass.get_config(v)
            else:
                sub = {"factory": klass, **_signature_defaults(klass)}
                sub.update({a: b for a, b in v.items() if a != "factory"})
                out[k] = _normalise(sub)
        else:
            out[k] = v
    return out


def _instantiate(config):
    if isinstance(config, dict):
        if "factory" in config:
            klass = resolve(config["factory"])
            kwargs = {k: _instantiate(v) for k, v in config.items() if k != "factory"}
            return klass(**kwargs)
        return {k: _instantiate(v) for k, v in config.items()}
    if isinstance(config, (list, tuple)):
        return type(config)(_instantiate(v) for v in config)
    return config
