"""Runtime policy of the host layer: the arithmetic and scheduling choices ABOVE the C ABI, in one recorded place.

The library (`libtssep_hip.so`) reads no environment variable and picks its kernels from the request alone
(`tssep_gemm_plan`); what is left to decide on the host is WHICH arithmetic a GEMM is asked in, which recurrence
family runs, and how the step is scheduled (side stream, folds, graph replay).  Rounds 1-4 read these from
`TSSEP_*` environment variables inside `hip_ops` -- a training run then depended on the shell it was started
from, and the product's default (exact-fp32 GEMMs) was not what `bench.py` measured.  Now:

  * `hip_ops` holds plain module attributes with the DEFAULTS below (the headline arithmetic: split-bf16 GEMMs,
    W-stationary split-bf16 recurrences) and reads no environment;
  * an experiment states deviations under ``eg.runtime`` in its YAML (``eg.runtime.gemm_precision=f32`` on the
    command line); `Experiment.finalize_dogmatic_config` writes the COMPLETE policy into the frozen
    ``config.yaml``, `Experiment.train` applies it and leaves ``log/runtime.json`` (policy, ABI version, device,
    and -- from the trainer -- the kernel plan of the first step) beside the checkpoints;
  * tools and the benchmark pass ``--runtime key=value ...``.

The reference has no counterpart (its arithmetic is whatever ATen / cuDNN do for fp32 tensors,
tssep/train/model.py:502-511); ``gemm_precision: f32`` + ``recurrence: stream`` is that arithmetic here."""
import json

import yaml

from .. import hip_ops as H

# key -> (hip_ops attribute, default, validator)
_POLICY = {
    # arithmetic
    "gemm_precision": ("GEMM_PRECISION", "bf16x3", lambda v: v in ("f32", "bf16x3", "bf16")),
    "wgrad_products": ("WGRAD_PRODUCTS", 3, lambda v: v in (2, 3)),
    "recurrence": ("RECURRENCE", "auto", lambda v: v in ("auto", "stream", "cluster", "onchip")),
    # recurrence scheduling (lstm_onchip.hip: 16-sequence groups in rotation)
    "onchip16": ("ONCHIP16", True, lambda v: isinstance(v, bool)),
    "onchip16_groups": ("ONCHIP16_GROUPS", 0, lambda v: v in (0, 1, 2, 4)),
    "onchip16_min_n": ("ONCHIP16_MIN_N", 1, lambda v: isinstance(v, int) and v >= 1),
    "onchip16_bwd": ("ONCHIP16_BWD", True, lambda v: isinstance(v, bool)),
    "onchip16_bwd_groups": ("ONCHIP16_BWD_GROUPS", 2, lambda v: v in (1, 2, 4)),
    # step scheduling
    "overlap_wgrad": ("OVERLAP_WGRAD", True, lambda v: isinstance(v, bool)),
    "side_stream": ("SIDE_STREAM", True, lambda v: isinstance(v, bool)),
    "side_stream_max_seqs": ("SIDE_STREAM_MAX_SEQS", 512, lambda v: isinstance(v, int) and v >= 0),
    # data parallel: per-layer gradient segments all-reduced during the backward instead of one collective behind it
    # (distributed.GradBucket.notify; off until a scaling curve exists)
    "bucketed_allreduce": ("BUCKETED_ALLREDUCE", False, lambda v: isinstance(v, bool)),
    "fold_tanh": ("FOLD_TANH", True, lambda v: isinstance(v, bool)),
    "fold_tail": ("FOLD_TAIL", 1, lambda v: v in (0, 1, 2, 3)),
    "prepare_derived": ("PREPARE_DERIVED", True, lambda v: isinstance(v, bool)),
    # hipGraph replay of forward + loss + backward in the Trainer: "auto" = batches of at most
    # `graph_max_utterances` utterances (launch-bound steps), "on", "off"
    "graph_step": ("GRAPH_STEP", "auto", lambda v: v in ("auto", "on", "off")),
    "graph_max_utterances": ("GRAPH_MAX_UTTERANCES", 32, lambda v: isinstance(v, int) and v >= 0),
}


def defaults():
    return {k: d for k, (_, d, _) in _POLICY.items()}


def current():
    """The policy in force (the attributes `hip_ops` reads)."""
    return {k: getattr(H, attr) for k, (attr, _, _) in _POLICY.items()}


def apply(settings=None, **more):
    """Set policy entries; -> the previous values of the entries touched (``apply(prev)`` restores them).
    Unknown keys and out-of-range values raise: a typo in a YAML must not train on the default silently."""
    settings = dict(settings or {}, **more)
    prev = {}
    for k, v in settings.items():
        if k not in _POLICY:
            raise KeyError(f"runtime: unknown key {k!r} (known: {', '.join(sorted(_POLICY))})")
        attr, _, ok = _POLICY[k]
        if isinstance(getattr(H, attr), bool) and v in (0, 1) and not isinstance(v, bool):
            v = bool(v)
        if not ok(v):
            raise ValueError(f"runtime: {k} = {v!r} is not a valid value")
        prev[k] = getattr(H, attr)
        setattr(H, attr, v)
    return prev


class applied:
    """``with applied(gemm_precision="f32"): ...`` -- tests and A/B tools."""

    def __init__(self, settings=None, **more):
        self.settings = dict(settings or {}, **more)

    def __enter__(self):
        self.prev = apply(self.settings)
        return self

    def __exit__(self, *exc):
        apply(self.prev)
        return False


def parse_overrides(items):
    """['gemm_precision=f32', 'fold_tail=0'] -> dict (YAML values, as on the experiment command line)."""
    out = {}
    for item in items or ():
        key, sep, value = item.partition("=")
        if not sep:
            raise ValueError(f"runtime override {item!r}: expected key=value")
        out[key.strip()] = yaml.safe_load(value)
    return out


def describe(device=None):
    """What a run should leave in its log directory: the policy, the library, the device."""
    import torch
    from .. import _lib
    d = dict(policy=current(), abi_version=int(_lib.lib().tssep_abi_version()), library=str(_lib.LIB_PATH))
    if torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        props = torch.cuda.get_device_properties(dev)
        d["device"] = dict(name=props.name, arch=getattr(props, "gcnArchName", None), cus=props.multi_processor_count,
                           memory_gb=round(props.total_memory / 2 ** 30, 1))
    return d


def summarise_plan(gemm_log, recurrence_log=()):
    """GEMM_LOG / RECURRENCE_LOG of one step -> {'gemm': {kernel: launches}, 'gemm_requests': [...], 'recurrence': {...}}:
    which kernels the library picked for THIS model and batch (`tssep_gemm_plan`), written once per run."""
    gemms, reqs, seen = {}, [], set()
    for name, M, N, K, d in gemm_log:
        gemms[str(name)] = gemms.get(str(name), 0) + 1
        key = (name, M, N, K, d.get("a_kmajor"), d.get("b_kmajor"), d.get("act"), d.get("c_remap"), d.get("splitk"))
        if key not in seen:
            seen.add(key)
            reqs.append(dict(kernel=str(name), M=M, N=N, K=K, layout="tn" if d.get("a_kmajor") else "nn" if d.get("b_kmajor") else "nt",
                             act=d.get("act"), remap=d.get("c_remap"), splitk=d.get("splitk"), precision=d.get("precision")))
    rec = {}
    for entry in recurrence_log:
        key = json.dumps(entry, sort_keys=True)
        rec[key] = rec.get(key, 0) + 1
    return dict(gemm=gemms, gemm_requests=reqs,
                recurrence=[dict(json.loads(k), launches=n) for k, n in rec.items()])
