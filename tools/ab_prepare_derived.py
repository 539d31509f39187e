"""bench.py with a module-level setting of tssep_amd.hip_ops changed first -- for alternating A/B runs of step-level
structure on one box:
    python tools/ab_prepare_derived.py {0|1} [bench.py arguments]               (PREPARE_DERIVED: weight layouts on a graph branch)
    python tools/ab_prepare_derived.py NAME=VALUE[,NAME=VALUE] [bench.py arguments]     (e.g. SIDE_STREAM_MAX_ROWS=0)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tssep_amd.hip_ops as H  # noqa: E402

spec = sys.argv[1]
for item in (spec.split(",") if "=" in spec else [f"PREPARE_DERIVED={spec}"]):
    name, value = item.split("=")
    assert hasattr(H, name), name
    setattr(H, name, type(getattr(H, name))(int(value)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
