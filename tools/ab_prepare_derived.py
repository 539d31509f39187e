"""cfg4 (8 utterances per GPU, graphed step) with and without the derived weight layouts built as a branch at the start
of the captured step (hip_ops.prepare_derived): `python tools/ab_prepare_derived.py [0|1] [bench.py arguments]`."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tssep_amd.hip_ops as H  # noqa: E402

H.PREPARE_DERIVED = bool(int(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
