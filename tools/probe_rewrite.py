"""Store-flavour probe (GPU box, under rocprofv3 --pmc WRITE_SIZE): does a line REWRITTEN inside the XCD's L2 reach the
memory side once, or every time?  (the backward recurrence's WRITE_SIZE is 1.8x its d(gates) stream.)

  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -o w -- python3 tools/probe_rewrite.py
  python tools/probe_rewrite.py --summarise out/.../w_counter_collection.csv > profiles/r4_store_flavour_probe.json

240 workgroups x 16 KB (the exchange working set of a backward launch is ~2.5 MB per XCD) rewritten 200 times:
3.9 MB of lines, 786 MB of stores per launch.  One launch per (flavour, pressure): the launches are told apart by
their order in the trace."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
NB, BYTES, REPS = 240, int(_os.environ.get("PROBE_BYTES", 16384)), int(_os.environ.get("PROBE_REPS", 200))
CASES = [(fl, 0, pr) for pr in (0, 65536) for fl in (0, 1, 2, 3, 4)] + \
        [(fl, rd, pr) for pr in (0, 65536) for fl in (1, 0) for rd in (1, 2, 3, 4)]          # a peer reads after every rewrite
NAMES = ["plain", "sc0", "sc1", "sc0 sc1", "nt"]
RNAMES = ["nobody reads", "sc1", "nt", "sc0 sc1", "sc0"]

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "probe_rewrite" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", r.get("Dispatch_ID", 0))))
    out = dict(what=f"tssep_probe_rewrite: {NB} workgroups x {BYTES // 1024} KB rewritten {REPS} times ({NB * BYTES / 2**20:.1f} MB of lines = "
                    f"{NB * BYTES / 8 / 2**20:.2f} MB per XCD, {NB * BYTES * REPS / 1e6:.0f} MB of stores per launch); pressure = bytes read "
                    "non-temporally between two rewrites + 2/3 as many written non-temporally; WRITE_SIZE includes the pressure's own writes",
               store_bytes=NB * BYTES * REPS, line_bytes=NB * BYTES, cases=[])
    assert len(rows) == len(CASES), (len(rows), len(CASES))
    for (fl, rd, pr), r in zip(CASES, rows):
        kb = float(r["Counter_Value"])
        kb -= (NB * REPS * (pr // 24 // 256 + (1 if pr else 0)) * 256 * 16) / 1024 if pr else 0      # ~ the pressure's own stores
        out["cases"].append(dict(flavour=NAMES[fl], peer_reads_with=RNAMES[rd], pressure_bytes_between_rewrites=pr, write_size_kb=kb,
                                 write_size_over_store_bytes=round(kb * 1024 / (NB * BYTES * REPS), 3),
                                 write_size_over_line_bytes=round(kb * 1024 / (NB * BYTES), 2)))
    print(json.dumps(out, indent=1))
    sys.exit(0)

import torch  # noqa: E402
from tssep_amd import _lib  # noqa: E402

L = _lib.lib()
buf = torch.zeros(NB * BYTES // 4, device="cuda")
src = torch.randn(2 * 256 * 1024 * 1024 // 4, device="cuda")          # lower half read, upper half written
sink = torch.zeros(4, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for fl, rd, pr in CASES:
    rc = L.tssep_probe_rewrite(buf.data_ptr(), NB, BYTES, REPS, fl, rd, src.data_ptr(), src.numel() * 2, pr, sink.data_ptr(), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
print("done")
