"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly the symbols
include/tssep_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re
import subprocess

from tssep_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_symbols():
    import __graft_entry__ as g
    g.build()
    protos = _lib.parse_header()
    assert len(protos) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T (tssep_\w+)", out))
    assert exported == set(protos), (exported ^ set(protos))
    L = _lib.lib()
    assert L.tssep_abi_version() == 4
    assert L.tssep_arch() == b"gfx950"


def test_library_reads_no_environment_variable():
    """VERDICT r3 #3 / header conventions: what runs is a function of the arguments.  The production library does not
    even IMPORT getenv (the experiment build `make exp`, selected with TSSEP_HIP_LIB, is the one with switches)."""
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    assert not re.search(r"\b(secure_)?getenv\b", out), out
    for f in sorted(os.listdir(os.path.join(ROOT, "tssep_amd", "csrc"))):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(ROOT, "tssep_amd", "csrc", f)).read()
            # every getenv sits behind the experiment build's macro
            depth, guarded = 0, []
            for line in text.splitlines():
                if line.startswith("#ifdef TSSEP_GEMM_EXP"):
                    guarded.append(depth)
                if line.startswith(("#if", "#ifdef", "#ifndef")):
                    depth += 1
                elif line.startswith("#endif"):
                    depth -= 1
                    if guarded and guarded[-1] == depth:
                        guarded.pop()
                elif line.startswith("#else") and guarded and guarded[-1] == depth - 1:
                    guarded.pop()
                if "getenv(" in line and not line.lstrip().startswith("//"):
                    assert guarded, f"{f}: getenv outside TSSEP_GEMM_EXP: {line.strip()}"


def test_host_side_helpers_match_oracle():
    from oracle import stft as ostft
    L = _lib.lib()
    for n in (160, 1000, 10_000, 64_000, 80_000, 480_000):
        assert L.tssep_stft_frames(n, 1024, 256, 1024, 1, 1) == ostft.num_frames(n)
    assert L.tssep_stft_frames(80_000, 1024, 256, 1024, 1, 1) == 316   # tssep/train/model.py:480


def test_header_is_plain_c():
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "tssep_hip.h")],
                   check=True)


def test_struct_layout_matches_c(tmp_path):
    src = tmp_path / "o.c"
    fields = [f for f, _ in _lib.GemmArgs._fields_]
    body = "".join(f'printf("%zu\\n", offsetof(tssep_gemm_args, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tssep_hip.h"\n'
                   f'int main(){{{body}printf("%zu\\n", sizeof(tssep_gemm_args));return 0;}}')
    exe = tmp_path / "o"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    import ctypes
    assert vals[:-1] == [getattr(_lib.GemmArgs, f).offset for f in fields]
    assert vals[-1] == ctypes.sizeof(_lib.GemmArgs)


def test_mvdr_size_queries_are_host_only():
    """tssep_mvdr_{partial,workspace}_bytes need no GPU: plan consistency and rejected shapes."""
    L = _lib.lib()
    for B, K, D, T, F in ((1, 4, 6, 253, 513), (1, 8, 6, 1878, 513), (8, 8, 6, 1878, 513), (2, 3, 1, 5, 3),
                          (1, 9, 8, 50, 66)):
        pb = L.tssep_mvdr_partial_bytes(B, K, D, T, F)
        unit = 8 * B * K * 2 * D * D * F                 # one time chunk of Hermitian partials
        assert pb > 0 and pb % unit == 0
        chunks = pb // unit
        assert 1 <= chunks <= (T + 15) // 16
        assert L.tssep_mvdr_workspace_bytes(B, K, D, T, F) >= pb + 16 * B * K * D * F
    assert L.tssep_mvdr_partial_bytes(1, 4, 9, 253, 513) == 0        # more than 8 channels
    assert L.tssep_mvdr_workspace_bytes(1, 0, 6, 253, 513) == 0
