"""ctypes binding of libtssep_hip.so (the C ABI declared in include/tssep_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift
from the ABI.  There is NO fallback: if the shared library is missing the import of any
compute op raises (run ``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "tssep_hip.h")
LIB_PATH = os.environ.get("TSSEP_HIP_LIB") or os.path.join(_HERE, "libtssep_hip.so")      # (override: kernel experiments)


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
        ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64),
        ("a_kmajor", ctypes.c_int32), ("b_kmajor", ctypes.c_int32),
        ("b_kshift", ctypes.c_int64), ("kperiod", ctypes.c_int64),
        ("bias", ctypes.c_void_p),
        ("act", ctypes.c_int32), ("accumulate", ctypes.c_int32),
        ("c_remap", ctypes.c_int32),
        ("c_T", ctypes.c_int64), ("c_K", ctypes.c_int64), ("c_sb", ctypes.c_int64),
        ("c_sk", ctypes.c_int64), ("c_st", ctypes.c_int64), ("c_cm", ctypes.c_int64),
        ("c_co", ctypes.c_int64),
        ("c_perm", ctypes.c_void_p), ("c_perm_ld", ctypes.c_int64),
        ("splitk", ctypes.c_int32), ("c_split_stride", ctypes.c_int64),
        ("precision", ctypes.c_int32),
        ("b_ones_col", ctypes.c_int32),
        ("aux", ctypes.c_void_p), ("ldaux", ctypes.c_int64),
    ]


class LstmSizes(ctypes.Structure):
    _fields_ = [("wih_p", ctypes.c_int64), ("bias_p", ctypes.c_int64),
                ("whh_f", ctypes.c_int64), ("whh_b", ctypes.c_int64)]


_SCALARS = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
            "float": ctypes.c_float, "double": ctypes.c_double}


def _ctype(decl):
    decl = decl.strip()
    decl = re.sub(r"/\*.*?\*/", "", decl).strip()
    if decl == "void":
        return None
    if "*" in decl:
        if "tssep_gemm_args" in decl:
            return ctypes.POINTER(GemmArgs)
        if "tssep_lstm_sizes" in decl:
            return ctypes.POINTER(LstmSizes)
        if decl.replace(" ", "").startswith("constchar*"):
            return ctypes.c_char_p
        return ctypes.c_void_p          # float*, int*, int32_t*, void* ... : raw device pointers
    base = decl.replace("const", "").split()
    # "int64_t N" -> type is the first token
    return _SCALARS[base[0]]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef struct.*?\}\s*\w+\s*;", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?\w+\s*\*?)\s*(tssep_\w+)\s*\(([^;{]*?)\)\s*;",
                         text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = [a for a in (x.strip() for x in args.replace("\n", " ").split(",")) if a]
        argtypes = [] if args == ["void"] else [_ctype(a) for a in args]
        protos[name] = (_ctype(ret) if ret.strip() != "int" else ctypes.c_int, argtypes)
    return protos


_lib = None


def lib():
    """Load (once) and return the bound library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64.so, and the HIP runtime this library binds to must be
    # the one torch's streams and allocations live in (loading ours first pulled /opt/rocm's copy into
    # the process as a second runtime: every launch on a torch stream then failed)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(tssep_amd has no CPU fallback).")
    dll = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(dll, name)          # AttributeError = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = dll
    return dll


_ERR = {-1: "invalid shape", -2: "misaligned pointer / leading dimension",
        -3: "unsupported configuration", -4: "kernel launch failed", -5: "NULL pointer"}


_SYNC_CHECK = os.environ.get("TSSEP_SYNC_CHECK") == "1"      # debugging: name the launch a device fault belongs to


def check(status, what):
    if _SYNC_CHECK:
        import sys
        import torch
        sys.stderr.write(f"[tssep] {what}\n"); sys.stderr.flush()
        torch.cuda.synchronize()
    if status != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(status, status)} (status {status})")
