"""Scan the gfx950 assembly of the library for a store-data hazard the compiler does not guard (found round 4):

    buffer_store_dwordx3/x4 vDATA, vOFF, s[..], sREG offen      <- more than 64 bits of data, soffset in an SGPR
    v_xxx  vD, ...        with vD inside vDATA                   <- VALU write in the very next issue slot

LLVM's hazard recognizer inserts the required wait state for >64-bit MUBUF stores only when soffset is NOT a register
(GCNHazardRecognizer::createsVALUHazard); on gfx950 the data registers of such a store were observed overwritten
before the store had read them (gemm_bf16x3_stream.hip: the integer offset of the next load reached C as a denormal in
lanes 12-15 of every 16 -- sporadic, ~100 elements per 50 M).  Usage:  python tools/scan_store_hazard.py [asm dir]
(builds the assembly with `hipcc --offload-device-only -S` per source when no directory is given)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tssep_amd", "csrc")
FLAGS = {"stft": ["-fno-slp-vectorize"], "lstm_onchip": ["-fno-slp-vectorize"]}


def build(outdir):
    procs = []
    for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        b = os.path.basename(f)[:-4]
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *FLAGS.get(b, []),
                                       "--offload-device-only", "-S", f, "-o", os.path.join(outdir, b + ".s")],
                                      stderr=subprocess.DEVNULL, cwd=SRC))
    for p in procs:
        p.wait()


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    hits = []
    lines = open(path).read().splitlines()
    ins = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    kernel = "?"
    names = {i: l[:-1] for i, l in enumerate(lines) if re.match(r"^_Z\w+:$", l)}
    for k, (i, l) in enumerate(ins):
        m = re.match(r"(buffer_store_dwordx[34]|buffer_store_dwordx2)\s+(\S+),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)", l)
        if not m or m.group(1).endswith("x2"):
            continue
        soff = m.group(5).rstrip(",")
        if not re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi|ttmp\d+", soff):
            continue                      # immediate / `off` soffset: the compiler guards that form itself
        data = vregs(m.group(2).rstrip(","))
        for ahead in (1, 2):            # one wait state is what the compiler inserts for the guarded forms; two are scanned
            if k + ahead >= len(ins):
                break
            j, nxt = ins[k + ahead]
            op = nxt.split()[0]
            if op.startswith(("s_nop", "s_waitcnt", "s_barrier")):
                break                   # an explicit wait state / a drain in between
            if not op.startswith("v_") or (op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and "_e64" not in op):
                continue
            parts = nxt.split(None, 1)
            if len(parts) < 2:          # an operand-less VALU instruction (v_nop) or a truncated last line: writes nothing
                continue
            dst = parts[1].split(",")[0].strip()
            if vregs(dst) & data:
                kn = max((n for n in names if n <= i), default=None)
                hits.append((os.path.basename(path), names.get(kn, "?")[:70], i + 1, l, nxt, ahead))
                break
    return hits


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else None
    tmp = None
    if d is None:
        tmp = tempfile.mkdtemp()
        build(tmp)
        d = tmp
    total = 0
    files = [f for f in sorted(glob.glob(os.path.join(d, "*.s"))) if "host-x86_64" not in f]
    assert files, f"no assembly under {d}"
    for f in files:
        for h in scan(f):
            total += 1
            print(f"{h[0]}:{h[2]}  {h[1]}  (+{h[5]})\n      {h[3]}\n      {h[4]}")
    print(f"{total} unguarded >64-bit buffer stores with a register soffset followed by a VALU write of their data")
    sys.exit(1 if total else 0)
