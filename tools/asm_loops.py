"""Instruction mix of the loops of one kernel in a -save-temps device assembly (build box, no GPU):
   python tools/asm_loops.py tssep_amd/csrc/build/<file>.s <substring of the mangled kernel name> [min MFMAs per loop]"""
import re, sys, collections

path, want = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l)]
for si, s in enumerate(starts):
    name = lines[s].split(":")[0]
    if want not in name:
        continue
    body = lines[s:starts[si + 1] if si + 1 < len(starts) else len(lines)]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if not (m and m.group(1) in labels and labels[m.group(1)] < i):
            continue
        c, v = collections.Counter(), collections.Counter()
        for b in body[labels[m.group(1)]:i]:
            b = b.strip()
            if not b or b[0] in ".;":
                continue
            op = b.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("ds_read"): c["ds_read"] += 1
            elif op.startswith("ds_write"): c["ds_write"] += 1
            elif op.startswith("buffer_") or op.startswith("global_"): c["vmem"] += 1
            elif op.startswith("scratch_"): c["scratch"] += 1
            elif op.startswith("v_accvgpr"): c["accvgpr"] += 1
            elif op.startswith("v_"): c["valu"] += 1; v[op] += 1
            elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif op.startswith("s_barrier"): c["barrier"] += 1
            elif op.startswith("s_"): c["salu"] += 1
        if c["mfma"] >= min_mfma:
            print(name[:70], m.group(1), dict(c))
            print("   ", v.most_common(12))
