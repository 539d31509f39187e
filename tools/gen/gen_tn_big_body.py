"""Generates the slot-by-slot stage body of csrc/gemm_bf16x3_tn_big.hip (between the GENERATED markers).
48 MFMA slots per stage (one k-step of 16): P1 = a_lo x b_hi, P2 = a_hi x b_lo, P3 = a_hi x b_hi over the 4 x 4
accumulators; the current stage's a_hi / b_lo transpose reads ride on P1, the NEXT stage's a_lo / b_hi on P3
(three LDS stages: the next stage is complete before this one starts); the 30 staging parts (10 pieces x {split,
split, write + reload}) are spread over the slots."""
import re, sys

mm = []
for (x, y) in (("al", "bh"), ("ah", "bl"), ("ah", "bh")):
    for i in range(4):
        for j in range(4):
            mm.append(f"{'MM1' if x == 'al' else 'MM'}({x}, {y}, {i}, {j});")
att = {k: [] for k in range(48)}
# current stage: a_hi / b_lo, in the order P2 needs them (ah0, bl0..3, ah1..3); two transpose reads each
cur = ["FA(ah, 0, 0)", "FB(bl, 0, 1)", "FB(bl, 1, 1)", "FB(bl, 2, 1)", "FB(bl, 3, 1)", "FA(ah, 1, 0)", "FA(ah, 2, 0)", "FA(ah, 3, 0)"]
for k, f in enumerate(cur):
    att[2 * k].append(f + ";")
# next stage's a_lo / b_hi into the *n registers during P3
nxt = ["NA(aln, 0, 1)", "NB(bhn, 0, 0)", "NB(bhn, 1, 0)", "NB(bhn, 2, 0)", "NB(bhn, 3, 0)", "NA(aln, 1, 1)", "NA(aln, 2, 1)", "NA(aln, 3, 1)"]
for k, f in enumerate(nxt):
    att[32 + 2 * k].append(f + ";")
parts = []
for i in range(8):
    parts += [f"SA1({i});", f"SA2({i});", f"SA3({i});"]
for i in range(2):
    parts += [f"SB1({i});", f"SB2({i});", f"SB3({i});"]
assert len(parts) == 30
free = [k for k in range(48) if not att[k]] + [k for k in range(16, 32)]
free = sorted(set(free))
# 30 parts over slots 1,3,5..(odd slots of P1/P3) and all of P2
order = [k for k in range(48) if k % 2 == 1 or 16 <= k < 32]
assert len(order) >= 30, len(order)
step = len(order) / 30.0
for n, p_ in enumerate(parts):
    att[order[int(n * step)]].append(p_)
lines = ["    " + mm[k] + " " + " ".join(att[k]) + (" " if att[k] else "") + "SLOT;" for k in range(48)]
body = "\n".join(lines)
path = sys.argv[1]
s = open(path).read()
a = s.index("// GENERATED-BODY-BEGIN")
b = s.index("// GENERATED-BODY-END")
s = s[:a] + "// GENERATED-BODY-BEGIN (tools/gen/gen_tn_big_body.py)\n" + body + "\n    " + s[b:]
open(path, "w").write(s)
print("slots written:", len(lines))
