"""Regenerates the round-3 results block of DESIGN.md section 5 (between the R3-RESULTS markers) from
tools/design_r3_results.tmpl.md and profiles/r3_* (run after tools/install_profiles.sh 3 and the default bench line)."""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(R, "profiles", f)
design = open(os.path.join(R, "DESIGN.md")).read()
s = open(os.path.join(R, "tools", "design_r3_results.tmpl.md")).read()
sweep = {json.loads(l)["config"]["batch_per_gpu"]: json.loads(l) for l in open(P("r3_batch_sweep.jsonl"))}
bs = [8, 32, 64, 128, 256, 384, 512, 768, 1152, 1536]
s = s.replace("R3_MS", " | ".join(("**%.1f**" if b == 768 else "%.1f") % sweep[b]["ms_per_step"] for b in bs))
s = s.replace("R3_FPS", " | ".join(("**%d k**" if b == 768 else "%d k") % round(sweep[b]["value"] / 1e3) for b in bs))
d = json.load(open(P("r3_bench_default.json")))
r, mh, tw, ex, cb = d["roofline"], d["roofline_mask_head"], d.get("two_product_wgrad"), d["exact_f32"], d["cpu_baseline"]
par = cb["parity_vs_hip"]
s = s.replace("R3_DEFAULT", (
    f"**{d['value'] / 1e6:.3f} M frames/s, {d['ms_per_step']:.1f} ms/step** (median {d['ms_per_step_median']:.1f}; round 2: 1.551 M / 125.2 ms); "
    f"`roofline`: {r['kernel']} {r['achieved']:.0f} TFLOP/s algorithmic = **{r['frac']:.3f}** of the 2.5 PFLOP/s dense bf16 peak over "
    f"{r['launches'] // d['steps']} launches per step ({r.get('tflop_per_step_launched', 0):.2f} TFLOP per step launched; 3× that issued); "
    f"`roofline_mask_head`: {mh['achieved'] / 1e3:.2f} TB/s = **{mh['frac']:.2f}** of 8 TB/s on the unfused mask head's bytes "
    f"(average launch {mh['avg_ms']:.2f} ms); `exact_f32`: {ex['value'] / 1e3:.0f} k frames/s, {ex['ms_per_step']:.1f} ms "
    f"({ex['roofline']['achieved']:.1f} TFLOP/s = {ex['roofline']['frac']:.2f} of the fp32 MFMA peak); `cpu_baseline`: {cb['value'] / 1e3:.2f} k frames/s on "
    f"{cb['cores']} threads (sweep {cb.get('thread_sweep_frames_per_s')}), single thread {cb['single_thread_value'] / 1e3:.2f} k; in-run parity: masks "
    f"{par['max_abs_mask_err']:.1e} abs, loss {par['rel_loss_err']:.1e} rel, gradients {par['max_rel_grad_err']:.1e} of each tensor's largest entry at worst "
    f"(median {par['median_rel_grad_err']:.1e})."))
if tw:
    p2 = tw.get("parity_vs_cpu_oracle") or {}
    s = s.replace("R3_TWO", f"{d['ms_per_step']:.1f} → {tw['ms_per_step']:.1f} ms ({tw['value'] / 1e6:.3f} M frames/s); masks unchanged, gradients "
                            f"{p2.get('max_rel_grad_err', float('nan')):.1e} at worst (median {p2.get('median_rel_grad_err', float('nan')):.1e})")
rows = [l.split() for l in open(P("r3_kernel_stats_default_b768_bf16x3.txt")) if not l.startswith(("#", "Name"))]
def tot(pred):
    return sum(int(x[-5]) for x in rows if len(x) > 6 and pred(" ".join(x[:-6]))) / 8 / 1e6
allk = tot(lambda n: True)
fam = {"big-tile": tot(lambda n: "big_kernel" in n and "tn_big" not in n), "streaming": tot(lambda n: "stream_kernel" in n),
       "8-wave tiled": tot(lambda n: "tall_kernel<" in n and "tn_tall" not in n), "weight gradients big-tile": tot(lambda n: "tn_big" in n),
       "weight gradients 256 x 160 (dW_hh)": tot(lambda n: "tn_w160" in n), "weight gradients 256 x 128": tot(lambda n: "tn_tall" in n),
       "weight gradients 128 x 128": tot(lambda n: "tn_kernel" in n)}
g = sum(fam.values()); rec_f = tot(lambda n: "onchip_fwd" in n or "onchip16_fwd" in n); rec_b = tot(lambda n: "onchip_bwd" in n or "onchip16_bwd" in n)
tail = tot(lambda n: "rfft_frames_kernel<true>" in n or "istft_kernel<true>" in n)
s = s.replace("R3_SHARES", f"{allk:.1f} ms of kernel time per step: GEMMs {100 * g / allk:.1f} % ({g:.1f} ms: " +
              ", ".join(f"{k} {v:.1f}" for k, v in fam.items()) + f"), recurrences {100 * (rec_f + rec_b) / allk:.1f} % (forward {rec_f:.1f}, backward {rec_b:.1f} ms), "
              f"fused tail {tail:.1f} ms, everything else {allk - g - rec_f - rec_b - tail:.1f} ms.")
m = json.load(open(P("r3_mfma_pmc.json")))["kernels"]
s = s.replace("R3_PMC", "MFMA-pipe busy share — " + ", ".join(f"`{re.sub('^gemm_bf16x3_', '', k)}` {100 * v['mfma_busy_frac']:.0f} %" for k, v in m.items()) + ".")
c4, c4n, c5 = (json.load(open(P(f))) for f in ("r3_bench_cfg4.json", "r3_bench_cfg4_nograph.json", "r3_bench_cfg5.json"))
s = s.replace("R3_OTHER", f"cfg4 (8 utterances per GPU, hipGraph replay) {c4['ms_per_step']:.2f} ms/step, {c4['value'] / 1e3:.0f} k frames/s (eager {c4n['ms_per_step']:.2f}; round 2: 14.11); "
              f"cfg5 (8 speakers × 30 s, batch 96) {c5['ms_per_step']:.1f} ms/step, {c5['value'] / 1e3:.0f} k frames/s, mask head {c5['roofline_mask_head']['frac']:.2f} (round 2: 259.6 ms, 0.45); "
              f"`--gemm f32` {ex['ms_per_step']:.1f} ms/step.")
a = design.index("<!-- R3-RESULTS-BEGIN"); a = design.index("\n", a) + 1
b = design.index("<!-- R3-RESULTS-END -->")
open(os.path.join(R, "DESIGN.md"), "w").write(design[:a] + s + design[b:])
print("filled; remaining placeholders:", re.findall(r"R3_[A-Z]+", s))
