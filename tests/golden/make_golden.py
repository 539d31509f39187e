"""Generate golden fixtures from the REFERENCE classes (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (read-only) -- it never travels to the GPU box; only the
.npz files written next to this script do.  The reference's absent third-party
packages (padertorch, paderbox, lazy_dataset, pb_bss) are replaced by the
in-memory stubs of SURVEY.md App. D: they provide base-class / identity roles
only, all arithmetic executed below is the reference's own torch code.
"""
import itertools
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_imports():
    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m
    pt = mod("padertorch", Configurable=type("Configurable", (), {}))
    ops = mod("padertorch.ops"); pt.ops = ops
    seq = mod("padertorch.ops.sequence",
              sequence_elementwise=lambda f, x, *a, **k: f(x, *a, **k))
    ops.sequence = seq
    seq.mask = mod("padertorch.ops.sequence.mask")
    mod("padertorch.contrib"); mod("padertorch.contrib.cb")
    mod("padertorch.contrib.cb.summary", ReviewSummary=dict)
    pb = mod("paderbox"); pb.utils = mod("paderbox.utils")
    pb.utils.iterable = mod("paderbox.utils.iterable", zip=zip)
    mod("lazy_dataset", new=lambda examples: examples)
    mod("pb_bss"); mod("pb_bss.testing"); mod("pb_bss.testing.random_utils")
    sys.path.insert(0, "/root/reference")
    import tssep.train
    tssep.train.feature_extractor = mod("tssep.train.feature_extractor")
    tssep.train.model = mod("tssep.train.model")


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def torch_bf_fixture(enhancer):
    """(7) TorchBF('mvdr_souden'), enhancer.py:140-265: target+interference masks, target mask
    only (batched), masking, explicit eps that clamps the trace."""
    torch.manual_seed(7)
    arrs = {}
    cases = (
        # tag, batch, K, M, D, T, F, mask dtype, ref, kwargs
        ("a", None, 3, 2, 4, 20, 5, torch.float32, 1, {}),
        ("b", 2, 2, 1, 6, 9, 66, torch.float64, 0, dict(masking=True, masking_eps=0.3)),
        ("c", None, 2, 1, 3, 12, 7, torch.float32, 2, dict(eps=2.5)),
        ("d", 1, 5, 2, 8, 17, 3, torch.float32, 7, dict(masking=True, masking_eps=0.0)),
    )
    for tag, b, K, M, D, T, F, mdt, ref, kw in cases:
        lead = () if b is None else (b,)
        Y = torch.randn(*lead, D, T, F, dtype=torch.complex128)
        m = torch.rand(*lead, K, M, T, F, dtype=mdt)
        out = enhancer.TorchBF(**kw)(m, {"Observation": Y, "reference_channel": ref}, None)
        arrs.update({f"{tag}_Y": Y, f"{tag}_m": m, f"{tag}_out": out,
                     f"{tag}_ref": np.array(ref),
                     f"{tag}_kw": np.array([kw.get("eps", -1.0), float(kw.get("masking", False)),
                                            kw.get("masking_eps", 0.0)])})
    x = torch.arange(1.0, 10.0).view(3, 3)                      # trace doctest, enhancer.py:107-125
    arrs["trace_3x3"] = enhancer.trace(x)
    arrs["trace_axes"] = enhancer.trace(x.view(3, 1, 3), axis1=0, axis2=2)
    npz("torch_bf", **arrs)


def main():
    _stub_imports()
    from tssep.train import net, rnnp, loss, enhancer, init_ckpt
    from tssep.data import DummyReader
    if sys.argv[1:] == ["torch_bf"]:
        return torch_bf_fixture(enhancer)

    # ---- (1) RNNP_packed, 2/3/4-D inputs (rnnp.py:63-76) + gradients ---------
    torch.manual_seed(1)
    m = rnnp.RNNP_packed(7, 1, 5, 6, 0)
    arrs = {"p." + k: v for k, v in m.state_dict().items()}
    for tag, shape in (("x3", (3, 9, 7)), ("x4", (2, 3, 9, 7)), ("x2", (9, 7))):
        x = torch.randn(shape, requires_grad=True)
        g = torch.randn(*shape[:-1], 6)
        y = m(x)
        m.zero_grad()
        (y * g).sum().backward()
        arrs.update({tag: x, tag + "_g": g, tag + "_y": y, tag + "_dx": x.grad})
        arrs.update({f"{tag}_dp.{k}": v.grad for k, v in m.named_parameters()})
    npz("rnnp", **arrs)

    # ---- (2) MaskEstimator_v2 grid -------------------------------------------
    B, T, D, F, E_cat = 2, 7, 12, 9, 4
    case = 0
    # ts_vad = 8 (the toy configuration's value, init_cfg_common.yaml:75) is generated AFTER the
    # original grid so that the seeds (100 + case index) of the earlier fixtures do not move
    grid = list(itertools.product(("mul", "cat"), (False, 3, 4), ("t", "tf"), (1, 2))) + \
        list(itertools.product(("mul", "cat"), (8,), ("t", "tf"), (1, 2)))
    for comb, ts_vad, res, nap in grid:
        if ts_vad is False and nap != 1:
            continue
        K = ts_vad if ts_vad else 3
        E = F if comb == "mul" else E_cat
        np.random.seed(100 + case)
        torch.manual_seed(100 + case)
        me = net.MaskEstimator_v2(
            idim=D, odim=F, layers=3, units=5, projs=6, dropout=0, nmask=1,
            pre_net="RNNP", aux_net=None, aux_net_output_size=E,
            combination=comb, ts_vad=ts_vad, output_resolution=res,
            random_speaker_order=True, num_averaged_permutations=nap)
        xs = torch.randn(B, T, D)
        aux = torch.rand(B, K, E)
        g = torch.randn(B, K, 1, T, F)
        rng_state = np.random.get_state()
        out = me(xs, [[a for a in ab] for ab in aux])
        # replay the RNG to record the permutations the forward drew (net.py:824-826)
        np.random.set_state(rng_state)
        perm = np.stack([np.random.permutation(K) for _ in range(B)])
        me.zero_grad()
        (out.mask * g).sum().backward()
        arrs = {"p." + k: v for k, v in me.state_dict().items()}
        arrs.update({"dp." + k: v.grad for k, v in me.named_parameters()})
        arrs.update(xs=xs, aux=aux, g=g, perm=perm, mask=out.mask, logit=out.logit,
                    embedding=out.embedding,
                    cfg=np.array([comb, str(ts_vad), res, str(nap)]))
        npz(f"me_{comb}_{ts_vad}_{res}_{nap}", **arrs)
        case += 1

    # ---- (3) Masking, LogMAE, VADSigmoidBCE ----------------------------------
    torch.manual_seed(3)
    mask = torch.rand(2, 3, 1, 5, 9)
    Obs = torch.randn(2, 1, 5, 9, dtype=torch.complex64)
    est = enhancer.Masking()(mask, {"reference_channel": 0, "Observation": Obs}, None)
    e = torch.randn(2, 3, 50); t = torch.randn(2, 3, 50)
    logit = torch.randn(2, 3, 5, 9); vad = (torch.rand(2, 3, 5) > 0.5).float()
    npz("enh_loss", mask=mask, Obs=Obs, est=est, e=e, t=t,
        logmae=loss.LogMAE(pit=False)(e, t), mae=loss.MAE(pit=False)(e, t),
        logit=logit, vad=vad, bce=loss.VADSigmoidBCE(pit=False)(logit, vad))

    # ---- (4) doctest KATs recomputed from seeds (loss.py:198-204,223-234,286-299)
    torch.manual_seed(0)
    target = torch.rand((2, 10000))
    estimate = target + 0.5 * torch.rand((2, 10000))
    kat = dict(mae=loss.MAE(pit=False)(estimate, target),
               logmae=loss.LogMAE(pit=False)(estimate, target))
    e2, t2 = estimate.clone(), target.clone()
    e2[1, :] = 0; t2[1, :] = 0
    kat["logmae_zero_row"] = loss.LogMAE(pit=False)(e2, t2)
    torch.manual_seed(0)
    target = torch.rand((2, 100, 257))
    estimate = target + 0.5 * torch.rand((2, 100, 257))
    l = loss.VADSigmoidBCE(pit=False, target="Speaker_reverberation_early")
    kat["bce"] = l(estimate, target)
    kat["bce10"] = l(((abs(target) > 0.05).float() - 0.5) * 10, target)
    kat["bce1"] = l(((abs(target) > 0.05).float() - 0.5) * 1, target)
    npz("kat_loss", **kat)

    # ---- (5) VAD -> SEP broadcast (init_ckpt.py:54-89) -----------------------
    np.random.seed(5); torch.manual_seed(5)
    kw = dict(idim=D, odim=F, layers=3, units=5, projs=6, dropout=0, nmask=1,
              pre_net="RNNP", aux_net=None, aux_net_output_size=F, combination="mul",
              ts_vad=4, random_speaker_order=False, num_averaged_permutations=1)
    vadm = net.MaskEstimator_v2(output_resolution="t", **kw)
    sepm = net.MaskEstimator_v2(output_resolution="tf", **kw)

    class _EG:  # the two attributes load_model_state_dict touches
        class trainer:
            pass
    holder = torch.nn.Module(); holder.mask_estimator = sepm
    _EG.trainer.model = holder
    ck = os.path.join("/tmp", "tssep_vad_ckpt.pth")
    hv = torch.nn.Module(); hv.mask_estimator = vadm
    torch.save({"model": hv.state_dict()}, ck)
    init_ckpt.InitCheckPointVAD2Sep(init_ckpt=ck).load_model_state_dict(_EG, ck)
    xs = torch.randn(B, T, D); aux = torch.rand(B, 4, F)
    lv = vadm(xs, [[a for a in ab] for ab in aux]).logit
    ls = sepm(xs, [[a for a in ab] for ab in aux]).logit
    arrs = {"vad." + k: v for k, v in hv.state_dict().items()}
    arrs.update({"sep." + k: v for k, v in holder.state_dict().items()})
    npz("vad2sep", xs=xs, aux=aux, logit_vad=lv, logit_sep=ls, **arrs)

    # ---- (6) DummyReader (data.py:58-146): checksums + one tiny full example --
    r = DummyReader(sample_rate=64, aux_size=20)
    ds = r("validate", None, ["observation", "speaker_reverberation_early_ch0", "vad"])
    ex = ds[1]
    arrs = dict(obs=ex["audio_data"]["observation"],
                early=ex["audio_data"]["speaker_reverberation_early_ch0"],
                vad=ex["audio_data"]["vad"], aux=ex["auxInput"])
    r16 = DummyReader()
    ds16 = r16("validate", None, ["observation", "speaker_reverberation_early_ch0"])
    for i in range(2):
        a = ds16[i]["audio_data"]
        arrs[f"sum16_{i}"] = np.array([
            np.float64(a["observation"].astype(np.float64).sum()),
            np.float64(np.abs(a["speaker_reverberation_early_ch0"]).astype(np.float64).sum()),
            np.float64(ds16[i]["auxInput"].sum())])
    vad = r._get_vad(71, 8)
    npz("dummy_reader", vad71=vad, **arrs)

    # ---- (7) TorchBF mask-based MVDR ------------------------------------------
    torch_bf_fixture(enhancer)


if __name__ == "__main__":
    main()
