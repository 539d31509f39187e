"""Module-level parity on a real MI355X: the drop-in classes (tssep_amd.train.*) against
(a) fixtures generated from the REFERENCE classes (tests/golden) and (b) the CPU oracle."""
import glob
import sys
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data as odata, model as omodel, net as onet  # noqa: E402
from test_gpu_kernels import close  # noqa: E402
from test_oracle import ME_CASES_ORDER  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
T_ = torch.as_tensor


def _load(module, g, pre="p."):
    sd = {k[len(pre):]: T_(v) for k, v in g.items() if k.startswith(pre)}
    missing = module.load_state_dict(sd, strict=True)
    return module.cuda()


@pytest.fixture(params=["f32", "bf16x3"])
def gemm_mode(request):
    from tssep_amd import hip_ops
    old = hip_ops.GEMM_PRECISION
    hip_ops.GEMM_PRECISION = request.param
    yield request.param
    hip_ops.GEMM_PRECISION = old


def test_rnnp_packed_against_reference_fixture(golden, gemm_mode):
    """Outputs of the reference's own RNNP_packed (tests/golden/make_golden.py), under both GEMM arithmetics: exact fp32
    (the reference's) and split-bf16 (the product's default; absolute floors x 10: 2^-16 per product instead of 2^-24)."""
    from tssep_amd.train.rnnp import RNNP_packed
    a = 1 if gemm_mode == "f32" else 10
    g = golden("rnnp")
    m = _load(RNNP_packed(7, 1, 5, 6, 0), g)
    for tag in ("x3", "x4", "x2"):
        x = T_(g[tag]).cuda().requires_grad_()
        y = m(x)
        close(y, g[tag + "_y"], rtol=1e-4 * a, atol=2e-6 * a, name=tag)
        m.zero_grad()
        (y * T_(g[tag + "_g"]).cuda()).sum().backward()
        close(x.grad, g[tag + "_dx"], rtol=1e-3, atol=2e-6 * a, name=tag + " dx")
        for k, p in m.named_parameters():
            close(p.grad, g[f"{tag}_dp.{k}"], rtol=1e-3, atol=5e-6 * a, name=f"{tag} d{k}")


ME_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "me_*.npz")))


@pytest.mark.parametrize("name", ME_CASES)
def test_mask_estimator_against_reference_fixture(golden, name, gemm_mode):
    from tssep_amd.train.net import MaskEstimator_v2
    a = 1 if gemm_mode == "f32" else 10          # (absolute floors: split-bf16 products carry 2^-16, not 2^-24)
    g = golden(name)
    comb, ts_vad, res, nap = [str(s) for s in g["cfg"]]
    ts_vad = False if ts_vad == "False" else int(ts_vad)
    E = g["aux"].shape[-1]
    me = _load(MaskEstimator_v2(idim=12, odim=9, layers=3, units=5, projs=6, combination=comb,
                                aux_net_output_size=E, ts_vad=ts_vad, output_resolution=res,
                                num_averaged_permutations=int(nap)), g)
    assert list(me.state_dict().keys()) == [k[2:] for k in g if k.startswith("p.")]
    np.random.seed(100 + ME_CASES_ORDER.index(name))     # same stream the reference forward consumed
    aux = T_(g["aux"]).cuda()
    out = me(T_(g["xs"]).cuda(), [[a for a in ab] for ab in aux])
    close(out.logit, g["logit"], rtol=1e-3, atol=5e-6 * a, name="logit")
    close(out.mask, g["mask"], rtol=1e-3, atol=2e-6 * a, name="mask")
    close(out.embedding, g["embedding"], name="embedding")
    (out.mask * T_(g["g"]).cuda()).sum().backward()
    for k, p in me.named_parameters():
        close(p.grad, g["dp." + k], rtol=2e-3, atol=5e-6 * a, name="d" + k)


def _example_batch(B, K, N, seed=0, E=513):
    rng = np.random.RandomState(seed)
    tgt = (rng.randn(B, K, N) * 0.1).astype(np.float32)
    vad = np.zeros((B, K, N), dtype=np.float32)
    for k in range(K):
        vad[:, k, k * N // (K + 1):(k + 2) * N // (K + 1)] = 1
    tgt *= vad
    obs = tgt.sum(1, keepdims=True) + 0.05 * rng.rand(B, 1, N).astype(np.float32)
    aux = rng.rand(B, K, E).astype(np.float32)
    return T_(obs), T_(aux), T_(tgt), T_(vad)


@pytest.mark.parametrize("res,loss_name", [("tf", "LogMAE"), ("t", "VADSigmoidBCE")])
def test_model_end_to_end_against_oracle(res, loss_name, gemm_mode):
    """STFT -> MFCC+log1p features -> mask estimator -> mask head -> iSTFT -> loss, forward and
    backward, HIP path vs CPU oracle on the same seeded inputs (1e-3 relative bar, fp32)."""
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    B, K, N, units, projs = 2, 4, 6000, 12, 16
    obs, aux, tgt, vad = _example_batch(B, K, N)
    torch.manual_seed(0)
    fe1 = fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40)
    fe2 = fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann")
    m = model.Model(
        fe=fe.ConcaternatedSTFTFeatures(fe1, fe2, size=1024, shift=256, window="hann"),
        reader=None,
        mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=units, projs=projs,
                                            combination="mul", aux_net_output_size=513, ts_vad=K,
                                            output_resolution=res),
        enhancer=enhancer.Masking(),
        loss=loss.LogMAE() if loss_name == "LogMAE" else loss.VADSigmoidBCE()).cuda()
    from tssep_amd.data import DummyReader
    m.reader = DummyReader()
    p = {"mask_estimator." + k: v.detach().cpu().clone().requires_grad_()
         for k, v in m.mask_estimator.state_dict().items()}
    from oracle import stft as ostft
    T = ostft.num_frames(N)
    Vad = vad.view(B, K, N)[..., ::256][..., :T]
    Vad = torch.nn.functional.pad(Vad, (0, T - Vad.shape[-1]))
    cfg = dict(odim=513, combination="mul", ts_vad=K, output_resolution=res)
    np.random.seed(3)
    o = omodel.forward_loss(p, obs, aux, tgt if loss_name == "LogMAE" else Vad, cfg=cfg,
                            loss=loss_name, fast=True)
    o["loss"].sum().backward()
    ex = dict(observation=obs.cuda(), auxInput=aux.cuda(), reference_channel=0,
              speaker_reverberation_early_ch0=tgt.cuda(), Vad=Vad.cuda(), dataset=["v"] * B)
    np.random.seed(3)
    out = m(ex)
    summary = m.review(ex, out)
    close(ex["Observation"], o["Observation"], rtol=1e-4, atol=1e-3, name="Observation")
    close(ex["Input"][..., 40:], o["Input"][..., 40:], rtol=1e-4, atol=2e-6, name="Input log1p")
    close(ex["Input"][..., :40], o["Input"][..., :40], rtol=1e-4, atol=5e-3, name="Input mfcc")
    close(out.logit, o["logit"], rtol=1e-3, atol=2e-5, name="logit")
    close(out.mask, o["mask"], rtol=1e-3, atol=1e-5, name="mask")
    close(out.stft_estimate, o["stft_estimate"], rtol=1e-3, atol=1e-3, name="stft_estimate")
    close(out.time_estimate, o["time_estimate"], rtol=1e-3, atol=1e-5, name="time_estimate")
    close(summary["loss"], o["loss"].sum(), rtol=1e-4, atol=1e-6, name="loss")
    summary["loss"].backward()
    for k, v in m.mask_estimator.named_parameters():
        ref = p["mask_estimator." + k].grad
        scale = float(ref.abs().max())
        close(v.grad, ref, rtol=1e-3, atol=1e-3 * scale + 1e-9, name="d" + k)


@pytest.mark.parametrize("combination,res,loss_name", [("cat", "tf", "LogMAE"), ("mul", "t", "VADSigmoidBCE"),
                                                       ("cat", "t", "VADSigmoidBCE")])
def test_production_size_cat_and_time_resolution_against_oracle(combination, res, loss_name, gemm_mode):
    """VERDICT r5 "missing" #4: `combination='cat'` (the class default, net.py:514, 879-894; auxiliary size 100) and
    `output_resolution='t'` with VADSigmoidBCE (net.py:957-967, loss.py:302-345) were only covered at 12 units; here at
    the PRODUCTION width -- units 300, projections 320, 553 input features, 4 speakers -- so that the kernels a real
    model selects (W-stationary recurrences at H = 300, the 320-wide GEMM tiles, the concatenating conditioning at
    553 + 100 columns, the 't' logit map at P = 320) run: masks, loss and EVERY parameter gradient against the oracle."""
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.data import DummyReader
    from tssep_amd import hip_ops as H
    from oracle import stft as ostft
    B, K, N, units, projs = 2, 4, 16000, 300, 320
    E = 100 if combination == "cat" else 513
    obs, aux, tgt, vad = _example_batch(B, K, N, E=E)
    torch.manual_seed(0)
    fe1 = fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40)
    fe2 = fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann")
    m = model.Model(
        fe=fe.ConcaternatedSTFTFeatures(fe1, fe2, size=1024, shift=256, window="hann"), reader=DummyReader(),
        mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=units, projs=projs, combination=combination,
                                            aux_net_output_size=E, ts_vad=K, output_resolution=res),
        enhancer=enhancer.Masking(), loss=loss.LogMAE() if loss_name == "LogMAE" else loss.VADSigmoidBCE()).cuda()
    p = {"mask_estimator." + k: v.detach().cpu().clone().requires_grad_() for k, v in m.mask_estimator.state_dict().items()}
    T = ostft.num_frames(N)
    Vad = vad.view(B, K, N)[..., ::256][..., :T]
    Vad = torch.nn.functional.pad(Vad, (0, T - Vad.shape[-1]))
    cfg = dict(odim=513, combination=combination, ts_vad=K, output_resolution=res)
    np.random.seed(3)
    o = omodel.forward_loss(p, obs, aux, tgt if loss_name == "LogMAE" else Vad, cfg=cfg, loss=loss_name, fast=True)
    o["loss"].sum().backward()
    ex = dict(observation=obs.cuda(), auxInput=aux.cuda(), reference_channel=0,
              speaker_reverberation_early_ch0=tgt.cuda(), Vad=Vad.cuda(), dataset=["v"] * B)
    np.random.seed(3)
    H.RECURRENCE_LOG = []
    try:
        out = m(ex)
        summary = m.review(ex, out)
        close(out.logit, o["logit"], rtol=1e-3, atol=5e-5, name="logit")
        close(out.mask, o["mask"], rtol=1e-3, atol=2e-5, name="mask")
        close(summary["loss"], o["loss"].sum(), rtol=1e-4, atol=1e-6, name="loss")
        summary["loss"].backward()
        rlog = list(H.RECURRENCE_LOG)
    finally:
        H.RECURRENCE_LOG = None
    H.check_cluster_errors()
    assert rlog, "no recurrence ran?"
    for k, v in m.mask_estimator.named_parameters():
        ref = p["mask_estimator." + k].grad
        assert ref is not None and v.grad is not None, k
        close(v.grad, ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()) + 1e-9, name="d" + k)


@pytest.mark.parametrize("res,loss_name", [("tf", "LogMAE"), ("t", "VADSigmoidBCE")])
def test_review_snapshot_branch(res, loss_name):
    """``create_snapshot`` (model.py:692-752, loss.py:148-169): the summary carries the audios / images the reference
    names -- estimates per batch entry, the observation, its STFT, the mask of the first example, the STFT estimate, one
    image per loss target (and, for the VAD loss, the mask framed by the target activity) -- each the selected batch
    entry as [freq-like, time]; the loss and its gradients are those of the step without the snapshot."""
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.data import DummyReader
    from oracle import stft as ostft
    B, K, N = 2, 3, 5000
    obs, aux, tgt, vad = _example_batch(B, K, N)
    T = ostft.num_frames(N)
    Vad = vad.view(B, K, N)[..., ::256][..., :T]
    Vad = torch.nn.functional.pad(Vad, (0, T - Vad.shape[-1]))
    torch.manual_seed(0)
    m = model.Model(
        fe=fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"), reader=DummyReader(),
        mask_estimator=net.MaskEstimator_v2(idim=513, odim=513, units=12, projs=16, combination="mul",
                                            aux_net_output_size=513, ts_vad=K, output_resolution=res),
        enhancer=enhancer.Masking(), loss=loss.LogMAE() if loss_name == "LogMAE" else loss.VADSigmoidBCE()).cuda()
    results = {}
    for snap in (False, True):
        m.zero_grad()
        m.create_snapshot = snap
        ex = dict(observation=obs.cuda(), auxInput=aux.cuda(), reference_channel=0,
                  speaker_reverberation_early_ch0=tgt.cuda(), Vad=Vad.cuda(), dataset=["v"] * B)
        np.random.seed(3)
        out = m(ex)
        summary = m.review(ex, out)
        summary["loss"].backward()
        results[snap] = (summary, {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None}, out)
    plain, snap = results[False][0], results[True][0]
    assert "audios" not in plain and "images" not in plain
    close(snap["loss"], plain["loss"], rtol=1e-6, atol=1e-7, name="loss with snapshot")
    for k, g in results[False][1].items():
        close(results[True][1][k], g, rtol=1e-5, atol=1e-7 * float(g.abs().max()) + 1e-12, name="d" + k)
    audios, images = snap["audios"], snap["images"]
    assert set(audios) == {f"Masking_audio_est_{i}" for i in range(B)} | {"Masking_audio_observation"}
    for a, sr in audios.values():
        assert sr == 16000 and a.dim() == 1 and a.numel() == N and abs(float(a.abs().max()) - 0.95) < 1e-5
    target_name = "Speaker_reverberation_early_ch0" if loss_name == "LogMAE" else "Vad"
    want = {"Masking_Observation", "v_Masking_mask", "Masking_stft_estimate", f"Masking_target_{target_name}"}
    if loss_name != "LogMAE":
        want.add("Masking_mask")
    assert set(images) == want, set(images)
    out = results[True][2]
    Fm = out.mask.shape[-1]
    assert images["Masking_Observation"].shape == (513, T)
    assert images["v_Masking_mask"].shape == (K * out.mask.shape[-3] * Fm, T)
    assert images["Masking_stft_estimate"].shape == (K * 513, T)
    assert images[f"Masking_target_{target_name}"].shape == (K * (513 if loss_name == "LogMAE" else 40), T)
    if loss_name != "LogMAE":
        assert images["Masking_mask"].shape == (K * out.mask.shape[-3] * (Fm + 80), T)
    for im in images.values():
        assert float(im.min()) >= 0.0 and float(im.max()) <= 1.0 + 1e-6
    # the mask image is the first example's mask, frequency axis upwards
    first = out.mask[0].detach().permute(2, 0, 1, 3).reshape(T, -1).t().flip(0).cpu()
    close(images["v_Masking_mask"], first.clamp(0, 1), rtol=0, atol=0, name="mask image")


def test_long_form_8_speakers_against_oracle():
    """BASELINE configs[4] sizes: 8 speakers, 30 s @ 16 kHz (T = 1878 frames), units 300 / projs 320,
    one utterance: the whole step (default split-bf16 GEMMs, on-chip recurrences at N = 8 and
    T = 1878) against the CPU oracle -- masks, loss and every parameter gradient."""
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.data import DummyReader
    from tssep_amd import hip_ops
    B, K, N = 1, 8, 480000
    obs, aux, tgt, _ = _example_batch(B, K, N, seed=2)
    torch.manual_seed(4)
    m = model.Model(
        fe=fe.ConcaternatedSTFTFeatures(
            fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40),
            fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"),
            size=1024, shift=256, window="hann"),
        reader=DummyReader(),
        mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=300, projs=320, combination="mul",
                                            aux_net_output_size=513, ts_vad=K, output_resolution="tf"),
        enhancer=enhancer.Masking(), loss=loss.LogMAE()).cuda()
    p = {"mask_estimator." + k: v.detach().cpu().clone().requires_grad_()
         for k, v in m.mask_estimator.state_dict().items()}
    np.random.seed(6)
    o = omodel.forward_loss(p, obs, aux, tgt, cfg=dict(odim=513, combination="mul", ts_vad=K,
                                                      output_resolution="tf"), fast=True)
    o["loss"].sum().backward()
    assert o["mask"].shape[-2] == 1878
    ex = dict(observation=obs.cuda(), auxInput=aux.cuda(), reference_channel=0,
              speaker_reverberation_early_ch0=tgt.cuda(), dataset=["long"] * B)
    old = hip_ops.GEMM_PRECISION
    hip_ops.GEMM_PRECISION = "bf16x3"
    try:
        np.random.seed(6)
        out = m(ex)
        summary = m.review(ex, out)
        summary["loss"].backward()
        hip_ops.check_cluster_errors()
    finally:
        hip_ops.GEMM_PRECISION = old
    assert tuple(out.mask.shape) == (B, K, 1, 1878, 513)
    close(out.mask, o["mask"], rtol=1e-3, atol=1e-5, name="mask @ cfg5")
    close(summary["loss"], o["loss"].sum(), rtol=1e-4, atol=1e-6, name="loss @ cfg5")
    for k, v in m.mask_estimator.named_parameters():
        ref = p["mask_estimator." + k].grad
        close(v.grad, ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()) + 1e-9, name="d" + k)


def test_end_to_end_known_answer_of_the_reference():
    """tssep/train/model.py:552-575: validate_LogMAE = 0.74156505 / 0.744494 for the default
    Model config (Log1pMaxNormAbsSTFT, cat, 8 speakers, units 10 / projs 12), 114038 params."""
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.data import DummyReader
    np.random.seed(0)
    torch.manual_seed(0)
    me = net.MaskEstimator_v2(idim=513, odim=513, units=10, projs=12, combination="cat",
                              aux_net_output_size=100)
    m = model.Model(fe=fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"),
                    reader=DummyReader(), mask_estimator=me, enhancer=enhancer.Masking(),
                    loss=loss.LogMAE())
    assert sum(p.numel() for p in m.parameters()) == 114038
    m = m.cuda()
    ex = m.prepare_validate_dataset(device="cuda", batch_size=2)[0]
    summary = m.review(ex, m(ex))
    got = [float(v) for v in summary["scalars"]["validate_LogMAE"]]
    np.testing.assert_allclose(got, [0.74156505, 0.744494], rtol=2e-4)
    assert float(torch.norm(ex["Input"])) == pytest.approx(58.8257, abs=5e-3)
    assert float(ex["Input"].abs().amax()) == pytest.approx(1.0, abs=1e-6)
    summary["loss"].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_vad_to_sep_checkpoint_broadcast(golden, tmp_path):
    """tssep/train/init_ckpt.py:54-89: SEP logits equal VAD logits in every frequency bin."""
    from tssep_amd.train import init_ckpt, net
    g = golden("vad2sep")
    kw = dict(idim=12, odim=9, units=5, projs=6, combination="mul", aux_net_output_size=9, ts_vad=4,
              random_speaker_order=False)
    vad = net.MaskEstimator_v2(output_resolution="t", **kw)
    sep = net.MaskEstimator_v2(output_resolution="tf", **kw)
    vad.load_state_dict({k[len("vad.mask_estimator."):]: T_(v) for k, v in g.items()
                         if k.startswith("vad.")})
    hv = torch.nn.Module(); hv.mask_estimator = vad
    hs = torch.nn.Module(); hs.mask_estimator = sep
    ck = tmp_path / "ckpt.pth"
    torch.save({"model": hv.state_dict()}, ck)

    class EG:
        class trainer:
            model = hs
    init_ckpt.InitCheckPointVAD2Sep(init_ckpt=str(ck))(EG)
    for k, v in hs.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), g["sep." + k])
    xs, aux = T_(g["xs"]).cuda(), T_(g["aux"]).cuda()
    lv = vad.cuda()(xs, aux).logit
    ls = sep.cuda()(xs, aux).logit
    close(lv, g["logit_vad"], rtol=1e-3, atol=5e-6, name="vad logit")
    close(ls, g["logit_sep"], rtol=1e-3, atol=5e-6, name="sep logit")
    close(ls, lv, rtol=0, atol=1e-6, name="sep == vad")


def test_full_size_properties(gemm_mode):
    """BASELINE cfg3 sizes (K=4, 4 s @ 16 kHz, H=300, P=320): parity with the oracle on the
    masks plus size-independent properties of the HIP path."""
    from tssep_amd.train import net
    from tssep_amd import hip_ops as H, functional as Fn
    B, K, N = 2, 4, 64000
    obs, aux, tgt, _ = _example_batch(B, K, N, seed=1)
    torch.manual_seed(1)
    me = net.MaskEstimator_v2(idim=553, odim=513, units=300, projs=320, combination="mul",
                              aux_net_output_size=513, ts_vad=K, output_resolution="tf").cuda()
    w, wsyn = Fn.windows("hann", 1024, 256, "cuda")
    X = H.stft_fwd(obs[:, 0].cuda(), w)
    assert X.shape == (B, 253, 513)
    # STFT -> iSTFT round trip
    xr, _ = H.istft_fwd(X, wsyn, N)
    close(xr, obs[:, 0], rtol=1e-4, atol=2e-5, name="round trip")
    from oracle import features as ofeat
    fb, dct = ofeat.mfcc_tables(1024)
    feat, _ = H.feat_fwd(X, fb.cuda(), dct.cuda(), 40)
    np.random.seed(5)
    out = me(feat, aux.cuda())
    p = {"mask_estimator." + k: v.detach().cpu() for k, v in me.state_dict().items()}
    np.random.seed(5)
    ref = onet.mask_estimator_forward(p, feat.cpu().contiguous(), aux, odim=513, combination="mul",
                                      ts_vad=K, output_resolution="tf", fast=True)
    close(out.mask, ref["mask"], rtol=1e-3, atol=1e-5, name="mask @ cfg3")
    # mask in (0,1); estimate magnitude never exceeds the observation
    assert float(out.mask.min()) > 0 and float(out.mask.max()) < 1
    mask, est = Fn.mask_head(out.logit[:, :, 0], X)
    assert bool((est.abs() <= X.abs()[:, None] * (1 + 1e-6)).all())
    # linearity of the iSTFT adjoint pair: <istft(E), y> == <E, istft^T(y)>
    y = torch.randn(B * K, N, device="cuda")
    t1 = (H.istft_fwd(est.reshape(B * K, 253, 513), wsyn, N)[0] * y).sum()
    dX = H.istft_bwd(y, wsyn, 253)
    t2 = (torch.view_as_real(est.reshape(B * K, 253, 513)) * torch.view_as_real(dX)).sum()
    assert float(t1) == pytest.approx(float(t2), rel=1e-3)


def test_recurrence_kernels_interchangeable_at_full_size():
    """The three recurrence kernels (streaming fp32, fp32 cluster, on-chip bf16x3) give the same
    masks and parameter gradients through the whole mask estimator at cfg3 sizes."""
    from tssep_amd.train import net
    from tssep_amd import hip_ops as H
    B, K, T = 3, 4, 253
    torch.manual_seed(2)
    me = net.MaskEstimator_v2(idim=553, odim=513, units=300, projs=320, combination="mul",
                              aux_net_output_size=513, ts_vad=K, output_resolution="tf").cuda()
    feat = torch.randn(B, T, 553, device="cuda")
    aux = torch.rand(B, K, 513, device="cuda")
    g = torch.randn(B, K, 1, T, 513, device="cuda") * 1e-3
    res = {}
    old = H.RECURRENCE
    try:
        for kind in ("stream", "cluster", "onchip"):
            H.RECURRENCE = kind
            me.zero_grad(set_to_none=True)
            np.random.seed(3)
            out = me(feat, aux)
            (out.mask * g).sum().backward()
            H.check_cluster_errors()
            res[kind] = (out.mask.detach().clone(), [p.grad.clone() for p in me.parameters()])
    finally:
        H.RECURRENCE = old
    for kind in ("cluster", "onchip"):
        close(res[kind][0], res["stream"][0], rtol=1e-3, atol=1e-6, name=kind + " mask")
        for (name, _), a, b in zip(me.named_parameters(), res[kind][1], res["stream"][1]):
            close(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max()) + 1e-9, name=kind + " " + name)


def test_step_is_bitwise_reproducible():
    """Same inputs, same seeds -> bit-identical masks and gradients, run after run (fixed-order
    reductions everywhere: split-K partials, granule sums, ordered-bits atomic maxima; dynamic work
    claiming and stream overlap do not touch the arithmetic)."""
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.data import DummyReader
    from tssep_amd.distributed import GradBucket
    from tssep_amd import hip_ops
    B, K, N = 6, 4, 64000
    obs, aux, tgt, _ = _example_batch(B, K, N, seed=9)
    torch.manual_seed(9)
    m = model.Model(
        fe=fe.ConcaternatedSTFTFeatures(
            fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40),
            fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"),
            size=1024, shift=256, window="hann"),
        reader=DummyReader(),
        mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=300, projs=320, combination="mul",
                                            aux_net_output_size=513, ts_vad=K, output_resolution="tf"),
        enhancer=enhancer.Masking(), loss=loss.LogMAE()).cuda()
    bucket = GradBucket(m.parameters())
    ex0 = dict(observation=obs.cuda(), auxInput=aux.cuda(), reference_channel=0,
               speaker_reverberation_early_ch0=tgt.cuda(), dataset=["r"] * B)
    old = hip_ops.GEMM_PRECISION
    hip_ops.GEMM_PRECISION = "bf16x3"
    runs = []
    try:
        for _ in range(3):
            bucket.zero()
            np.random.seed(11)
            ex = dict(ex0)
            out = m(ex)
            m.review(ex, out)["loss"].backward()
            bucket.sync()
            torch.cuda.synchronize()
            hip_ops.check_cluster_errors()
            runs.append((out.mask.detach().clone(), bucket.flat.clone()))
    finally:
        hip_ops.GEMM_PRECISION = old
    for mask, flat in runs[1:]:
        assert torch.equal(mask, runs[0][0])
        assert torch.equal(flat, runs[0][1])


def test_direct_grad_sink_matches_autograd():
    """Weight gradients accumulated straight into the GradBucket on the side stream equal the
    gradients returned through autograd (and accumulate over two backward passes)."""
    from tssep_amd.distributed import GradBucket
    from tssep_amd.train import net
    from tssep_amd import hip_ops
    B, K, T = 2, 4, 9
    torch.manual_seed(0)
    me = net.MaskEstimator_v2(idim=24, odim=17, units=12, projs=8, combination="mul",
                              aux_net_output_size=17, ts_vad=K, output_resolution="tf").cuda()
    xs = torch.randn(B, T, 24, device="cuda")
    aux = torch.rand(B, K, 17, device="cuda")
    g = torch.randn(B, K, 1, T, 17, device="cuda")

    def run():
        np.random.seed(1)
        (me(xs, aux).mask * g).sum().backward()

    run()
    ref = [p.grad.clone() for p in me.parameters()]
    me.zero_grad(set_to_none=True)
    bucket = GradBucket(me.parameters())
    assert hip_ops.OVERLAP_WGRAD
    run()
    run()
    bucket.sync()
    torch.cuda.synchronize()
    for (name, p), r in zip(me.named_parameters(), ref):
        assert p.grad.data_ptr() == p._tssep_grad_sink.data_ptr()
        close(p.grad, 2 * r, rtol=1e-4, atol=1e-6 + 1e-5 * float(r.abs().max()), name=name)


@pytest.mark.parametrize("combine", [0, 2])
def test_tanh_fold_survives_a_second_consumer_and_a_hook(combine):
    """the Tanh backward folded into the consumer's d(input) GEMM never forms d(activation).
    With an auxiliary loss on the activation (a second consumer) the producer must still see BOTH gradients
    correctly scaled, and with a hook / retain_grad on it the fold must not be taken at all (the hook receives the
    true gradient).  Reference semantics: plain autograd through nn.Tanh (net.py:623-625)."""
    from tssep_amd import functional as Fn
    torch.manual_seed(3)
    K = combine or 1
    B, T, I, Hh, hd = 2, 7, 12, 8, 8
    N = B * K
    l0, p0 = torch.nn.LSTM(I, Hh, bidirectional=True, batch_first=True).cuda(), torch.nn.Linear(2 * Hh, hd).cuda()
    l1, p1 = torch.nn.LSTM(hd * K, Hh, bidirectional=True, batch_first=True).cuda(), torch.nn.Linear(2 * Hh, hd).cuda()
    params = [*l0.parameters(), *p0.parameters(), *l1.parameters(), *p1.parameters()]
    x0 = torch.randn(N * T, I, device="cuda")
    g_out = torch.randn(B * T, hd, device="cuda")
    g_aux = torch.randn(B * T if combine else N * T, hd * K, device="cuda")

    def run(fold, aux_loss, hook):
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_()
        h = Fn.rnnp_layer(x, l0, p0, N, T, act=1, combine=combine, dz_given=fold)
        seen = []
        if hook == "hook":
            h.register_hook(lambda g: seen.append(g.clone()))
        elif hook == "retain":
            h.retain_grad()
        y = Fn.rnnp_layer(h, l1, p1, B, T, in_tanh=(K if fold else 0))
        loss = (y * g_out).sum()
        if aux_loss:
            loss = loss + (h * g_aux).sum()
        loss.backward()
        if hook == "retain":
            seen.append(h.grad.clone())
        return [x.grad.clone()] + [p.grad.clone() for p in params], seen

    for aux_loss in (False, True):
        ref, ref_seen = run(False, aux_loss, "hook")
        for hook in (None, "hook", "retain"):
            got, seen = run(True, aux_loss, hook)
            for i, (a, b) in enumerate(zip(got, ref)):
                close(a, b, rtol=1e-4, atol=1e-6 + 1e-5 * float(b.abs().max()), name=f"aux={aux_loss} hook={hook} grad {i}")
            if hook:
                assert len(seen) == 1
                close(seen[0], ref_seen[0], rtol=1e-4, atol=1e-6, name=f"aux={aux_loss} {hook}: observed d(activation)")
    # and the fold IS taken when nobody watches (the consumer answers on the link with a dummy for its input)
    x = x0.clone().requires_grad_()
    h = Fn.rnnp_layer(x, l0, p0, N, T, act=1, combine=combine, dz_given=True)
    assert getattr(h, "_tssep_tanh_link", None) is not None and not h._tssep_tanh_link.observed()
    h.register_hook(lambda g: None)
    assert h._tssep_tanh_link.observed()


@pytest.mark.parametrize("combine", [0, 2])
def test_tanh_fold_with_two_folded_consumers(combine):
    """an activation that feeds TWO consumers which both fold its Tanh backward into their d(input) GEMM: each
    leaves its d(pre-activation) on the producer's link -- the second must ADD to the first, not replace it (the dummies
    autograd sums for the activation are zeros either way, so nothing else would notice the loss).  Reference semantics:
    plain autograd through nn.Tanh (net.py:623-625)."""
    from tssep_amd import functional as Fn
    torch.manual_seed(5)
    K = combine or 1
    B, T, I, Hh, hd = 2, 7, 12, 8, 8
    N = B * K
    l0, p0 = torch.nn.LSTM(I, Hh, bidirectional=True, batch_first=True).cuda(), torch.nn.Linear(2 * Hh, hd).cuda()
    cons = [(torch.nn.LSTM(hd * K, Hh, bidirectional=True, batch_first=True).cuda(), torch.nn.Linear(2 * Hh, hd).cuda())
            for _ in range(2)]
    params = [*l0.parameters(), *p0.parameters()] + [q for l, p in cons for q in (*l.parameters(), *p.parameters())]
    x0 = torch.randn(N * T, I, device="cuda")
    g_out = [torch.randn(B * T, hd, device="cuda") for _ in cons]

    def run(fold):
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_()
        h = Fn.rnnp_layer(x, l0, p0, N, T, act=1, combine=combine, dz_given=fold)
        loss = sum((Fn.rnnp_layer(h, l, p, B, T, in_tanh=(K if fold else 0)) * g).sum() for (l, p), g in zip(cons, g_out))
        loss.backward()
        return [x.grad.clone()] + [p.grad.clone() for p in params]

    ref, got = run(False), run(True)
    for i, (a, b) in enumerate(zip(got, ref)):
        close(a, b, rtol=1e-4, atol=1e-6 + 1e-5 * float(b.abs().max()), name=f"grad {i}")


def test_fused_adam_matches_torch_adam_with_clipping():
    """tssep_adam_step == torch.nn.utils.clip_grad_norm_(10) + torch.optim.Adam (the reference's
    optimizer settings, tssep/train/experiment.py:147-150) over three steps."""
    from tssep_amd.train.optimizer import Adam
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5))
    ref = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5))
    ref.load_state_dict(lin.state_dict())
    lin.cuda()
    opt = Adam(gradient_clipping=0.5, lr=1e-2)
    opt.set_parameters(lin.parameters())
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for it in range(3):
        x = torch.randn(64, 33) * (5 if it == 0 else 0.01)      # first step clips, later ones do not
        opt.zero_grad()
        lin(x.cuda()).pow(2).sum().backward()
        norm = float(opt.step())
        ropt.zero_grad()
        ref(x).pow(2).sum().backward()
        rnorm = float(torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5))
        ropt.step()
        assert norm == pytest.approx(rnorm, rel=1e-4)
        for p, q in zip(lin.parameters(), ref.parameters()):
            close(p, q, rtol=1e-4, atol=1e-6, name=f"param after step {it}")


def test_optimizer_skips_the_update_when_a_recurrence_timed_out():
    """the only guard against a W-stationary launch that gave up on a peer (err[0]) was a host read at
    summary / checkpoint time -- steps in between trained on garbage.  The fused clip + Adam launch now reads the flag on
    the DEVICE and skips the update (tssep_adam_step_guarded): parameters and moments stay those of the last good step."""
    from tssep_amd import hip_ops
    from tssep_amd.train.optimizer import Adam
    torch.manual_seed(0)
    lin = torch.nn.Linear(33, 17).cuda()
    opt = Adam(gradient_clipping=10.0, lr=1e-2)
    opt.set_parameters(lin.parameters())
    x = torch.randn(8, 33, device="cuda")
    opt.zero_grad()
    lin(x).pow(2).sum().backward()
    before = opt.flat_param.clone()
    flag = hip_ops._err_flag(opt.flat_param.device)
    flag[0] = 3                                   # what a timed-out forward recurrence leaves
    try:
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat_param, before) and float(opt.exp_avg.abs().max()) == 0.0
        with pytest.raises(RuntimeError, match="timed out"):
            hip_ops.check_cluster_errors(opt.flat_param.device)      # (reads and clears the flag)
    finally:
        flag[0] = 0
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(opt.flat_param, before) and float(opt.exp_avg.abs().max()) > 0.0


def test_optimizer_state_survives_a_checkpoint_round_trip():
    """Adam.state_dict / load_state_dict carry the moments per parameter (the flat buffers pad every tensor to a 256-byte
    boundary: distributed.flat_offsets): a second optimizer over a copy of the model continues bit for bit after loading,
    also from the single unpadded flat tensor that checkpoints written before the aligned layout hold; the weights sit
    on 256-byte boundaries and the gaps of every flat buffer stay zero."""
    from tssep_amd.train.optimizer import Adam
    torch.manual_seed(1)
    a = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).cuda()
    b = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).cuda()
    c = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).cuda()
    oa = Adam(gradient_clipping=0.5, lr=1e-2)
    oa.set_parameters(a.parameters())
    xs = [torch.randn(64, 33, device="cuda") for _ in range(4)]

    def step(m, o, x):
        o.zero_grad()
        m(x).pow(2).sum().backward()
        o.step()

    for x in xs[:2]:
        step(a, oa, x)
    assert all(p.data_ptr() % 256 == 0 and p.grad.data_ptr() % 256 == 0 for p in a.parameters())
    used = sum(p.numel() for p in a.parameters())
    assert oa.flat_param.numel() > used
    sd = oa.state_dict()
    legacy = dict(step=sd["step"], exp_avg=torch.cat([t.reshape(-1) for t in sd["exp_avg"]]),
                  exp_avg_sq=torch.cat([t.reshape(-1) for t in sd["exp_avg_sq"]]))
    for m, state in ((b, sd), (c, legacy)):
        m.load_state_dict(a.state_dict())
        o = Adam(gradient_clipping=0.5, lr=1e-2)
        o.set_parameters(m.parameters())
        o.load_state_dict(state)
        m._opt = o
    for x in xs[2:]:
        step(a, oa, x)
        step(b, b._opt, x)
        step(c, c._opt, x)
    for p, q, r in zip(a.parameters(), b.parameters(), c.parameters()):
        assert torch.equal(p, q) and torch.equal(p, r)
    for flat in (oa.flat_param, oa.exp_avg, oa.exp_avg_sq, oa.bucket.flat):      # the gaps never move
        assert float(flat.abs().sum()) == pytest.approx(
            float(sum(t.abs().sum() for t in oa._per_parameter(flat))), rel=1e-6)


@pytest.mark.parametrize("arithmetic", ["bf16x3", "f32"])
def test_toy_experiments_tsvad_then_tssep(tmp_path, arithmetic):
    """BASELINE configs[0]/[1]: the toy TS-VAD run (8 speakers, 2 averaged permutations), then TS-SEP
    initialised from its checkpoint, each as the reference's two child processes -- ``init with ...`` then
    ``with config.yaml`` inside the storage dir (tssep/exp/run_tsvad.py:54-71, run_tssep.py:56-74; the
    reference's tests/test_exp.py:135-162 asserts the same flow does not raise).  Under the product's default
    arithmetic (split-bf16 GEMMs: what bench.py measures -- no override, no environment variable) and under the
    reference's (``eg.runtime.gemm_precision=f32``); either way the policy is frozen into config.yaml and the run leaves
    log/runtime.json + log/kernel_plan.json."""
    import json
    import subprocess
    import sys
    import yaml
    from tssep_amd.exp import run_tsvad, run_tssep
    fast = ["eg.trainer.stop_trigger=[3,iteration]", "eg.trainer.checkpoint_trigger=[3,iteration]",
            "eg.trainer.summary_trigger=[1,iteration]"]
    if arithmetic != "bf16x3":
        fast.append(f"eg.runtime.gemm_precision={arithmetic}")
    vad_dir = run_tsvad.main(storage_dir=tmp_path / "tsvad", overrides=fast)
    hist = json.loads((vad_dir / "log" / "history.json").read_text())
    assert hist["iteration"] == 3 and len(hist["loss"]) == 3 and all(np.isfinite(l) for _, l in hist["loss"])
    assert yaml.safe_load((vad_dir / "config.yaml").read_text())["eg"]["runtime"]["gemm_precision"] == arithmetic
    rt = json.loads((vad_dir / "log" / "runtime.json").read_text())
    assert rt["policy"]["gemm_precision"] == arithmetic and rt["abi_version"] >= 2 and rt["device"]["cus"] > 0
    plan = json.loads((vad_dir / "log" / "kernel_plan.json").read_text())
    assert plan["policy"]["gemm_precision"] == arithmetic and plan["gemm"]
    assert (set(plan["gemm"]) == {"f32"}) == (arithmetic == "f32"), plan["gemm"]
    assert hist.get("graph_replays", 0) >= 1          # single toy utterances: the trainer replays graphs (runtime.graph_step auto)
    ck = vad_dir / "checkpoints" / "ckpt_best_loss.pth"
    for f in ("config.yaml", "Makefile", "python_history.txt", "log/model.txt"):
        assert (vad_dir / f).exists(), f
    sd = torch.load(ck, map_location="cpu")
    assert sd["model"]["mask_estimator.post_net.linear2.weight"].shape == (8, 42)
    assert "fe.fe1.mel_scale.fb" in sd["model"] and "fe.fe1.dct_mat" in sd["model"]      # reference keys
    sep_dir = run_tssep.main(storage_dir=tmp_path / "tssep", checkpoint=ck, overrides=fast)
    hist2 = json.loads((sep_dir / "log" / "history.json").read_text())
    assert hist2["iteration"] == 3 and all(np.isfinite(l) for _, l in hist2["loss"])
    sd2 = torch.load(sep_dir / "checkpoints" / "ckpt_latest.pth", map_location="cpu")
    assert sd2["model"]["mask_estimator.post_net.linear2.weight"].shape == (8 * 513, 42)
    # a second call finds config.yaml, skips init and resumes from ckpt_latest: nothing left to do
    run_tssep.main(storage_dir=sep_dir, checkpoint=ck, overrides=fast)
    assert json.loads((sep_dir / "log" / "history.json").read_text())["iteration"] == 3
    # the Makefile's `run` with one override: continues from iteration 3 to 4 instead of re-initialising
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-m", "tssep_amd.train.run", "with", "config.yaml",
                    "eg.trainer.stop_trigger=[4,iteration]"], cwd=sep_dir, check=True,
                   env=dict(os.environ, PYTHONPATH=root))
    assert json.loads((sep_dir / "log" / "history.json").read_text())["iteration"] == 4
    assert any((sep_dir / "backup").iterdir())                      # the changed config.yaml was backed up


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_torch_bf_class_against_reference_fixture(golden, tag):
    """tssep_amd.train.enhancer.TorchBF, same constructor / call as enhancer.py:140-265."""
    from tssep_amd.train.enhancer import TorchBF
    g = golden("torch_bf")
    eps, masking, masking_eps = g[tag + "_kw"]
    bf = TorchBF("mvdr_souden", masking=bool(masking), masking_eps=float(masking_eps),
                 eps=None if eps < 0 else float(eps))
    assert bf.name == "TorchBF"
    ex = {"Observation": T_(g[tag + "_Y"]).cuda(), "reference_channel": int(g[tag + "_ref"])}
    with torch.no_grad():
        est = bf(T_(g[tag + "_m"]).cuda(), ex, None)
    assert est.dtype == torch.complex128
    close(est, g[tag + "_out"], rtol=1e-9, atol=1e-12, name=tag)


def test_enhancer_classes_checks(golden):
    from tssep_amd.train import enhancer as E
    g = golden("torch_bf")
    m, Y = T_(g["a_m"]).cuda(), T_(g["a_Y"]).cuda()
    with pytest.raises(AssertionError):
        E.TorchBF("gev")
    with pytest.raises(AssertionError):
        E.TorchBF()(m, {"Observation": Y.to(torch.complex64), "reference_channel": 0}, None)
    with pytest.raises(NotImplementedError):
        E.TorchBF()(m.clone().requires_grad_(), {"Observation": Y, "reference_channel": 0}, None)
    assert E.Dummy()(m, {}, None) is None
    out = E.Nothing()(m, {"Observation": Y, "reference_channel": 1}, None)
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(Y[1][None]))
    x = torch.arange(1.0, 10.0).view(3, 3)
    assert float(E.trace(x)) == 15.0 and E.trace(x.view(3, 1, 3), axis1=0, axis2=2).tolist() == [15.0]


def test_device_loader_pinned_async_h2d():
    """tssep_amd.dataset.DeviceLoader: batches arrive on the GPU bit-identical and in order, pinned staging
    buffers are reused (depth + 1 allocations per key, not one per batch), errors of the producer surface."""
    from tssep_amd import dataset as D
    rng = np.random.RandomState(0)
    host = [dict(observation=rng.randn(3, 1, 5000).astype(np.float32),
                 auxInput=rng.rand(3, 4, 513).astype(np.float32), example_id=[f"e{i}"] * 3, reference_channel=0)
            for i in range(9)]
    dl = D.DeviceLoader(D.new(host), "cuda:0", ("observation", "auxInput"), depth=2)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                               # consumer on a non-default stream
        got = [(ex["observation"].clone(), ex["auxInput"].clone(), ex["example_id"]) for ex in dl]
    torch.cuda.synchronize()
    assert len(got) == 9 and dl.stats["batches"] == 9
    for (o, a, ids), h in zip(got, host):
        assert o.is_cuda and torch.equal(o.cpu(), torch.as_tensor(h["observation"]))
        assert torch.equal(a.cpu(), torch.as_tensor(h["auxInput"])) and ids == h["example_id"]
    assert dl.stats["pinned_allocations"] == 2 * 3            # 2 keys x (depth + 1) slots
    assert len(dl[:2]) == 2

    def boom():
        yield host[0]
        raise RuntimeError("reader failed")
    with pytest.raises(RuntimeError, match="reader failed"):
        list(D.DeviceLoader(boom(), "cuda:0", ("observation",)))


def test_device_loader_large_batches_survive_queued_steps():
    """The >= 32 MB branch of DeviceLoader (every real training batch): the host runs several steps ahead
    of the GPU (Trainer.train has no per-step sync) and drops each batch right after queueing its kernels.
    The device block must not be recycled for a later batch's DMA while queued kernels still read it:
    every batch is summed AFTER a long queued delay and the sums must be those of the host data."""
    from tssep_amd import dataset as D
    rows, cols = 24, 400_000                                  # 38.4 MB per batch, 24 >= 2 * copy_threads
    host = [dict(observation=np.full((rows, cols), float(i + 1), dtype=np.float32), reference_channel=0)
            for i in range(8)]
    for i, h in enumerate(host):
        h["observation"][:, ::7] += 0.25 * i
    dl = D.DeviceLoader(D.new(host), "cuda:0", ("observation",), depth=2)
    busy = torch.randn(4096, 4096, device="cuda")
    sums = []
    for ex in dl:
        x = ex["observation"]
        for _ in range(12):                                   # ~ tens of ms of queued work per "step"
            busy = torch.tanh(busy @ busy) * 0.5
        sums.append(x.double().sum() + 0 * busy[0, 0].double())
        del x, ex                                             # the step's only reference goes away here
    torch.cuda.synchronize()
    want = [float(h["observation"].astype(np.float64).sum()) for h in host]
    got = [float(v) for v in sums]
    assert got == want, (got, want)


def test_training_through_the_input_pipeline():
    """prepare_train_dataset(device=cuda) -> DeviceLoader -> Model.forward/review/backward."""
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    torch.manual_seed(0)
    m = model.Model(
        fe=fe.ConcaternatedSTFTFeatures(
            fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40),
            fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"),
            size=1024, shift=256, window="hann"),
        reader=DummyReader(sample_rate=1600, train_examples=5, aux_size=513),
        mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=8, projs=8, combination="mul",
                                            aux_net_output_size=513, ts_vad=8, output_resolution="tf"),
        enhancer=enhancer.Masking(), loss=loss.LogMAE()).cuda()
    m.train()
    ds = m.prepare_train_dataset(device="cuda:0", batch_size=2)
    losses = []
    for ex in ds:
        assert ex["observation"].is_cuda and ex["speaker_reverberation_early_ch0"].is_cuda
        out = m(ex)
        l = m.review(ex, out)["loss"]
        l.backward()
        losses.append(float(l))
    assert len(losses) == 3 and all(np.isfinite(losses))


# ---- data parallel on the HIP path: two processes on ONE GPU --------------------------------------------
def _dp_model():
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    torch.manual_seed(11)
    # units < 128 -> streaming recurrence: no W-stationary launches, which two processes sharing a GPU
    # must not run concurrently (include/tssep_hip.h).  log1p features only: the MFCC dB floor is over the
    # LOCAL batch by design and would couple the utterances of a shard.
    return model.Model(
        fe=fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"), reader=DummyReader(),
        mask_estimator=net.MaskEstimator_v2(idim=513, odim=513, units=16, projs=12, combination="mul",
                                            aux_net_output_size=513, ts_vad=4, output_resolution="tf",
                                            random_speaker_order=False),
        enhancer=enhancer.Masking(), loss=loss.LogMAE()).cuda()


def _dp_batch(lo, hi):
    rng = np.random.RandomState(5)
    tgt = 0.1 * rng.randn(4, 4, 6000).astype(np.float32)
    obs = tgt.sum(1, keepdims=True) + 0.01 * rng.rand(4, 1, 6000).astype(np.float32)
    aux = rng.rand(4, 4, 513).astype(np.float32)
    return dict(observation=torch.as_tensor(obs[lo:hi]).cuda(), auxInput=torch.as_tensor(aux[lo:hi]).cuda(),
                speaker_reverberation_early_ch0=torch.as_tensor(tgt[lo:hi]).cuda(), reference_channel=0,
                dataset=["dp"] * (hi - lo))


def _dp_step(lo, hi):
    from tssep_amd.train.optimizer import Adam
    m = _dp_model()
    opt = Adam(gradient_clipping=10.0)
    opt.set_parameters(m.parameters())
    opt.zero_grad()
    ex = _dp_batch(lo, hi)
    loss = m.review(ex, m(ex))["loss"]
    loss.backward()
    opt.bucket.all_reduce()              # joins the side stream; SUM over ranks when a group exists
    torch.cuda.synchronize()
    return opt.bucket.flat.cpu().numpy(), float(loss.detach())       # numpy: picklable by value through the queue


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from tssep_amd.distributed import shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, loss = _dp_step(*shard_range(4, rank, world))
        q.put((rank, flat, loss))
    except BaseException as e:           # surface the failure in the parent instead of a queue timeout
        q.put((rank, repr(e), None))
    dist.destroy_process_group()


def test_data_parallel_hip_step_matches_single_process():
    """SURVEY 8e on the product path: each of two ranks runs the HIP forward + backward on its half of the
    utterances, GradBucket.all_reduce sums the flat gradients; the result is the single-process gradient
    of the whole batch (loss summed over the batch, tssep/train/model.py:669)."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p_ in procs:
        p_.join(60)
    for r in res:
        assert r[2] is not None, r[1]
    ref, loss = _dp_step(0, 4)
    scale = float(np.abs(ref).max())
    assert scale > 1e-6
    for _, flat, _ in res:
        assert float(np.abs(flat - ref).max()) <= 2e-5 * scale, float(np.abs(flat - ref).max()) / scale
    assert abs(res[0][2] + res[1][2] - loss) <= 1e-5 * abs(loss)


def _dp_guard_worker(rank, world, port, q):
    import torch.distributed as dist
    from tssep_amd import hip_ops
    from tssep_amd.distributed import shard_range
    from tssep_amd.train.optimizer import Adam
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _dp_model()
        opt = Adam(gradient_clipping=10.0, lr=1e-2)
        opt.set_parameters(m.parameters())
        res = []
        for bad in (True, False):
            opt.zero_grad()
            ex = _dp_batch(*shard_range(4, rank, world))
            m.review(ex, m(ex))["loss"].backward()
            before = opt.flat_param.clone()
            flag = hip_ops._err_flag(opt.flat_param.device)
            flag[0] = 3 if (bad and rank == 1) else 0          # ONLY rank 1's recurrence "timed out"
            opt.step()
            torch.cuda.synchronize()
            flag[0] = 0
            res.append((bool(torch.equal(opt.flat_param, before)), float(opt.bucket.guard[0])))
        q.put((rank, res, opt.flat_param.cpu().numpy()))
    except BaseException as e:           # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))
    dist.destroy_process_group()


def test_optimizer_guard_is_global_over_the_ranks():
    """the Adam guard used to read only THIS rank's err[0] although the gradient it protects had been summed
    over the ranks -- the rank whose recurrence timed out skipped the update, its peers applied the garbage.  The flag
    now rides in the gradient all-reduce (GradBucket.guard): when ONLY rank 1 raises it, BOTH ranks skip the update
    (parameters bit-identical to before the step, replicas still equal); the next, clean step updates both."""
    import socket
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p_ in procs:
        p_.join(60)
    for r in res:
        assert r[2] is not None, r[1]
    for rank, (bad, good), _ in res:
        assert bad[0] is True and bad[1] != 0.0, (rank, bad)        # the update was skipped on this rank too
        assert good[0] is False and good[1] == 0.0, (rank, good)    # a clean step applies
    assert np.array_equal(res[0][2], res[1][2])                      # the replicas never diverged


def test_trainer_mixes_eager_and_graph_micro_steps_in_one_virtual_minibatch():
    """with `virtual_minibatch_size` > 1 an EAGER micro-step (its weight gradients still accumulating into the
    bucket on the side stream) may be followed by a GRAPH replay whose captured weight-gradient nodes accumulate into
    the same views -- ordered only against the launch stream until GraphedStep joined the side stream in front of the
    replay.  Alternating eager / graph micro-steps over several virtual minibatches gives the all-eager run's gradient
    bucket bit for bit."""
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.train.graph import GraphedStep
    from tssep_amd.train.optimizer import Adam

    def build():
        torch.manual_seed(4)
        m = model.Model(
            fe=fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"), reader=DummyReader(),
            mask_estimator=net.MaskEstimator_v2(idim=513, odim=513, units=24, projs=16, combination="mul",
                                                aux_net_output_size=513, ts_vad=3, output_resolution="tf",
                                                random_speaker_order=False),
            enhancer=enhancer.Masking(), loss=loss.LogMAE()).cuda()
        opt = Adam(gradient_clipping=10.0, lr=1e-3)
        opt.set_parameters(m.parameters())
        return m, opt

    rng = np.random.RandomState(9)
    batches = []
    for _ in range(6):
        tgt = 0.1 * rng.randn(2, 3, 5000).astype(np.float32)
        batches.append(dict(observation=torch.as_tensor(tgt.sum(1, keepdims=True)).cuda(),
                            auxInput=torch.as_tensor(rng.rand(2, 3, 513).astype(np.float32)).cuda(),
                            speaker_reverberation_early_ch0=torch.as_tensor(tgt).cuda(), reference_channel=0, dataset=["a", "b"]))

    def run(mixed):
        m, opt = build()
        g = GraphedStep(m, opt, zero_grad=False) if mixed else None
        opt.zero_grad()
        buckets = []
        for i, ex in enumerate(batches):
            if mixed and i % 2 == 1:
                _, summ = g(dict(ex))
                assert set(summ["scalars"]) == {"a_LogMAE", "b_LogMAE"}      # (names of THIS batch)
            else:
                ex = dict(ex)                                # (the forward caches Observation / Input in the example it is given)
                m.review(ex, m(ex))["loss"].backward()
            if i % 2 == 1:                                   # virtual minibatch of two micro-steps: eager, then graph
                opt.bucket.sync()
                torch.cuda.synchronize()
                buckets.append(opt.bucket.flat.clone())
                opt.step()
                opt.zero_grad()
        torch.cuda.synchronize()
        return buckets, opt.flat_param.clone()

    be, pe = run(False)
    bm, pm = run(True)
    for a, b in zip(be, bm):
        assert torch.equal(a, b)
    assert torch.equal(pe, pm)


@pytest.mark.parametrize("units", [16, 128])
def test_graphed_step_replays_the_eager_step(units):
    """tssep_amd.train.graph.GraphedStep: forward + loss + backward replayed as one hipGraph gives the eager
    step's loss, masks and flat gradient -- for the streaming (units 16) and the W-stationary (units 128:
    device-side launch epoch) recurrences, on the side stream too -- and the speaker permutation is NOT
    frozen into the graph: every replay consumes np.random exactly like an eager forward (net.py:824-826)."""
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    from tssep_amd.train.graph import GraphedStep
    from tssep_amd.train.optimizer import Adam
    from tssep_amd import hip_ops as H
    torch.manual_seed(2)
    m = model.Model(
        fe=fe.ConcaternatedSTFTFeatures(
            fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40),
            fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"), size=1024, shift=256, window="hann"),
        reader=DummyReader(),
        mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=units, projs=24, combination="mul",
                                            aux_net_output_size=513, ts_vad=4, output_resolution="tf",
                                            random_speaker_order=True),
        enhancer=enhancer.Masking(), loss=loss.LogMAE()).cuda()
    m.train()
    opt = Adam(gradient_clipping=10.0)
    opt.set_parameters(m.parameters())
    rng = np.random.RandomState(9)
    B, K, N = 3, 4, 5000
    tgt = 0.1 * rng.randn(B, K, N).astype(np.float32)
    ex = dict(observation=T_(tgt.sum(1, keepdims=True) + 0.01 * rng.rand(B, 1, N).astype(np.float32)).cuda(),
              auxInput=T_(rng.rand(B, K, 513).astype(np.float32)).cuda(),
              speaker_reverberation_early_ch0=T_(tgt).cuda(), reference_channel=0, dataset=["g"] * B)

    def eager(seed):
        np.random.seed(seed)
        opt.zero_grad()
        out = m(dict(ex))
        l = m.review(dict(ex), out)["loss"]
        l.backward()
        opt.bucket.sync()
        torch.cuda.synchronize()
        return float(l.detach()), out.mask.detach().clone(), opt.bucket.flat.clone()

    ref5, ref6 = eager(5), eager(6)
    assert float((ref5[1] - ref6[1]).abs().max()) > 1e-4           # the permutation matters (ts_vad layer)
    g = GraphedStep(m, opt)
    g(dict(ex))                                                    # warm-up, capture, first replay
    for seed, ref in ((5, ref5), (6, ref6), (5, ref5)):
        np.random.seed(seed)
        out, summary = g(dict(ex))
        state = np.random.get_state()[1][:4].copy()
        torch.cuda.synchronize()
        np.random.seed(seed)
        net.MaskEstimator_v2.draw_permutations(B, K)
        assert (np.random.get_state()[1][:4] == state).all()       # one draw of B permutations per replay
        assert float(summary["loss"]) == pytest.approx(ref[0], rel=1e-6)
        close(out.mask, ref[1].detach().cpu().numpy(), rtol=1e-5, atol=1e-6, name="mask")
        scale = float(ref[2].abs().max())
        assert float((opt.bucket.flat - ref[2]).abs().max()) <= 1e-5 * scale
    H.check_cluster_errors("cuda")
    assert g.replays == 4 and len(g._graphs) == 1
    # a new input shape is a new graph; new DATA in the same shape is copied into the static inputs
    ex2 = {k: (v[:2] if isinstance(v, torch.Tensor) else v) for k, v in ex.items()}
    ex2["dataset"] = ["g"] * 2
    g(ex2)
    assert len(g._graphs) == 2
    ex3 = dict(ex, observation=ex["observation"] * 0.5)
    np.random.seed(5)
    out3, s3 = g(ex3)
    l3 = float(s3["loss"])
    np.random.seed(5)
    opt.zero_grad()
    l3e = float(m.review(dict(ex3), m(dict(ex3)))["loss"])
    assert l3 == pytest.approx(l3e, rel=1e-6) and abs(l3 - ref5[0]) > 1e-4
    # the weight packs / transposes are built INSIDE the graph (as a branch at its start: hip_ops.prepare_derived) from
    # the weights of the moment: after optimizer steps a replay still equals the eager step
    assert H.PREPARED_HITS > 0
    for _ in range(3):
        np.random.seed(5)
        g(dict(ex))
        opt.step()
    ref7 = eager(7)
    assert abs(ref7[0] - ref5[0]) > 1e-6 * abs(ref5[0])           # the weights did move
    np.random.seed(7)
    out7, s7 = g(dict(ex))
    torch.cuda.synchronize()
    assert float(s7["loss"]) == pytest.approx(ref7[0], rel=1e-6)
    close(out7.mask, ref7[1].detach().cpu().numpy(), rtol=1e-5, atol=1e-6, name="mask after optimizer steps")
    assert float((opt.bucket.flat - ref7[2]).abs().max()) <= 1e-5 * float(ref7[2].abs().max())
    H.check_cluster_errors("cuda")


@pytest.mark.parametrize("units", [24, 128])
def test_trainer_graph_step_is_bit_identical_to_the_eager_trainer(units, tmp_path):
    """`Trainer` replays forward + loss + backward as a hipGraph for small batches
    (runtime.graph_step = auto | on; train/graph.py) -- merged, not a patch.  A TS-SEP configuration (random speaker
    order, one permutation, LogMAE; tssep/train/experiment.py:135-151 with virtual_minibatch_size = 3) trained for 24
    iterations over batches of two chunk lengths, eagerly and through graphs, from the same weights and np.random seed:
    the loss of EVERY iteration and the final parameters are bit-identical -- the graphs draw the speaker permutations from
    np.random exactly like the eager step (the capture's own draw is put back), accumulate into the trainer's virtual
    minibatch, and the passes a capture needs leave the half-filled gradient bucket as they found it.  units 24: streaming
    recurrence; 128: the W-stationary kernels (device-side launch epoch)."""
    import json
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net, runtime
    from tssep_amd.train.optimizer import Adam
    from tssep_amd.train.trainer import Trainer

    def build():
        torch.manual_seed(21)
        return model.Model(
            fe=fe.ConcaternatedSTFTFeatures(
                fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40),
                fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"), size=1024, shift=256, window="hann"),
            reader=DummyReader(),
            mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=units, projs=24, combination="mul",
                                                aux_net_output_size=513, ts_vad=4, output_resolution="tf",
                                                random_speaker_order=True, num_averaged_permutations=1),
            enhancer=enhancer.Masking(), loss=loss.LogMAE())

    rng = np.random.RandomState(31)
    data = []
    for i in range(6):
        B, K, N = 2, 4, (5000 if i % 2 else 7300)
        tgt = 0.1 * rng.randn(B, K, N).astype(np.float32)
        data.append(dict(observation=T_(tgt.sum(1, keepdims=True) + 0.01 * rng.rand(B, 1, N).astype(np.float32)).cuda(),
                         auxInput=T_(rng.rand(B, K, 513).astype(np.float32)).cuda(),
                         speaker_reverberation_early_ch0=T_(tgt).cuda(), reference_channel=0, dataset=["tg"] * B))

    class Dataset(list):
        def __iter__(self):
            return (dict(ex) for ex in list.__iter__(self))

    runs = {}
    for mode in ("off", "on", "auto"):
        with runtime.applied(graph_step=mode):
            np.random.seed(77)
            tr = Trainer(build(), tmp_path / mode, Adam(gradient_clipping=10.0, lr=1e-3), summary_trigger=(1, "iteration"),
                         checkpoint_trigger=(1000, "iteration"), stop_trigger=(24, "iteration"), virtual_minibatch_size=3)
            hist = tr.train(Dataset(data), device=0)
            torch.cuda.synchronize()
            hist_file = json.loads((tmp_path / mode / "log" / "history.json").read_text())
            plan = json.loads((tmp_path / mode / "log" / "kernel_plan.json").read_text())
            runs[mode] = ([l for _, l in hist], tr.optimizer.flat_param.clone(), hist_file, plan, tr.optimizer.step_count)
    l_off, p_off, h_off, plan, steps = runs["off"]
    assert len(l_off) == 24 and steps == 8 and all(np.isfinite(l_off)) and len(set(l_off)) > 12
    assert "graph_replays" not in h_off
    for mode in ("on", "auto"):
        l_g, p_g, h_g, _, steps_g = runs[mode]
        assert steps_g == 8
        assert h_g["graph_replays"] == 23 and h_g["graphs"] == 2 and h_g["graph_eager_steps"] == 0, h_g   # (step 1: the eager plan-logging step)
        assert l_g == l_off, [(i, a, b) for i, (a, b) in enumerate(zip(l_g, l_off)) if a != b]
        assert torch.equal(p_g, p_off), float((p_g - p_off).abs().max())
    # the plan of the first step is on record: arithmetic, GEMM kernels by request, recurrence family
    assert plan["policy"]["gemm_precision"] == "bf16x3" and sum(plan["gemm"].values()) >= 20 and plan["gemm_requests"]
    kernels = {r["kernel"] for r in plan["recurrence"]}      # (H = 128: the interleaved forward, the 32-sequence backward)
    assert kernels == ({"onchip16_bf16x3", "onchip32_bf16x3"} if units == 128 else {"stream_f32"}), plan["recurrence"]


def test_fused_tail_equals_materialised_chain():
    """Model.review on an untouched ForwardOutput runs sigmoid -> masking -> istft (-> |e - t| sums) as
    one fused kernel; touching out.mask / out.stft_estimate first (snapshots, custom code) takes the
    chain of separate kernels.  Same loss, same time estimate, same parameter gradients; and the lazy
    fields give the eager tensors."""
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, feature_extractor as fe, loss, model, net
    for loss_cls in (loss.LogMAE, loss.MAE):
        torch.manual_seed(4)
        m = model.Model(
            fe=fe.ConcaternatedSTFTFeatures(
                fe.TorchMFCC(size=1024, shift=256, window="hann", output_size=40),
                fe.Log1pMaxNormAbsSTFT(size=1024, shift=256, window="hann"), size=1024, shift=256, window="hann"),
            reader=DummyReader(),
            mask_estimator=net.MaskEstimator_v2(idim=553, odim=513, units=12, projs=10, combination="mul",
                                                aux_net_output_size=513, ts_vad=4, output_resolution="tf",
                                                random_speaker_order=False),
            enhancer=enhancer.Masking(), loss=loss_cls()).cuda()
        rng = np.random.RandomState(3)
        B, K, N = 2, 4, 7000
        tgt = 0.1 * rng.randn(B, K, N).astype(np.float32)
        ex0 = dict(observation=T_(tgt.sum(1, keepdims=True) + 0.01 * rng.rand(B, 1, N).astype(np.float32)).cuda(),
                   auxInput=T_(rng.rand(B, K, 513).astype(np.float32)).cuda(),
                   speaker_reverberation_early_ch0=T_(tgt).cuda(), reference_channel=0, dataset=["f"] * B)
        res = {}
        for mode in ("fused", "materialised"):
            m.zero_grad()
            ex = dict(ex0)
            out = m(ex)
            assert not out.materialised and "<lazy>" in repr(out)
            if mode == "materialised":
                assert tuple(out.mask.shape) == (B, K, 1, ex["Observation"].shape[-2], 513)
                assert out.stft_estimate.dtype == torch.complex64 and out.materialised
            s_ = m.review(ex, out)
            s_["loss"].backward()
            if mode == "fused":
                assert not out.materialised                       # the step never needed them
                assert getattr(out.time_estimate, "_tssep_absdiff", None) is not None
            res[mode] = (float(s_["loss"].detach()), out.time_estimate.detach().clone(),
                         [p.grad.clone() for p in m.parameters()], out.mask.detach().clone(),
                         out.stft_estimate.detach().clone())
        a, b = res["fused"], res["materialised"]
        assert a[0] == pytest.approx(b[0], rel=1e-5)
        close(a[1], b[1], rtol=1e-5, atol=2e-6, name="time estimate")
        for ga, gb in zip(a[2], b[2]):
            close(ga, gb, rtol=1e-4, atol=1e-6 * float(gb.abs().max()) + 1e-12, name="gradient")
        close(a[3], b[3], rtol=0, atol=0, name="lazy mask")
        close(a[4], b[4], rtol=0, atol=0, name="lazy estimate")


def test_data_parallel_trainer_two_ranks_on_one_gpu(tmp_path):
    """The rank-aware Trainer / Experiment end to end: `init` once, then TWO processes run
    `python -m tssep_amd.train.run with config.yaml` under a torchrun-style environment (gloo staged through the
    host, because RCCL refuses two ranks on one device; the toy model's 40 units use the streaming
    recurrence).  Each rank trains on its half of the toy utterances, gradients are summed by the flat-bucket
    all-reduce, the trainer itself verifies that the replicas are still identical at the end, and only rank 0
    writes the checkpoint."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = os.path.join(root, "tssep_amd", "exp")
    sd = tmp_path / "dp"
    env = dict(os.environ, PYTHONPATH=root)
    fast = ["eg.trainer.stop_trigger=[3,iteration]", "eg.trainer.checkpoint_trigger=[3,iteration]",
            "eg.trainer.summary_trigger=[1,iteration]", "eg.trainer.virtual_minibatch_size=1"]
    subprocess.run([sys.executable, "-m", "tssep_amd.train.run", "init", "with", os.path.join(exp, "toy_common.yaml"),
                    os.path.join(exp, "toy_tsvad.yaml"), f"eg.trainer.storage_dir={sd}", *fast], check=True, env=env,
                   cwd=tmp_path, stdout=subprocess.DEVNULL)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), TSSEP_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-m", "tssep_amd.train.run", "with", "config.yaml"], cwd=sd, env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p_.communicate(timeout=900)[0] for p_ in procs]
    for rank, (p_, o) in enumerate(zip(procs, outs)):
        assert p_.returncode == 0, f"rank {rank}:\n{o[-3000:]}"
    hist = json.loads((sd / "log" / "history.json").read_text())          # written by rank 0 only
    assert hist["iteration"] == 3 and all(np.isfinite(l) for _, l in hist["loss"])
    ck = torch.load(sd / "checkpoints" / "ckpt_latest.pth", map_location="cpu")
    assert ck["iteration"] == 3 and "mask_estimator.post_net.linear2.weight" in ck["model"]
    assert len(list((sd / "checkpoints").glob("ckpt_*.pth"))) >= 2         # ckpt_3 + the links, once


def test_rccl_communicator_single_rank():
    """What a one-GPU box can show of the production backend: `nccl` (= RCCL) initialises with the arguments
    `distributed.init_from_env` passes (device_id, long timeout), the host-side gloo control group is created next to it,
    and the collectives of the data-parallel step run on it -- the in-place SUM all-reduce of a flat fp32 gradient of the
    real size on the compute stream (between two kernels of that stream), a parameter broadcast, the MIN / MAX replica
    check, the int64 control exchange on the gloo group.  One rank: RCCL refuses two ranks per device; the N > 1 logic
    is covered with gloo (test_distributed.py, the two-rank tests above)."""
    import socket
    import subprocess
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    code = textwrap.dedent("""
        import datetime, os, sys, torch
        import torch.distributed as dist
        sys.path.insert(0, %r)
        from tssep_amd import distributed as D
        torch.cuda.set_device(0)
        timeout = datetime.timedelta(seconds=600)
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=timeout, device_id=torch.device("cuda", 0))
        D._make_control_group("nccl", timeout)
        assert D._CONTROL is not None and dist.get_backend() == "nccl"
        flat = torch.arange(10_850_000, device="cuda", dtype=torch.float32) * 1e-3      # 41 MiB, the K = 4 model's bucket
        want = flat * 2 + 1
        flat.mul_(2)                                   # a kernel in front of the collective on the same stream
        w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        w.wait()
        flat.add_(1)                                   # ... and one behind it
        assert torch.equal(flat, want)
        p = torch.randn(1000, device="cuda"); q = p.clone()
        dist.broadcast(p, src=0)
        assert torch.equal(p, q)
        stat = torch.stack([p.double().sum(), (p.double() ** 2).sum()])
        lo, hi = stat.clone(), stat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
        t, group = D._control_tensor([3, 0])
        assert not t.is_cuda and group is D._CONTROL
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        assert t.tolist() == [3, 0]
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        print("rccl ok")
    """ % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_default_flags_print_one_json_line():
    """`python bench.py` with its default flags (split-bf16 line + the exact-fp32 secondary line + the rooflines)
    at a small batch: ONE JSON line on stdout with the contract's keys.  (The exact-fp32 leg once crashed on a
    variable the main leg defined later -- nothing else in the suite runs that leg.)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "8", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--oracle-slice", "4", "--graph", "off"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_mask_head", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["roofline_mask_head"]["bound"] == "hbm" and "chain" in d["roofline_mask_head"]
    mh = d["roofline_mask_head"]
    # `frac` is the fused kernels' OWN bytes over their time (physical); the unfused head's bytes over the
    # fused time is `frac_effective`; the stand-alone head (the HBM-bound kernel of the north star) is measured beside it
    assert 0 < mh["frac"] < 1 and mh["own_bytes_per_launch"] < mh["algorithmic_bytes_per_launch"] and mh["frac"] < mh["frac_effective"]
    assert mh["frac"] == pytest.approx(mh["own_bytes_per_launch"] / (mh["avg_ms"] * 1e-3) / 1e9 / 8000.0, rel=2e-2)
    sh = mh["standalone_head"]
    assert 0 < sh["fwd"]["frac"] < 1 and 0 < sh["bwd"]["frac"] < 1 and sh["bytes_per_launch"] == mh["algorithmic_bytes_per_launch"]
    # the timed batch itself is checked in the same run; VERDICT r5 #3: gradients at the 1e-3 bar, and the
    # BACKWARD against the CPU oracle too (the batch once more with 0 / 1 loss weights = the gradient of the slice)
    pb = d["parity_at_headline_batch"]
    assert pb["batch"] == 8 and pb["within_bars"] is True and pb["max_abs_mask_err"] < 1e-3 and pb["max_rel_grad_err"] < 1e-3
    assert pb["bar_gradients"] == 1e-3
    ob = pb["against_cpu_oracle"]["backward"]
    assert ob["utterances"] == 4 and ob["max_rel_grad_err"] < 1e-3 and len(ob["worst_five"]) == 5
    assert pb["against"]["gemm_kernels"] == ["f32"] and "f32" not in pb["headline_arithmetic"]["gemm_kernels"]
    assert d["roofline"]["traffic_algorithmic"] > 0          # (at batch 8 the dominant MFMA kernel is a recurrence: 40 / 44 B per cell)
    e = d["f32_gemms_bf16x3_recurrence"]
    assert e is not None and e["roofline"]["peak"] == pytest.approx(157.3) and e["dtype"] == "f32 GEMMs + bf16x3 recurrence"
    # the fp32 end-to-end bookend: exact-fp32 GEMMs AND the exact-fp32 recurrence (VERDICT r3 #5a)
    rw = d["reference_width"]
    assert rw is not None and rw["dtype"] == "f32" and rw["roofline"]["peak"] == pytest.approx(157.3) and rw["value"] > 0
    assert d["two_product_wgrad"] is not None and d["two_product_wgrad"]["value"] > 0


_TWO_RANK_CHECKSUMS = {}


@pytest.mark.parametrize("workload,batch,graph,bucketed", [("cfg3", 4, "off", False), ("cfg4", 8, "on", False),
                                                           ("cfg3", 4, "off", True), ("cfg4", 8, "on", True)])
def test_bench_two_ranks_on_one_gpu(workload, batch, graph, bucketed):
    """The N > 1 path of bench.py exactly as the driver launches it (`python -m torch.distributed.run ... bench.py
    --gpus 2`): rank-0 broadcast of the parameters, per-rank shards, barrier + max-over-ranks timing, the flat
    gradient all-reduce inside the optimizer step, ONE JSON line from rank 0 with n_gpus = 2 and the doubled global
    batch.  Two ranks share this box's single GPU, so the backend is gloo (RCCL refuses two ranks per device) and the
    recurrences are the streaming ones (two processes must not run W-stationary launches concurrently).
    cfg4 + graph on: the per-GPU shard of configs[3] as the replayed hipGraph with the all-reduce +
    optimizer outside it -- the replicas must still agree after the timed steps, i.e. the all-reduce sees the complete
    gradient of replay i and replay i + 1 starts from the updated weights.
    bucketed (round 6, `--runtime bucketed_allreduce=true`): the five layers' segments are all-reduced as their backward
    completes, last layer first, the rest (alignment tail, guard slot) in the optimizer step; a graph replay reduces
    after the replay.  Two ranks: a + b = b + a, so the parameters after the run are those of the flat all-reduce BIT FOR
    BIT (`parameter_checksum` of the un-bucketed case of the same workload, which runs first)."""
    import json
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, TSSEP_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--batch", str(batch),
                        "--recurrence", "stream", "--no-cpu-baseline", "--no-exact-f32", "--graph", graph,
                        *(["--runtime", "bucketed_allreduce=true"] if bucketed else [])],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 2 * batch
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert (d["config"]["hip_graph"] is not None) == (graph == "on")
    c = d["config"]["collective"]
    # (graph on: the two timed replays + the two eager steps behind them that time the kernels for the roofline)
    assert c["process_group_world_size"] == 2 and c["replicas_agree"] is True and c["launches"] == (4 if graph == "on" else 2)
    assert c["allreduce_ms"] > 0 and c["bytes"] > 4e7
    # what the first 8-GPU run should explain by itself: per-rank step times, bus bandwidth, the library behind "nccl"
    assert len(c["ms_per_step_by_rank"]) == 2 and all(v > 0 for v in c["ms_per_step_by_rank"])
    assert c["allreduce_busbw_gbps"] > 0 and "collective_library" in c and "environment" in c
    assert c["bucketed"]["enabled"] is bucketed
    if bucketed:
        assert c["bucketed"]["segments"] == 5
        # (the last step of either run is an eager one: every layer reported, last layer first)
        assert c["bucketed"]["reduced_during_backward_in_order"] == [4, 3, 2, 1, 0], c["bucketed"]
        if (workload, graph) in _TWO_RANK_CHECKSUMS:
            assert c["parameter_checksum"] == _TWO_RANK_CHECKSUMS[(workload, graph)]
    else:
        _TWO_RANK_CHECKSUMS[(workload, graph)] = c["parameter_checksum"]


def test_headline_batch_gemm_requests_on_every_covering_kernel():
    """VERDICT r4 #1b: round 3's streaming GEMM corrupted ~100 of 50 M elements per launch ONLY where a workgroup walks
    several tiles -- i.e. at the batch the headline is timed on, never at the batch-4 / batch-24 sizes the parity checks
    ran at (there the library picks other kernels).  Here every GEMM request of ONE forward + backward of the DEFAULT
    model at batch 768 (rnnp.py:88-96,146-161, net.py:663-666: M up to 777 216 rows) is replayed on fresh operands
      * on the library's own choice and on EVERY other kernel that covers it (tssep_gemm_f32_on): bit-identical results
        for single-pass requests (the family's contract: same k order, same epilogue arithmetic);
      * and on the exact-fp32 MFMA kernel (precision 0, gemm.hip -- different code, no operand split): within the
        split-bf16 bound, relative to the largest entry; split-K weight gradients compared after summing the partials."""
    import ctypes
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sweep_gemm_shapes as sw
    from tssep_amd import _lib, hip_ops
    old = hip_ops.GEMM_PRECISION
    hip_ops.GEMM_PRECISION = "bf16x3"
    try:
        m = sw.build(300, 320, 4)
        reqs = sw.requests_of_a_step(m, 4, 768)
    finally:
        hip_ops.GEMM_PRECISION = old
    del m
    torch.cuda.empty_cache()
    assert len(reqs) >= 20 and max(d["M"] for d, _ in reqs) == 768 * 4 * 253, len(reqs)
    nt_family = ("pipe", "tall2", "tall4", "tall4_xcol", "big", "big_p", "big_p320", "stream", "nt_w160")
    tn_family = ("pipe", "tn", "tn_tall", "tn_big", "tn_p320", "tn_w160", "tn_h160")
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    seen, bit_checks = set(), 0
    for d, _count in reqs:
        r = sw.Replay(d)
        S = max(d["splitk"], 1)

        def result(C):
            """what the consumer of this request reads: the tensor, or the sum of the split-K partials"""
            return C.view(S, -1).sum(0) if S > 1 else C
        got = r.prepare("auto")
        assert got is not None, d
        _call, C_auto, g_auto, first_auto = got
        if first_auto is None:          # an accumulating store: `prepare` launched once onto zeros, C IS the result
            first_auto = C_auto
        choice = hip_ops.gemm_plan(g_auto, "auto")
        seen.add(choice)
        # the exact-fp32 kernel on the same operands (it has no virtual ones column: those requests -- the weight gradients
        # that carry the bias gradient -- take the library's choice as the reference for the other kernels, and their
        # ones column is checked against the column sums of dY in fp64)
        C32 = torch.zeros(r.celems, device="cuda")
        g32 = r.args(C32)
        g32.precision = 0
        rc = L.tssep_gemm_f32(ctypes.byref(g32), st)
        if rc == 0:
            ref = result(C32)
        else:
            assert rc == -3 and d["b_ones_col"], (rc, d)
            ref = result(first_auto).clone()
            ldp = d["c_split_stride"] // d["M"] if S > 1 else d["ldc"]
            ones = ref.view(d["M"], ldp)[:, d["N"] - 1].double()
            want = r.A[:d["K"], :d["M"]].double().sum(0)
            assert float((ones - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-6, d
        scale = float(ref.abs().max())
        assert scale > 0 and bool(torch.isfinite(ref).all()), d
        err = float((result(first_auto) - ref).abs().max())
        assert err <= 1e-4 * scale + 1e-6, (choice, err, scale, d)
        fam = tn_family if d["a_kmajor"] else nt_family
        for k in fam:
            if k == choice:
                continue
            res = r.prepare(k)
            if res is None:
                continue
            seen.add(k)
            out_k = res[3] if res[3] is not None else res[1]
            if S == 1 and not (d["N"] % 256 == 1 or {k, choice} & {"tall4_xcol"}):
                assert torch.equal(torch.nan_to_num(out_k), torch.nan_to_num(first_auto)), (k, choice, d)
                bit_checks += 1
            else:
                err = float((result(out_k) - ref).abs().max())
                assert err <= 1e-4 * scale + 1e-6, (k, err, scale, d)
            del res
        del r, got, C_auto, first_auto, C32, ref
        torch.cuda.empty_cache()
    # the kernels of the headline step were all exercised at their headline shapes
    assert {"big_p", "big_p320", "tn_big", "tn_p320", "tn_w160", "tn_h160"} <= seen, seen
    assert bit_checks >= 20, bit_checks


@pytest.mark.parametrize("units,projs,K", [(300, 320, 4), (128, 256, 8), (512, 320, 4), (256, 256, 4)])
def test_gemm_shape_sweep_parity_and_interchangeable_kernels(units, projs, K):
    """another `units` / `projs` / speaker count (net.py:504-509) goes through the same dispatcher.  Per
    model size: the real model against the CPU oracle (masks, loss, gradients), then every GEMM request of a step
    replayed on the library's choice AND on every other kernel that covers it (tssep_gemm_f32_on) -- bit-identical
    results for single-pass requests; TFLOP/s per request are recorded (tools/sweep_gemm_shapes.py writes
    profiles/r4_gemm_shape_sweep.jsonl with the bench's batch)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sweep_gemm_shapes as sw
    from tssep_amd import hip_ops
    old = hip_ops.GEMM_PRECISION
    hip_ops.GEMM_PRECISION = "bf16x3"
    try:
        lines = sw.sweep([(units, projs, K)], B=24, reps=1)
    finally:
        hip_ops.GEMM_PRECISION = old
    assert len(lines) >= 20
    kernels = {l["choice"] for l in lines}
    assert kernels <= set(hip_ops.GEMM_KERNELS) - {"auto", "f32"} and len(kernels) >= 4, kernels
    for l in lines:
        assert l["tflops_choice"] > 0 and l["parity_vs_oracle"]["max_abs_mask_err"] < 1e-3
