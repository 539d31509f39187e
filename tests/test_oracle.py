"""Pin the CPU oracle: against fixtures generated from the reference classes
(tests/golden/make_golden.py) and against the reference's doctest numbers."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import data as odata, features, loss as oloss, model as omodel, net as onet
from oracle import enhancer as oenh, rnnp as ornnp, stft as ostft

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
T = torch.as_tensor


def _params(g, pre="p."):
    return {k[len(pre):]: T(v) for k, v in g.items() if k.startswith(pre)}


def test_stft_doctest_numbers():
    # tssep/train/feature_extractor.py:197-202
    rng = np.random.RandomState(0)
    f = features.log1p_max_norm_abs(ostft.stft(rng.uniform(0, 1, size=10_000), window="blackman"))
    assert f.shape == (43, 513)
    assert np.mean(f) == pytest.approx(0.03461471931132962, rel=1e-12)
    assert np.min(f) == pytest.approx(1.0003006801514706e-06, rel=1e-9)
    assert np.max(f) == 1.0
    assert np.std(f) == pytest.approx(0.051645387514742555, rel=1e-12)
    assert ostft.num_frames(80000) == 316          # tssep/train/model.py:480
    # feature_extractor.py:194-196
    k = features.log1p_max_norm_abs(np.array([[1, 5], [3 + 4j, -5]]))
    np.testing.assert_allclose(k, [[0.29539453, 1.0], [1.0, 1.0]], atol=1e-8)


def test_istft_inverts_stft_and_matches_autograd_adjoint():
    x = torch.randn(2, 3, 3000, dtype=torch.float64)
    X = ostft.stft(x)
    assert (ostft.istft(X, num_samples=3000) - x).abs().max() < 1e-12
    # <istft(X), y> == <X, istft^T y> is what the HIP backward implements
    X = torch.randn(1, 9, 513, dtype=torch.complex128, requires_grad=True)
    y = torch.randn(1, 1500, dtype=torch.float64)
    (ostft.istft(X, num_samples=1500) * y).sum().backward()
    assert X.grad.shape == X.shape


def test_rnnp_against_reference_fixture(golden):
    g = golden("rnnp")
    p = _params(g)
    for tag in ("x3", "x4", "x2"):
        x = T(g[tag]).requires_grad_()
        pp = {k: v.clone().requires_grad_() for k, v in p.items()}
        y = ornnp.rnnp(x, pp, "")
        np.testing.assert_allclose(y.detach().numpy(), g[tag + "_y"], rtol=1e-5, atol=1e-6)
        yf = ornnp.rnnp(x, pp, "", fast=True)
        np.testing.assert_allclose(yf.detach().numpy(), g[tag + "_y"], rtol=1e-5, atol=1e-6)
        (y * T(g[tag + "_g"])).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), g[tag + "_dx"], rtol=1e-4, atol=1e-6)
        for k, v in pp.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"{tag}_dp.{k}"], rtol=1e-4, atol=1e-6)


ME_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "me_*.npz")))


@pytest.mark.parametrize("name", ME_CASES)
def test_mask_estimator_against_reference_fixture(golden, name):
    g = golden(name)
    comb, ts_vad, res, nap = [str(s) for s in g["cfg"]]
    ts_vad = False if ts_vad == "False" else int(ts_vad)
    p = {"mask_estimator." + k: v.requires_grad_() for k, v in _params(g).items()}
    out = onet.mask_estimator_forward(
        p, T(g["xs"]), T(g["aux"]), odim=9, combination=comb, ts_vad=ts_vad,
        output_resolution=res, num_averaged_permutations=int(nap), perm=g["perm"])
    np.testing.assert_allclose(out["logit"].detach().numpy(), g["logit"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(out["mask"].detach().numpy(), g["mask"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["embedding"].numpy(), g["embedding"])
    (out["mask"] * T(g["g"])).sum().backward()
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), g["dp." + k[len("mask_estimator."):]],
                                   rtol=2e-4, atol=2e-6, err_msg=k)


def test_mask_estimator_consumes_np_random_like_reference(golden):
    # net.py:824-826: one np.random.permutation(K) per batch entry, batch order
    g = golden("me_mul_4_tf_1")
    p = {"mask_estimator." + k: v for k, v in _params(g).items()}
    np.random.seed(100 + ME_CASES_ORDER.index("me_mul_4_tf_1"))
    torch.manual_seed(0)
    # the generator seeds np.random and then builds the module (no np draws) -> same stream
    out = onet.mask_estimator_forward(p, T(g["xs"]), T(g["aux"]), odim=9, combination="mul",
                                      ts_vad=4, output_resolution="tf")
    np.testing.assert_array_equal(out["perm"], g["perm"])
    np.testing.assert_allclose(out["logit"].numpy(), g["logit"], rtol=1e-5, atol=2e-6)


# generation order of make_golden.py (seed = 100 + case index)
ME_CASES_ORDER = [f"me_{c}_{v}_{r}_{n}" for c in ("mul", "cat") for v in (False, 3, 4)
                  for r in ("t", "tf") for n in (1, 2) if not (v is False and n != 1)] + \
                 [f"me_{c}_8_{r}_{n}" for c in ("mul", "cat") for r in ("t", "tf") for n in (1, 2)]


def test_enhancer_and_losses_against_reference_fixture(golden):
    g = golden("enh_loss")
    np.testing.assert_allclose(oloss.masking(T(g["mask"]), T(g["Obs"])).numpy(), g["est"], rtol=1e-6)
    np.testing.assert_allclose(oloss.log_mae(T(g["e"]), T(g["t"])).numpy(), g["logmae"], rtol=1e-6)
    np.testing.assert_allclose(oloss.mae(T(g["e"]), T(g["t"])).numpy(), g["mae"], rtol=1e-6)
    np.testing.assert_allclose(oloss.vad_sigmoid_bce(T(g["logit"]), T(g["vad"])).numpy(), g["bce"],
                               rtol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_torch_bf_against_reference_fixture(golden, tag):
    """TorchBF('mvdr_souden') of the reference (enhancer.py:215-265): target + interference masks,
    target mask only (batched, masking), an eps that clamps the trace, 8 channels."""
    g = golden("torch_bf")
    eps, masking, masking_eps = g[tag + "_kw"]
    got = oenh.torch_bf(g[tag + "_m"], g[tag + "_Y"], int(g[tag + "_ref"]),
                        eps=None if eps < 0 else float(eps), masking=bool(masking),
                        masking_eps=float(masking_eps))
    assert got.dtype == np.complex128 and got.shape == g[tag + "_out"].shape
    np.testing.assert_allclose(got, g[tag + "_out"], rtol=1e-9, atol=1e-12)


def test_torch_bf_trace_and_invariances(golden):
    g = golden("torch_bf")
    x = np.arange(1.0, 10.0).reshape(3, 3)                       # enhancer.py:107-125
    assert oenh.trace(x) == g["trace_3x3"] == 15.0
    np.testing.assert_array_equal(oenh.trace(x.reshape(3, 1, 3), axis1=0, axis2=2), g["trace_axes"])
    # Souden MVDR is invariant to a rescaling of either PSD matrix, i.e. of either mask
    m, Y = g["a_m"].astype(np.float64), g["a_Y"]
    base = oenh.torch_bf(m, Y, 1)
    m2 = m.copy(); m2[:, 0] *= 3.0; m2[:, 1] *= 0.25
    np.testing.assert_allclose(oenh.torch_bf(m2, Y, 1), base, rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        oenh.torch_bf(np.ones((2, 3, 4, 5)), Y, 0)


def test_loss_doctest_known_answers(golden):
    # tssep/train/loss.py:198-204, 223-234, 286-299
    torch.manual_seed(0)
    target = torch.rand((2, 10000))
    estimate = target + 0.5 * torch.rand((2, 10000))
    assert float(oloss.mae(estimate, target)) == pytest.approx(0.5018, abs(5e-5))
    assert float(oloss.log_mae(estimate, target)) == pytest.approx(-0.2995, abs=5e-5)
    assert float(oloss.log_mae(target, target)) == -np.inf
    estimate[1, :] = 0
    target[1, :] = 0
    assert float(oloss.log_mae(estimate, target)) == pytest.approx(-0.5980, abs=5e-5)
    k = golden("kat_loss")
    assert float(k["logmae"]) == pytest.approx(-0.2995, abs=5e-5)
    # BCE with the target already in VAD form (prepare_target is off the hot path)
    torch.manual_seed(0)
    tgt = torch.rand((2, 100, 257))
    vad = (abs(tgt).sum(-1) / abs(tgt).sum(-1).amax(-1, keepdim=True) > 0.05).float()
    est = ((abs(tgt) > 0.05).float() - 0.5) * 10
    assert float(oloss.vad_sigmoid_bce(est, vad)) == pytest.approx(float(k["bce10"]), rel=1e-5)


def test_vad2sep_broadcast(golden):
    g = golden("vad2sep")
    vad = _params(g, "vad.")
    sep = _params(g, "sep.")
    b = oloss.vad2sep_broadcast(vad, {k: v.shape for k, v in sep.items()})
    for k in sep:
        np.testing.assert_array_equal(b[k].numpy(), sep[k].numpy())
    kw = dict(odim=9, combination="mul", ts_vad=4, random_speaker_order=False)
    lv = onet.mask_estimator_forward(vad, T(g["xs"]), T(g["aux"]), output_resolution="t", **kw)
    ls = onet.mask_estimator_forward(b, T(g["xs"]), T(g["aux"]), output_resolution="tf", **kw)
    np.testing.assert_allclose(lv["logit"].numpy(), g["logit_vad"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ls["logit"].numpy(), g["logit_sep"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ls["logit"].numpy(), lv["logit"].numpy(), atol=1e-6)


def test_dummy_reader(golden):
    g = golden("dummy_reader")
    ex = odata.dummy_example(1, sample_rate=64, aux_size=20)
    np.testing.assert_array_equal(ex["observation"], g["obs"])
    np.testing.assert_array_equal(ex["speaker_reverberation_early_ch0"], g["early"])
    np.testing.assert_array_equal(ex["vad"], g["vad"])
    np.testing.assert_array_equal(ex["auxInput"], g["aux"])
    np.testing.assert_array_equal(odata.get_vad(71, 8), g["vad71"])
    for i in range(2):
        e = odata.dummy_example(i)
        s = [e["observation"].astype(np.float64).sum(),
             np.abs(e["speaker_reverberation_early_ch0"]).astype(np.float64).sum(),
             e["auxInput"].sum()]
        np.testing.assert_allclose(s, g[f"sum16_{i}"], rtol=1e-12)


def test_end_to_end_known_answer():
    # tssep/train/model.py:552-575: 114038 parameters, validate_LogMAE
    # 0.74156505 / 0.744494, ||Input|| 58.8257, std 0.0960, max 1.
    np.random.seed(0)
    torch.manual_seed(0)
    p = omodel.init_mask_estimator_params(idim=513, odim=513, units=10, projs=12,
                                          combination="cat", aux_size=100)
    assert sum(v.numel() for v in p.values()) == 114038
    exs = [odata.dummy_example(s) for s in (0, 1)]
    obs = T(np.stack([e["observation"] for e in exs]))
    aux = T(np.stack([e["auxInput"] for e in exs]))
    tgt = T(np.stack([e["speaker_reverberation_early_ch0"] for e in exs]))
    cfg = dict(odim=513, combination="cat", ts_vad=False, output_resolution="tf")
    o = omodel.forward_loss(p, obs, aux, tgt, cfg=cfg, mfcc=False, fast=True)
    np.testing.assert_allclose(o["loss"].numpy(), [0.74156505, 0.744494], rtol=2e-5)
    assert float(torch.norm(o["Input"])) == pytest.approx(58.8257, abs=2e-3)
    assert float(torch.std(o["Input"])) == pytest.approx(0.0960, abs=5e-5)
    assert float(o["Input"].abs().amax()) == 1.0


def test_mfcc_restatement_properties():
    # torchaudio is absent: parity unpinned.  Structural facts SURVEY App. A.2 records.
    fb, dct = features.mfcc_tables(1024)
    assert fb.shape == (513, 40) and dct.shape == (40, 40)
    assert int((fb.sum(0) == 0).sum()) == 7          # f_max wraps to 15600 Hz > Nyquist
    np.testing.assert_allclose((dct.t() @ dct).numpy(), np.eye(40), atol=1e-5)  # ortho
    X = torch.randn(2, 6, 513, dtype=torch.complex64)
    m3 = features.torch_mfcc(X, fb, dct)
    assert m3.shape == (2, 6, 40)
    # 3-D input: the top_db floor is taken over the whole batch (torchaudio quirk)
    db = 10 * torch.log10(torch.clamp((abs(X) ** 2) @ fb, min=1e-10))
    ref = torch.clamp(db, min=float(db.max()) - 80.0) @ dct
    np.testing.assert_allclose(m3.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_mfcc_restatement_against_independent_paths():
    """torchaudio 2.0.2 is absent (parity unpinned, oracle/features.py header); what CAN be checked is checked
    against independent routes: the DCT table against scipy's DCT-II, the mel filterbank against a
    per-filter float64 construction from the HTK formula, the whole transform against power -> filters -> dB ->
    scipy DCT, and AmplitudeToDB's batching rule (tssep/train/feature_extractor_torchaudio.py:93-106 feeds a 3-D
    [B, n_mels, T] tensor, which torchaudio treats as the channels of ONE item: one floor for the whole batch; a
    2-D input gets its own floor)."""
    import scipy.fft
    fb, dct = features.mfcc_tables(1024)
    # create_dct(n_mfcc, n_mels, 'ortho')[n, k] = DCT-II basis: x @ dct == scipy.fft.dct(x, type=2, norm='ortho')
    eye = np.eye(40)
    np.testing.assert_allclose(dct.numpy(), scipy.fft.dct(eye, type=2, norm="ortho", axis=-1), atol=3e-6)
    x = np.random.RandomState(0).randn(5, 40)
    np.testing.assert_allclose(x @ dct.double().numpy(), scipy.fft.dct(x, type=2, norm="ortho", axis=-1), atol=2e-5)
    # HTK mel filters, one triangle at a time in float64: centres equally spaced in mel between 40 Hz and
    # 16000 - 400 = 15600 Hz (the negative f_max wraps, feature_extractor_torchaudio.py:57-60), bins at k * 8000 / 512
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)         # noqa: E731
    inv = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)       # noqa: E731
    pts = inv(np.linspace(mel(40.0), mel(15600.0), 42))
    freqs = np.arange(513) * (8000.0 / 512)
    want = np.zeros((513, 40))
    for m in range(40):
        lo, c, hi = pts[m:m + 3]
        want[:, m] = np.maximum(0.0, np.minimum((freqs - lo) / (c - lo), (hi - freqs) / (hi - c)))
    np.testing.assert_allclose(fb.numpy(), want, atol=2e-5)
    dead = np.nonzero(want.sum(0) == 0)[0]
    assert dead.tolist() == list(range(33, 40))                  # 7 filters lie wholly above Nyquist (App. A.2)
    assert (fb.numpy()[:, dead] == 0).all() and pts[33] >= 8000.0 > pts[32]      # a filter is dead when its lower edge is at or above Nyquist
    # the whole transform by the independent route, float64
    rng = np.random.RandomState(1)
    X = (rng.randn(3, 9, 513) + 1j * rng.randn(3, 9, 513)) * np.array([1.0, 1e-2, 3e-5])[:, None, None]
    got = features.torch_mfcc(torch.as_tensor(X.astype(np.complex64)), fb, dct).numpy()
    db = 10.0 * np.log10(np.maximum(((np.abs(X.astype(np.complex64)).astype(np.float64)) ** 2) @ want, 1e-10))
    floor_batch = db.max() - 80.0                                # 3-D input: ONE floor for the batch
    ref = scipy.fft.dct(np.maximum(db, floor_batch), type=2, norm="ortho", axis=-1)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-3)
    assert (db[2] < floor_batch).all() and (db[0] > floor_batch).any()      # the quiet utterance is all floor
    np.testing.assert_allclose(got[2, :, 1:], 0.0, atol=2e-3)               # a constant's DCT: only c0 is non-zero
    # 2-D input (one utterance): its own floor -> the quiet utterance is NOT flattened
    got2 = features.torch_mfcc(torch.as_tensor(X[2].astype(np.complex64)), fb, dct).numpy()
    ref2 = scipy.fft.dct(np.maximum(db[2], db[2].max() - 80.0), type=2, norm="ortho", axis=-1)
    np.testing.assert_allclose(got2, ref2, rtol=2e-4, atol=2e-3)
    assert np.abs(got2[:, 1:]).max() > 1.0


def test_torch_mfcc_against_an_independent_third_party_implementation():
    """SURVEY 8a2 / VERDICT r5 "missing" #2: torchaudio 2.0.2 (what `TorchMFCC` wraps, feature_extractor_torchaudio.py:57-106)
    is not in the image and the reference holds no MFCC value, so the oracle's MFCC stays "parity unpinned" by the rules.
    What CAN be checked without it (round 6): the whole chain against code nobody here wrote -- the mel filterbank and the
    dB conversion of `transformers.audio_utils` (HuggingFace's numpy port of torchaudio's `melscale_fbanks` /
    librosa's `power_to_db`, installed in this image) and scipy's DCT-II: |STFT|^2 -> mel (HTK and Slaney scales, with
    and without Slaney area normalisation, the reference's f_max = sample_rate - 400 quirk) -> 10 log10 with the 80-dB
    floor below the maximum of the WHOLE 3-D batch (torchaudio's batching rule) -> orthonormal DCT, on the FFT sizes
    the configurations use (1024 shipped, 400 TorchMFCC's default, 512)."""
    import warnings
    import pytest
    import scipy.fft
    au = pytest.importorskip("transformers.audio_utils")
    from oracle import features as ofeat
    rng = np.random.RandomState(5)
    for size in (1024, 400, 512):
        F = size // 2 + 1
        X = torch.as_tensor((rng.randn(3, 37, F) + 1j * rng.randn(3, 37, F)).astype(np.complex64)) * \
            torch.as_tensor(10.0 ** rng.uniform(-4, 1, size=(3, 37, 1)).astype(np.float32))       # 100 dB of level spread: the floor bites
        for scale, norm in (("htk", None), ("slaney", None), ("slaney", "slaney"), ("htk", "slaney")):
            fb, dct = ofeat.mfcc_tables(size, mel_scale=scale, mel_norm=norm)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")               # (f_max above Nyquist leaves empty filters: the reference's quirk)
                ref_fb = au.mel_filter_bank(F, 40, 40.0, 16000.0 - 400.0, 16000, norm=norm, mel_scale=scale)
            assert float(np.abs(fb.numpy() - ref_fb).max()) < 5e-6, (size, scale, norm)
            power = np.abs(X.numpy().astype(np.complex128)) ** 2                                  # [B, T, F]
            mel = np.einsum("btf,fm->bmt", power, ref_fb.astype(np.float64))                      # [B, n_mels, T]
            db = au.power_to_db(mel, reference=1.0, min_value=1e-10, db_range=80.0)               # max over the whole array
            ref = scipy.fft.dct(db, type=2, norm="ortho", axis=1).transpose(0, 2, 1)              # [B, T, n_mfcc]
            got = ofeat.torch_mfcc(X, fb, dct).numpy()
            assert got.shape == ref.shape == (3, 37, 40)
            assert float(np.abs(got - ref).max()) < 1e-3, (size, scale, norm)      # (fp32 against fp64: 2e-4 at values up to 270)
            assert float(db.min()) == pytest.approx(float(db.max()) - 80.0, abs=1e-9)             # the floor was active
