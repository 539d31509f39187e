"""Host-side logic that needs no GPU: kernel selection policy, split-K choice, workspace sizes,
argument validation of the C ABI (status codes instead of launches)."""
import ctypes

import numpy as np
import pytest
import torch

from tssep_amd import _lib, hip_ops as H


def test_recurrence_policy():
    old = H.RECURRENCE
    try:
        H.RECURRENCE = "auto"
        assert H.recurrence_kernel(8, 300, False) == "onchip"
        assert H.recurrence_kernel(1536, 300, False) == "onchip"
        assert H.recurrence_kernel(8, 300, True) == "onchip"
        assert H.recurrence_kernel(32, 300, True) == "onchip"
        # no cliff at 2 046 frames any more (the step field of the exchange tags wraps); the limit is the
        # 32-bit lane offset into the gate tensor -- 14 913 frames for the 16-sequence kernels, 7 215 for the 32-sequence
        # ones (what this GPU-less process assumes) -- and a fallback beyond it is announced
        assert H.recurrence_kernel(32, 300, True, T=1878) == "onchip" and H.recurrence_kernel(32, 300, True, T=4000) == "onchip"
        L = _lib.lib()
        assert L.tssep_lstm_onchip_max_steps(300, 16) == 14913 and L.tssep_lstm_onchip_max_steps(300, 32) == 7215
        import pytest
        with pytest.warns(RuntimeWarning, match="streaming fp32 kernel runs instead"):
            assert H.recurrence_kernel(32, 300, True, T=20000) == "stream"
        assert H.recurrence_kernel(33, 300, True) == "onchip"
        assert H.recurrence_kernel(768, 300, True) == "onchip"
        # small or unsupported hidden sizes fall back to the streaming kernels
        assert H.recurrence_kernel(64, 12, False) == "stream"
        assert H.recurrence_kernel(64, 512, False) == "stream"
        H.RECURRENCE = "stream"
        assert H.recurrence_kernel(8, 300, False) == "stream"
        H.RECURRENCE = "onchip"
        assert H.recurrence_kernel(8, 512, True) == "stream"       # not supported -> fallback
    finally:
        H.RECURRENCE = old


def test_splitk_choice():
    """tssep_gemm_wgrad_splits (host-only: runs without a GPU): the split count follows the kernel the library's own
    dispatcher picks for the weight gradient (tssep_gemm_plan), never a second copy of its conditions."""
    # weight gradients (K = rows) are split in multiples of 8; tiny ones are not
    s = H.pick_splitk(2400, 513, 194304)
    assert s > 1 and s % 8 == 0 and s <= 64
    assert H.pick_splitk(24, 12, 35) == 1
    # the big-tile weight-gradient kernel (512 x 128 tiles, one workgroup per CU, a K slab per XCD): as many slabs per
    # XCD as fill its 32 CUs best -- 15 tiles -> 2 slabs (30 / 32); N = 514 = 4 x 128 + 2 and 1281 = 10 x 128 + 1 keep
    # their last columns on the VALU of the last full column tile: 20 and 50 tiles -> 3 slabs (60 / 64, 150 / 160)
    old = H.GEMM_PRECISION
    H.GEMM_PRECISION = "bf16x3"
    try:
        # round 5: the dW_ih GEMMs whose input width is 320 q or 256 q (+ 1) (+ the ones column) run the eight-wave
        # workgroups of tn_w160 (256 x 320 / 256 x 256 tiles, one per CU, a K slab per XCD): the smallest multiple of 8
        # within 7 % of the best fill of whole rounds of 32 -- 10, 20 and 40 tiles -> 3 slabs (30 / 32, 60 / 64, 120 / 128)
        assert H.pick_splitk(2400, 321, 777216, ones_col=True) == 24 and H.pick_splitk(2400, 514, 777216, ones_col=True) == 24
        assert H.pick_splitk(2400, 1281, 194304, ones_col=True) == 24
        # the pre-net's N = 553 + 1: a ragged second 320-column tile (640 columns, as many as five 128-wide tiles), column 552
        # and the ones column on the VALU: 20 tiles -> 3 slabs; N = 873 + 1 (960 against 896 columns) stays on the 512 x 128 tile
        assert H.pick_splitk(2400, 554, 194304, ones_col=True) == 24
        # dW_hh: both 160-column tiles in one 256 x 320 workgroup: 5 tiles -> 6 slabs per XCD (30 / 32)
        assert H.pick_splitk(1200, 300, 777216, shifted=True) == 48 and H.pick_splitk(1200, 300, 194304, shifted=True) == 48
        # the projection weight gradients (M = 320): round 5, with enough rows, the eight-wave workgroups on the SWAPPED problem
        # (three 256 x 320 tiles -> 80 splits, 30 workgroups per XCD); short K: the 320 x 128 tile, 5 column tiles, two per CU
        assert H.pick_splitk(320, 601, 777216, ones_col=True) == 80 and H.pick_splitk(320, 601, 25600, ones_col=True) == 96
        assert H.pick_splitk(640, 601, 777216, ones_col=True) == 32              # (M = 640 fits the 128-row tiles: 25 tiles)
        # one 128 x 128 tile with a long K never gets more splits than it has K-tile groups of 8
        assert H.pick_splitk(100, 100, 16 * 512) == 64 and H.pick_splitk(100, 100, 16 * 2048) <= 256
    finally:
        H.GEMM_PRECISION = old
    # the exact-fp32 GEMM has one kernel: the general rule
    s = H.pick_splitk(2400, 321, 777216)
    assert s >= 8 and s % 8 == 0


def test_wgrad_split_count_and_kernel_are_a_fixed_point_over_many_shapes():
    """ADVICE r4 + round 5's new dispatch rules: for a grid of weight-gradient requests (the model's shapes, their
    neighbours, other unit / projection / speaker counts, with and without the ones column and the time shift) the split
    count the library recommends is one the kernel it then plans ACCEPTS (same kernel at 8 and at S splits, or a second
    round settled it), a multiple of 8 for the kernels that keep a K slab on one XCD, never more than one split per 8 K
    tiles -- host-only (plan queries)."""
    import ctypes
    from tssep_amd._lib import GemmArgs
    old = H.GEMM_PRECISION
    H.GEMM_PRECISION = "bf16x3"
    try:
        def plan_at(M, N, K, S, ones, shifted):
            g = GemmArgs()
            g.A = g.B = 0x1000
            g.C = 0x2000
            g.M, g.N, g.K = M, N, K
            g.lda, g.ldb, g.a_kmajor, g.b_kmajor = H.round_up(M, 4), H.round_up(N, 4), 1, 1
            g.ldc = H.round_up(N, 4)
            g.splitk, g.b_ones_col, g.precision = S, int(ones), 1
            g.c_split_stride = M * g.ldc
            if shifted:
                g.b_kshift, g.kperiod = -1, 253
            name = H.gemm_plan(g, "auto")
            # the split count the rule OF THAT kernel gives the request (tssep_gemm_wgrad_split_rule, ABI 4)
            rule = int(H._lib.lib().tssep_gemm_wgrad_split_rule(ctypes.byref(g), H.GEMM_KERNELS[name])) if name else None
            return name, rule

        n = fallbacks = 0
        for K in (8 * 253, 32 * 253, 320 * 253, 768 * 253, 3072 * 253):
            for M in (320, 513, 640, 1200, 1280, 2052, 2400, 4104):
                for N, ones in ((300, False), (301, True), (321, True), (320, False), (514, True), (554, True), (557, True), (601, True),
                                (600, False), (1281, True), (2561, True), (874, True), (130, False)):
                    for shifted in (False, True):
                        if shifted and ones:
                            continue
                        S = H.pick_splitk(M, N, K, shifted=shifted, ones_col=ones)
                        ktiles = (K + 15) // 16
                        assert 1 <= S <= max(1, ktiles // 8) or S == 8, (M, N, K, S)
                        k_at_s, rule_s = plan_at(M, N, K, S, ones, shifted)
                        assert k_at_s is not None, (M, N, K, S)
                        # the kernel planned AT S is the kernel whose rule produced S (the former check
                        # compared two identical queries) -- or the library found no fixed point and fell back to 8
                        if rule_s != S:
                            assert S == 8, (k_at_s, M, N, K, S, rule_s)
                            fallbacks += 1
                        if k_at_s in ("tn_big", "tn_p320", "tn_w160", "tn_h160") and ktiles >= 64 * 8:
                            assert S % 8 == 0, (k_at_s, M, N, K, S)
                        n += 1
        assert n > 500 and fallbacks <= n // 20, (n, fallbacks)
    finally:
        H.GEMM_PRECISION = old


def test_gemm_plan_names_the_kernel_without_a_gpu():
    """tssep_gemm_plan: the library's choice is a function of the request alone (no environment variable, VERDICT r3
    #3) -- the shapes of the default step land on the kernels DESIGN 4.1 names."""
    import ctypes
    from tssep_amd._lib import GemmArgs

    def plan(M, N, K, wgrad=False, shifted=False, act=0, remap=False, ones=False, prec=1, force="auto"):
        g = GemmArgs()
        g.A = g.B = 0x1000
        g.C = 0x2000
        g.M, g.N, g.K = M, N, K
        if wgrad:
            g.lda, g.ldb, g.a_kmajor, g.b_kmajor = H.round_up(M, 4), H.round_up(N, 4), 1, 1
            g.splitk, g.b_ones_col = 8, int(ones)
            if shifted:
                g.b_kshift, g.kperiod = -1, 253
        else:
            g.lda = g.ldb = H.round_up(K, 4)
        g.ldc = H.round_up(N, 4)
        g.act, g.precision = act, prec
        if remap:
            g.c_remap, g.c_T, g.c_K, g.c_sb, g.c_sk, g.c_st, g.ldc = 1, 253, 4, 253 * 4 * N, N, 4 * N, 0
        return H.gemm_plan(g, force)

    R = 768 * 253
    # input projections: the persistent big tile (plain / bias / Tanh store) since round 4's second half, any K
    assert plan(R, 2400, 556) == "big_p" and plan(4 * R, 2400, 516) == "big_p" and plan(4 * R, 2400, 320) == "big_p"
    assert plan(R, 1280, 2400, remap=True) == "big_p"                                # ... also with a remapped store (< 2 GB)
    assert plan(R, 1280, 2400, remap=True, act=1) == "big_p" and plan(4 * R, 5120, 2400, remap=True) == "big"      # (4 GB: 64-bit addresses)
    assert plan(R, 513, 600) == "big_p" and plan(R + 8, 513, 600) == "big"           # pre-net projection: 2 tiles + a VALU column
    # N = 320 (Tanh projections, with and without the speaker combination) and N = 600: the 192 x 320 persistent tile
    assert plan(4 * R, 320, 600, act=1) == "big_p320" and plan(4 * R, 320, 600, act=1, remap=True) == "big_p320"
    assert plan(4 * R, 600, 320) == "big_p320" and plan(4 * R, 1000, 320) == "big_p"  # dgrad proj dh: 640 columns computed for 600
    assert plan(4 * R, 780, 320) == "stream"                                         # ... 1024 for 780 is too many
    # round 5: the dW_ih GEMMs on the eight-wave workgroups of tn_w160 wherever 320-wide column tiles pad no more than 128-wide ones
    assert plan(2400, 557, R, wgrad=True, ones=True) == "tn_w160" and plan(2400, 554, R, wgrad=True, ones=True) == "tn_w160"
    assert plan(2400, 514, 4 * R, wgrad=True, ones=True) == "tn_w160" and plan(2400, 321, 4 * R, wgrad=True, ones=True) == "tn_w160"
    assert plan(2400, 1281, R, wgrad=True, ones=True) == "tn_w160" and plan(2400, 874, R, wgrad=True, ones=True) == "tn_big"
    assert plan(1200, 300, 4 * R, wgrad=True, shifted=True) == "tn_w160"
    assert plan(320, 601, 4 * R, wgrad=True, ones=True) == "tn_w160" and plan(320, 601, 100 * 253, wgrad=True, ones=True) == "tn_h160"
    assert plan(513, 601, R, wgrad=True, ones=True) == "tn"
    assert plan(100, 50, 30) == "pipe" and plan(100, 50, 30, prec=0) == "f32"
    # round 4, from the shape sweep (profiles/r4_gemm_shape_sweep*.jsonl): the logit layer (N = 4 x 513, K = projs) on the
    # 160-wide tile, its weight gradient (M = 2052) and the 8-speaker one (4104) on the big weight-gradient tile, one /
    # two column tiles (projs = 256) on the big tile, never the eight-wave 256 x 256 tile below K = 448
    assert plan(R, 2052, 320, remap=True) == "big_p" and plan(2052, 320, R, wgrad=True) == "tn_w160"      # (round 5: 2304 rows of 256-row tiles)
    assert plan(1800, 320, R, wgrad=True) == "tn_p320"                               # (13.8 % padding on 256-row tiles: the 192-row tile)
    assert plan(4104, 256, R // 2, wgrad=True) == "tn_big"
    assert plan(4 * R, 256, 1024, act=1) == "big_p" and plan(4 * R, 256, 256, act=1) == "big_p"
    assert plan(R // 2, 4104, 256, remap=True) == "tall2"
    # occupancy: the 8-utterance shard of the 8-GPU configuration (2024 / 8096 rows) takes the 128 x 128 tiles where the big
    # tiles cannot fill the chip; 380 big tiles (1.48 resident rounds) go to the persistent kernel
    assert plan(8096, 320, 2400, act=1) == "pipe" and plan(2024, 2400, 1280) == "stream" and plan(2024, 513, 600) == "pipe"
    assert plan(1200, 300, 2024, wgrad=True, shifted=True) in ("tn", "tn_tall") and plan(R // 2, 256, 512) == "stream"
    # naming a kernel: honoured when it covers the request, refused (None) when it does not
    assert plan(4 * R, 2400, 320, force="big") == "big" and plan(4 * R, 2400, 320, force="tall4") == "tall4"
    assert plan(4 * R, 320, 600, act=1, force="stream") is None                      # no Tanh in the streaming kernel
    assert plan(100, 50, 30, force="big") is None and plan(100, 50, 30, prec=0, force="pipe") is None
    L = _lib.lib()
    assert L.tssep_gemm_kernel_name(6) == b"big" and L.tssep_gemm_kernel_name(14) == b"big_p" and L.tssep_gemm_kernel_name(99) == b"?"


def test_onchip_workspace_sizes():
    L = _lib.lib()
    assert L.tssep_lstm_onchip_supported(300) == 1 and L.tssep_lstm_onchip_supported(320) == 0
    f = int(L.tssep_lstm_onchip_xbuf_bytes(768, 300, 0))
    b = int(L.tssep_lstm_onchip_xbuf_bytes(768, 300, 1))
    # 48 work items x 2 slots x 5 workgroups x 32 sequences x (64 | 320) values, two per 8-byte granule,
    # + header: the backward's per-XCD working set (6 clusters) is 2.5 MB, inside the 4-MB L2
    assert f == 1024 + 48 * 2 * 5 * 32 * 64 * 4
    assert b == 1024 + 48 * 2 * 5 * 32 * 320 * 4
    assert (b - 1024) // 8 < 4 << 20
    assert int(L.tssep_lstm_onchip_pack_floats(300, 0)) > 0


def test_interleaved_recurrence_group_choice():
    """tssep_blstm_onchip16_groups: two groups of 16 sequences per cluster where that divides the number of groups and still
    gives each of the 48 XCD-local clusters (256 CUs, 5 workgroups each) a bundle, else one; 0 where the kernel does not
    apply.  (Four groups stay available through `groups`; two are faster since they request their operands early.)"""
    L = _lib.lib()
    assert L.tssep_blstm_onchip16_groups(3072, 300, 256) == 2          # 384 items / 2 = 192 bundles = 4 per cluster
    assert L.tssep_blstm_onchip16_groups(768, 300, 256) == 2           # 96 items / 2 = 48 bundles: one per cluster
    assert L.tssep_blstm_onchip16_groups(1536, 300, 256) == 2
    assert L.tssep_blstm_onchip16_groups(384, 300, 256) == 1           # 48 items: two groups would leave half the clusters idle
    assert L.tssep_blstm_onchip16_groups(200, 300, 256) == 1           # 13 groups of 16: nothing else divides
    assert L.tssep_blstm_onchip16_groups(768, 302, 256) == 0 and L.tssep_blstm_onchip16_groups(768, 400, 256) == 0
    assert L.tssep_blstm_onchip16_groups(768, 300, 16) == 0            # not even one cluster per XCD
    assert int(L.tssep_lstm_onchip16_xbuf_bytes(3072, 300)) == 1024 + 2 * 192 * 2 * 5 * 16 * 64 * 4
    assert int(L.tssep_lstm_onchip16_pack_floats(300)) == 2 * 5 * 8 * 2 * 10 * 2 * 64 * 4
    # argument validation of the launcher without a GPU
    assert L.tssep_blstm_onchip16_fwd(None, None, None, 0, 0, None, None, None, 1, 1, 300, 256, 0, 0, None) < 0


def test_abi_argument_validation_without_gpu():
    """Invalid arguments are rejected with a status before anything is launched."""
    L = _lib.lib()
    g = _lib.GemmArgs()
    assert L.tssep_gemm_f32(ctypes.byref(g), None) < 0                      # null operands
    assert L.tssep_blstm_onchip_fwd(None, None, None, 0, 0, None, None, None, 1, 1, 300, 256, 0, None) < 0
    assert L.tssep_stft_frames(64000, 1024, 256, 1024, 1, 1) == 253
    assert L.tssep_stft_frames(480000, 1024, 256, 1024, 1, 1) == 1878


def test_fft_plans_without_gpu():
    """tssep_stft_plan / tssep_fft_twiddles are host-only: 1 = the 1024 / 256 plan of the shipped configs, 2 = the general
    plan (even sizes, size / 2 = 2^a 3^b 5^c <= 2048, shift <= min(size, 512)), 0 = not built; the twiddle tables of both
    plans share one layout ([size / 2] exp(-2 pi i k / (size / 2)), then [size / 2 + 1] exp(-2 pi i k / size))."""
    L = _lib.lib()
    assert L.tssep_stft_plan(1024, 256) == 1
    for size, shift in ((512, 128), (400, 200), (256, 64), (2048, 512), (960, 240), (1000, 250), (60, 20), (1024, 128)):
        assert L.tssep_stft_plan(size, shift) == 2, (size, shift)
    for size, shift in ((1022, 256), (514, 128), (1023, 256), (4098, 512), (2048, 1024), (512, 600), (14, 7), (512, 0)):
        assert L.tssep_stft_plan(size, shift) == 0, (size, shift)
    for size in (1024, 400, 60):
        nh = size // 2
        buf = np.zeros(2 * (nh + nh + 1), dtype=np.float32)
        assert L.tssep_fft_twiddles(size, buf.ctypes.data_as(ctypes.c_void_p)) == 0
        tw = buf.view(np.complex64)
        np.testing.assert_allclose(tw[:nh], np.exp(-2j * np.pi * np.arange(nh) / nh), atol=1e-6)
        np.testing.assert_allclose(tw[nh:], np.exp(-2j * np.pi * np.arange(nh + 1) / size), atol=1e-6)
    buf = np.zeros(4, dtype=np.float32)
    assert L.tssep_fft_twiddles(1022, buf.ctypes.data_as(ctypes.c_void_p)) < 0


def test_side_stream_rows_policy():
    assert H.SIDE_STREAM_MAX_SEQS == 512


def test_runtime_policy_is_recorded_not_read_from_the_environment(monkeypatch):
    """the host layer's policy (arithmetic, recurrence family, folds, side stream, graph replay) lives in
    tssep_amd.train.runtime -- defaults = what bench.py measures (split-bf16 GEMMs), deviations come from `eg.runtime` in
    the YAML and are frozen into config.yaml; hip_ops reads no TSSEP_* environment variable."""
    import inspect
    from tssep_amd.train import experiment, runtime
    src = inspect.getsource(H)
    assert "os.environ" not in src and "getenv" not in src and "import os" not in src
    assert runtime.defaults()["gemm_precision"] == "bf16x3" == H.GEMM_PRECISION
    assert runtime.current() == runtime.defaults()
    monkeypatch.setenv("TSSEP_GEMM_PRECISION", "f32")            # ignored: not a channel any more
    assert runtime.current()["gemm_precision"] == "bf16x3"
    with runtime.applied(gemm_precision="f32", fold_tail=0, onchip16_bwd=0):
        assert (H.GEMM_PRECISION, H.FOLD_TAIL, H.ONCHIP16_BWD) == ("f32", 0, False)
    assert runtime.current() == runtime.defaults()
    with pytest.raises(KeyError):
        runtime.apply(gemm_precison="f32")                       # a typo must not train on the default silently
    with pytest.raises(ValueError):
        runtime.apply(gemm_precision="fp8")
    assert runtime.parse_overrides(["gemm_precision=f32", "side_stream_max_seqs=0", "fold_tanh=false"]) == \
        dict(gemm_precision="f32", side_stream_max_seqs=0, fold_tanh=False)
    # the frozen configuration carries the COMPLETE policy, with the experiment's deviations
    cfg = experiment.Experiment.get_config({"runtime": {"gemm_precision": "f32"}})
    assert cfg["runtime"] == dict(runtime.defaults(), gemm_precision="f32")
    assert experiment.Experiment.get_config({})["runtime"] == runtime.defaults()


def test_lazy_dataset_stages_and_threaded_prefetch():
    """tssep_amd.dataset: the lazy_dataset calls of model.py:182-337 (map / shuffle(reshuffle) / batch /
    prefetch / catch): order-preserving threads, a new order per pass, exceptions at their position."""
    import time
    import numpy as np
    from tssep_amd import dataset as D
    ds = D.new([{"i": i} for i in range(10)])
    seen = []

    def f(ex):
        seen.append(ex["i"])
        return {"i": ex["i"], "sq": ex["i"] ** 2}

    chain = (ds.map(f).shuffle(reshuffle=True, rng=np.random.RandomState(0)).batch(3)
             .map(lambda b: [e["sq"] for e in b]).prefetch(3, 6))
    a, b = list(chain), list(chain)
    assert [len(x) for x in a] == [3, 3, 3, 1] and len(chain) == 4
    assert sorted(sum(a, [])) == sorted(sum(b, [])) == [i * i for i in range(10)] and a != b
    assert chain[:1] and len(chain[:1]) == 1

    def slow(ex):
        time.sleep(0.02 * (5 - ex["i"] % 5))
        return ex["i"]

    t0 = time.time()
    assert list(ds.map(slow).prefetch(4, 8)) == list(range(10))          # in order ...
    assert time.time() - t0 < 0.45                                        # ... and in parallel (0.6 s serial)

    def bad(ex):
        if ex["i"] == 4:
            raise D.FilterException()
        if ex["i"] == 7:
            raise ValueError("boom")
        return ex["i"]

    got = []
    with pytest.raises(ValueError):
        for v in ds.map(bad).prefetch(2, 4, catch_filter_exception=True):
            got.append(v)
    assert got == [0, 1, 2, 3, 5, 6]
    assert list(ds.map(bad).catch((D.FilterException, ValueError))) == [0, 1, 2, 3, 5, 6, 8, 9]
    with pytest.raises(TypeError):
        len(ds.catch())
    fixed = ds.shuffle(reshuffle=False, rng=np.random.RandomState(1))
    assert list(fixed) == list(fixed)
    assert [e["i"] for e in ds.sort(lambda e: -e["i"])] == list(range(9, -1, -1))


def test_prepare_dataset_follows_the_reference_stages():
    """Model.prepare_dataset (model.py:182-337) on the DummyReader: collate shapes, reshuffle per epoch
    when training, sorted test run, unbatched validation examples."""
    from tssep_amd.data import DummyReader
    from tssep_amd.train import enhancer, loss, model
    m = model.Model(fe=None, reader=DummyReader(sample_rate=800, train_examples=7), mask_estimator=None,
                    enhancer=enhancer.Masking(), loss=loss.LogMAE())
    m.train()
    np.random.seed(0)
    ds = m.prepare_train_dataset(device=None, batch_size=3)
    e1, e2 = [ex["example_id"] for ex in ds], [ex["example_id"] for ex in ds]
    assert [len(b) for b in e1] == [3, 3, 1] and sorted(sum(e1, [])) == sorted(sum(e2, [])) and e1 != e2
    ex = ds[:1][0]
    assert ex["observation"].shape == (3, 1, 4000) and ex["speaker_reverberation_early_ch0"].shape == (3, 8, 4000)
    assert ex["auxInput"].shape == (3, 8, 100) and ex["reference_channel"] == 0
    val = m.prepare_validate_dataset(device=None, batch_size=None, prefetch=False)
    assert [e["example_id"] for e in val] == [f"dummy_id_{i}" for i in range(4)]
    srt = m.prepare_dataset("train", None, training=True, sort=True, batch_size=2, prefetch=False)
    assert [e["example_id"] for e in srt][0] == ["dummy_id_0", "dummy_id_1"]
