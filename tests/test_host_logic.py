"""Host-side logic that needs no GPU: kernel selection policy, split-K choice, workspace sizes,
argument validation of the C ABI (status codes instead of launches)."""
import ctypes

import pytest
import torch

from tssep_amd import _lib, hip_ops as H


def test_recurrence_policy():
    old = H.RECURRENCE
    try:
        H.RECURRENCE = "auto"
        assert H.recurrence_kernel(8, 300, False) == "onchip"
        assert H.recurrence_kernel(1536, 300, False) == "onchip"
        assert H.recurrence_kernel(8, 300, True) == "cluster"
        assert H.recurrence_kernel(32, 300, True) == "cluster"
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
    # big-M forward GEMMs are never split; weight gradients (K = rows) are, in multiples of 8
    assert H.pick_splitk(194304, 2400, 513) == 1
    s = H.pick_splitk(2400, 513, 194304)
    assert s > 1 and s % 8 == 0 and s <= 64
    assert H.pick_splitk(24, 12, 35) == 1


def test_onchip_workspace_sizes():
    L = _lib.lib()
    assert L.tssep_lstm_onchip_supported(300) == 1 and L.tssep_lstm_onchip_supported(320) == 0
    f = int(L.tssep_lstm_onchip_xbuf_bytes(768, 300, 0))
    b = int(L.tssep_lstm_onchip_xbuf_bytes(768, 300, 1))
    # 48 work items x 2 slots x 5 workgroups x 32 sequences x (64 | 320) granules of 8 bytes + header
    assert f == 1024 + 48 * 2 * 5 * 32 * 64 * 8
    assert b == 1024 + 48 * 2 * 5 * 32 * 320 * 8
    assert int(L.tssep_lstm_onchip_pack_floats(300, 0)) > 0


def test_abi_argument_validation_without_gpu():
    """Invalid arguments are rejected with a status before anything is launched."""
    L = _lib.lib()
    g = _lib.GemmArgs()
    assert L.tssep_gemm_f32(ctypes.byref(g), None) < 0                      # null operands
    assert L.tssep_blstm_onchip_fwd(None, None, None, 0, 0, None, None, None, 1, 1, 300, 256, 0, None) < 0
    assert L.tssep_stft_frames(64000, 1024, 256, 1024, 1, 1) == 253
    assert L.tssep_stft_frames(480000, 1024, 256, 1024, 1, 1) == 1878


def test_side_stream_rows_policy():
    assert H.SIDE_STREAM_MAX_ROWS == 768 * 253
