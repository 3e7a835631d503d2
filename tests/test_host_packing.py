"""Host functions of the split-operand kernels (no GPU): the weight packers of include/atvsnet_hip.h that arrange a TF kernel as
two fp16 pieces per weight (w = g0 + g1 / 2048) in the lane order of
v_mfma_f32_16x16x32_f16, their size queries, and the argument checks the launch entry points make before they touch the
HIP runtime."""
import ctypes

import numpy as np
import pytest
import torch

from atvsnet_amd import _lib

OK, ERR_NULL, ERR_SHAPE = 0, -1, -2


def _f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def _pack(name, w, cin, cout):
    L = _lib.lib()
    pb = ctypes.c_long()
    assert getattr(L, name + '_pack_size')(cin, cout, ctypes.byref(pb)) == OK and pb.value > 0
    buf = np.full(pb.value, 0xAB, np.uint8)          # the packer must write every byte
    w = np.ascontiguousarray(w, np.float32)
    assert getattr(L, name + '_pack')(w.ctypes.data_as(ctypes.c_void_p), cin, cout, buf.ctypes.data_as(ctypes.c_void_p)) == OK
    return buf


def _check_pieces(p, want):
    """p: (2 pieces, ...) fp16 with the piece axis FIRST; want: the weights (float32) in the same arrangement.
    g0 = fp16(w) (round to nearest even), g1 = fp16((w - g0) * 2048) -- the kernels' own split (conv_xb.hip xb_split2) -- and
    g0 + g1 / 2048 is the weight to 2^-22."""
    g0, g1 = p
    want = want.astype(np.float32)
    assert np.array_equal(g0, want.astype(np.float16))
    assert np.array_equal(g1, ((want - g0.astype(np.float32)) * np.float32(2048.0)).astype(np.float16))
    back = g0.astype(np.float64) + g1.astype(np.float64) / 2048.0
    assert np.abs(back - want).max() <= np.abs(want).max() * 2.0 ** -21


@pytest.mark.parametrize('cin,cout', [(32, 32), (64, 128), (128, 64)])
def test_conv1x1_b_pack_layout(cin, cout):
    """atvs_conv1x1_b_pack: packed[chunk][tile n][piece][lane = q * 16 + co16][e] = piece of w[32 chunk + 8 q + e][16 n + co16],
    one zero chunk behind the last."""
    w = np.random.default_rng(cin + cout).standard_normal((cin, cout)).astype(np.float32)
    buf = _pack('atvs_conv1x1_b', w, cin, cout)
    nt = cout // 16
    p = buf.view(np.float16).reshape(cin // 32 + 1, nt, 2, 4, 16, 8)            # chunk, n, piece, q, co16, e
    assert not p[-1].any()
    want = w.reshape(cin // 32, 4, 8, nt, 16).transpose(0, 3, 1, 4, 2)              # chunk, n, q, co16, e
    _check_pieces(np.moveaxis(p[:-1], 2, 0), want)


@pytest.mark.parametrize('name,cin,cout', [('atvs_conv3d_b', 32, 32), ('atvs_conv3d_b', 64, 64), ('atvs_conv3d_b', 16, 32),
                                           ('atvs_conv3d_s2b', 16, 32), ('atvs_conv3d_s2b', 32, 64)])
def test_conv3d_b_pack_layout(name, cin, cout):
    """atvs_conv3d_b_pack / atvs_conv3d_s2b_pack: packed[chunk][step j][tile n][piece][lane = q * 16 + co16][e] = piece of
    w[tap 2 j + (q >> 1)][16 chunk + 8 (q & 1) + e][16 n + co16], zeros for the 28th tap, 16 zero bytes behind."""
    w = np.random.default_rng(cin * 3 + cout).standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
    buf = _pack(name, w, cin, cout)
    assert not buf[-16:].any()
    nt, nch = cout // 16, cin // 16
    p = buf[:-16].view(np.float16).reshape(nch, 14, nt, 2, 2, 2, 16, 8)          # chunk, j, n, piece, tap half, ci half, co16, e
    w28 = np.concatenate([w.reshape(27, cin, cout), np.zeros((1, cin, cout), np.float32)])
    want = w28.reshape(14, 2, nch, 2, 8, nt, 16).transpose(2, 0, 5, 1, 3, 6, 4)    # chunk, j, n, tap half, ci half, co16, e
    _check_pieces(np.moveaxis(p, 3, 0), want)


@pytest.mark.parametrize('cin,cout', [(16, 8), (32, 16), (64, 16)])
def test_deconv_up_b_pack_holds_every_weight_once(cin, cout):
    """atvs_deconv_up_b_pack: whatever the order (output classes x fragment steps), the two piece images are aligned, piece 0
    is the fp16 of a weight, piece 1 the fp16 of its scaled residual, and every weight of the [3,3,3,Cout,Cin] kernel appears at
    least once (classes re-use taps), zeros elsewhere, 16 zero bytes behind."""
    w = np.random.default_rng(cin + 7 * cout).standard_normal((3, 3, 3, cout, cin)).astype(np.float32)
    buf = _pack('atvs_deconv_up_b', w, cin, cout)
    assert not buf[-16:].any()
    vals = buf[:-16].view(np.float16).reshape(-1, 2, 64, 8)          # (chunk x step, piece, lane, e)
    g0 = w.astype(np.float16)
    g1 = ((w - g0.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    assert set(np.unique(g0.view(np.uint16)).tolist()) <= set(np.unique(vals[:, 0].view(np.uint16)).tolist())
    # piece 1 sits where piece 0 sits, and is the residual of THAT weight
    pairs = set(zip(g0.view(np.uint16).ravel().tolist(), g1.view(np.uint16).ravel().tolist())) | {(0, 0)}
    got = set(zip(vals[:, 0].view(np.uint16).ravel().tolist(), vals[:, 1].view(np.uint16).ravel().tolist()))
    assert got <= pairs
    assert np.count_nonzero(vals[:, 0]) >= w.size


def test_split16_entry_points_check_their_arguments_first():
    """supported() truth tables, size queries and NULL / shape checks (all return before any launch)."""
    L = _lib.lib()
    pb = ctypes.c_long()
    assert L.atvs_conv1x1_b_supported(128, 128) == 1 and L.atvs_conv1x1_b_supported(48, 128) == 0
    assert L.atvs_conv1x1_b_supported(128, 16) == 0
    assert L.atvs_conv3d_b_supported(32, 32) == 1 and L.atvs_conv3d_b_supported(8, 32) == 0
    assert L.atvs_conv3d_b_supported(32, 16) == 0 and L.atvs_conv3d_s2b_supported(16, 32) == 1
    assert L.atvs_conv3d_s2b_supported(8, 16) == 0
    assert L.atvs_deconv_up_b_supported(16, 8) == 1 and L.atvs_deconv_up_b_supported(16, 32) == 0
    assert L.atvs_deconv_up_b_supported(24, 8) == 0
    for name, bad in (('atvs_conv1x1_b', (48, 128)), ('atvs_conv3d_b', (8, 32)), ('atvs_conv3d_s2b', (16, 48)),
                      ('atvs_deconv_up_b', (16, 32))):
        assert getattr(L, name + '_pack_size')(bad[0], bad[1], ctypes.byref(pb)) == ERR_SHAPE
        assert getattr(L, name + '_pack_size')(32, 32, None) == ERR_NULL
        assert getattr(L, name + '_pack')(None, 32, 32, None) == ERR_NULL
    n = None
    assert L.atvs_conv3d_b_f32(n, n, n, n, n, 1, 8, 8, 8, 32, 32, 32, 0, 0, n) == ERR_NULL
    assert L.atvs_conv3d_s2b_f32(n, n, n, n, n, 1, 8, 8, 8, 16, 32, 32, 0, 0, n) == ERR_NULL
    assert L.atvs_deconv_up_b_f32(n, n, n, n, 1, 8, 8, 8, 16, 8, 8, 0, 0, 16, 0, n) == ERR_NULL
    assert L.atvs_conv1x1_b_f32(n, n, n, n, n, 0, n, n, 1, ctypes.c_long(64), 32, 32, 32, 0, 0, n) == ERR_NULL
    assert L.atvs_refine_stems_f32(n, n, n, n, n, n, n, n, 1, 8, 8, 32, n) == ERR_NULL
    assert L.atvs_conv3d_8to1(n, n, n, 1, 8, 8, 8, n) == ERR_NULL
    one = (ctypes.c_float * 4)()
    ptr = ctypes.cast(one, ctypes.c_void_p)
    # valid pointers, unsupported shapes: rejected by the checks in front of the launch
    assert L.atvs_conv3d_b_f32(ptr, ptr, n, ptr, n, 1, 8, 8, 8, 8, 32, 32, 0, 0, n) == ERR_SHAPE
    assert L.atvs_conv3d_s2b_f32(ptr, ptr, n, ptr, n, 1, 8, 8, 8, 16, 48, 48, 0, 0, n) == ERR_SHAPE
    assert L.atvs_deconv_up_b_f32(ptr, ptr, ptr, n, 1, 8, 8, 8, 16, 32, 32, 0, 0, 32, 0, n) == ERR_SHAPE
    assert L.atvs_conv1x1_b_f32(ptr, ptr, n, n, n, 0, ptr, n, 1, ctypes.c_long(64), 48, 32, 32, 0, 0, n) == ERR_SHAPE
    assert L.atvs_refine_stems_f32(ptr, ptr, n, ptr, ptr, ptr, ptr, n, 0, 8, 8, 32, n) == ERR_SHAPE
