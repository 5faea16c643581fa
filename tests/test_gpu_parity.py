"""GPU parity tests (-m gpu): every call goes through the C ABI of libbasic_dsp_hip.so and is
compared with the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): bit-exact for index moves and single-rounding elementwise
ops; rel-L2 <= 1e-6 for f32 FFT/convolution against the f64 oracle; 1e-12 for f64.
"""
import os

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu

bd = pytest.importorskip("basic_dsp_amd")
from basic_dsp_amd import DspVec  # noqa: E402
from basic_dsp_amd import vector as V  # noqa: E402


def _as_real(a):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        return a.astype(np.complex128).view(np.float64)
    return a.astype(np.float64)


def rel_l2(got, ref):
    got, ref = _as_real(got), _as_real(ref)
    return np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)


def tol_for(dtype):
    return 1e-6 if dtype == np.float32 else 1e-12


def test_gpu_present_and_native_library_loaded():
    assert bd.lib.bdsp_hip_has_gpu_support_f32() == 1, bd.last_error()
    assert bd.lib.bdsp_hip_has_gpu_support_f64() == 1


# ------------------------------------------------------------------ config C1: scale + offset
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1000, 65536, 100003])
def test_real_scale_offset_bit_exact(n, dtype):
    # BASELINE config 1: real DspVec, scale(2.5) then offset(-1.25); elementary.rs:283-342
    x = orc.fill_uniform(n, 201511141, -10, 10, dtype)
    v = DspVec(x)
    assert v.scale(2.5) == 0 and v.offset(-1.25) == 0
    ref = orc.real_offset(orc.real_scale(x, 2.5), -1.25)
    assert np.array_equal(v.data(), ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_complex_elementwise_bit_exact(dtype):
    x = orc.fill_uniform(2 * 5001, 7, -10, 10, dtype)
    y = orc.fill_uniform(2 * 5001, 8, -10, 10, dtype)
    v = DspVec(x, is_complex=True)
    assert v.scale(complex(0.5, -1.5)) == 0
    assert np.array_equal(v.data(), orc.complex_scale(x, 0.5, -1.5))
    v = DspVec(x, is_complex=True)
    assert v.offset(complex(3.0, -2.0)) == 0
    assert np.array_equal(v.data(), orc.complex_offset(x, 3.0, -2.0))
    v = DspVec(x, is_complex=True)
    assert v.offset(1.5) == 0  # real offset on a complex vector adds (f, 0): elementary.rs:291-297
    assert np.array_equal(v.data(), orc.real_offset(x, 1.5, True))
    v = DspVec(x, is_complex=True)
    assert v.conj() == 0
    assert np.array_equal(v.data(), orc.conj(x))
    for op, name in enumerate(["add", "sub", "mul", "div"]):
        for cplx in (True, False):
            a, b = DspVec(x, is_complex=cplx), DspVec(y, is_complex=cplx)
            assert getattr(a, name)(b) == 0
            code, ref = orc.binary(x, y, cplx, op)
            assert code == 0
            if name == "div" and cplx:
                np.testing.assert_allclose(a.data(), ref, rtol=4 * np.finfo(dtype).eps)
            else:
                assert np.array_equal(a.data(), ref), (name, cplx)


def test_binary_op_error_codes():
    a = DspVec(np.zeros(8, np.float32), is_complex=True)
    b = DspVec(np.zeros(6, np.float32), is_complex=True)
    assert a.mul(b) == 1                      # InputMustHaveTheSameSize (elementary.rs:392)
    c = DspVec(np.zeros(8, np.float32), is_complex=False)
    assert a.add(c) == 2                      # InputMetaDataMustAgree (number space)
    d = DspVec(np.zeros(8, np.float32), is_complex=True, delta=2.0)
    assert a.add(d) == 2                      # delta ratio outside 0.9..1.1
    e = DspVec(np.zeros(8, np.float32), is_complex=True, domain=V.FREQ)
    assert a.add(e) == 2                      # domain differs


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_complex_to_real_maps(dtype):
    x = orc.fill_uniform(2 * 4099, 11, -10, 10, dtype)
    eps = np.finfo(dtype).eps
    for kind, name in enumerate(["magnitude", "magnitude_squared", "to_real", "to_imag", "phase"]):
        v = DspVec(x, is_complex=True)
        assert getattr(v, name)() == 0
        assert not v.is_complex() and len(v) == 4099
        ref = orc.complex_to_real(x, kind)
        if name in ("to_real", "to_imag", "magnitude_squared"):
            assert np.array_equal(v.data(), ref)
        else:
            np.testing.assert_allclose(v.data(), ref, rtol=4 * eps, atol=4 * eps)
    r = DspVec(x[:10], is_complex=False)
    assert r.magnitude() == -1 and r.is_erroneous()  # assert_complex! poisons (complex_to_real.rs:352-362)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_multiply_complex_exponential(dtype):
    x = orc.fill_uniform(2 * 3000, 13, -10, 10, dtype)
    v = DspVec(x, is_complex=True, delta=0.5)
    assert v.multiply_complex_exponential(0.02, 0.3) == 0
    k = np.arange(3000)
    # a and b are multiplied by delta in T first (complex_ops.rs:83-84)
    a, b = float(dtype(0.02) * dtype(0.5)), float(dtype(0.3) * dtype(0.5))
    ref = x.astype(np.float64).view(np.complex128) * np.exp(1j * (a * k + b))
    assert rel_l2(v.data(), ref.view(np.float64)) < (2e-7 if dtype == np.float32 else 1e-14)
    # and it tracks the reference's running product to the reference's own accuracy
    refrun = orc.multiply_complex_exponential(x, 0.02, 0.3, 0.5)
    assert rel_l2(v.data(), refrun) < (5e-5 if dtype == np.float32 else 1e-11)


# ------------------------------------------------------------------ index moves (bit-exact)
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("points", [1, 2, 9, 10, 4097, 65536])
def test_swap_halves_and_shifts(points, cplx):
    e = 2 if cplx else 1
    x = orc.fill_uniform(points * e, 3, -10, 10, np.float32)
    for name, fwd in (("swap_halves", True), ("fft_shift", True), ("ifft_shift", False)):
        v = DspVec(x, is_complex=cplx, domain=V.FREQ)
        assert getattr(v, name)() == 0
        assert np.array_equal(v.data(), orc.swap_halves(x, cplx, fwd)), (name, points)
    v = DspVec(x, is_complex=cplx)
    assert v.reverse() == 0
    assert np.array_equal(v.data(), orc.reverse(x, cplx))


def test_zero_pad_interleave_mirror():
    for cplx in (False, True):
        e = 2 if cplx else 1
        for n, p in ((10, 24), (11, 20), (5, 6)):
            x = orc.fill_uniform(n * e, n, -10, 10, np.float64)
            for opt in (V.PAD_END, V.PAD_SURROUND, V.PAD_CENTER):
                v = DspVec(x, is_complex=cplx)
                assert v.zero_pad(p, opt) == 0
                code, ref = orc.zero_pad(x, cplx, p, opt, buffered=True)
                assert code == 0 and np.array_equal(v.data(), ref), (cplx, n, p, opt)
            v = DspVec(x, is_complex=cplx)
            assert v.zero_pad(n, V.PAD_END) == 7  # InvalidArgumentLength
        x = orc.fill_uniform(7 * e, 5, -1, 1, np.float32)
        v = DspVec(x, is_complex=cplx)
        assert v.zero_interleave(3) == 0
        assert np.array_equal(v.data(), orc.zero_interleave(x, cplx, 3))
    x = orc.fill_uniform(2 * 6, 9, -1, 1, np.float32)
    v = DspVec(x, is_complex=True, domain=V.FREQ)
    assert v.mirror() == 0
    assert np.array_equal(v.data(), orc.mirror(x))
    r = DspVec(x[:6], is_complex=False)
    assert r.to_complex() == 0 and r.is_complex()
    assert np.array_equal(r.data(), orc.zero_interleave(x[:6], False, 2))


# ------------------------------------------------------------------ windows
def _ulp_of_10(dtype):
    return float(np.spacing(np.asarray(10.0, dtype=dtype)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("wid", [0, 1, 2, 3, 4])
def test_windows(wid, dtype):
    # window_functions.rs:26-132 applied by time.rs:32-66; inputs are in [-10, 10).  Triangular, Hamming and Hann
    # are held to 4 ulp of 10 (same formula in T; libm and device cos differ by an ulp or two of the value).  The
    # four-term Blackman-Harris sum cancels to 6e-5 at its ends, where an ulp of the cosines is 1e-3 of the window
    # value: 30 ulp of 10 there.
    ulps = 30 if wid == 2 else 4
    for cplx, points in ((True, 1000), (False, 1001), (True, 65537)):
        e = 2 if cplx else 1
        x = orc.fill_uniform(points * e, 17, -10, 10, dtype)
        oid, alpha = (1, 0.5) if wid == 4 else (wid, 0.54)
        v = DspVec(x, is_complex=cplx)
        assert v.apply_window(wid) == 0
        ref = orc.apply_window(x, cplx, oid, alpha)
        np.testing.assert_allclose(v.data(), ref, rtol=0, atol=ulps * _ulp_of_10(dtype))
        # unapply_window divides by the same window (time.rs:50-66): against the oracle's division of the
        # oracle's product, and back to the input wherever the window is not tiny
        assert v.unapply_window(wid) == 0
        back = orc.apply_window(ref, cplx, oid, alpha, unapply=True)
        w = orc.apply_window(np.ones_like(x), cplx, oid, alpha).astype(np.float64)
        ok = np.abs(w) > 1e-2
        assert ok.sum() > 0.7 * ok.size  # (Blackman-Harris is below 1e-2 over its outer 10 % on either side)
        got = v.data().astype(np.float64)
        np.testing.assert_allclose(got[ok], back.astype(np.float64)[ok], rtol=0,
                                   atol=4 * ulps * _ulp_of_10(dtype) / 1e-2)
        assert rel_l2(got[ok], x.astype(np.float64)[ok]) < (3e-6 if dtype == np.float32 else 1e-14)


# ------------------------------------------------------------------ FFT
FFT_SIZES = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536,
             1 << 17, 1 << 20, 1 << 21]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", FFT_SIZES)
def test_plain_fft_and_ifft_pow2(n, dtype):
    x = orc.fill_uniform(2 * n, 201511212 + n, -10, 10, dtype)
    ref = np.fft.fft(x.astype(np.float64).view(np.complex128))
    v = DspVec(x, is_complex=True, delta=0.25)
    assert v.plain_fft() == 0
    assert v.domain() == V.FREQ and v.is_complex() and v.points() == n
    assert rel_l2(v.datac(), ref) < tol_for(dtype), n
    assert v.delta() == pytest.approx(0.25 * n)  # time_freq/mod.rs:54-55
    assert v.plain_ifft() == 0
    assert v.domain() == V.TIME
    got = v.data().astype(np.float64) / n
    assert rel_l2(got, x) < 2 * tol_for(dtype)


def test_fft_matches_oracle_restatement_small():
    # the oracle's own FFT (pinned to the Octave golden vectors) agrees with numpy's, so numpy can
    # stand in at the large sizes above
    for n in (64, 1000, 4096):
        x = orc.fill_uniform(2 * n, n, -10, 10, np.float64)
        assert rel_l2(orc.fft(x), np.fft.fft(x.view(np.complex128)).view(np.float64)) < 1e-13


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [3, 5, 6, 7, 12, 100, 1000, 1023, 4097, 10000, 12289, 100000])
def test_fft_any_length_bluestein(n, dtype):
    x = orc.fill_uniform(2 * n, 99 + n, -10, 10, dtype)
    ref = np.fft.fft(x.astype(np.float64).view(np.complex128))
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    assert rel_l2(v.datac(), ref) < tol_for(dtype), n  # (north_star: 1e-6 rel for one f32 transform; measured 1.5e-7 ... 7e-7)
    assert v.plain_ifft() == 0
    assert rel_l2(v.data().astype(np.float64) / n, x) < 2 * tol_for(dtype)  # (two transforms)


def test_golden_fft_vector64_on_gpu():
    # tests/time_freq_test.rs:46-120 and :123-197 (Octave golden vectors), through the C ABI
    import json
    import os
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
    n = np.arange(64, dtype=np.float64)
    sig = np.cos(2 * np.pi * 0.1 * n + 0.25)
    cplx = np.zeros(128)
    cplx[0::2] = sig
    v = DspVec(cplx, is_complex=True)
    assert v.fft() == 0 and v.magnitude() == 0
    np.testing.assert_allclose(v.data(), kats["fft_vector64"]["arrays"][-1], atol=1e-6)
    v = DspVec(cplx, is_complex=True)
    assert v.windowed_fft(V.WINDOW_HAMMING) == 0 and v.magnitude() == 0
    np.testing.assert_allclose(v.data(), kats["windowed_fft_vector64"]["arrays"][-1], atol=1e-6)
    # real input is zero-interleaved first (time_to_freq.rs:147-150)
    v = DspVec(sig, is_complex=False)
    assert v.fft() == 0 and v.magnitude() == 0
    np.testing.assert_allclose(v.data(), kats["fft_vector64"]["arrays"][-1], atol=1e-6)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [64, 1000, 4096, 8192, 16384, 1 << 20, 1 << 21])
def test_fft_ifft_with_fused_shift_window_scale(n, dtype):
    x = orc.fill_uniform(2 * n, 5 + n, -10, 10, dtype)
    xd = x.astype(np.float64)
    # fft = plain_fft + fft_shift (time_to_freq.rs:158-165)
    v = DspVec(x, is_complex=True)
    assert v.fft() == 0
    ref = orc.swap_halves(orc.fft(xd), True, True)
    assert rel_l2(v.data(), ref) < tol_for(dtype)
    # ifft = scale(1/n) -> ifft_shift -> plain_ifft (freq_to_time.rs:160-168); round trip
    assert v.ifft() == 0
    assert rel_l2(v.data(), xd) < 2 * tol_for(dtype)
    # windowed_fft (Hann) then windowed_ifft restores the signal except where the window is ~0
    v = DspVec(x, is_complex=True)
    assert v.windowed_fft(V.WINDOW_HANN) == 0
    ref = orc.swap_halves(orc.fft(orc.apply_window(xd, True, 1, 0.5)), True, True)
    assert rel_l2(v.data(), ref) < tol_for(dtype)
    v = DspVec(x, is_complex=True)
    assert v.windowed_fft(V.WINDOW_HAMMING) == 0 and v.windowed_ifft(V.WINDOW_HAMMING) == 0
    assert rel_l2(v.data(), xd) < (2e-5 if dtype == np.float32 else 1e-10)


def test_fft_type_state_errors():
    v = DspVec(np.zeros(16, np.float32), is_complex=True, domain=V.FREQ)
    assert v.plain_fft() == -1 and v.is_erroneous()   # time_to_freq.rs:140-145
    v = DspVec(np.zeros(16, np.float32), is_complex=True, domain=V.TIME)
    assert v.plain_ifft() == -1 and v.is_erroneous()  # freq_to_time.rs:142-147


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_b1_fft_host_slices(dtype):
    # GpuSupport::fft (gpu_support/mod.rs:35): in place on a host slice, unnormalised both ways.  The sizes walk through every
    # staging route of round 6: kernels reading and writing the pinned stage (one-kernel powers of two incl. 8192, two-pass
    # ones, smooth lengths resident -- 1000 and 3000 in the register-resident kernel -- and four-step: 5000, 10 000), pinned
    # copies (chirp-z: 100, 1009, 100 003), the 1 MiB limit (131 072 f32 points) and the pageable path above it
    for n in (16, 4096, 8192, 16384, 1 << 16, 100, 1000, 1009, 3000, 5000, 10000, 100003, 1 << 17, 1 << 18, 1 << 21):
        x = orc.fill_uniform(2 * n, n, -10, 10, dtype)
        got = V.gpu_fft(x.copy())
        ref = np.fft.fft(x.astype(np.float64).view(np.complex128)).view(np.float64)
        assert rel_l2(got, ref) < tol_for(dtype)
        back = V.gpu_fft(got.copy(), inverse=True)
        assert rel_l2(back.astype(np.float64) / n, x) < 2 * tol_for(dtype)


# ------------------------------------------------------------------ convolution
CONV_CASES = [(100, 6), (1000, 17), (5000, 64), (12288, 33), (4096, 1), (4097, 1024), (10000, 1025),
              (65536, 1024), (3073 * 3, 1024), (50000, 2), (20000, 257),
              # the block kernel stores whole 256-point rows from row ceil((M-1)/256) on: tap counts either side of
              # every row boundary, the largest filter it takes (3073), vectors shorter than one block, lengths that
              # leave 1 / V-1 outputs in the last block (V = 4096 - 256 ceil((M-1)/256))
              (9000, 256), (9000, 258), (7000, 513), (30000, 769), (30000, 2049), (40000, 3073), (3073, 3073),
              (3840 * 2 + 1, 200), (3840 * 3 - 1, 129), (3072 * 4, 1024), (3072 * 4 + 1, 1024), (2, 2), (1, 1)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,m", CONV_CASES)
def test_convolve_signal_complex_vs_direct_oracle(n, m, dtype):
    x = orc.fill_uniform(2 * n, 201601171 + n, -10, 10, dtype)
    h = orc.fill_uniform(2 * m, 201601172 + m, -1, 1, dtype) / dtype(m)
    ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), True)
    v, hv = DspVec(x, is_complex=True), DspVec(h, is_complex=True)
    assert v.convolve_signal(hv) == 0
    assert len(v) == 2 * n
    assert rel_l2(v.data(), ref) < tol_for(dtype), (n, m)


@pytest.mark.parametrize("taps", [5, 300, 1024])
def test_convolve_batch_through_device_api_matches_single_vectors(taps):
    """bdsp_hip_dev_convolve on a batch (what the C5 shard and the matrix API use): the block space of all vectors is
    one index range inside the kernel; every vector must equal its own single-vector convolution, wrap-around blocks
    included, and vectors of a length that is not a multiple of the block step."""
    import ctypes as C
    import torch
    lib = bd.lib
    sp = bd._lib.torch_stream_arg()
    for n, nvec in ((10000, 7), (3072 * 5 + 17, 3), (100, 5)):
        if taps > n:
            continue
        rows = np.stack([orc.fill_uniform(2 * n, 31 + r + n, -10, 10, np.float32) for r in range(nvec)])
        h = (orc.fill_uniform(2 * taps, 77 + taps, -1, 1, np.float32) / np.float32(taps)).astype(np.float32)
        dx, dh = torch.from_numpy(rows).cuda(), torch.from_numpy(h).cuda()
        dy = torch.empty_like(dx)
        assert lib.bdsp_hip_dev_convolve(0, dx.data_ptr(), dy.data_ptr(), n, nvec, dh.data_ptr(), taps, sp) == 0
        got = dy.cpu().numpy()
        for r in range(nvec):
            ref = orc.convolve_direct(rows[r].astype(np.float64), h.astype(np.float64), True)
            assert rel_l2(got[r], ref) < 1e-6, (n, nvec, r)
        # the prepared-spectrum entry point takes the same path with the delay applied as a linear phase
        spec = torch.empty(2 * lib.bdsp_hip_conv_spectrum_points(), device="cuda", dtype=torch.float32)
        assert lib.bdsp_hip_dev_conv_prepare(0, dh.data_ptr(), taps, spec.data_ptr(), sp) == 0
        dy.zero_()
        assert lib.bdsp_hip_dev_convolve_prepared(0, dx.data_ptr(), dy.data_ptr(), n, nvec, spec.data_ptr(), taps, sp) == 0
        got2 = dy.cpu().numpy()
        assert rel_l2(got2, got) < 1e-6


def test_convolve_signal_kats_on_gpu():
    # convolution.rs:819-842 shift identities incl. wrap-around, :885-898 overlap_discard == scalar
    a = np.zeros(20, np.float32)
    a[0::2] = np.arange(10)
    b = np.zeros(20, np.float32)
    b[8] = 1.0
    v, hv = DspVec(a, is_complex=True), DspVec(b, is_complex=True)
    assert v.convolve_signal(hv) == 0 and v.magnitude() == 0
    np.testing.assert_allclose(v.data(), np.arange(10), atol=1e-4)
    b = np.zeros(6, np.float32)
    b[4] = 1.0
    v, hv = DspVec(a, is_complex=True), DspVec(b, is_complex=True)
    assert v.convolve_signal(hv) == 0 and v.magnitude() == 0
    np.testing.assert_allclose(v.data(), [9, 0, 1, 2, 3, 4, 5, 6, 7, 8], atol=1e-4)
    a = np.zeros(200, np.float32)
    a[0::2] = np.arange(100)
    taps = np.zeros(12, np.float32)
    taps[0::2] = [0.1, 0.2, 0.3, 0.5, 0.1, 0.2]
    v, hv = DspVec(a, is_complex=True), DspVec(taps, is_complex=True)
    assert v.convolve_signal(hv) == 0
    code, ref, _ = orc.convolve_signal(a, taps, True)
    np.testing.assert_allclose(v.data(), ref, atol=1e-4 * 50)


def test_convolve_signal_real_and_errors():
    x = orc.fill_uniform(5000, 3, -10, 10, np.float32)
    h = orc.fill_uniform(31, 4, -1, 1, np.float32)
    v, hv = DspVec(x), DspVec(h)
    assert v.convolve_signal(hv) == 0 and not v.is_complex()
    ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), False)
    assert rel_l2(v.data(), ref) < 1e-6
    assert DspVec(x).convolve_signal(DspVec(x, is_complex=True)) == 2     # meta data must agree
    assert DspVec(x, domain=V.FREQ).convolve_signal(DspVec(h, domain=V.FREQ)) == 5  # must be time
    assert DspVec(h).convolve_signal(DspVec(x)) == 7                      # points < imp points


REAL_CONV_CASES = [(100, 6), (5000, 31), (4097, 1024), (3840 + 1, 200), (3840 * 2, 200), (3840 * 2 + 1, 129),
                   (3840 * 3 - 1, 129), (3072 * 5, 1024), (3072 * 5 + 7, 1025), (30000, 769), (40000, 3073), (9000, 258),
                   (3073, 3073), (2, 2), (1, 1)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,m", REAL_CONV_CASES)
def test_convolve_signal_real_vs_direct_oracle(n, m, dtype):
    """Real vectors with real taps: two real blocks share one complex transform pair in the block kernel.  Odd and
    even numbers of real blocks (the last pair half empty), lengths one sample past / short of a block boundary,
    tap counts either side of a 256-row boundary, vectors shorter than a block."""
    x = orc.fill_uniform(n, 201601181 + n, -10, 10, dtype)
    h = orc.fill_uniform(m, 201601182 + m, -1, 1, dtype) / dtype(m)
    ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), False)
    v, hv = DspVec(x), DspVec(h)
    assert v.convolve_signal(hv) == 0 and not v.is_complex() and len(v) == n
    assert rel_l2(v.data(), ref) < tol_for(dtype), (n, m)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_b1_gpu_convolve_vector(dtype):
    # ocl/mod.rs:549-564 compares inside the returned range; ours covers the whole vector
    x = orc.fill_uniform(2 * 20000, 1, -10, 10, dtype)
    h = orc.fill_uniform(2 * 100, 2, -1, 1, dtype)
    target, rng = V.gpu_convolve_vector(x, h, True)
    assert rng == (0, x.size)
    ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), True)
    assert rel_l2(target, ref) < tol_for(dtype)
    xr, hr = x[:15000], h[:77]
    target, rng = V.gpu_convolve_vector(xr, hr, False)
    assert rng == (0, xr.size)
    assert rel_l2(target, orc.convolve_direct(xr.astype(np.float64), hr.astype(np.float64), False)) < tol_for(dtype)
    assert V.gpu_convolve_vector(h, x, True) == (None, None)  # declines taps longer than the signal
    # the other staging routes of round 6: long taps (pinned copies, the multi-pass long-filter path), a signal above the
    # 1 MiB staging limit (pageable copies), real data above it
    for n, m, cplx in ((30000, 4000, True), (200000, 300, True), (400000, 500, False)):
        e = 2 if cplx else 1
        xx = orc.fill_uniform(e * n, 3 + n, -10, 10, dtype)
        hh = orc.fill_uniform(e * m, 4 + m, -1, 1, dtype) / dtype(m)
        target, rng = V.gpu_convolve_vector(xx, hh, cplx)
        assert rng == (0, xx.size)
        v = DspVec(xx, is_complex=cplx)
        assert v.convolve_signal(DspVec(hh, is_complex=cplx)) == 0
        assert np.array_equal(target, v.data()), (n, m, cplx)      # the same kernels on the same data: the same bits
        first = n - 2048
        ref = orc.convolve_direct(xx.astype(np.float64), hh.astype(np.float64), cplx, first, 2048)
        assert rel_l2(target[e * first:], ref) < tol_for(dtype), (n, m, cplx)


def _min_time(fn, reset=None, reps=25):
    import time
    best = 1e9
    for _ in range(reps):
        if reset is not None:
            reset()
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def test_b1_size_policy_declines_small_jobs_and_matches_a_fresh_measurement():
    """Round 6: the B1 boundary declines what the caller's CPU does faster than a host round trip -- is_supported_fft_len below
    FFT_MIN_LEN (-> rustfft, time_freq/mod.rs:41-44), gpu_convolve_vector -> None below CONV_MIN_WORK where the reference's next
    choice is its direct form (convolution.rs:530-541).  The shipped thresholds (profiles/r06_b1_crossover.txt) must be within a
    factor of two of a fresh measurement on THIS box, by the rule they were set with: fft -- the shortest power of two whose B1
    call takes at most half of numpy's (pocketfft's) time; convolution -- where the round trip ties with the oracle's scalar
    loop on one core.  bdsp_hip_fft_* itself still transforms any length, and a declined convolution is computed once the
    policy is lifted."""
    import ctypes as C
    L = bd._lib
    get, put = bd.lib.bdsp_hip_b1_policy_get, bd.lib.bdsp_hip_b1_policy_set
    defaults = [get(k) for k in range(4)]
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    try:
        # --- what is declined, what never is
        x = orc.fill_uniform(2 * 6000, 5, -10, 10, np.float32)
        h3 = orc.fill_uniform(2 * 3, 6, -1, 1, np.float32)
        assert V.gpu_convolve_vector(x, h3, True) == (None, None)                      # 6000 x 3 complex taps: the scalar loop wins
        assert V.gpu_convolve_vector(x[:11000], h3[:3], False) == (None, None)          # real data, 3 taps
        h200 = orc.fill_uniform(2 * 200, 7, -1, 1, np.float32) / 200
        y, rng = V.gpu_convolve_vector(x, h200, True)                                   # the reference would run overlap_discard: never declined
        assert rng == (0, x.size) and rel_l2(y, orc.convolve_direct(x.astype(np.float64), h200.astype(np.float64), True)) < 1e-6
        assert put(L.B1_CONV_MIN_WORK_F32, 0) == 0
        y, rng = V.gpu_convolve_vector(x, h3, True)
        assert rng == (0, x.size) and rel_l2(y, orc.convolve_direct(x.astype(np.float64), h3.astype(np.float64), True)) < 1e-6
        assert bd.lib.bdsp_hip_is_supported_fft_len_f32(1, 2 * 4096) == 0
        assert rel_l2(V.gpu_fft(x[:2 * 4096].copy()), orc.fft(x[:2 * 4096].astype(np.float64))) < 1e-6   # ... but fft() works at any length
        # --- the thresholds against a fresh measurement (a measurement on a shared box can be disturbed: up to three attempts,
        # each a complete fresh measurement; one of them within the factor of two passes)
        for k in range(4):
            put(k, 0)

        def fft_crossover(dtype, sfx):
            cdt = np.complex64 if dtype == np.float32 else np.complex128
            fft = getattr(bd.lib, "bdsp_hip_fft_" + sfx)
            wins = {}
            for n in (1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
                x0 = orc.fill_uniform(2 * n, n, -1, 1, dtype)
                xg = x0.copy()
                xc, out = x0.view(cdt), np.empty(n, cdt)
                fft(1, P(xg), xg.size, 0)
                tg = _min_time(lambda: fft(1, P(xg), xg.size, 0), lambda: np.copyto(xg, x0))
                tc = _min_time(lambda: np.fft.fft(xc, out=out))
                wins[n] = tg <= 0.5 * tc
            cross = None
            for n in sorted(wins, reverse=True):
                if not wins[n]:
                    break
                cross = n
            return cross, wins

        def conv_crossover(dtype, sfx):
            conv = getattr(bd.lib, "bdsp_hip_convolve_vector_" + sfx)
            oconv = orc._fn("orc_convolve_signal", dtype)
            oconv.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
            oconv.restype = C.c_int
            rs, re, path = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
            m = 4
            h = orc.fill_uniform(2 * m, 9, -1, 1, dtype)
            for n in (4096, 8192, 16384, 32768, 65536, 131072):
                x = orc.fill_uniform(2 * n, n + 1, -1, 1, dtype)
                y = np.zeros_like(x)
                assert conv(1, P(x), x.size, P(y), y.size, P(h), h.size, C.byref(rs), C.byref(re)) == 1
                tg = _min_time(lambda: conv(1, P(x), x.size, P(y), y.size, P(h), h.size, C.byref(rs), C.byref(re)))
                tc = _min_time(lambda: oconv(P(x), x.size, P(h), h.size, 1, P(y), C.byref(path)), reps=15)
                assert path.value == 4  # the reference's scalar loop
                if tg <= tc:
                    return n * m
            return None

        for dtype, sfx, key in ((np.float32, "f32", L.B1_FFT_MIN_LEN_F32), (np.float64, "f64", L.B1_FFT_MIN_LEN_F64)):
            seen = []
            for attempt in range(3):
                cross, wins = fft_crossover(dtype, sfx)
                seen.append((cross, wins))
                if cross is not None and defaults[key] // 2 <= 2 * cross <= defaults[key] * 2:
                    break
            else:
                raise AssertionError((sfx, "fft crossover (points) in three measurements", seen, "default (scalars)", defaults[key]))
        for dtype, sfx, key in ((np.float32, "f32", L.B1_CONV_MIN_WORK_F32), (np.float64, "f64", L.B1_CONV_MIN_WORK_F64)):
            seen = []
            for attempt in range(3):
                cross = conv_crossover(dtype, sfx)
                seen.append(cross)
                if cross is not None and defaults[key] // 2 <= cross <= defaults[key] * 2:
                    break
            else:
                raise AssertionError((sfx, "convolution crossover (points x taps) in three measurements", seen, "default", defaults[key]))
    finally:
        for k, v in enumerate(defaults):
            put(k, v)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,m", [(20000, 100), (50000, 1024), (12345, 17)])
def test_b1_overlap_discard_drop_in(n, m, dtype):
    # Drive GpuSupport::overlap_discard exactly as the reference's caller does
    # (convolution.rs:326-343, 376-412, 453-458) and compare the assembled result with the oracle.
    x = orc.fill_uniform(2 * n, 5, -10, 10, dtype)
    h = orc.fill_uniform(2 * m, 6, -1, 1, dtype) / dtype(m)
    fft_len = max(orc.next_power_of_two(m), orc.next_power_of_two(4 * (m - 1)))
    step = fft_len - (m - 1)
    hpad = np.zeros(2 * fft_len, dtype)
    hpad[:2 * m] = h
    h_freq = orc.fft(hpad)
    remainder_len = n - n % fft_len
    tmp = np.zeros(2 * fft_len, dtype)
    head = orc.convolve_direct(x, h, True, 0, m // 2)
    tmp[:head.size] = head
    end = orc.convolve_direct(x, h, True, n - remainder_len // 2, remainder_len // 2)
    sig = x.copy()
    pos = V.gpu_overlap_discard(sig, tmp, h_freq, 2 * m, 2 * step) // 2
    sig[2 * (pos - step + m // 2):2 * (pos + m // 2)] = tmp[2 * (m - 1):2 * fft_len]
    sig[2 * (n - remainder_len // 2):] = end
    code, ref = orc.overlap_discard(x, h, 0)
    assert code == 0
    assert rel_l2(sig, ref) < 4 * tol_for(dtype)
    # failure is reported out of band (shim/hip.rs checks bdsp_hip_last_error(), not the position): an unsupported
    # argument combination leaves a message, and the next good call clears it
    with pytest.raises(bd._lib.BackendError):
        V.gpu_overlap_discard(x.copy(), tmp.copy(), h_freq[:-2].copy(), 2 * m, 2 * step)   # fft_len not a power of two
    assert bd._lib.last_error() != ""
    assert V.gpu_overlap_discard(x.copy(), tmp.copy(), h_freq, 2 * m, 2 * step) // 2 == pos and bd._lib.last_error() == ""


# ------------------------------------------------------------------ interpolatef
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cplx", [True, False])
def test_interpolatef_both_paths(cplx, dtype):
    e = 2 if cplx else 1
    tol = 2e-6 if dtype == np.float32 else 1e-13
    x = orc.fill_uniform(e * 3000, 201602221, -10, 10, dtype)
    for fid, rolloff, factor, delay, conv_len in [(1, 0.35, 4.0, 0.0, 12), (0, 0.0, 2.0, 0.0, 30),
                                                  (1, 0.35, 3.0, 0.5, 10), (0, 0.0, 13.0 / 6.0, 0.0, 8),
                                                  (1, 0.2, 1.5, 0.25, 5)]:
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, factor, delay, conv_len, rolloff) == 0
        ref, path = orc.interpolatef(x, cplx, fid, rolloff, dtype(factor), delay, conv_len)
        assert len(v) == ref.size, (factor, len(v), ref.size)
        assert rel_l2(v.data(), ref) < tol, (fid, factor, delay, conv_len, path)
    # small vector -> scalar path, conv_len clamp (interpolation.rs:399-404)
    t = np.zeros(12, dtype)
    t[6] = 1.0
    v = DspVec(t, is_complex=True)
    assert v.interpolatef(0, 2.0, 1.0, 6) == 0
    ref, _ = orc.interpolatef(t, True, 0, 0.0, 2.0, 1.0, 6)
    np.testing.assert_allclose(v.data(), ref, atol=1e-5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cplx", [True, False])
def test_interpolatef_fractional_factor_kernel_singularities_and_fallback(cplx, dtype):
    """The fractional-factor kernel of round 4, k_interp_scalar_v2 (the reference's scalar path, interpolation.rs:92-131):
    one quotient per tap, cos(pi beta (j0 + k)) from an LDS table, and the raised cosine's two removable singularities
    (conv_types.rs:406-424) by SELECTS on the accumulated j.  With factor 2.5 and delay 0 every fifth output has an integer
    j0 = -conv_len, so its taps walk through j == 0 and |j| == 1 / (2 beta) exactly (beta 0.25 -> 2, beta 0.5 -> 1); an integer
    delay does the same for every such output with another offset.  Then tap counts whose cos / sin table no longer fits the
    48 KB of LDS the launcher allows: those take the first-generation kernel (one rotation per tap in double)."""
    e = 2 if cplx else 1
    tol = 2e-6 if dtype == np.float32 else 1e-13
    x = orc.fill_uniform(e * 3000, 201602223, -10, 10, dtype)
    for fid, rolloff, factor, delay, conv_len in [(1, 0.25, 2.5, 0.0, 12), (1, 0.5, 2.5, 0.0, 12), (1, 0.25, 2.5, 3.0, 9),
                                                  (1, 0.5, 1.25, -1.0, 4), (0, 0.0, 2.5, 0.0, 12), (0, 0.0, 1.25, 2.0, 30),
                                                  (1, 0.35, 48.0 / 44.1, 0.3, 20), (1, 0.125, 2.5, 0.0, 6)]:
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, factor, delay, conv_len, rolloff) == 0
        ref, path = orc.interpolatef(x, cplx, fid, rolloff, dtype(factor), delay, conv_len)
        assert path == 0  # the reference's scalar path
        assert len(v) == ref.size
        got = v.data()
        assert rel_l2(got, ref) < tol, (fid, rolloff, factor, delay, conv_len, rel_l2(got, ref))
        if delay == 0.0 and factor == 2.5:
            # the outputs whose taps sit ON the singularities, on their own (a wrong select would be lost in the norm of 7500)
            idx = np.arange(0, ref.size // e, 5)
            g, r = got.reshape(-1, e)[idx], ref.reshape(-1, e)[idx]
            assert rel_l2(g.ravel(), r.ravel()) < tol, (fid, rolloff, "outputs with integer j")
    # 2 * (2 L + 1) table entries beyond 48 KB: L = 3100 in f32 (49.6 KB), 1600 in f64 (51.2 KB)
    big = 3100 if dtype == np.float32 else 1600
    x = orc.fill_uniform(e * 7000, 201602224, -10, 10, dtype)
    for fid, rolloff in [(0, 0.0), (1, 0.35)]:
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, 1.5, 0.25, big, rolloff) == 0
        ref, path = orc.interpolatef(x, cplx, fid, rolloff, dtype(1.5), 0.25, big)
        assert path == 0 and len(v) == ref.size
        # (f32: the reference -- and the oracle -- take sin(pi * j) of a ROUNDED product; at |j| ~ 3000 that argument is off by
        # 1e-4 rad, which the kernel's exact sinpi does not reproduce: 6201 such taps add up to a few 1e-6 of the result)
        assert rel_l2(v.data(), ref) < (2e-5 if dtype == np.float32 else 1e-12), (fid, big, rel_l2(v.data(), ref))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cplx", [True, False])
def test_interpolatef_fractional_factor_packed_kernel(cplx, dtype):
    """Round 6: k_interp_frac_pk, the fractional path with two taps per packed instruction (f64: the same structure on pairs of
    scalar operations).  Its three shortcuts each get
    the case that would break them: (a) the per-launch mask of tap pairs that may hold j == 0 or a tap near / at the raised
    cosine's second singularity -- delays that move those taps to other pairs, roll-offs whose near range spans many taps (0.02)
    or none; (b) the no-wrap fast path -- a vector so short that most waves cross its end, and the 127-tap limit of the mask
    (conv_len 63 packed, 64 the round-4 kernel); (c) z * (w, w) instead of the reference's spelled-out complex product -- inf and
    NaN in the data must reach exactly the outputs and components they reach in the oracle (interpolation.rs:92-131)."""
    e = 2 if cplx else 1
    tol = 2e-6 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(e * 5000, 201606001, -10, 10, dtype)
    for fid, rolloff, factor, delay, conv_len in [(1, 0.35, 2.5, 7.25, 12), (1, 0.35, 2.5, -5.5, 12), (1, 0.02, 2.5, 0.0, 30), (1, 0.9, 1.7, 0.1, 9),
                                                  (1, 0.35, 1.088, 0.0, 63), (1, 0.35, 1.088, 0.0, 64), (0, 0.0, 2.5, 11.0, 12), (0, 0.0, 0.75, 0.5, 63),
                                                  (1, 0.25, 0.6, 2.0, 1), (0, 0.0, 3.3, 0.0, 0)]:
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, factor, delay, conv_len, rolloff) == 0
        with orc.exact_weights():
            ref, path = orc.interpolatef(x, cplx, fid, rolloff, dtype(factor), delay, conv_len)
        assert path == 0 and len(v) == ref.size
        assert rel_l2(v.data(), ref) < tol, (fid, rolloff, factor, delay, conv_len, rel_l2(v.data(), ref))
    # a short vector: 300 points, 81 taps -- most outputs read across the end of the vector
    xs = orc.fill_uniform(e * 300, 201606002, -10, 10, dtype)
    for fid, rolloff in [(0, 0.0), (1, 0.35)]:
        v = DspVec(xs, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, 2.5, 0.3, 40, rolloff) == 0
        ref, path = orc.interpolatef(xs, cplx, fid, rolloff, dtype(2.5), 0.3, 40)
        assert path == 0 and rel_l2(v.data(), ref) < tol
    # inf / NaN in the data
    xn = x.copy()
    xn[e * 1000] = np.inf
    xn[e * 2000 + (e - 1)] = np.nan
    xn[e * 3000] = -np.inf
    v = DspVec(xn, is_complex=cplx, delta=1.0)
    assert v.interpolatef(1, 2.5, 0.0, 12, 0.35) == 0
    with np.errstate(all="ignore"):
        ref, _ = orc.interpolatef(xn, cplx, 1, 0.35, dtype(2.5), 0.0, 12)
    got = v.data()
    # (WHICH non-finite value an output holds is not compared: on outputs whose tap arguments are integers the kernel's exact sinpi
    # makes the weights exactly 0 where the reference's sin(pi * j) of a rounded product leaves 1e-8 -- inf * 0 is NaN, inf * 1e-8 inf)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    fin = np.isfinite(ref)
    assert 0 < np.count_nonzero(~fin) < 400 and rel_l2(got[fin], ref[fin]) < tol


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cplx", [True, False])
def test_raised_cosine_next_to_its_second_singularity(cplx, dtype):
    """Roll-off 0.2 puts the raised cosine's second singularity at |x| = 2.5, and with a delay of 0.3 or 0.5 the accumulated
    tap arguments land ON it or an ulp BESIDE it (-6 - 0.8 + 0.3).  Beside it the reference's expression
    sin(pi x) cos(pi x beta) / (pi x) / (1 - (2 beta x)^2) (conv_types.rs:419-421) cancels in numerator and denominator and
    keeps no correct digit -- the literal oracle returns 0.1098 where the weight is 0.1000 -- so a result built on such a tap
    cannot be compared with the reference's.  Every raised-cosine evaluation of the library takes the cancellation-free
    form there (dsp_funcs.h rc_near_num) and is held to the oracle's EXACT-weights mode (same lattice of arguments in T,
    weights in long double); the literal oracle must be the one that is off.  Both interpolatef paths."""
    e = 2 if cplx else 1
    tol = 2e-6 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(e * 3000, 201602225, -10, 10, dtype)
    literal_off = 0
    for factor, delay, conv_len, rolloff in [(2.5, 0.3, 6, 0.2), (2.5, 0.5, 6, 0.2), (48.0 / 44.1, 0.5, 16, 0.2), (2.0, 0.5, 6, 0.2),
                                             (4.0, 0.5, 12, 0.2), (2.5, 0.5, 8, 0.5), (3.0, 0.25, 9, 0.4)]:
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(1, factor, delay, conv_len, rolloff) == 0
        with orc.exact_weights():
            ref, path = orc.interpolatef(x, cplx, 1, rolloff, dtype(factor), delay, conv_len)
        lit, _ = orc.interpolatef(x, cplx, 1, rolloff, dtype(factor), delay, conv_len)
        got = v.data()
        assert len(v) == ref.size
        assert rel_l2(got, ref) < tol, (factor, delay, conv_len, rolloff, path, rel_l2(got, ref), rel_l2(lit, ref))
        literal_off += rel_l2(lit, ref) > 10 * tol
    if dtype == np.float64:
        assert literal_off >= 2  # (the reference's own arithmetic is what fails these cases, by 1e-6 ... 1e-3)


@pytest.mark.parametrize("cplx", [True, False])
def test_interpolatef_f64_integer_factors_ragged_lengths(cplx):
    # f64, integer factors with and without a blocked inner kernel (2, 4, 8 / 16), ragged lengths (the last workgroup's
    # tile ends inside the vector, an odd first inner output), conv_len from 1 to 46, a delay, both functions -- each
    # against the oracle's "simd" path at the f64 tolerance
    e = 2 if cplx else 1
    cases = [(1, 0.35, 2, 0.0, 12, 3000), (1, 0.35, 4, 0.0, 12, 3001), (0, 0.0, 8, 0.0, 12, 1777), (0, 0.0, 16, 0.0, 5, 1500),
             (1, 0.2, 4, 0.25, 1, 2047), (0, 0.0, 2, -0.5, 40, 4099), (1, 0.5, 4, 0.0, 44, 2500), (0, 0.0, 16, 0.0, 45, 3000),
             (1, 0.35, 8, 0.0, 46, 9001), (0, 0.0, 4, 0.0, 12, 70001)]
    for fid, rolloff, factor, delay, conv_len, points in cases:
        x = orc.fill_uniform(e * points, 201602221 + points, -10, 10, np.float64)
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, float(factor), delay, conv_len, rolloff) == 0
        ref, path = orc.interpolatef(x, cplx, fid, rolloff, np.float64(factor), delay, conv_len)
        assert path == 1  # the reference would take its "simd" path (interpolation.rs:411-445)
        assert len(v) == ref.size
        got = v.data()
        assert rel_l2(got, ref) < 1e-13, (fid, factor, delay, conv_len, points, rel_l2(got, ref))
        assert np.max(np.abs(got - ref)) < 1e-11, (factor, conv_len, points)


# ------------------------------------------------------------------ batch (config C5, one GPU)
def test_batch_shard_on_gpu_matches_oracle():
    import torch
    from basic_dsp_amd.batch import process_shard_gpu
    nvec, points, m = 5, 1 << 14, 200
    host = np.stack([orc.fill_uniform(2 * points, 201511212 + v, -10, 10, np.float32) for v in range(nvec)])
    h = orc.fill_uniform(2 * m, 201601172, -1, 1, np.float32) / np.float32(m)
    out = process_shard_gpu(torch.from_numpy(host).cuda(), torch.from_numpy(h).cuda(), points)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for v in range(nvec):
        y = orc.convolve_direct(host[v].astype(np.float64), h.astype(np.float64), True)
        assert rel_l2(out[v], orc.fft(y)) < 2e-6, v


def test_b3_device_pointer_api():
    """bdsp_hip_dev_* on torch-owned memory and torch's stream (what bench.py uses)."""
    import ctypes as C
    import torch
    lib = bd.lib
    n = 1 << 16
    x = orc.fill_uniform(2 * n, 77, -10, 10, np.float32)
    d = torch.from_numpy(x).cuda()
    s = torch.empty_like(d)
    sp = bd._lib.torch_stream_arg()
    flag = C.c_int(0)
    assert lib.bdsp_hip_dev_fft(0, d.data_ptr(), s.data_ptr(), n, 1, bd._lib.FFT_SHIFT_OUT | bd._lib.FFT_MAGNITUDE,
                                1.0, -1, 0.0, C.byref(flag), sp) == 0
    torch.cuda.synchronize()
    got = (s if flag.value else d).cpu().numpy()[:n]
    ref = orc.magnitude(orc.swap_halves(orc.fft(x.astype(np.float64)), True, True))
    assert rel_l2(got, ref) < 1e-6
    # the same fused pair (shift + magnitude output) on batches of smooth lengths: the register-resident mixed-radix kernels'
    # staged output path (100 = 10 10 two stages, 1000 and 3000 three), the general kernel (1001 = 7 11 13)
    for m, rows in ((100, 700), (250, 300), (1000, 300), (3000, 70), (1001, 50)):
        xb = orc.fill_uniform(2 * m * rows, 78 + m, -10, 10, np.float32).reshape(rows, 2 * m)
        d = torch.from_numpy(xb.copy()).cuda()
        s = torch.empty_like(d)
        assert lib.bdsp_hip_dev_fft(0, d.data_ptr(), s.data_ptr(), m, rows, bd._lib.FFT_SHIFT_OUT | bd._lib.FFT_MAGNITUDE,
                                    1.0, -1, 0.0, C.byref(flag), sp) == 0
        torch.cuda.synchronize()
        got = (s if flag.value else d).cpu().numpy().ravel()[:m * rows].reshape(rows, m)
        for k in (0, rows // 2, rows - 1):
            ref = orc.magnitude(orc.swap_halves(orc.fft(xb[k].astype(np.float64)), True, True))
            assert rel_l2(got[k], ref) < 1e-6, (m, k)
    d = torch.from_numpy(x).cuda()
    assert lib.bdsp_hip_dev_real_scale(0, d.data_ptr(), 2 * n, 2.5, sp) == 0
    assert lib.bdsp_hip_dev_real_offset(0, d.data_ptr(), 2 * n, 0, -1.25, sp) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), orc.real_offset(orc.real_scale(x, 2.5), -1.25))


# ------------------------------------------------------------------ FFT-domain interpolation family (a14)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_interpolatei_interpolate_decimatei(dtype):
    tol = 5e-6 if dtype == np.float32 else 1e-11
    for cplx in (True, False):
        e = 2 if cplx else 1
        for points, factor in ((6, 2), (1024, 4), (1000, 3)):
            x = orc.fill_uniform(points * e, 31 + points, -10, 10, dtype)
            for fid, ro in ((0, 0.0), (1, 0.4)):
                v = DspVec(x, is_complex=cplx)
                assert v.interpolatei(fid, factor, ro) == 0
                code, ref = orc.interpolatei(x.astype(np.float64), cplx, fid, ro, factor)
                assert len(v) == ref.size and v.is_complex() == cplx
                assert rel_l2(v.data(), ref) < tol, (cplx, points, factor, fid)
        for points, dest, delay in ((6, 12, 0.0), (7, 14, 0.0), (6, 13, 0.0), (13, 6, 0.0), (4096, 8192, 0.0),
                                    (1000, 1500, 0.0), (2048, 1024, 0.0), (6, 12, 1.0), (512, 2048, 0.3)):
            x = orc.fill_uniform(points * e, 77 + points, -10, 10, dtype)
            v = DspVec(x, is_complex=cplx, delta=0.5)
            assert v.interpolate(V.CONV_SINC, dest, delay) == 0
            code, ref, nd = orc.interpolate(x.astype(np.float64), cplx, 0, 0.0, dest, delay, 0.5)
            assert len(v) == ref.size
            assert rel_l2(v.data(), ref) < (2e-5 if dtype == np.float32 else 1e-10), (cplx, points, dest, delay)
            assert v.delta() == pytest.approx(nd, rel=1e-6)
            v = DspVec(x, is_complex=cplx)
            assert v.interpft(dest) == 0
            code, ref, _ = orc.interpolate(x.astype(np.float64), cplx, -1, 0.0, dest, 0.0, 1.0)
            assert rel_l2(v.data(), ref) < (2e-5 if dtype == np.float32 else 1e-10)
        x = orc.fill_uniform(1001 * e, 5, -10, 10, dtype)
        for factor, delay in ((2, 1), (3, 0), (7, 5), (2000, 3)):
            v = DspVec(x, is_complex=cplx)
            assert v.decimatei(factor, delay) == 0
            assert np.array_equal(v.data(), orc.decimatei(x, cplx, factor, delay))


def test_multiply_frequency_response_gpu():
    x = np.ones(10, np.float32)
    v = DspVec(x, is_complex=True, domain=V.FREQ)
    assert v.multiply_frequency_response(V.CONV_RAISED_COSINE, 2.0, 1.0) == 0
    np.testing.assert_allclose(v.data(), [0, 0, 1, 1, 2, 2, 1, 1, 0, 0], atol=1e-4)  # convolution.rs:633-639
    for cplx, n in ((True, 1001), (False, 4096)):
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 3, -1, 1, np.float64)
        v = DspVec(x, is_complex=cplx, domain=V.FREQ)
        assert v.multiply_frequency_response(V.CONV_RAISED_COSINE, 1.7, 0.35) == 0
        np.testing.assert_allclose(v.data(), orc.multiply_frequency_response(x, cplx, 1, 0.35, 1.7, False), atol=1e-12)
    t = DspVec(np.ones(8, np.float32), is_complex=True, domain=V.TIME)
    assert t.multiply_frequency_response(0, 1.0) == -1  # must be in frequency domain (convolution.rs:590-593)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_symmetric_fft_family(dtype):
    # tests/real_test.rs:582-605: mirror(plain_sfft(x)) == plain_fft(to_complex(x)); plain_sifft inverts it
    n = 1001
    x = orc.fill_uniform(n, 201511210, -10, 10, dtype)
    tol = 1e-6 if dtype == np.float32 else 1e-12
    v = DspVec(x)
    assert v.plain_sfft() == 0 and v.is_complex() and v.domain() == V.FREQ and v.points() == n // 2 + 1
    full = np.fft.fft(x.astype(np.float64))
    assert rel_l2(v.datac(), full[:n // 2 + 1]) < 2 * tol
    assert v.mirror() == 0 and v.points() == n
    assert rel_l2(v.datac(), full) < 2 * tol
    v = DspVec(x)
    assert v.plain_sfft() == 0 and v.plain_sifft() == 0
    assert not v.is_complex() and len(v) == n
    assert rel_l2(v.data().astype(np.float64) / n, x) < 4 * tol
    # odd lengths the register-resident mixed-radix kernels are built for (round 6): real input on the way in, the real part
    # straight out of the inverse transform (FFT_OUT_REAL) on the way back
    for m in (45, 225, 375, 1125, 3375):
        xm = orc.fill_uniform(m, 201511211 + m, -10, 10, dtype)
        v = DspVec(xm)
        assert v.plain_sfft() == 0 and v.points() == m // 2 + 1
        assert rel_l2(v.datac(), np.fft.fft(xm.astype(np.float64))[:m // 2 + 1]) < 2 * tol, m
        assert v.plain_sifft() == 0 and not v.is_complex() and len(v) == m
        assert rel_l2(v.data().astype(np.float64) / m, xm) < 4 * tol, m
    assert DspVec(x[:1000]).plain_sfft() == 9                       # InputMustHaveAnOddLength
    assert DspVec(np.zeros(10, dtype), is_complex=True).plain_sfft() == 5   # must be real time data
    assert DspVec(np.zeros(10, dtype), is_complex=True).plain_sifft() == 6  # must be frequency domain
    bad = DspVec(np.array([1.0, 0.5, 2.0, 0.0], dtype), is_complex=True, domain=V.FREQ)
    assert bad.plain_sifft() == 8                                            # first bin must be real


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [9, 1001, 4097, 65537])
def test_shifted_and_windowed_symmetric_transforms_against_oracle(n, dtype):
    # sfft / windowed_sfft (time_to_freq.rs:232-298): zero-interleave -> [window] -> fft (= plain_fft + fft_shift)
    # -> keep the first n/2+1 bins of the SHIFTED spectrum (unmirror!, :178-186).
    # sifft / windowed_sifft (freq_to_time.rs:226-247): scale(1/points) and ifft_shift of the HALF spectrum ->
    # plain_sifft (mirror -> inverse transform -> real part) -> [unapply_window].  The oracle side is composed
    # from its pinned fft / swap_halves / apply_window / mirror restatements.
    tol = 1e-6 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(n, 201511213 + n, -10, 10, dtype)
    xd = x.astype(np.float64)
    p = n // 2 + 1
    cplx = orc.zero_interleave(xd, False, 2)

    v = DspVec(x)
    assert v.sfft() == 0 and v.is_complex() and v.domain() == V.FREQ and v.points() == p
    ref = orc.swap_halves(orc.fft(cplx), True, True)[:2 * p]
    assert rel_l2(v.data(), ref) < 2 * tol

    for wid, oid, alpha in ((V.WINDOW_HAMMING, 1, 0.54), (V.WINDOW_HANN, 1, 0.5), (V.WINDOW_TRIANGULAR, 0, 0.0),
                            (V.WINDOW_BLACKMAN_HARRIS, 2, 0.0)):
        v = DspVec(x)
        assert v.windowed_sfft(wid) == 0 and v.points() == p
        ref = orc.swap_halves(orc.fft(orc.apply_window(cplx, True, oid, alpha)), True, True)[:2 * p]
        assert rel_l2(v.data(), ref) < 2 * tol, wid

    # a half spectrum whose first bin AFTER ifft_shift is real (index p/2 before it)
    h = orc.fill_uniform(2 * p, 77 + n, -10, 10, dtype)
    h[2 * (p // 2) + 1] = 0
    hd = h.astype(np.float64)

    def oracle_sifft(hd):
        y = orc.complex_scale(hd, 1.0 / p, 0.0)
        y = orc.swap_halves(y, True, False)
        return orc.fft(orc.mirror(y), inverse=True)[0::2]

    v = DspVec(h, is_complex=True, domain=V.FREQ)
    assert v.sifft() == 0 and not v.is_complex() and v.domain() == V.TIME and len(v) == n
    assert rel_l2(v.data(), oracle_sifft(hd)) < 2 * tol

    v = DspVec(h, is_complex=True, domain=V.FREQ)
    assert v.windowed_sifft(V.WINDOW_HAMMING) == 0 and len(v) == n
    ref = orc.apply_window(oracle_sifft(hd), False, 1, 0.54, unapply=True)
    assert rel_l2(v.data(), ref) < 4 * tol

    # an imaginary first bin after the shift is rejected as in plain_sifft (freq_to_time.rs:203-211)
    bad = h.copy()
    bad[2 * (p // 2) + 1] = 1.0
    assert DspVec(bad, is_complex=True, domain=V.FREQ).sifft() == 8


# ------------------------------------------------------------------ correlate, convolve(function), real interpolation
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_correlate_and_prepare_argument(dtype):
    # the doc example and KAT of correlation.rs:52-62, 201-215
    a = DspVec(np.array([1, 1, 2, 1, 3, 1], dtype), is_complex=True)
    b = DspVec(np.array([4, 1, 5, 1, 6, 1], dtype), is_complex=True)
    assert b.prepare_argument_padded() == 0
    assert b.domain() == V.FREQ and b.points() == 5
    assert a.correlate(b) == 0
    np.testing.assert_allclose(a.data(), [7, 5, 19, 8, 35, 9, 25, 4, 13, 1], atol=1e-4)
    assert a.delta() == 1.0 and a.domain() == V.TIME
    tol = 2e-6 if dtype == np.float32 else 1e-12
    for n in (100, 1000, 4097):
        x = orc.fill_uniform(2 * n, 11 + n, -10, 10, dtype)
        y = orc.fill_uniform(2 * n, 12 + n, -10, 10, dtype)
        for padded in (True, False):
            arg = DspVec(y, is_complex=True)
            assert (arg.prepare_argument_padded() if padded else arg.prepare_argument()) == 0
            _, ref_arg = orc.prepare_argument(y.astype(np.float64), padded)
            assert rel_l2(arg.data(), ref_arg) < tol
            v = DspVec(x, is_complex=True)
            code = v.correlate(arg)
            ref_code, ref = orc.correlate(x.astype(np.float64), ref_arg)
            assert code == ref_code == (0 if padded else 7)  # an unpadded argument is not longer: zero_pad_b fails
            if code == 0:
                assert rel_l2(v.data(), ref) < tol, (n, padded)
    # type-state errors (correlation.rs:134-146): both report InputMustBeInTimeDomain and poison
    t = DspVec(np.ones(8, dtype), is_complex=True)
    notprepared = DspVec(np.ones(16, dtype), is_complex=True)
    assert t.correlate(notprepared) == 5 and t.is_erroneous()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_convolve_with_function(dtype):
    # KATs convolution.rs:651-702
    x = np.zeros(10, dtype)
    x[5] = 1.0
    v = DspVec(x)
    assert v.convolve(V.CONV_RAISED_COSINE, 0.2, 5, rolloff=0.35) == 0
    np.testing.assert_allclose(v.data(), [0.0, 0.2171850639713355, 0.4840621929215732, 0.7430526238101408,
                                          0.9312114164253432, 1.0, 0.9312114164253432, 0.7430526238101408,
                                          0.4840621929215732, 0.2171850639713355], atol=1e-4)
    tol = 2e-6 if dtype == np.float32 else 1e-12
    for cplx in (True, False):
        e = 2 if cplx else 1
        # (points, L): table longer than the vector, equal, overlap-save path, long vector
        for points, L, ratio in ((11, 5, 0.5), (8, 8, 0.25), (7, 20, 0.3), (5000, 12, 0.25), (70000, 300, 0.1)):
            xx = orc.fill_uniform(points * e, 900 + points, -10, 10, dtype)
            for fid, ro in ((0, 0.0), (1, 0.35)):
                v = DspVec(xx, is_complex=cplx)
                assert v.convolve(fid, ratio, L, rolloff=ro) == 0
                ref = orc.convolve_function(xx.astype(np.float64), cplx, fid, ro, ratio, L)
                assert rel_l2(v.data(), ref) < tol, (cplx, points, L, fid)
    # the callback variant samples the function on the host
    xx = orc.fill_uniform(2 * 1000, 4, -10, 10, dtype)
    v = DspVec(xx, is_complex=True)
    assert v.convolve(lambda t: float(np.sinc(t)), 0.5, 9) == 0
    ref = orc.convolve_function(xx.astype(np.float64), True, 0, 0.0, 0.5, 9)
    assert rel_l2(v.data(), ref) < tol
    f = DspVec(xx, is_complex=True, domain=V.FREQ)
    assert f.convolve(0, 0.5, 3) == -1  # assert_time!


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_interpolate_lin_hermite_bit_exact(dtype):
    # KATs real_interpolation.rs:198-237
    x = np.array([-1.0, -2.0, -1.0, 0.0, 1.0, 3.0, 4.0], dtype)
    v = DspVec(x)
    assert v.interpolate_lin(4.0) == 0
    np.testing.assert_allclose(v.data(), [-1.0, -1.25, -1.5, -1.75, -2.0, -1.75, -1.5, -1.25, -1.0, -0.75, -0.5,
                                          -0.25, 0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 2.5, 3.0, 3.25, 3.5, 3.75,
                                          4.0], atol=1e-6)
    for n, factor, delay in ((7, 4.0, 0.0), (7, 3.0, 0.0), (1000, 2.5, 0.0), (4097, 7.0, 0.0), (100000, 1.37, 0.0),
                             (513, 0.5, 0.0), (2000, 3.0, 0.25)):
        xx = orc.fill_uniform(n, 50 + n, -10, 10, dtype)
        for hermite in (False, True):
            v = DspVec(xx)
            code = v.interpolate_hermite(factor, delay) if hermite else v.interpolate_lin(factor, delay)
            assert code == 0
            ref = (orc.interpolate_hermite if hermite else orc.interpolate_lin)(xx, factor, delay)
            assert len(v) == ref.size
            assert np.array_equal(v.data(), ref), (n, factor, delay, hermite)
    c = DspVec(np.ones(8, dtype), is_complex=True)
    assert c.interpolate_lin(2.0) == -1  # complex input poisons the vector (real_interpolation.rs:47-50)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_smaller_vector_ops_custom_windows_misc(dtype):
    for cplx in (False, True):
        x = orc.fill_uniform(1200, 1, -10, 10, dtype)
        y = orc.fill_uniform(24, 2, 1, 10, dtype)
        for op, name in enumerate(("add_smaller", "sub_smaller", "mul_smaller", "div_smaller")):
            v = DspVec(x, is_complex=cplx)
            assert getattr(v, name)(DspVec(y, is_complex=cplx)) == 0
            _, ref = orc.binary(x, np.tile(y, 50), cplx, op)
            assert np.array_equal(v.data(), ref), (cplx, name)
        v = DspVec(x, is_complex=cplx)
        assert v.add_smaller(DspVec(orc.fill_uniform(14, 2, 1, 10, dtype), is_complex=cplx)) == 7
    # callback windows: sampled on the host, applied on the device; must equal the built-in Hamming
    x = orc.fill_uniform(2 * 1001, 8, -10, 10, dtype)
    ham = lambda n, length: 0.54 - 0.46 * np.cos(2 * np.pi * n / (length - 1))  # noqa: E731
    for sym in (True, False):
        a, b = DspVec(x, is_complex=True), DspVec(x, is_complex=True)
        assert a.apply_custom_window(ham, sym) == 0 and b.apply_window(V.WINDOW_HAMMING) == 0
        assert rel_l2(a.data(), b.data()) < (1e-6 if dtype == np.float32 else 1e-14)
        assert a.unapply_custom_window(ham, sym) == 0
        assert rel_l2(a.data(), x) < (1e-6 if dtype == np.float32 else 1e-14)
    a, b = DspVec(x, is_complex=True), DspVec(x, is_complex=True)
    assert a.windowed_custom_fft(ham) == 0 and b.windowed_fft(V.WINDOW_HAMMING) == 0
    assert rel_l2(a.data(), b.data()) < (2e-6 if dtype == np.float32 else 1e-12)
    assert a.windowed_custom_ifft(ham) == 0 and b.windowed_ifft(V.WINDOW_HAMMING) == 0
    assert rel_l2(a.data(), b.data()) < (2e-5 if dtype == np.float32 else 1e-10)
    r = orc.fill_uniform(1001, 9, -10, 10, dtype)
    a, b = DspVec(r), DspVec(r)
    assert a.windowed_custom_sfft(ham) == 0 and b.windowed_sfft(V.WINDOW_HAMMING) == 0
    assert rel_l2(a.data(), b.data()) < (2e-6 if dtype == np.float32 else 1e-12)
    # ... and both equal the oracle's zero-interleave -> window -> fft -> fft_shift -> first n/2+1 bins
    rc = orc.zero_interleave(r.astype(np.float64), False, 2)
    ref = orc.swap_halves(orc.fft(orc.apply_window(rc, True, 1, 0.54)), True, True)[:2 * (1001 // 2 + 1)]
    assert rel_l2(a.data(), ref) < (2e-6 if dtype == np.float32 else 1e-12)
    assert rel_l2(b.data(), ref) < (2e-6 if dtype == np.float32 else 1e-12)
    # callback frequency response equals the built-in raised cosine
    xs = orc.fill_uniform(2 * 1000, 3, -1, 1, dtype)
    rc = lambda t: float(orc.conv_freq(1, 0.35, t, np.float64))  # noqa: E731
    a, b = DspVec(xs, is_complex=True, domain=V.FREQ), DspVec(xs, is_complex=True, domain=V.FREQ)
    assert a.multiply_frequency_response_fn(rc, 1.7) == 0
    assert b.multiply_frequency_response(V.CONV_RAISED_COSINE, 1.7, 0.35) == 0
    assert rel_l2(a.data(), b.data()) < (1e-6 if dtype == np.float32 else 1e-13)
    # complex_divide, set_value, allocated_len
    v = DspVec(xs, is_complex=True)
    assert v.complex_divide(2.0 - 1.5j) == 0
    assert rel_l2(v.datac(), xs.view(np.complex64 if dtype == np.float32 else np.complex128) / (2.0 - 1.5j)) < \
        (1e-6 if dtype == np.float32 else 1e-14)
    v.set_value(3, 42.0)
    assert v.data()[3] == 42.0 and v.allocated_len() >= len(v)


# ------------------------------------------------------------------ matrix / batch API
def _mat_rows(rows, n, seed, dtype, cplx):
    e = 2 if cplx else 1
    return np.stack([orc.fill_uniform(n * e, seed + r, -10, 10, dtype) for r in range(rows)])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_matrix_batched_ops_equal_per_row_vector_ops(dtype):
    from basic_dsp_amd import DspMat
    tol = 2e-6 if dtype == np.float32 else 1e-12
    for cplx, rows, n in ((True, 3, 4096), (True, 5, 1000), (False, 4, 2048), (True, 2, 1 << 14)):
        a = _mat_rows(rows, n, 100 + n, dtype, cplx)
        # elementwise + complex->real: bit-exact against the oracle applied row by row
        m = DspMat(a, is_complex=cplx)
        assert (m.rows(), m.row_len(), m.row_points()) == (rows, a.shape[1], n)
        assert m.scale(2.5) == 0 and m.offset(-1.25) == 0
        ref = np.stack([orc.real_offset(orc.real_scale(r, 2.5), -1.25, cplx) for r in a])
        assert np.array_equal(m.data(), ref)
        if cplx:
            assert m.magnitude_squared() == 0
            assert np.array_equal(m.data(), np.stack([orc.complex_to_real(r, 1) for r in ref]))
        # matrix (.) matrix and matrix (.) vector
        b = _mat_rows(rows, n, 7, dtype, cplx)
        m = DspMat(a, is_complex=cplx)
        assert m.mul(DspMat(b, is_complex=cplx)) == 0
        assert np.array_equal(m.data(), np.stack([orc.binary(x, y, cplx, 2)[1] for x, y in zip(a, b)]))
        m = DspMat(a, is_complex=cplx)
        assert m.add(DspVec(b[0], is_complex=cplx)) == 0
        assert np.array_equal(m.data(), np.stack([orc.binary(x, b[0], cplx, 0)[1] for x in a]))
        # transforms: every row equals the single-vector path
        for name, args in (("plain_fft", ()), ("fft", ()), ("windowed_fft", (V.WINDOW_HAMMING,))):
            m = DspMat(a, is_complex=cplx)
            assert getattr(m, name)(*args) == 0 and m.is_complex() and m.domain() == V.FREQ
            got = m.data()
            for r in range(rows):
                v = DspVec(a[r], is_complex=cplx)
                assert getattr(v, name)(*args) == 0
                assert rel_l2(got[r], v.data()) < tol, (name, cplx, rows, n, r)
            assert m.delta() == pytest.approx(float(n))
            inv = {"plain_fft": "plain_ifft", "fft": "ifft", "windowed_fft": "windowed_ifft"}[name]
            assert getattr(m, inv)(*args) == 0
            back = m.data()
            scale = n if name == "plain_fft" else 1
            expect = a if cplx else np.stack([np.stack([r, np.zeros_like(r)], -1).reshape(-1) for r in a])
            assert rel_l2(back / scale, expect) < (2e-5 if dtype == np.float32 else 1e-10), (name, cplx)
        # shared-filter convolution, shift, window, zero_pad, interpolatef
        h = orc.fill_uniform(33 * (2 if cplx else 1), 5, -1, 1, dtype)
        m = DspMat(a, is_complex=cplx)
        assert m.convolve_signal(DspVec(h, is_complex=cplx)) == 0
        got = m.data()
        for r in range(rows):
            ref = orc.convolve_direct(a[r].astype(np.float64), h.astype(np.float64), cplx)
            assert rel_l2(got[r], ref) < tol
        m = DspMat(a, is_complex=cplx)
        assert m.fft_shift() == 0 and m.apply_window(V.WINDOW_BLACKMAN_HARRIS) == 0
        ref = np.stack([orc.apply_window(orc.swap_halves(r, cplx, True), cplx, 2) for r in a])
        assert rel_l2(m.data(), ref) < (1e-6 if dtype == np.float32 else 1e-14)
        m = DspMat(a, is_complex=cplx)
        assert m.zero_pad(n + 37, V.PAD_SURROUND) == 0
        assert np.array_equal(m.data(), np.stack([orc.zero_pad(r, cplx, n + 37, 1, True)[1] for r in a]))
        m = DspMat(a, is_complex=cplx)
        assert m.interpolatef(V.CONV_SINC, 2.0, 0.0, 8) == 0
        got = m.data()
        for r in range(rows):
            ref, _ = orc.interpolatef(a[r].astype(np.float64), cplx, 0, 0.0, 2.0, 0.0, 8)
            assert rel_l2(got[r], ref) < (2e-6 if dtype == np.float32 else 1e-12)
        # rows in and out
        m = DspMat(a, is_complex=cplx)
        assert np.array_equal(m.get_row(1).data(), a[1])
        assert m.set_row(0, DspVec(b[1], is_complex=cplx)) == 0
        assert np.array_equal(m.data()[0], b[1])
        assert m.set_row(0, DspVec(b[1][:-2], is_complex=cplx)) == 7


def test_matrix_convolve_signal_mimo_kats():
    from basic_dsp_amd import DspMat
    # matrix/src/time_freq.rs:587-657: delay and channel-swapping impulse-response matrices
    x = np.zeros((2, 11), np.float32)
    x[0, 5], x[1, 5] = 0.5, 2.0
    empty = DspVec(np.zeros(3, np.float32))
    delay = DspVec(np.array([1.0, 0.0, 0.0], np.float32))
    m = DspMat(x)
    assert m.convolve_signal([[delay, empty], [empty, delay]]) == 0
    exp = np.zeros((2, 11), np.float32)
    exp[0, 4], exp[1, 4] = 0.5, 2.0
    np.testing.assert_allclose(m.data(), exp, atol=1e-4)
    m = DspMat(x)
    assert m.convolve_signal([[empty, delay], [delay, empty]]) == 0
    exp[0, 4], exp[1, 4] = 2.0, 0.5
    np.testing.assert_allclose(m.data(), exp, atol=1e-4)
    assert m.convolve_signal([[empty, delay]]) == 7  # needs rows x rows responses (mod.rs:373-375)
    # random complex 3x3 system against the oracle's direct form
    a = _mat_rows(3, 5000, 3, np.float64, True)
    hs = [[orc.fill_uniform(2 * 17, 10 * n + r, -1, 1, np.float64) for r in range(3)] for n in range(3)]
    m = DspMat(a, is_complex=True)
    assert m.convolve_signal([[DspVec(h, is_complex=True) for h in row] for row in hs]) == 0
    got = m.data()
    for n in range(3):
        ref = sum(orc.convolve_direct(a[r], hs[n][r], True) for r in range(3))
        assert rel_l2(got[n], ref) < 1e-12


def test_b1_entry_points_are_thread_safe():
    # GpuSupport<T> is called from whichever thread owns the vector (T: Send + Sync):
    # concurrent host threads must not corrupt each other's results
    import threading
    n, m = 20000, 65
    cases = []
    for k in range(8):
        x = orc.fill_uniform(2 * n, 1000 + k, -10, 10, np.float32)
        h = orc.fill_uniform(2 * m, 2000 + k, -1, 1, np.float32)
        cases.append((x, h, orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), True),
                      orc.fft(x.astype(np.float64))))
    errors = []

    def worker(k):
        x, h, ref_conv, ref_fft = cases[k]
        try:
            for _ in range(5):
                got, rng = V.gpu_convolve_vector(x, h, True)
                assert rng == (0, x.size) and rel_l2(got, ref_conv) < 2e-6
                assert rel_l2(V.gpu_fft(x.copy()), ref_fft) < 2e-6
                v = DspVec(x, is_complex=True)
                assert v.plain_fft() == 0 and rel_l2(v.data(), ref_fft) < 2e-6
        except Exception as exc:  # noqa: BLE001
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_hip_graph_capture_and_replay():
    import time
    from basic_dsp_amd import Graph, lib
    # config C1: real f32, 65 536 samples, scale(2.5) then offset(-1.25) -- two launch-bound kernels
    x = orc.fill_uniform(65536, 201511141, -10, 10, np.float32)
    v = DspVec(x)
    g = Graph.capture(lambda: (v.scale(2.5), v.offset(-1.25)))   # one warm-up run; capturing records, not runs
    ref = orc.real_offset(orc.real_scale(x, 2.5), -1.25)
    assert np.array_equal(v.data(), ref)
    g.launch()
    ref = orc.real_offset(orc.real_scale(ref, 2.5), -1.25)
    assert np.array_equal(v.data(), ref)
    # a transform chain: fft -> ifft is the identity, so replaying it leaves the vector unchanged
    c = DspVec(orc.fill_uniform(2 * 16384, 5, -10, 10, np.float32), is_complex=True)
    before = c.data()
    g2 = Graph.capture(lambda: (c.fft(), c.ifft()))
    for _ in range(5):
        g2.launch()
    assert rel_l2(c.data(), before) < 2e-5
    # replay is cheaper than issuing the calls one by one
    w = DspVec(x)
    w.scale(1.0); w.offset(0.0)
    g3 = Graph.capture(lambda: (w.scale(1.0), w.offset(0.0)))
    lib.bdsp_hip_synchronize(None)
    t0 = time.perf_counter()
    for _ in range(200):
        w.scale(1.0); w.offset(0.0)
    lib.bdsp_hip_synchronize(None)
    t_direct = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        g3.launch()
    lib.bdsp_hip_synchronize(None)
    t_graph = (time.perf_counter() - t0) / 200
    print("C1 scale+offset: %.1f us direct, %.1f us as a graph" % (t_direct * 1e6, t_graph * 1e6))
    assert np.array_equal(w.data(), x)
    # a longer chain (12 kernels)
    def chain():
        for _ in range(6):
            w.scale(1.0); w.offset(0.0)
    g4 = Graph.capture(chain)
    lib.bdsp_hip_synchronize(None)
    t0 = time.perf_counter()
    for _ in range(100):
        chain()
    lib.bdsp_hip_synchronize(None)
    t_direct = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for _ in range(100):
        g4.launch()
    lib.bdsp_hip_synchronize(None)
    t_graph = (time.perf_counter() - t0) / 100
    print("12-kernel chain: %.1f us direct, %.1f us as a graph" % (t_direct * 1e6, t_graph * 1e6))
    # a graph replays addresses: a captured sequence that leaves a vector's (live, trade) buffers swapped is refused
    big = DspVec(orc.fill_uniform(2 * (1 << 21), 6, -10, 10, np.float32), is_complex=True)
    ref3 = np.fft.fft(big.datac().astype(np.complex128))
    warm = DspVec(orc.fill_uniform(2 * (1 << 21), 7, -10, 10, np.float32), is_complex=True)
    assert warm.plain_fft() == 0 and warm.plain_ifft() == 0   # twiddle tables and workspace of this length exist now
    with pytest.raises(bd.BackendError):
        Graph.capture(lambda: big.plain_fft(), warmup=False)   # three passes: the result ends in the trade buffer
    # ... the capture attempt still ran nothing twice and left the library usable
    assert big.domain() == V.FREQ
    g5 = Graph.capture(lambda: (big.plain_ifft(), big.plain_fft()), warmup=False)  # two trades cancel
    # a captured sequence that fails half way is dropped (bdsp_hip_capture_abort) and the next capture is accepted
    def boom():
        w.scale(1.0)
        raise RuntimeError("the caller's own failure inside a capture")
    with pytest.raises(RuntimeError):
        Graph.capture(boom, warmup=False)
    g6 = Graph.capture(lambda: (w.scale(1.0), w.offset(0.0)))
    g6.launch()
    # a capture whose owning thread is GONE (it exited between begin and end): every other thread is refused by begin (one
    # capture per process) and by abort (not theirs, the stream still records) -- bdsp_hip_capture_reset is the way out
    import ctypes as C
    import threading
    rc = []
    th = threading.Thread(target=lambda: (w.scale(1.0), rc.append(bd.lib.bdsp_hip_capture_begin(C.c_void_p(None))), w.scale(1.0)))
    th.start()
    th.join()
    assert rc == [0]
    assert bd.lib.bdsp_hip_capture_begin(C.c_void_p(None)) <= -100 and "already open" in bd._lib.last_error()
    assert bd.lib.bdsp_hip_capture_abort(C.c_void_p(None)) <= -100 and "another thread" in bd._lib.last_error()
    assert bd.lib.bdsp_hip_capture_reset(C.c_void_p(None)) == 0
    assert bd.lib.bdsp_hip_capture_reset(C.c_void_p(None)) == 0   # (nothing open: a no-op)
    g8 = Graph.capture(lambda: (w.scale(1.0), w.offset(0.0)))
    g8.launch()
    del g8
    # a chirp-z plan cannot be BUILT inside a capture (it synchronises the stream): a clean error, not a broken capture
    prime = DspVec(orc.fill_uniform(2 * 10007, 8, -10, 10, np.float32), is_complex=True)
    with pytest.raises(bd.BackendError, match="warm the plan"):
        Graph.capture(lambda: bd._lib.check(prime.plain_fft()), warmup=False)
    g7 = Graph.capture(lambda: (bd._lib.check(prime.plain_ifft()), bd._lib.check(prime.plain_fft())))   # warmed: fine
    g7.launch()
    del g, g2, g3, g4, g5, g6, g7   # releases the workspace blocks and plans the graphs pinned
    w2 = DspVec(x)
    assert w2.scale(2.0) == 0 and np.array_equal(w2.data(), orc.real_scale(x, 2.0))
    del ref3


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_empty_and_tiny_vectors_through_every_operation(dtype):
    # the reference's tests run every operation on empty and 1..3-point vectors too (tests/tools/mod.rs);
    # nothing may crash, lengths and metadata must stay consistent
    ops = [
        ("scale", (2.0,)), ("offset", (1.0,)), ("conj", ()), ("magnitude", ()), ("magnitude_squared", ()),
        ("to_real", ()), ("to_imag", ()), ("phase", ()), ("reverse", ()), ("swap_halves", ()), ("fft_shift", ()),
        ("ifft_shift", ()), ("mirror", ()), ("apply_window", (V.WINDOW_HAMMING,)), ("plain_fft", ()), ("fft", ()),
        ("windowed_fft", (V.WINDOW_HANN,)), ("zero_interleave", (3,)), ("zero_pad", (7, V.PAD_CENTER)),
        ("interpolatef", (V.CONV_SINC, 2.0, 0.0, 4)), ("interpolatei", (V.CONV_SINC, 2)),
        ("decimatei", (2, 0)), ("multiply_complex_exponential", (0.5, 0.25)),
        ("convolve", (V.CONV_SINC, 0.5, 3)), ("prepare_argument", ()),
    ]
    for cplx in (True, False):
        e = 2 if cplx else 1
        for points in (0, 1, 2, 3):
            x = orc.fill_uniform(points * e, 9, -1, 1, dtype)
            for name, args in ops:
                v = DspVec(x, is_complex=cplx) if points else DspVec(is_complex=cplx, dtype=dtype, length=0)
                code = getattr(v, name)(*args)
                assert isinstance(code, int) and code > -100, (name, cplx, points, code)
                d = v.data()
                assert d.size == len(v) and np.all(np.isfinite(d) | np.isnan(d))
            # frequency-domain entry points
            for name, args in (("plain_ifft", ()), ("ifft", ()), ("multiply_frequency_response", (V.CONV_SINC, 1.0)),
                               ("plain_sifft", ())):
                v = DspVec(x, is_complex=True, domain=V.FREQ) if points and cplx else \
                    DspVec(is_complex=True, dtype=dtype, length=0, domain=V.FREQ)
                code = getattr(v, name)(*args)
                assert isinstance(code, int) and code > -100, (name, points, code)
            # binary operations and convolution with tiny operands
            a = DspVec(x, is_complex=cplx) if points else DspVec(is_complex=cplx, dtype=dtype, length=0)
            b = DspVec(x, is_complex=cplx) if points else DspVec(is_complex=cplx, dtype=dtype, length=0)
            assert a.mul(b) > -100 and a.convolve_signal(b) > -100 and len(a) == points * e


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_convolve_signal_long_filters(dtype):
    # more than 1025 taps: overlap-save with L = next_pow2(4 (M-1)) on the batched FFT (the reference's
    # overlap_discard takes any imp_len, convolution.rs:292-462)
    tol = 2e-6 if dtype == np.float32 else 1e-11
    for cplx, n, m in ((True, 200000, 1026), (True, 200000, 5000), (True, 300000, 65537), (False, 150000, 3000),
                       (False, 120000, 5000), (True, 40000, 40000)):
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 77 + m, -10, 10, dtype)
        h = orc.fill_uniform(m * e, 78 + m, -1, 1, dtype) / dtype(np.sqrt(m))
        v = DspVec(x, is_complex=cplx)
        assert v.convolve_signal(DspVec(h, is_complex=cplx)) == 0
        got = v.data()
        x64, h64 = x.astype(np.float64), h.astype(np.float64)
        xc = x64 if cplx else np.stack([x64, np.zeros_like(x64)], -1).reshape(-1)
        hc = h64 if cplx else np.stack([h64, np.zeros_like(h64)], -1).reshape(-1)
        code, ref = orc.overlap_discard(xc, hc, orc.next_power_of_two(m), fair=True)
        assert code == 0
        ref = ref if cplx else ref[0::2]
        assert rel_l2(got, ref) < tol, (cplx, n, m)
        # and two windows against the direct form (start incl. wrap-around, end)
        for first in (0, n - 300):
            d = orc.convolve_direct(x64, h64, cplx, first, 300)
            assert rel_l2(got[first * e:(first + 300) * e], d) < tol, (cplx, n, m, first)


def test_c_client_of_the_abi(tmp_path):
    # the boundary is a C ABI: build a plain-C client against include/basic_dsp_hip.h and run it
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "facade_demo")
    libdir = os.path.join(root, "basic_dsp_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-D_GNU_SOURCE", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "facade_demo.c"), "-L", libdir,
                           "-lbasic_dsp_hip", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c abi demo ok" in out.stdout


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,rows", [(1024, 17000), (2048, 8300), (4096, 4200), (8192, 600), (1000, 3000)])
def test_large_batches_take_the_persistent_kernels(n, rows, dtype):
    # many rows: k_fft_wg_batch (n = 1024..4096), k_fft_wg4 (8192, f32), the fused Bluestein kernel (1000);
    # every row is checked against numpy's f64 transform
    from basic_dsp_amd import DspMat
    if dtype == np.float64 and rows * n > 20_000_000:
        rows //= 2
    rng = np.random.default_rng(n + rows)
    a = (rng.standard_normal((rows, 2 * n)) * 3).astype(dtype)
    m = DspMat(a, is_complex=True)
    assert m.plain_fft() == 0
    got = m.data().astype(np.float64).view(np.complex128)
    ref = np.fft.fft(a.astype(np.float64).view(np.complex128), axis=1)
    err = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < (2e-6 if dtype == np.float32 else 1e-12), (n, rows, float(err.max()), int(err.argmax()))
    assert m.plain_ifft() == 0
    back = m.data().astype(np.float64) / n
    assert rel_l2(back, a) < (2e-6 if dtype == np.float32 else 1e-12)


def test_three_pass_fft_with_a_batch():
    from basic_dsp_amd import DspMat
    n, rows = 1 << 21, 3
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((rows, 2 * n)) * 3).astype(np.float32)
    m = DspMat(a, is_complex=True)
    assert m.fft() == 0  # fused fft_shift on the last pass
    got = m.data().astype(np.float64).view(np.complex128)
    ref = np.fft.fftshift(np.fft.fft(a.astype(np.float64).view(np.complex128), axis=1), axes=1)
    err = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert err.max() < 2e-6


def test_getters_into_a_destination_vector():
    # facade32.rs:564-668: source consumed, destination resized to `points` reals, 9 on success (convert_void)
    x = orc.fill_uniform(2 * 1000, 21, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    for name, kind in (("get_real", 2), ("get_imag", 3), ("get_magnitude", 0), ("get_magnitude_squared", 1),
                       ("get_phase", 4)):
        d = DspVec(np.zeros(5, np.float32))
        assert getattr(v, name)(d) == 9
        assert len(d) == 1000 and not d.is_complex()
        ref = orc.complex_to_real(x, kind)
        if kind in (1, 2, 3):
            assert np.array_equal(d.data(), ref)
        else:
            np.testing.assert_allclose(d.data(), ref, rtol=2e-6, atol=2e-6)
    assert np.array_equal(v.data(), x)  # the Python mirror works on a clone
    d = DspVec(np.zeros(4, np.float32), is_complex=True)
    assert v.get_real(d) == 9 and len(d) == 0  # a complex destination is emptied


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_statistics_sums_dot_products(dtype):
    # KATs: statistics.rs:44-65, :84-91, :113-128, dot_products.rs:338-388
    z = DspVec(np.array([1, 2, 3, 4, 5, 6], dtype), is_complex=True)
    s = z.statistics()
    assert s["sum"] == 9 + 12j and s["count"] == 3 and s["average"] == 3 + 4j
    assert abs(s["rms"] - (3.4027193 + 4.3102784j)) < 1e-4
    assert (s["min"], s["min_index"], s["max"], s["max_index"]) == (1 + 2j, 0, 5 + 6j, 2)
    code, parts = z.statistics_split(2)
    assert code == 0 and parts[0]["sum"] == 6 + 8j and parts[1]["sum"] == 3 + 4j
    assert z.statistics_split(17)[0] == 7
    assert z.sum() == 9 + 12j and z.sum_sq() == -21 + 88j
    r = DspVec(np.array([1, 2, 3], dtype))
    assert r.dot_product(DspVec(np.array([1, 2, 3], dtype))) == (0, 14.0)
    c = DspVec(np.array([1, 0, 3, 0], dtype), is_complex=True)
    assert c.dot_product(DspVec(np.array([1, 0, 3, 0], dtype), is_complex=True)) == (0, 10 + 0j)
    assert c.dot_product(r)[0] == 2 and r.dot_product(r)[0] == 0
    # large vectors against the oracle (sequential sums in T there, double accumulation here)
    tol = 2e-5 if dtype == np.float32 else 1e-12
    for cplx, n in ((False, 1_000_003), (True, 700_001)):
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 5 + n, -10, 10, dtype)
        x[e * 123456] = 77.0   # a unique maximum
        if not cplx:
            x[654321] = -88.0  # and a unique minimum
        v = DspVec(x, is_complex=cplx)
        got = v.statistics()
        ref = (orc.complex_statistics if cplx else orc.real_statistics)(x.astype(np.float64))
        assert got["count"] == ref["count"] == n
        for k in ("sum", "average", "rms"):
            assert abs(got[k] - ref[k]) <= tol * max(1.0, abs(ref[k])) * 50, (k, got[k], ref[k])
        for k in ("min", "max", "min_index", "max_index"):
            assert got[k] == ref[k], (k, got[k], ref[k])
        p = v.statistics(prec=True)
        assert abs(p["sum"] - ref["sum"]) <= 1e-9 * max(1.0, abs(ref["sum"])) * (1e3 if dtype == np.float32 else 1)
        ssum, ssq = v.sum(), v.sum_sq()
        rsum, rsq = orc.vec_sum(x.astype(np.float64), cplx), orc.vec_sum(x.astype(np.float64), cplx, True)
        assert abs(ssum - rsum) <= tol * 50 * max(1.0, abs(rsum)) and abs(ssq - rsq) <= tol * abs(rsq) * 50
        y = orc.fill_uniform(n * e, 9, -1, 1, dtype)
        code, d = v.dot_product(DspVec(y, is_complex=cplx))
        rd = orc.dot(x.astype(np.float64), y.astype(np.float64), cplx)
        assert code == 0 and abs(d - rd) <= tol * 50 * max(1.0, abs(rd))
        code, parts = v.statistics_split(3)
        assert code == 0
        for b in range(3):
            rb = (orc.complex_statistics if cplx else orc.real_statistics)(x.astype(np.float64), b, 3)
            assert parts[b]["count"] == rb["count"] and parts[b]["max_index"] == rb["max_index"]
            assert parts[b]["min_index"] == rb["min_index"]
            assert abs(parts[b]["sum"] - rb["sum"]) <= tol * 50 * max(1.0, abs(rb["sum"]))
    # first occurrence wins ties; an empty vector reports count 0 and NaN averages
    t = DspVec(np.array([2, 7, 7, -3, -3, 0], dtype)).statistics()
    assert (t["max_index"], t["min_index"]) == (1, 3)
    e0 = DspVec(dtype=dtype, length=0).statistics()
    assert e0["count"] == 0 and np.isnan(e0["average"]) and e0["min"] == np.inf and e0["max"] == -np.inf


# ---- the facade's remaining families: per-element math, differences / running sums, pairs, split / merge, callbacks
_MATH_DOMAINS = {  # real input ranges that keep the function real-valued
    "sqrt": (0.0, 50.0), "square": (-10, 10), "ln": (1e-3, 50.0), "exp": (-10, 10), "sin": (-10, 10), "cos": (-10, 10),
    "tan": (-1.4, 1.4), "asin": (-0.99, 0.99), "acos": (-0.99, 0.99), "atan": (-10, 10), "sinh": (-8, 8),
    "cosh": (-8, 8), "tanh": (-8, 8), "asinh": (-10, 10), "acosh": (1.01, 50.0), "atanh": (-0.99, 0.99),
    "abs": (-10, 10), "ln_approx": (1e-3, 50.0), "exp_approx": (-10, 10), "sin_approx": (-10, 10),
    "cos_approx": (-10, 10)}
_MATH_ARGS = {"powf": ((0.1, 10.0), 2.5), "root": ((0.1, 10.0), 3.0), "log": ((1e-3, 50.0), 10.0),
              "expf": ((-3, 3), 10.0), "wrap": ((-20, 20), 4.0), "log_approx": ((1e-3, 50.0), 10.0),
              "expf_approx": ((-3, 3), 10.0), "powf_approx": ((0.1, 10.0), 2.5)}


def _oracle_math(x, cplx, name, arg):
    key = {"ln_approx": "ln", "exp_approx": "exp", "sin_approx": "sin", "cos_approx": "cos", "log_approx": "log"}.get(name, name)
    if name == "root":
        key, arg = "powf", 1.0 / arg
    return orc.math(x.astype(np.float64), cplx, key, arg)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_math_family(dtype):
    # doc-test KATs (trigonometry_and_powers.rs:14-189)
    v = DspVec(np.array([1, 4, 9, 16, 25], dtype))
    assert v.sqrt() == 0 and list(v.data()) == [1, 2, 3, 4, 5]
    v = DspVec(np.array([1, 2, 3], dtype))
    assert v.expf(10.0) == 0
    np.testing.assert_allclose(v.data(), [10, 100, 1000], rtol=1e-6)
    v = DspVec(np.array([1, 8, 27], dtype))
    assert v.root(3.0) == 0
    np.testing.assert_allclose(v.data(), [1, 2, 3], rtol=1e-6)
    tol = 3e-6 if dtype == np.float32 else 1e-13
    n = 100_003
    for name, (lo, hi) in _MATH_DOMAINS.items():
        x = orc.fill_uniform(n, 31, lo, hi, dtype)
        v = DspVec(x)
        assert getattr(v, name)() == 0, name
        ref = _oracle_math(x, False, name, 0.0)
        assert np.max(np.abs(v.data() - ref) / (np.abs(ref) + 1.0)) < tol, name
    for name, ((lo, hi), arg) in _MATH_ARGS.items():
        x = orc.fill_uniform(n, 32, lo, hi, dtype)
        v = DspVec(x)
        assert getattr(v, name)(arg) == 0, name
        ref = _oracle_math(x, False, name, arg)
        assert np.max(np.abs(v.data() - ref) / (np.abs(ref) + 1.0)) < tol * 4, name
    # complex vectors: TrigOps / PowerOps follow num-complex's formulas, the RealOps family poisons
    ctol = 2e-5 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(2 * n, 33, -3, 3, dtype)
    for name in ("sqrt", "square", "ln", "exp", "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh",
                 "asinh", "acosh", "atanh"):
        v = DspVec(x, is_complex=True)
        assert getattr(v, name)() == 0, name
        ref = _oracle_math(x, True, name, 0.0)
        assert rel_l2(v.data(), ref) < ctol, name
    for name, arg in (("powf", 2.5), ("root", 3.0), ("log", 10.0), ("expf", 7.0)):
        v = DspVec(x, is_complex=True)
        assert getattr(v, name)(arg) == 0, name
        assert rel_l2(v.data(), _oracle_math(x, True, name, arg)) < ctol, name
    for name in ("abs", "ln_approx", "sin_approx"):
        v = DspVec(x, is_complex=True)
        assert getattr(v, name)() == -1 and len(v) == 0, name
    for name in ("wrap", "unwrap", "powf_approx"):
        v = DspVec(x, is_complex=True)
        assert getattr(v, name)(2.0) == -1, name
    # special values of the complex square root (sign of zero) and powf(0)
    z = DspVec(np.array([-4, 0, -4, -0.0, 0, 2, 0, -2, 9, 0], dtype), is_complex=True)
    assert z.sqrt() == 0
    np.testing.assert_allclose(z.data(), [0, 2, 0, -2, 1, 1, 1, -1, 3, 0], atol=1e-6)
    z = DspVec(np.array([3, 4, -1, 2], dtype), is_complex=True)
    assert z.powf(0.0) == 0 and list(z.data()) == [1, 0, 1, 0]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_diff_cum_sum_wrap_unwrap(dtype):
    # KATs diff_sum.rs:18-53, real_ops.rs:49-65
    v = DspVec(np.array([2, 3, 2, 6], dtype))
    assert v.diff() == 0 and list(v.data()) == [1, -1, 4]
    v = DspVec(np.array([2, 2, 3, 3, 5, 5], dtype), is_complex=True)
    assert v.diff_with_start() == 0 and list(v.data()) == [2, 2, 1, 1, 2, 2]
    v = DspVec(np.array([2, 1, -1, 4], dtype))
    assert v.cum_sum() == 0 and list(v.data()) == [2, 3, 2, 6]
    v = DspVec(np.arange(1, 9, dtype=dtype))
    assert v.wrap(4.0) == 0 and list(v.data()) == [1, 2, 3, 0, 1, 2, 3, 0]
    assert v.unwrap(4.0) == 0 and list(v.data()) == [1, 2, 3, 4, 5, 6, 7, 8]
    for cplx, n in ((False, 1), (False, 4097), (False, 300_001), (True, 1), (True, 5000), (True, 262_147)):
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 40 + n, -10, 10, dtype)
        for with_start in (False, True):
            v = DspVec(x, is_complex=cplx)
            assert (v.diff_with_start() if with_start else v.diff()) == 0
            assert np.array_equal(v.data(), orc.diff(x, cplx, with_start)), (cplx, n, with_start)   # bit-exact
        v = DspVec(x, is_complex=cplx)
        assert v.cum_sum() == 0
        ref = np.cumsum(x.astype(np.float64).reshape(-1, e), axis=0).reshape(-1)
        scale = np.max(np.abs(ref)) + 1.0
        # f32: the double running sum rounded once; f64: a different (blocked) summation order than numpy's
        assert np.max(np.abs(v.data() - ref)) / scale < (2e-7 if dtype == np.float32 else 1e-13), (cplx, n)
        # ... and the reference's running sum in T (its own rounding) stays within its error bound of ours
        seq = orc.cum_sum(x, cplx)
        assert np.max(np.abs(seq - ref)) / scale < (n * 1e-7 if dtype == np.float32 else n * 1e-16)
    # unwrap: a phase ramp wrapped into (-pi, pi], and random data (the recurrence is sequential: bit-exact)
    t = np.arange(50_001, dtype=np.float64) * 0.37
    for data, div in ((np.angle(np.exp(1j * t)).astype(dtype), 2 * np.pi), (orc.fill_uniform(20_000, 5, -30, 30, dtype), 7.0),
                      (orc.fill_uniform(9000, 6, -1, 1, dtype), 0.25)):
        v = DspVec(data)
        assert v.unwrap(dtype(div)) == 0
        assert np.array_equal(v.data(), orc.unwrap(data, dtype(div)))
    # quotients near integers and large ratios exercise the exact-remainder path and its library fallback
    big = (orc.fill_uniform(4000, 8, -1, 1, dtype) * dtype(1e6)).astype(dtype)
    big[::7] = np.round(big[::7] / 3) * 3
    for data, div in ((big, 3.0), (big, 1e-3), (orc.fill_uniform(3000, 9, -100, 100, dtype), 0.1)):
        v = DspVec(data)
        assert v.unwrap(dtype(div)) == 0
        assert np.array_equal(v.data(), orc.unwrap(data, dtype(div)))
    v = DspVec(np.angle(np.exp(1j * t)).astype(dtype))
    v.unwrap(dtype(2 * np.pi))
    np.testing.assert_allclose(v.data(), t, atol=2e-2 if dtype == np.float32 else 1e-9)
    # empty vectors stay empty
    e0 = DspVec(dtype=dtype, length=0)
    assert e0.diff() == 0 and e0.cum_sum() == 0 and e0.unwrap(1.0) == 0 and len(e0) == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pairs_split_merge_map(dtype):
    n = 70_001
    x = orc.fill_uniform(2 * n, 51, -10, 10, dtype)
    z = DspVec(x, is_complex=True)
    a, b = DspVec(dtype=dtype, length=0), DspVec(dtype=dtype, length=3)
    assert z.get_real_imag(a, b) == 9 and np.array_equal(a.data(), x[0::2]) and np.array_equal(b.data(), x[1::2])
    assert z.get_mag_phase(a, b) == 9
    mag, ph = orc.get_mag_phase(x.astype(np.float64))
    tol = 2e-6 if dtype == np.float32 else 1e-14
    assert rel_l2(a.data(), mag) < tol and np.max(np.abs(b.data() - ph)) < tol * 4
    w = DspVec(dtype=dtype, length=0, is_complex=True)
    assert w.set_mag_phase(a, b) == 0 and len(w) == 2 * n and rel_l2(w.data(), x) < tol * 4
    assert w.set_real_imag(b, a) == 0 and np.array_equal(w.data()[0::2], b.data()) and np.array_equal(w.data()[1::2], a.data())
    assert w.set_real_imag(a, DspVec(np.zeros(5, dtype))) == 7
    c = DspVec(dtype=dtype, length=4, is_complex=True)
    assert DspVec(x[:10]).get_real_imag(a, b) == 9 and len(a) == 0 and len(b) == 0     # real source empties both
    assert z.get_real_imag(a, c) == 9 and len(a) == 0 and len(c) == 0                  # complex target too
    # split_into / merge (data_reorganization.rs:185-212 and round trips)
    v = DspVec(np.arange(1, 11, dtype=dtype))
    t = [DspVec(dtype=dtype, length=0), DspVec(dtype=dtype, length=0)]
    assert v.split_into(t) == 9 and list(t[0].data()) == [1, 3, 5, 7, 9] and list(t[1].data()) == [2, 4, 6, 8, 10]
    m = DspVec(dtype=dtype, length=0)
    assert m.merge([DspVec(np.array([1, 2], dtype)), DspVec(np.array([1, 2], dtype))]) == 0 and list(m.data()) == [1, 1, 2, 2]
    assert DspVec(np.arange(9, dtype=dtype)).split_into(t) == 7 and v.split_into([]) == 7 and m.merge([]) == 7
    assert m.merge([DspVec(np.zeros(2, dtype)), DspVec(np.zeros(3, dtype))]) == 7
    for cplx, parts in ((False, 3), (True, 3), (True, 7), (False, 16)):
        e = 2 if cplx else 1
        xx = orc.fill_uniform(parts * 3001 * e, 60 + parts, -10, 10, dtype)
        src = DspVec(xx, is_complex=cplx)
        tg = [DspVec(dtype=dtype, length=0, is_complex=cplx) for _ in range(parts)]
        assert src.split_into(tg) == 9
        code, ref = orc.split_into(xx, cplx, parts)
        assert code == 0 and all(np.array_equal(tg[k].data(), ref[k]) for k in range(parts))
        back = DspVec(dtype=dtype, length=0, is_complex=cplx)
        assert back.merge(tg) == 0 and np.array_equal(back.data(), xx)
    # host callbacks: map_inplace / map_aggregate
    r = DspVec(x[:1000])
    assert r.map_inplace(lambda val, i: val * 2 + i) == 0
    np.testing.assert_allclose(r.data(), x[:1000] * 2 + np.arange(1000), rtol=1e-6)
    zc = DspVec(x[:1000], is_complex=True)
    assert zc.map_inplace(lambda val, i: val * 1j + i) == 0
    ref = (x[0:1000:2] + 1j * x[1:1000:2]) * 1j + np.arange(500)
    np.testing.assert_allclose(zc.data()[0::2] + 1j * zc.data()[1::2], ref, rtol=1e-6)
    assert DspVec(x[:10]).map_inplace(lambda val, i: val) == 0 and zc.clone().map_inplace(lambda val, i: val) == 0
    code, total = DspVec(x[:1000]).map_aggregate(lambda val, i: val * i, lambda p, q: p + q)
    assert code == 0 and abs(total - float(np.dot(x[:1000].astype(np.float64), np.arange(1000)))) < 1e-3 * 1000
    code, best = zc.map_aggregate(lambda val, i: (abs(val), i), max)
    assert code == 0 and best[1] == int(np.argmax(np.abs(ref)))
    assert DspVec(dtype=dtype, length=0).map_aggregate(lambda val, i: val, max)[0] == 12
    assert zc.clone()._fn("map_aggregate_real") is not None


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_callback_variants_of_convolution_and_interpolation(dtype):
    tol = 5e-6 if dtype == np.float32 else 1e-11
    sinc = lambda t: float(np.sinc(t))
    x = orc.fill_uniform(2 * 3000, 71, -10, 10, dtype)
    # complex impulse response: a real-valued one reproduces the built-in, a complex one the direct sum
    for points, L in ((3000, 9), (11, 5), (7, 20)):
        xx = x[: 2 * points]
        v = DspVec(xx, is_complex=True)
        assert v.convolve_complex(lambda t: complex(np.sinc(t), 0.0), 0.5, L) == 0
        ref = orc.convolve_function(xx.astype(np.float64), True, 0, 0.0, 0.5, L)
        assert rel_l2(v.data(), ref) < tol, (points, L)
        h = lambda t: complex(np.sinc(t), 0.25 * t)
        v = DspVec(xx, is_complex=True)
        assert v.convolve_complex(h, 0.5, L) == 0
        zz = xx[0::2].astype(np.float64) + 1j * xx[1::2]
        Lc = min(L, points)
        ref = np.array([sum(zz[(i + m) % points] * h(-m * 0.5) for m in range(-Lc, Lc + 1)) for i in range(points)])
        got = v.data()[0::2] + 1j * v.data()[1::2]
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, (points, L)
    assert DspVec(x[:100]).convolve_complex(lambda t: 1.0, 0.5, 3) == -1                 # assert_complex!
    # complex frequency response (natural-order axis j / max * ratio, scaled by ratio)
    for points in (1000, 1001):
        xx = x[: 2 * points]
        f = DspVec(xx, is_complex=True, domain=V.FREQ)
        fr = lambda t: complex(1.0 / (1.0 + t * t), 0.5 * t)
        assert f.multiply_frequency_response_complex(fr, 0.5) == 0
        maxv = (points - points % 2) / 2
        hh = np.array([0.5 * fr((-maxv + i) / maxv * 0.5) for i in range(points)])
        ref = (xx[0::2].astype(np.float64) + 1j * xx[1::2]) * hh
        got = f.data()[0::2] + 1j * f.data()[1::2]
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol
    assert DspVec(x[:100], is_complex=True).multiply_frequency_response_complex(lambda t: 1.0, 0.5) == -1
    # interpolation with a callback equals the built-in function of the same shape
    for cplx in (True, False):
        e = 2 if cplx else 1
        xx = x[: 600 * e]
        a, b = DspVec(xx, is_complex=cplx), DspVec(xx, is_complex=cplx)
        assert a.interpolatei_custom(lambda t: 1.0 if abs(t) <= 1.0 else 0.0, 3) == 0 and b.interpolatei(V.CONV_SINC, 3) == 0
        assert rel_l2(a.data(), b.data()) < tol
        a, b = DspVec(xx, is_complex=cplx), DspVec(xx, is_complex=cplx)
        assert a.interpolate_custom(lambda t: 1.0 if abs(t) <= 1.0 else 0.0, 1500, 0.0) == 0
        assert b.interpolate(V.CONV_SINC, 1500, 0.0) == 0
        assert rel_l2(a.data(), b.data()) < tol and a.delta() == b.delta()
        for factor, L in ((4.0, 12), (2.5, 8)):      # table path / scalar path
            a, b = DspVec(xx, is_complex=cplx), DspVec(xx, is_complex=cplx)
            assert a.interpolatef_custom(sinc, factor, 0.0, L) == 0 and b.interpolatef(V.CONV_SINC, factor, 0.0, L) == 0
            assert len(a) == len(b) and rel_l2(a.data(), b.data()) < tol * 4, (cplx, factor)


def test_b1_convolve_vector_pipelined_transfers():
    """Above 2^20 complex points gpu_convolve_vector pipelines upload / blocks / download in chunks: the result must
    equal the device-resident path bit for bit (same blocks, same kernel) and the oracle on windows incl. both ends."""
    for n, m in (((1 << 20) + 12345, 257), ((1 << 21), 1024), (3_000_001, 3), ((1 << 20) + 77, 3073), ((1 << 20) + 5, 2500)):
        x = orc.fill_uniform(2 * n, 77 + n, -10, 10, np.float32)
        h = orc.fill_uniform(2 * m, 78, -1, 1, np.float32) / m
        y, rng = V.gpu_convolve_vector(x, h, True)
        assert rng == (0, 2 * n)
        v = DspVec(x, is_complex=True)
        assert v.convolve_signal(DspVec(h, is_complex=True)) == 0
        assert np.array_equal(y, v.data()), (n, m)
        for start in (0, n // 2 - 50, n - 300):
            ref = orc.convolve_direct(x.astype(np.float64), h.astype(np.float64), True, start, 300)
            assert rel_l2(y[2 * start: 2 * (start + 300)], ref) < 1e-6, (n, m, start)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mixed_radix_fft_lengths_and_options(dtype):
    """2,3,5,7-smooth lengths take the mixed-radix Stockham path (workgroup-resident up to 4096 / 2048 points, four-step
    above): every radix, both forms, every fused option, batches through the matrix API -- against the oracle's DFT."""
    from basic_dsp_amd import DspMat
    tol = 1e-6 if dtype == np.float32 else 1e-12
    for n in (6, 7, 9, 10, 11, 12, 13, 14, 15, 21, 22, 26, 35, 49, 60, 105, 121, 143, 169, 210, 343, 625, 729, 1000, 1001,
              2002, 14641, 28561, 143000, 1536, 2187, 2401, 3000, 3125, 4000,
              4200, 5000, 6000, 6561, 10000, 16807, 30000, 65610, 100000, 250047, 360000):
        x = orc.fill_uniform(2 * n, 300 + n, -10, 10, dtype)
        v = DspVec(x, is_complex=True)
        assert v.plain_fft() == 0
        ref = orc.fft(x.astype(np.float64))
        assert rel_l2(v.data(), ref) < tol, n
        assert v.plain_ifft() == 0
        assert rel_l2(v.data() / n, x) < tol * 2, n
    for n in (30, 1000, 2187, 3000, 12000, 16807, 100000):      # 2187 = 3^7 and 16807 = 7^5 are odd: rotations by n/2 and n - n/2
        x = orc.fill_uniform(2 * n, 17 + n, -10, 10, dtype)
        xf = x.astype(np.float64)
        # fft (shifted), ifft (scaled, unshifted), windowed pair, magnitude, real input
        v = DspVec(x, is_complex=True)
        assert v.fft() == 0
        ref = np.array(orc.fft(xf)); ref = orc.swap_halves(ref, True, True)
        assert rel_l2(v.data(), ref) < tol, n
        assert v.ifft() == 0 and rel_l2(v.data(), x) < tol * 2, n
        v = DspVec(x, is_complex=True)
        assert v.windowed_fft(V.WINDOW_HAMMING) == 0 and v.windowed_ifft(V.WINDOW_HAMMING) == 0
        assert rel_l2(v.data(), x) < tol * 20, n
        v = DspVec(x, is_complex=True)
        assert v.plain_fft() == 0
        spec = v.data().copy()
        assert v.magnitude() == 0
        np.testing.assert_allclose(v.data(), np.hypot(spec[0::2], spec[1::2]), rtol=1e-5)
        r = DspVec(x[:n])
        assert r.plain_fft() == 0 and len(r) == 2 * n
        zr = np.zeros(2 * n); zr[0::2] = xf[:n]
        assert rel_l2(r.data(), orc.fft(zr)) < tol, n
    # batches of smooth lengths
    for rows, n in ((37, 100), (5, 3000), (3, 50000)):
        xs = orc.fill_uniform(2 * n * rows, 9, -10, 10, dtype).reshape(rows, 2 * n)
        m = DspMat(xs, is_complex=True)
        assert m.plain_fft() == 0
        got = m.data()
        for k in (0, rows // 2, rows - 1):
            assert rel_l2(got[k], orc.fft(xs[k].astype(np.float64))) < tol, (rows, n, k)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_register_resident_three_stage_batches(dtype):
    """Round 6: k_mr_reg3 -- transforms of n = R0 R1 R2 (1000 = 10 10 10, 360 = 10 6 6, 2000 = 20 10 10, 3000 = 20 15 10 with
    512-thread workgroups ...) in registers, persistent workgroups, two LDS exchanges -- and k_mr_reg2, its two-stage sibling
    for n = R0 R1 < 300 (100 = 10 10, 240 = 16 15 ...).  EVERY built length, batches of 515 ... 2053 rows (a ragged count, so that the last workgroup's second transform is empty), forward and inverse: rows against the
    f64 oracle; real rows (the LDS-staged input path) against the same rows as complex data (the plain path), which pins every
    row and the row order; fft() / ifft() with their fused shift and scale, odd lengths included.  Matches
    time_freq/mod.rs:47-58 (any length), time_to_freq.rs:158-165."""
    tol = 1e-6 if dtype == np.float32 else 1e-12
    # every built length: the table of basic_dsp_amd/csrc/mixed_radix_reg3.h (tools/gen_reg3_table.py), restated
    import itertools
    best = {}
    for t in itertools.combinations_with_replacement([4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25], 3):
        n = t[0] * t[1] * t[2]
        if n > 4096 or n < 300 or n & (n - 1) == 0 or n // min(t) > 512:
            continue
        key = (n // min(t) <= 256, min(t), -max(t))
        if n not in best or key > best[n]:
            best[n] = key
    lengths = [n for n in sorted(best) if dtype == np.float32 or n <= 2048]
    assert len(lengths) == (67 if dtype == np.float32 else 51) and all(n in lengths for n in (360, 1000, 2000)) and (3000 in lengths) == (dtype == np.float32)
    # ... and the two-stage kernel k_mr_reg2 (mixed_radix_reg2.h): every n = R0 R1 < 300, the same radix set
    short = sorted({a * b for a in (4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25) for b in (4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 25)
                    if a * b < 300 and (a * b) & (a * b - 1)})
    assert len(short) == 31 and short[0] == 20 and 100 in short and 250 in short
    from basic_dsp_amd import DspMat
    for n in short + lengths:
        rows = 2053 if n < 300 else (1027 if n <= 1200 else 515)
        xs = orc.fill_uniform(2 * n * rows, 600 + n, -10, 10, dtype).reshape(rows, 2 * n)
        m = DspMat(xs, is_complex=True)
        assert m.plain_fft() == 0
        got = m.data()
        for k in (0, 1, rows // 2, rows - 2, rows - 1):
            assert rel_l2(got[k], orc.fft(xs[k].astype(np.float64))) < tol, (n, k, rel_l2(got[k], orc.fft(xs[k].astype(np.float64))))
        # REAL rows (real input is a fused option: the kernel's LDS-staged input path) against the same rows as complex data with
        # zero imaginary parts (its plain path): same values to rounding, row for row -- which pins every row and the row order
        xr = np.ascontiguousarray(xs[:, :n])
        zc = np.zeros_like(xs)
        zc[:, 0::2] = xr
        a, b = DspMat(zc, is_complex=True), DspMat(xr, is_complex=False)
        assert a.plain_fft() == 0 and b.plain_fft() == 0
        assert rel_l2(a.data().ravel(), b.data().ravel()) < 4 * tol, n
        assert m.plain_ifft() == 0
        back = m.data() / n
        assert rel_l2(back.ravel(), xs.ravel()) < 4 * tol, (n, "round trip")
        # the fused options the kernel stages through LDS: fft() = transform + fft_shift (odd lengths rotate by n - n/2),
        # ifft() = scale(1/n) + ifft_shift + inverse transform (time_to_freq.rs:158-165, freq_to_time.rs:160-168)
        f = DspMat(xs, is_complex=True)
        assert f.fft() == 0
        gf = f.data()
        for k in (0, rows - 1):
            ref = orc.swap_halves(orc.fft(xs[k].astype(np.float64)), True, True)
            assert rel_l2(gf[k], ref) < tol, (n, k, "fft() with the shift fused")
        assert f.ifft() == 0
        assert rel_l2(f.data().ravel(), xs.ravel()) < 4 * tol, (n, "fft() -> ifft()")
        if n in (45, 100, 250, 375, 1000, 1875, 2000, 3000, 3375):
            # windows ride in the same staging loops: windowed_fft = window + transform + shift, every reference window
            for wid, orc_id, alpha in ((V.WINDOW_TRIANGULAR, 0, 0.0), (V.WINDOW_HAMMING, 1, 0.54), (V.WINDOW_BLACKMAN_HARRIS, 2, 0.0), (V.WINDOW_HANN, 1, 0.5)):
                w = DspMat(xs[:9], is_complex=True)
                assert w.windowed_fft(wid) == 0
                ref = orc.swap_halves(orc.fft(orc.apply_window(xs[8].astype(np.float64), True, orc_id, alpha)), True, True)
                assert rel_l2(w.data()[8], ref) < tol, (n, wid, "windowed_fft")
            w = DspMat(xs[:9], is_complex=True)
            assert w.windowed_fft(V.WINDOW_HAMMING) == 0 and w.windowed_ifft(V.WINDOW_HAMMING) == 0
            assert rel_l2(w.data().ravel(), xs[:9].ravel()) < 20 * tol, (n, "windowed_fft -> windowed_ifft")


def test_b1_convolve_vector_from_concurrent_threads():
    """GpuSupport functions are called from whatever thread owns the vector: four threads run the pipelined
    gpu_convolve_vector (its own streams and helper thread per call) at once; results equal the sequential ones."""
    import threading
    n = (1 << 20) + 333
    xs = [orc.fill_uniform(2 * n, 10 + k, -10, 10, np.float32) for k in range(4)]
    h = orc.fill_uniform(2 * 700, 99, -1, 1, np.float32) / 700
    ref = [V.gpu_convolve_vector(x, h, True)[0] for x in xs]
    out = [None] * 4

    def work(k):
        for _ in range(3):
            out[k] = V.gpu_convolve_vector(xs[k], h, True)[0]
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(np.array_equal(out[k], ref[k]) for k in range(4))


def test_randomised_differential_run_of_the_hot_path():
    """Fifteen seconds of tools/fuzz_hot_path.py (seeded): random lengths, tap counts, factors, precisions, families."""
    import subprocess, sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([_sys.executable, os.path.join(root, "tools", "fuzz_hot_path.py"), "15", "2024"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
