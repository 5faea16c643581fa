"""Parity at BASELINE.json's FULL sizes, through the C ABI on the GPU.

The oracle cannot produce a full 16M-point x 1024-tap convolution in seconds, so these tests combine
(a) oracle comparisons where the oracle is fast enough (whole 1M/4M-point transforms, windows of the
16M-point convolution, single DFT bins) with (b) size-independent properties: round trips, Parseval,
linearity, shift identities.  Tolerances are north_star's: 1e-6 rel-L2 for f32, 1e-12 for f64.
"""
import os

import numpy as np
import pytest

import oracle_lib as orc
from basic_dsp_amd import DspVec
from basic_dsp_amd import vector as V

pytestmark = pytest.mark.gpu

SEED_C2, SEED_C3_X, SEED_C3_H, SEED_C4 = 201511212, 201601171, 201601172, 201602221


def rel_l2(got, ref):
    got = np.asarray(got, np.float64).ravel()
    ref = np.asarray(ref, np.float64).ravel()
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))


def test_c2_fft_magnitude_1m_vs_oracle():
    n = 1 << 20
    x = orc.fill_uniform(2 * n, SEED_C2, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0 and v.magnitude() == 0
    ref = orc.magnitude(orc.fft(x.astype(np.float64)))
    assert len(v) == n and rel_l2(v.data(), ref) < 1e-6


def test_c3_convolution_16m_windows_and_identities():
    n, m = 1 << 24, 1024
    x = orc.fill_uniform(2 * n, SEED_C3_X, -10, 10, np.float32)
    h = (orc.fill_uniform(2 * m, SEED_C3_H, -1, 1, np.float32) / np.float32(m)).astype(np.float32)
    hv = DspVec(h, is_complex=True)
    v = DspVec(x, is_complex=True)
    assert v.convolve_signal(hv) == 0
    y = v.data()
    # (i) direct-form oracle (f64) on windows: start and end (both wrap around), block seams, interior
    x64, h64 = x.astype(np.float64), h.astype(np.float64)
    for first in (0, 3072 - 8, 5_000_000, 11_184_810, n - 4096):
        ref = orc.convolve_direct(x64, h64, True, first, 4096)
        assert rel_l2(y[2 * first:2 * (first + 4096)], ref) < 1e-6, first
    # (i') the whole vector against the oracle's own overlap-save schedule in f64 (the tail-free variant;
    # the oracle tests pin it to the direct form and to the reference's schedule on small sizes)
    code, ref = orc.overlap_discard(x64, h64, orc.next_power_of_two(m), fair=True)
    assert code == 0 and rel_l2(y, ref) < 1e-6
    del ref
    # (ii) linearity on the full vector: conv(2.5 x + z) == 2.5 conv(x) + conv(z)
    z = orc.fill_uniform(2 * n, SEED_C3_X + 7, -10, 10, np.float32)
    vz = DspVec(z, is_complex=True)
    assert vz.convolve_signal(hv) == 0
    mix = DspVec(x, is_complex=True)
    assert mix.scale(2.5) == 0 and mix.add(DspVec(z, is_complex=True)) == 0 and mix.convolve_signal(hv) == 0
    assert rel_l2(mix.data(), 2.5 * y.astype(np.float64) + vz.data()) < 1e-6
    # (iii) shift identity at full size (convolution.rs:819-842): taps [0, 0, 1] delay by one sample with
    # wrap-around; a single tap [1] is the identity
    d = np.zeros(2 * 3, np.float32)
    d[4] = 1.0
    s = DspVec(x, is_complex=True)
    assert s.convolve_signal(DspVec(d, is_complex=True)) == 0
    assert rel_l2(s.data(), np.roll(x, 2)) < 1e-6
    one = DspVec(np.array([1.0, 0.0], np.float32), is_complex=True)
    s = DspVec(x, is_complex=True)
    assert s.convolve_signal(one) == 0
    assert rel_l2(s.data(), x) < 1e-6


def test_fft_16m_bins_parseval_roundtrip():
    n = 1 << 24
    x = orc.fill_uniform(2 * n, SEED_C3_X, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    X = v.datac().astype(np.complex128)
    xc = x.view(np.complex64).astype(np.complex128)
    # selected bins against the DFT definition evaluated in f64 (exact phase via integer k*n mod N)
    idx = np.arange(n, dtype=np.int64)
    scale = np.sqrt(n) * 10.0  # typical bin magnitude
    for k in (0, 1, 4097, n // 2, n - 1, 12_345_678):
        ph = (idx * k) % n
        ref = np.sum(xc * np.exp(-2j * np.pi * ph / n))
        assert abs(X[k] - ref) / scale < 2e-6, k
    # the whole spectrum against the f64 oracle transform (north_star tolerance 1e-6 rel-L2)
    ref = orc.fft(x.astype(np.float64))
    assert rel_l2(X.view(np.float64), ref) < 1e-6
    del ref
    # Parseval
    e_t, e_f = np.sum(np.abs(xc) ** 2), np.sum(np.abs(X) ** 2) / n
    assert abs(e_f - e_t) / e_t < 1e-6
    # round trip and delta bookkeeping
    assert v.delta() == float(n)
    assert v.plain_ifft() == 0 and v.scale(1.0 / n) == 0
    assert rel_l2(v.data(), x) < 2e-6
    # a pure tone lands in one bin
    k0 = 1_000_003
    tone = np.exp(2j * np.pi * ((idx * k0) % n) / n).astype(np.complex64)
    t = DspVec(tone, is_complex=True)
    assert t.plain_fft() == 0
    T = t.datac()
    assert abs(T[k0] - n) / n < 1e-6
    T[k0] = 0
    assert np.linalg.norm(T) / n < 1e-6


def test_c4_f64_windowed_fft_and_interpolatef_4m():
    n = 1 << 22
    x = orc.fill_uniform(2 * n, SEED_C4, -10, 10, np.float64)
    # (i) windowed_fft(Hann) against the oracle: window -> fft -> fft_shift
    v = DspVec(x, is_complex=True)
    assert v.windowed_fft(V.WINDOW_HANN) == 0
    ref = orc.swap_halves(orc.fft(orc.apply_window(x, True, 1, 0.5)), True, True)
    assert rel_l2(v.data(), ref) < 1e-12
    assert v.windowed_ifft(V.WINDOW_HANN) == 0
    got = v.data()
    # the Hann window tends to zero at both ends (w(n) ~ (pi n / N)^2): un-applying it amplifies the
    # transform's rounding error by 1/w there, so the round trip is checked where w > 0.03
    edge = 2 * (n // 16)
    assert rel_l2(got[edge:-edge], x[edge:-edge]) < 1e-11
    # (ii) interpolatef(RC 0.35, x4, delay 0, conv_len 12) against the oracle (same tap windows)
    v = DspVec(x, is_complex=True)
    assert v.interpolatef(V.CONV_RAISED_COSINE, 4.0, 0.0, 12, rolloff=0.35) == 0
    ref, path = orc.interpolatef(x, True, 1, 0.35, 4.0, 0.0, 12)
    assert path == 1 and len(v) == ref.size == 8 * n
    assert rel_l2(v.data(), ref) < 1e-13


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cplx", [True, False])
def test_interpolatef_fractional_factor_2m_points(cplx, dtype):
    """The fractional-factor path (k_interp_scalar_v2; the reference's scalar path, interpolation.rs:92-131) at 2 000 000
    points -> 5 000 000: sinc and raised cosine, real and complex, both precisions, the whole output against the oracle
    in the same precision (the reference computes its sampling positions i / factor in T).  Roll-off 0.25 and delay 0: the
    outputs with an integer position walk over both removable singularities (conv_types.rs:406-424)."""
    e = 2 if cplx else 1
    n = 2_000_000
    x = orc.fill_uniform(e * n, SEED_C4 + 5, -10, 10, dtype)
    for fid, rolloff in [(0, 0.0), (1, 0.25)]:
        v = DspVec(x, is_complex=cplx, delta=1.0)
        assert v.interpolatef(fid, 2.5, 0.0, 12, rolloff) == 0
        ref, path = orc.interpolatef(x, cplx, fid, rolloff, dtype(2.5), 0.0, 12)
        assert path == 0 and len(v) == ref.size == e * 5_000_000
        got = v.data()
        assert rel_l2(got, ref) < (2e-6 if dtype == np.float32 else 1e-13), (fid, rel_l2(got, ref))
        # both ends of the result wrap around the input (WrappingIterator, mod.rs:725-786)
        assert rel_l2(got[:e * 64], ref[:e * 64]) < (2e-6 if dtype == np.float32 else 1e-13)
        assert rel_l2(got[-e * 64:], ref[-e * 64:]) < (2e-6 if dtype == np.float32 else 1e-13)
        del got, ref, v


def test_c5_shard_of_64_1m_vectors_against_the_oracle():
    """BASELINE config C5, one GPU's shard: 64 vectors of 1 048 576 complex f32 points through the batched
    convolve_signal(1024 taps) -> plain_fft (basic_dsp_amd.batch.process_shard_gpu: two launches for the whole
    shard).  One vector in eight is checked against the oracle's f64 overlap-save + transform of the same f32
    input, three more against the single-vector path of the library, and the chunked pipeline
    (scatter_process_gather_chunked, world size 1) must reproduce the shard bit for bit."""
    import torch
    from basic_dsp_amd.batch import process_shard_gpu, scatter_process_gather_chunked
    nvec, n, m = 64, 1 << 20, 1024
    rows = np.stack([orc.fill_uniform(2 * n, SEED_C2 + r, -10, 10, np.float32) for r in range(nvec)])
    taps = (orc.fill_uniform(2 * m, SEED_C3_H, -1, 1, np.float32) / np.float32(m)).astype(np.float32)
    dev_rows, dev_taps = torch.from_numpy(rows).cuda(), torch.from_numpy(taps).cuda()
    out = process_shard_gpu(dev_rows, dev_taps, n)
    assert out.data_ptr() != dev_rows.data_ptr()
    assert np.array_equal(dev_rows.cpu().numpy(), rows)  # the compute step leaves its input alone (round 3 overwrote it)
    # no device-wide synchronize here: .cpu() on torch's stream must already be ordered behind the kernels (they run
    # on the stream process_shard_gpu was given, torch's current one)
    out = out.cpu().numpy()
    for r in range(0, nvec, 8):
        code, y = orc.overlap_discard(rows[r].astype(np.float64), taps.astype(np.float64), 0, fair=True)
        assert code == 0
        assert rel_l2(out[r], orc.fft(y)) < 1e-6, r
    for r in (1, 31, nvec - 1):
        v = DspVec(rows[r], is_complex=True)
        assert v.convolve_signal(DspVec(taps, is_complex=True)) == 0 and v.plain_fft() == 0
        assert rel_l2(out[r], v.data()) < 1e-6
    piped = scatter_process_gather_chunked(dev_rows, dev_taps, n, process_shard_gpu, chunk_vectors=8)
    assert np.array_equal(piped.cpu().numpy(), out)


@pytest.mark.parametrize("bits,dtype", [(19, np.float32), (21, np.float32), (22, np.float32), (19, np.float64), (21, np.float64), (22, np.float64)])
def test_two_pass_lengths_2m_and_4m_with_every_fused_option(bits, dtype):
    """2^21 and 2^22 points run as TWO passes of long columns (plan_passes, fft_impl.h; round 5: 2^22 f32 = 4096-point
    columns in 4-wide tiles, then 1024-point columns), and from 2^19 points on the last pass of a two-pass plan runs in
    place (capi.cpp fft_two_buffers: the result is left in the trade buffer).  Against the oracle's f64 transform:
    plain_fft, fft (shift fused into the last pass), windowed_fft (window fused into the first), plain_fft -> magnitude
    (reshaping output: not in place), the round trip through ifft, and a REAL vector's plain_fft (real input read by the
    first pass)."""
    n = 1 << bits
    tol = 1e-6 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(2 * n, SEED_C2 + bits, -10, 10, dtype)
    ref = orc.fft(x.astype(np.float64))
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0 and rel_l2(v.data(), ref) < tol
    assert v.plain_ifft() == 0 and rel_l2(v.data() / n, x) < 2 * tol
    v = DspVec(x, is_complex=True)
    assert v.fft() == 0 and rel_l2(v.data(), orc.swap_halves(ref, True, True)) < tol
    assert v.ifft() == 0 and rel_l2(v.data(), x) < 2 * tol
    v = DspVec(x, is_complex=True)
    assert v.windowed_fft(V.WINDOW_HAMMING) == 0
    w = orc.apply_window(x.astype(np.float64), True, 1, 0.54)
    assert rel_l2(v.data(), orc.swap_halves(orc.fft(w), True, True)) < tol
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0 and v.magnitude() == 0
    assert len(v) == n and rel_l2(v.data(), orc.magnitude(ref)) < tol
    xr = x[:n].copy()
    v = DspVec(xr, is_complex=False)
    assert v.plain_fft() == 0 and v.is_complex() and len(v) == 2 * n
    xc = np.zeros(2 * n, np.float64)
    xc[0::2] = xr
    assert rel_l2(v.data(), orc.fft(xc)) < tol


@pytest.mark.parametrize("bits,dtype", [(13, np.float32), (19, np.float32), (20, np.float32), (22, np.float32), (23, np.float32), (14, np.float64), (19, np.float64), (21, np.float64)])
def test_every_reference_window_fused_into_the_first_global_pass(bits, dtype):
    """Above 4096 points all four reference windows (triangular, Hamming / Hann, Blackman-Harris, rectangular) are applied
    in the registers of the first global pass (k_fft_pass: two sincospi per thread + host constants for the cosine
    windows, the harmonics of Blackman-Harris by Chebyshev recurrences), with and without the fused ifft_shift that
    renames the registers.  Against the oracle's window (window_functions.rs:26-132, evaluated symmetrically) and f64
    transform."""
    n = 1 << bits
    tol = 1e-6 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(2 * n, SEED_C2 + 7 * bits, -10, 10, dtype)
    for wid, oid, alpha in ((V.WINDOW_TRIANGULAR, 0, 0.0), (V.WINDOW_HAMMING, 1, 0.54), (V.WINDOW_HANN, 1, 0.5),
                            (V.WINDOW_BLACKMAN_HARRIS, 2, 0.0), (V.WINDOW_RECTANGULAR, 3, 0.0)):
        v = DspVec(x, is_complex=True)
        assert v.windowed_fft(wid) == 0
        w = orc.apply_window(x.astype(np.float64), True, oid, alpha)
        assert rel_l2(v.data(), orc.swap_halves(orc.fft(w), True, True)) < tol, wid
        # windowed_ifft = ifft (scale + ifft_shift fused into the first pass) then the division by the window: the
        # round trip restores the signal where the window is not ~0
        if wid in (V.WINDOW_HAMMING, V.WINDOW_RECTANGULAR):
            assert v.windowed_ifft(wid) == 0
            assert rel_l2(v.data(), x) < (2e-5 if dtype == np.float32 else 1e-9), wid
        # windowed_ifft by itself (the division by the window in the registers of the LAST global pass) against the
        # oracle's ifft + unapply_window of the same spectrum.  A division amplifies the window's own rounding (1e-7 in
        # f32, absolute) by 1 / w, and the triangular and Blackman-Harris windows are small towards the ends (2e-4 at
        # n/64): compare the central half for those two, all but the first and last 1/64 for the others
        spec = orc.fill_uniform(2 * n, SEED_C2 + 11 * bits + wid, -10, 10, dtype)
        v = DspVec(spec, is_complex=True, domain=1)
        assert v.windowed_ifft(wid) == 0
        t = orc.fft(orc.swap_halves(spec.astype(np.float64), True, False), inverse=True) / n
        t = orc.apply_window(t, True, oid, alpha, unapply=True)
        cut = 2 * (n // 4) if wid in (V.WINDOW_TRIANGULAR, V.WINDOW_BLACKMAN_HARRIS) else 2 * (n // 64)
        assert rel_l2(v.data()[cut:-cut], t[cut:-cut]) < 4 * tol, wid


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fft_domain_interpolation_family_at_1m_points(dtype):
    """interpolatei / interpolate / interpft at sizes where every transform runs in global passes.  interpolatei takes the
    N-point transform and reads its spectrum `factor` times under the frequency response (the transform of the
    zero-interleaved vector IS that periodic repetition); interpolate / interpft put the linear phase, the centred
    zero-padding and the response into one resampling trip (capi.cpp op_interpolatei / op_interpolate,
    elementwise.hip k_spectrum_resample).  Against the oracle's literal restatement of interpolation.rs:484-605."""
    n = 1 << 20
    for cplx in (True, False):
        e = 2 if cplx else 1
        x = orc.fill_uniform(n * e, 4711 + e, -10, 10, dtype)
        for fid, ro, factor in ((0, 0.0, 2), (1, 0.35, 3)):
            v = DspVec(x, is_complex=cplx)
            assert v.interpolatei(fid, factor, ro) == 0
            code, ref = orc.interpolatei(x.astype(np.float64), cplx, fid, ro, factor)
            assert code == 0 and len(v) == ref.size and v.is_complex() == cplx
            assert rel_l2(v.data(), ref) < (5e-6 if dtype == np.float32 else 1e-11), (cplx, fid, factor)
        for dest, delay in ((3 * n // 2, 0.0), (2 * n, 0.3), (2 * n + 1, 0.0)):
            v = DspVec(x, is_complex=cplx, delta=0.5)
            assert v.interpolate(V.CONV_SINC, dest, delay) == 0
            code, ref, nd = orc.interpolate(x.astype(np.float64), cplx, 0, 0.0, dest, delay, 0.5)
            assert len(v) == ref.size and v.delta() == pytest.approx(nd, rel=1e-6)
            assert rel_l2(v.data(), ref) < (2e-5 if dtype == np.float32 else 1e-10), (cplx, dest, delay)
        v = DspVec(x, is_complex=cplx)
        assert v.interpft(2 * n) == 0
        code, ref, _ = orc.interpolate(x.astype(np.float64), cplx, -1, 0.0, 2 * n, 0.0, 1.0)
        assert rel_l2(v.data(), ref) < (2e-5 if dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("n,dtype", [(1_000_003, np.float32), (2_000_003, np.float32), (1_000_003, np.float64)])
def test_bluestein_lengths_over_two_pass_transforms(n, dtype):
    """Lengths with a large prime factor run Bluestein's chirp-z on power-of-two transforms of m >= 2n - 1 points:
    1 000 003 -> m = 2^21, 2 000 003 -> m = 2^22, the two lengths whose plan changed to two passes this round (the f32
    2^21 one finishes in its scratch buffer).  Against numpy's f64 transform of the same input; round trip."""
    x = orc.fill_uniform(2 * n, 77 + n, -10, 10, dtype)
    ref = np.fft.fft(x.astype(np.float64).view(np.complex128)).view(np.float64)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    assert rel_l2(v.data(), ref) < (2e-6 if dtype == np.float32 else 1e-12)
    assert v.plain_ifft() == 0
    assert rel_l2(v.data().astype(np.float64) / n, x) < (4e-6 if dtype == np.float32 else 1e-12)


def test_c2_batch_of_40_fft_magnitude_in_one_call():
    """40 x 1 048 576-point complex f32 plain_fft -> magnitude in ONE device call (config C2 batched): 320 MB of data in
    one piece (rounds 2-4 walked such a batch in cache-sized chunks of vectors; on valid data that bought nothing, DESIGN.md
    4.2).  The magnitudes come back compact (1M reals per vector at the head of the data buffer): every vector must equal
    the single-vector fused call bit for bit, vector 0 and the last one also the oracle's f64 transform."""
    import ctypes as C
    import torch
    from basic_dsp_amd import _lib
    from basic_dsp_amd._lib import FFT_MAGNITUDE
    lib, sp = _lib.lib, _lib.torch_stream_arg()
    nvec, n = 40, 1 << 20
    rows = np.stack([orc.fill_uniform(2 * n, SEED_C2 + 7 * r, -10, 10, np.float32) for r in range(nvec)])
    data = torch.from_numpy(rows).cuda().reshape(-1)
    scratch = torch.empty_like(data)
    flag = C.c_int(-1)
    assert lib.bdsp_hip_dev_fft(0, data.data_ptr(), scratch.data_ptr(), n, nvec, FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp) == 0
    assert flag.value == 0  # result in the data buffer
    got = data[: nvec * n].reshape(nvec, n).cpu().numpy()
    one, one_scratch = torch.empty(2 * n, device="cuda"), torch.empty(2 * n, device="cuda")
    for r in range(nvec):
        one.copy_(torch.from_numpy(rows[r]))
        assert lib.bdsp_hip_dev_fft(0, one.data_ptr(), one_scratch.data_ptr(), n, 1, FFT_MAGNITUDE, 1.0, -1, 0.0, C.byref(flag), sp) == 0
        single = (one_scratch if flag.value else one)[:n].cpu().numpy()
        assert np.array_equal(got[r], single), r
    for r in (0, nvec - 1):
        assert rel_l2(got[r], orc.magnitude(orc.fft(rows[r].astype(np.float64)))) < 1e-6, r


@pytest.mark.gpu
def test_mixed_radix_three_million_points():
    """3 000 000 = 2^6 3 5^6 points: the four-step mixed-radix form with 4-wide tiles (factors 1500 x 2000), against
    the oracle's f64 transform of the same f32 input; round trip."""
    n = 3_000_000
    x = orc.fill_uniform(2 * n, 424242, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    ref = orc.fft(x.astype(np.float64))
    got = v.data().astype(np.float64)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-6
    assert v.plain_ifft() == 0
    back = v.data().astype(np.float64) / n
    assert np.linalg.norm(back - x) / np.linalg.norm(x) < 2e-6


@pytest.mark.parametrize("n,dtype", [(2_000_000, np.float64), (3_000_000, np.float64), (3 * (1 << 20), np.float64), (10_000_000, np.float64), (6_000_000, np.float32), (16_000_000, np.float32)])
def test_mixed_radix_three_stockham_passes(n, dtype):
    """Round 5: smooth lengths whose four-step form would need tiles narrower than four columns (f64 beyond factors of 1024
    points, i.e. above 10^6; f32 beyond 2048, about 4.2M points) or has no two-factor split at all (10^7 in f64) run as
    THREE global Stockham passes with any smooth super-radix (k_mr_gpass: n = r0 r1 r2, tiles at most 8 columns wide in f32 and 4 in f64 (64-byte runs);
    the result ends in the trade buffer).  Until round 5 the chirp-z path served them at 2.6 ... 3.6 times the time and twice
    the error.  plain_fft against the oracle's f64 transform, fft -> ifft round trip with the shifts fused, and a
    Hann-windowed transform."""
    tol = 1e-6 if dtype == np.float32 else 1e-12
    x = orc.fill_uniform(2 * n, 515151 + n % 97, -10, 10, dtype)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    ref = orc.fft(x.astype(np.float64))
    assert rel_l2(v.data(), ref) < tol
    del v
    v = DspVec(x, is_complex=True)
    assert v.fft() == 0
    assert rel_l2(v.data(), orc.swap_halves(ref, True, True)) < tol
    assert v.ifft() == 0 and rel_l2(v.data(), x) < 2 * tol
    del ref
    if n <= 4_000_000:
        v = DspVec(x, is_complex=True)
        assert v.windowed_fft(V.WINDOW_HANN) == 0
        w = orc.apply_window(x.astype(np.float64), True, 1, 0.5)
        assert rel_l2(v.data(), orc.swap_halves(orc.fft(w), True, True)) < tol
        # a REAL vector (the first pass reads n reals, the second writes n complex values over them) and the magnitude
        # of its spectrum (the last pass writes n reals)
        xr = x[:n].copy()
        v = DspVec(xr, is_complex=False)
        assert v.plain_fft() == 0 and v.is_complex() and len(v) == 2 * n
        xc = np.zeros(2 * n, np.float64)
        xc[0::2] = xr
        refr = orc.fft(xc)
        assert rel_l2(v.data(), refr) < tol
        assert v.magnitude() == 0 and rel_l2(v.data(), orc.magnitude(refr)) < tol


@pytest.mark.gpu
def test_f64_and_real_convolution_4m_whole_output():
    """The f64 block kernel (filter spectrum and twiddles in registers, taps transformed in the kernel) and the real
    two-for-one variant at 4M points: whole output against the oracle's f64 overlap-save, windows against the direct form."""
    n, m = 1 << 22, 1024
    x = orc.fill_uniform(2 * n, SEED_C3_X + 1, -10, 10, np.float64)
    h = orc.fill_uniform(2 * m, SEED_C3_H + 1, -1, 1, np.float64) / m
    v = DspVec(x, is_complex=True)
    assert v.convolve_signal(DspVec(h, is_complex=True)) == 0
    y = v.data()
    code, ref = orc.overlap_discard(x, h, orc.next_power_of_two(m), fair=True)
    assert code == 0 and rel_l2(y, ref) < 1e-12
    for first in (0, 3072 - 8, n // 2, n - 4096):
        assert rel_l2(y[2 * first:2 * (first + 4096)], orc.convolve_direct(x, h, True, first, 4096)) < 1e-12, first
    for dtype, tol in ((np.float32, 1e-6), (np.float64, 1e-12)):
        xr = orc.fill_uniform(n, SEED_C3_X + 2, -10, 10, dtype)
        hr = (orc.fill_uniform(m - 1, SEED_C3_H + 2, -1, 1, dtype) / dtype(m)).astype(dtype)   # odd tap count
        r = DspVec(xr)
        assert r.convolve_signal(DspVec(hr)) == 0
        yr = r.data()
        for first in (0, n // 3, n - 4096):
            ref = orc.convolve_direct(xr.astype(np.float64), hr.astype(np.float64), False, first, 4096)
            assert rel_l2(yr[first:first + 4096], ref) < tol, (dtype, first)


@pytest.mark.gpu
def test_bench_py_runs_both_modes_and_prints_the_contract_fields():
    """bench.py end to end on this GPU (few steps): the headline mode and --mode c5, one JSON line each with the
    contract's fields, `roofline` and (headline) `cpu_baseline`; `--gpus 1` is accepted as the driver passes it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "2",
                        "--prewarm", "0.02", "--cpu-sample-points", str(1 << 18)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "ranks_seen"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["steps"] == 16 and d["value"] > 1000
    assert d["roofline"]["bound"] == "hbm" and 0.05 < d["roofline"]["frac"] < 1.0
    # the step and the transform in the contract object too (round-3 verdict): fractions of the same 8 TB/s
    r = d["roofline"]
    for k in ("step_frac", "fft_frac_algorithmic", "fft_traffic_ratio", "frac_rocprof", "fft_passes", "profile"):
        assert k in r, k
    assert 0.02 < r["step_frac"] < r["frac"] and 0.02 < r["fft_frac_algorithmic"] < 1.0 and r["fft_passes"] == 3
    assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
    # round-4 verdict: clock and power sampled IN the run (or null), never a remembered sentence; the fraction against the
    # achievable 6.29 TB/s; the kernel that dominates the step BY TIME with its own fraction; the first-call cost
    for k in ("sclk_mhz", "socket_power_w", "power_cap_w", "limited_by", "clock_power_samples", "frac_vs_achievable", "dominant_by_time"):
        assert k in r, k
    assert abs(r["frac_vs_achievable"] - r["frac"] * 8000.0 / 6290.0) < 1e-9
    dom = r["dominant_by_time"]
    assert "k_fft_pass" in dom["kernel"] and 0.5 < dom["share_of_step"] < 0.9 and abs(dom["frac"] - r["fft_frac_algorithmic"]) < 1e-12
    assert dom["frac"] < r["frac"]  # the transform, not the block kernel, is what holds the step back
    cps = r["clock_power_samples"]
    if cps["source"] is not None:
        assert cps["prewarm"]["samples"] >= 3 and 300 < r["sclk_mhz"] < 3000 and 50 < r["socket_power_w"] < 2000
        assert r["limited_by"] is None or "sampled in this run" in r["limited_by"]
    else:
        assert r["sclk_mhz"] is None and r["limited_by"] is None
    fc = d["config"]["first_call_ms"]
    assert fc and "error" not in fc, fc
    assert fc["plain_fft_first_ms"] > fc["plain_fft_third_ms"] > 0 and fc["convolve_signal_first_ms"] >= fc["convolve_signal_third_ms"] > 0
    # the two kernels' event-derived durations add up to (almost) the step: an over-subtracted event-pair overhead would
    # inflate both fractions (seen at --steps 200 before the overhead became the smallest of its samples)
    k = d["kernels"]
    assert 0.85 * d["ms_per_step"] < k["conv_ms"] + k["fft_ms"] <= 1.02 * d["ms_per_step"], (k, d["ms_per_step"])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["fair_allcores_Msamples_s"] > 0
    # round 5: the noise floor -- window 0 is the contract's timed region, nine more windows of the same K steps follow
    vw = d["value_windows"]
    assert vw["n"] == 10 and vw["steps_per_window"] == 16 and vw["first_is_value"] is True
    assert len(vw["values"]) == len(vw["conv_ms"]) == len(vw["fft_ms"]) == len(vw["sclk_mhz"]) == len(vw["ms_per_step"]) == 10
    assert vw["values"][0] == d["value"] and vw["min"] <= vw["median"] <= vw["max"] and vw["min"] <= d["value"] <= vw["max"]
    assert abs(vw["conv_ms"][0] - k["conv_ms"]) < 1e-12 and abs(vw["fft_ms"][0] - k["fft_ms"]) < 1e-12
    assert all(0.5 * k["conv_ms"] < c < 2 * k["conv_ms"] for c in vw["conv_ms"]) and vw["spread_pct"] >= 0
    assert "c5_end_to_end" not in d  # (no process group at a plain N = 1: nothing changes)
    # round 6: the line certifies the output of the step it timed -- the last step's convolution result against the f64 direct
    # form on three windows, six bins of its spectrum against the DFT definition, Parseval; and each rank's own rate
    sc = d["self_check"]
    assert sc["ok"] is True and sc["rerun_spectrum_bit_identical_to_timed"] is True, sc
    assert 0 < sc["conv_rel_l2_max"] < 1e-6 and 0 < sc["fft_bin_err_max"] < 2e-6 and sc["parseval_rel"] < 1e-6, sc
    assert len(sc["conv_windows"]) == 3 and sc["conv_windows"][0][0] == 0 and sc["conv_windows"][-1][1] == 1 << 24 and len(sc["fft_bins"]) == 6
    assert sc["seconds"] < 5.0, sc["seconds"]
    assert d["value_by_rank"] == [d["value"]]
    # ... and a wrong output makes the run fail AFTER the line is printed (test hook: one value of the checked y is changed)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--prewarm", "0.02", "--windows", "0",
                        "--no-cpu-baseline", "--no-first-call", "--test-corrupt-self-check"], env=dict(env, BDSP_BENCH_CORRUPT_SELF_CHECK="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 5 and "self-check FAILED" in p.stderr, (p.returncode, p.stderr[-1500:])
    bad = json.loads(p.stdout.strip().splitlines()[-1])["self_check"]
    assert bad["ok"] is False and bad["conv_rel_l2_max"] > 1e-3 and bad["rerun_spectrum_bit_identical_to_timed"] is True
    # (the flag alone, without the environment variable, changes nothing; --no-self-check leaves the key null)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--prewarm", "0.02", "--windows", "0",
                        "--no-cpu-baseline", "--no-first-call", "--no-self-check"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["self_check"] is None
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "c5", "--steps", "4", "--warmup", "1",
                        "--prewarm", "0.02", "--vectors-per-gpu", "16", "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["config"]["vectors_per_gpu"] == 16 and d["c5_end_to_end"]["vectors"] == 16 and d["c5_end_to_end"]["ms"] > 0
    assert d["c5_end_to_end"]["expected_ms"] is None  # (world size 1: nothing travels)
    assert d["self_check"]["ok"] is True and d["self_check"]["vector_checked"] == 15, d["self_check"]


def test_bench_py_two_and_three_ranks_through_its_own_launcher_on_one_gpu():
    """`python bench.py --gpus N` starts N ranks itself (the driver's command shape).  RCCL cannot put two ranks on
    one device, so the test hook BDSP_BENCH_SHARE_GPU=1 keeps every rank on GPU 0 and runs the control collectives over
    gloo: everything else -- the child processes, the rendezvous, the barriers around the timed region, the maximum
    over ranks, `ranks_seen`, the whole-job aggregate -- is the real N-rank path on real hardware.  Also under
    torch.distributed.run, the launcher the driver uses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["BDSP_BENCH_SHARE_GPU"] = "1"

    def free_port():
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return str(sk.getsockname()[1])
    one = None
    for ranks in (1, 2, 3):
        # the environment variable ALONE must change nothing (ranks == 1 runs without the flag: a plain line with a value)
        hook = ["--test-share-gpu"] if ranks > 1 else []
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "30", "--warmup", "3",
                            "--prewarm", "0.05", "--no-cpu-baseline", "--e2e-vectors-per-gpu", "4"] + hook, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (ranks, p.stderr[-2000:])
        lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]  # rank 0 alone prints
        d = json.loads(lines[0])
        assert d["n_gpus"] == ranks and d["ranks_seen"] == ranks and d["steps"] == 30 and d["scaling"] == "weak"
        if ranks == 1:
            one = d["value"]
            assert "test_hook" not in d and "TEST HOOK" not in d["config"]["parallelism"]
        else:
            # a line produced under the hook is marked and carries NO value (all ranks shared one GPU)
            assert d["test_hook"] is True and d["value"] is None
            # the ranks share one GPU here, so the whole-job rate stays near the one-rank rate (it is N x work in ~N x time)
            assert 0.5 * one < d["test_hook_value"] < 1.6 * one, (ranks, one, d["test_hook_value"])
            assert "TEST HOOK" in d["config"]["parallelism"]
            # round 5: in the default mode the ranks of a process group ALSO run the path's one exchange -- rank 0 scatters a
            # C5-shaped batch in chunks, every rank convolves + transforms, the spectra come back -- and rank 0 checks the
            # first and last chunk of every peer bit for bit (here over gloo with host tensors: one GPU)
            e = d["c5_end_to_end"]
            assert e["peers"] == ranks - 1 and e["verified_rows"] == 4 * (ranks - 1) and e["vectors"] == 4 * ranks
            assert e["ms"] > 0 and e["Msamples_s"] > 0 and e["chunk_vectors"] == 2 and "gloo" in e["transport"]
            # round 6: what the wires alone would cost (4 vectors in chunks of 2: 2 + 2 rounds of 16 MiB per link at 153 GB/s)
            assert abs(e["expected_ms"] - 4 * (2 << 23) / 153e9 * 1e3) < 1e-9 and e["link_model"]["bound"] == "xgmi link" and e["link_model"]["rounds"] == 4
            assert d["value_by_rank"] is None and d["self_check"]["ok"] is True
            assert d["value_windows"]["values"] is None and len(d["value_windows"]["ms_per_step"]) == 10
    # ... and a gathered row that differs from rank 0's own computation ends every rank non-zero
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--prewarm", "0.02",
                        "--no-cpu-baseline", "--test-share-gpu", "--e2e-vectors-per-gpu", "4", "--windows", "1"],
                       env=dict(env, BDSP_BENCH_CORRUPT_E2E="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "differ from rank 0's own computation" in p.stderr, (p.returncode, p.stderr[-1500:])
    # ... and a leg that HANGS (a rank never joins it) does not cost the record its measurements: after --e2e-timeout rank 0
    # prints the line -- every other figure was final before the leg started -- with an error in the leg's place
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--prewarm", "0.02",
                        "--no-cpu-baseline", "--test-share-gpu", "--e2e-vectors-per-gpu", "4", "--windows", "1", "--e2e-timeout", "20"],
                       env=dict(env, BDSP_BENCH_HANG_E2E="1"), capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    # (round 6: the ranks leave with exit code 4 AFTER the line, so the launcher -- and the driver -- see the failure)
    assert p.returncode != 0 and "exited with code 4" in p.stderr and len(lines) == 1, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    d = json.loads(lines[0])
    assert "hung" in d["c5_end_to_end"]["error"] and d["ranks_seen"] == 2 and d["test_hook_value"] > 0 and len(d["value_windows"]["ms_per_step"]) == 2
    # the same two ranks under torch.distributed.run, as the driver launches them
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", free_port(), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20",
                        "--warmup", "3", "--prewarm", "0.05", "--no-cpu-baseline", "--test-share-gpu"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2
    # a mismatch between --gpus and the launcher's world size must fail loudly
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", free_port(), os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0



def _bin_by_definition(xc64, k, n, chunk=1 << 22):
    """X[k] = sum_j x[j] exp(-2 pi i (j k mod n) / n) in f64, in chunks (the phase table of 2^27 points would be 2 GiB)."""
    acc = 0.0 + 0.0j
    for a in range(0, n, chunk):
        idx = np.arange(a, min(a + chunk, n), dtype=np.int64)
        acc += np.sum(xc64[a:a + chunk].astype(np.complex128) * np.exp(-2j * np.pi * ((idx * k) % n) / n))
    return acc


def test_above_2_24_points_fft_2_25_whole_spectrum_and_round_trip():
    """2^25 points: beyond 2^24 the f32 inter-pass twiddle argument 2e/n is no longer exact in float and unit_root<float>
    takes its double branch (fft_impl.h); three passes of 512 / 256 / 256-point columns.  Whole spectrum against the
    oracle's f64 transform of the same f32 input (north_star: 1e-6), Parseval, round trip.  The reference accepts any
    length (vector/src/vector_types/time_freq/mod.rs:32-63)."""
    n = 1 << 25
    x = orc.fill_uniform(2 * n, SEED_C3_X + 25, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    X = v.data()
    ref = orc.fft_pow2_mt(x.astype(np.float64), False, max(1, len(os.sched_getaffinity(0))))
    assert rel_l2(X, ref) < 1e-6
    del ref
    e_t = float(np.sum(x.astype(np.float64) ** 2))
    e_f = float(np.sum(X.astype(np.float64) ** 2)) / n
    assert abs(e_f - e_t) / e_t < 1e-6
    del X
    assert v.plain_ifft() == 0 and v.scale(1.0 / n) == 0
    assert rel_l2(v.data(), x) < 2e-6


def test_above_2_24_points_fft_2_27_bins_and_parseval():
    """2^27 points (1 GiB per buffer): two bins against the DFT definition in f64, Parseval, and the round trip."""
    n = 1 << 27
    x = orc.fill_uniform(2 * n, SEED_C3_X + 27, -10, 10, np.float32)
    v = DspVec(x, is_complex=True)
    assert v.plain_fft() == 0
    X = v.datac()
    xc = x.view(np.complex64)
    scale = np.sqrt(n) * 10.0
    for k in (1, n // 3):
        assert abs(X[k] - _bin_by_definition(xc, k, n)) / scale < 3e-6, k
    e_t = float(np.sum(x.astype(np.float64) ** 2))
    e_f = 0.0
    for a in range(0, n, 1 << 24):  # (in pieces: the f64 copy of the whole spectrum would be 2 GiB)
        e_f += float(np.sum(np.abs(X[a:a + (1 << 24)].astype(np.complex128)) ** 2))
    assert abs(e_f / n - e_t) / e_t < 1e-6
    del X
    assert v.plain_ifft() == 0 and v.scale(1.0 / n) == 0
    assert rel_l2(v.data(), x) < 3e-6


def test_above_2_24_points_convolution_2_25_with_257_taps():
    """convolve_signal on 2^25 points with 257 taps (R0 = 1: 3840 outputs per block, 8739 blocks; the block kernel's
    32-bit element indices are good to 2^31): three windows -- both wrap-around ends and an interior seam -- against the
    f64 direct form (convolution.rs:304-461)."""
    n, m = 1 << 25, 257
    x = orc.fill_uniform(2 * n, SEED_C3_X + 26, -10, 10, np.float32)
    h = (orc.fill_uniform(2 * m, SEED_C3_H + 26, -1, 1, np.float32) / np.float32(m)).astype(np.float32)
    c = DspVec(x, is_complex=True)
    assert c.convolve_signal(DspVec(h, is_complex=True)) == 0
    y = c.data()
    x64, h64 = x.astype(np.float64), h.astype(np.float64)
    for first in (0, n - 2000, (n // 2 // 3840) * 3840 - 1000):
        ref = orc.convolve_direct(x64, h64, True, first, 2000)
        assert rel_l2(y[2 * first:2 * (first + 2000)], ref) < 1e-6, first



def test_block_kernel_dispatch_order_assumptions_still_hold():
    """The fused block kernel gives the workgroups dispatched FIRST a larger share of the blocks (43 / 37 / 20 %) and
    maps workgroup b to XCD b mod 8 -- properties OBSERVED on this firmware (the oldest wave is issued first), not
    architectural ones.  If a driver or firmware flips the dispatch order the skew silently costs
    ~10 us per launch: time the default against equal shares (given per call through bdsp_hip_dev_convolve_ex) and fail
    if the skew has become more than 5 % SLOWER (median of five interleaved pairs); the CU count must stay a multiple of the 8 XCDs the block-to-XCD map assumes.
    The shares only move blocks between workgroups: the output is bit-identical."""
    import ctypes as C
    import torch
    import basic_dsp_amd as bd
    lib = bd.lib
    cus = lib.bdsp_hip_compute_units()
    assert cus > 0 and cus % 8 == 0, cus
    n, m = 1 << 24, 1024
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    xs = [torch.rand(2 * n, generator=g, device=dev, dtype=torch.float32) * 20 - 10 for _ in range(3)]
    taps = (torch.rand(2 * m, generator=g, device=dev, dtype=torch.float32) * 2 - 1) / m
    y = torch.empty(2 * n, device=dev, dtype=torch.float32)
    sp = bd._lib.torch_stream_arg()

    def run(i, shares=(-1, -1)):
        bd._lib.check(lib.bdsp_hip_dev_convolve_ex(0, xs[i % 3].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m,
                                                   shares[0], shares[1], sp))

    def timed(reps, shares):
        e0, e1 = lib.bdsp_hip_event_create(), lib.bdsp_hip_event_create()
        lib.bdsp_hip_event_record(e0, sp)
        for i in range(reps):
            run(i, shares)
        lib.bdsp_hip_event_record(e1, sp)
        torch.cuda.synchronize()
        ms = C.c_float(0)
        lib.bdsp_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        lib.bdsp_hip_event_destroy(e0)
        lib.bdsp_hip_event_destroy(e1)
        return ms.value / reps * 1e3
    assert lib.bdsp_hip_dev_convolve_ex(0, xs[0].data_ptr(), y.data_ptr(), n, 1, taps.data_ptr(), m, 60, 45, sp) == 7  # refused
    for i in range(2500):  # clock ramp
        run(i)
    torch.cuda.synchronize()
    t_skew, t_equal = [], []
    for rep in range(5):  # interleaved, so a drifting clock hits both alike
        t_skew.append(timed(300, (-1, -1)))
        t_equal.append(timed(300, (33, 33)))
    run(0, (33, 33))
    torch.cuda.synchronize()
    y_equal = y.clone()
    run(0)  # the override did not outlive its call
    torch.cuda.synchronize()
    assert torch.equal(y, y_equal)
    skew, equal = sorted(t_skew)[2], sorted(t_equal)[2]
    print("block kernel: default shares %.1f us, equal shares %.1f us" % (skew, equal))
    assert skew < 1.05 * equal, (t_skew, t_equal)


RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import datetime
import torch
import torch.distributed as dist
from basic_dsp_amd.batch import process_shard_gpu, scatter_process_gather, scatter_process_gather_chunked

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
# control collectives of bench.py, through librccl
one = torch.ones(1, device=dev, dtype=torch.int64)
dist.all_reduce(one)
assert int(one.item()) == 1
t = torch.tensor([3.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
dist.barrier()
# a point-to-point pair to itself, the primitive the scatter / gather rounds are made of
a = torch.arange(1024, device=dev, dtype=torch.float32)
b = torch.zeros_like(a)
for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
    w.wait()
torch.cuda.synchronize()
assert torch.equal(a, b)
# the C5 drivers with group = WORLD: broadcast_object_list + broadcast run through RCCL, the pipeline on the side stream
n, m, nvec = 1 << 16, 257, 12
g = torch.Generator(device=dev); g.manual_seed(11)
batch = torch.rand((nvec, 2 * n), generator=g, device=dev, dtype=torch.float32) * 20 - 10
taps = (torch.rand(2 * m, generator=g, device=dev, dtype=torch.float32) * 2 - 1) / m
want = process_shard_gpu(batch.clone(), taps, n).clone()
for rep in range(3):   # repeated: a recycled side-stream block would show up as a mismatch
    got = scatter_process_gather_chunked(batch.clone(), taps, n, process_shard_gpu, chunk_vectors=5, group=dist.group.WORLD, device=dev)
    torch.cuda.synchronize()
    assert torch.equal(got, want), rep
got = scatter_process_gather(batch.clone(), taps, n, process_shard_gpu, group=dist.group.WORLD, device=dev)
torch.cuda.synchronize()
assert torch.equal(got, want)
maps = open("/proc/self/maps").read()
assert "librccl" in maps, "RCCL was never loaded"
dist.barrier()
dist.destroy_process_group()
print("RCCL_WS1_OK")
"""


@pytest.mark.gpu
def test_rccl_runs_on_the_one_gpu_at_world_size_one(tmp_path):
    """The first RCCL call this code makes must not happen on the 8-GPU box: ONE fresh child process joins an "nccl"
    process group of world size 1 on GPU 0 and runs, through librccl, the control collectives of bench.py (all_reduce
    SUM / MAX, barrier), a batched isend/irecv pair, and both C5 drivers of basic_dsp_amd.batch with group = WORLD
    (broadcast_object_list, broadcast; the chunked pipeline on its side stream, three times, bit-identical to the
    plain shard).  Then bench.py itself under --init-dist, headline and --mode c5: `ranks_seen` comes out of an RCCL
    all-reduce.  (A process that has touched the GPU is never re-executed: fresh children only.)"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def env_for_child():
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BDSP_BENCH_SHARE_GPU")}
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        return env
    script = tmp_path / "rccl_child.py"
    script.write_text(RCCL_CHILD)
    p = subprocess.run([sys.executable, str(script), root], env=env_for_child(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WS1_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    for extra in ([], ["--mode", "c5", "--vectors-per-gpu", "16"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--init-dist", "--steps", "8", "--warmup", "2",
                            "--prewarm", "0.02", "--no-cpu-baseline"] + extra, env=env_for_child(), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (extra, p.stderr[-3000:])
        d = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["value"] > 1000
        if extra:
            assert d["c5_end_to_end"]["vectors"] == 16 and d["c5_end_to_end"]["ms"] > 0 and "C5" in d["metric"]
        else:
            # round 5: the verified scatter / compute / gather leg of the DEFAULT mode, here through librccl at world size 1
            # (8 vectors of 2^20 points, chunks of 2, rank 0's own first and last chunk recomputed and compared bit for bit)
            e = d["c5_end_to_end"]
            assert e["peers"] == 0 and e["verified_rows"] == 4 and e["vectors"] == 8 and e["ms"] > 0 and "RCCL" in e["transport"]
