"""Pins the CPU oracle against the reference's own known-answer vectors.

Every expected array comes from tests/golden/reference_kats.json (transcribed from the reference's
tests by tools/extract_golden.py; `source` holds file:line).  The scenario around each vector is
re-stated here from the cited reference test.  Tolerances are the reference's own.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kats.json")) as f:
    KATS = json.load(f)


def kat(name, idx=-1):
    return np.array(KATS[name]["arrays"][idx])


def to_complex(real):
    """RealToComplex::to_complex: interleave with zero imaginary parts."""
    out = np.zeros(2 * len(real), dtype=np.asarray(real).dtype)
    out[0::2] = real
    return out


def sinusoid64():
    # tests/time_freq_test.rs:221-231 new_sinusoid_vector: cos(2*pi*0.1*n + 0.25), n = 0..63
    v = orc.real_scale(np.arange(64, dtype=np.float64) * 0.1, 2.0 * np.pi)
    v = orc.real_offset(v, 0.25)
    return np.cos(v)


def ref_fft(x):
    """TimeToFrequencyDomainOperations::fft = plain_fft + fft_shift (time_to_freq.rs:158-165)"""
    return orc.swap_halves(orc.fft(x), True, True)


def ref_ifft(x):
    """ifft = scale(1/points) -> ifft_shift -> plain_ifft (freq_to_time.rs:160-168)"""
    points = x.size // 2
    y = orc.real_scale(x, 1.0 / points)
    y = orc.swap_halves(y, True, False)
    return orc.fft(y, inverse=True)


# ---------------------------------------------------------------- FFT golden vectors (Octave)
def test_fft_vector64():
    got = orc.magnitude(ref_fft(to_complex(sinusoid64())))
    np.testing.assert_allclose(got, kat("fft_vector64"), rtol=0, atol=1e-6)


def test_windowed_fft_vector64():
    x = orc.apply_window(to_complex(sinusoid64()), True, 1, 0.54)
    got = orc.magnitude(ref_fft(x))
    np.testing.assert_allclose(got, kat("windowed_fft_vector64"), rtol=0, atol=1e-6)


def test_fft_ifft_roundtrip64():
    # tests/time_freq_test.rs:199-207 fft_ifft_vector64, tol 1e-6
    x = to_complex(sinusoid64())
    np.testing.assert_allclose(ref_ifft(ref_fft(x)), x, rtol=0, atol=1e-6)


def test_windowed_roundtrip64():
    # tests/time_freq_test.rs:209-219: windowed_fft then windowed_ifft restores the signal
    x = to_complex(sinusoid64())
    f = ref_fft(orc.apply_window(x, True, 1, 0.54))
    t = orc.apply_window(ref_ifft(f), True, 1, 0.54, unapply=True)
    np.testing.assert_allclose(t, x, rtol=0, atol=1e-6)


def test_window_real_vs_complex():
    # tests/time_freq_test.rs:35-44: windowing a complex copy then to_real == windowing the real
    x = sinusoid64()
    c = orc.apply_window(to_complex(x), True, 1, 0.54)
    assert np.array_equal(orc.complex_to_real(c, 2), orc.apply_window(x, False, 1, 0.54))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 12, 17, 64, 100, 243, 1000, 1024])
def test_fft_any_length_matches_naive_dft(n, dtype):
    # a3: unnormalised DFT for ANY N (rustfft semantics); radix-2 and Bluestein vs O(N^2) sum
    x = orc.fill_uniform(2 * n, 42 + n, -10, 10, dtype)
    tol = 2e-5 if dtype == np.float32 else 1e-12
    for inv in (False, True):
        ref = orc.dft_naive(x.astype(np.float64), inv)
        got = orc.fft(x, inv).astype(np.float64)
        assert np.linalg.norm(got - ref) <= tol * max(np.linalg.norm(ref), 1e-30)


def test_doc_3point_dft():
    # time_to_freq.rs:30-38 doc test: plain_fft of [1,0,-0.5,0.8660254,-0.5,-0.8660254] = [0,0,3,0,0,0]
    x = np.array([1.0, 0.0, -0.5, 0.8660254, -0.5, -0.8660254])
    np.testing.assert_allclose(orc.fft(x), [0, 0, 3, 0, 0, 0], atol=1e-4)
    # freq_to_time.rs:32-40: plain_ifft of [0,0,1,0,0,0] = [1,0,-0.5,0.8660254,-0.5,-0.8660254]
    np.testing.assert_allclose(orc.fft(np.array([0.0, 0, 1, 0, 0, 0]), True),
                               [1, 0, -0.5, 0.8660254, -0.5, -0.8660254], atol=1e-4)


# ---------------------------------------------------------------- windows / conv functions
@pytest.mark.parametrize("name,wid", [("triangular_window32_test", 0), ("hamming_window32_test", 1),
                                      ("blackmanharris_window32_test", 2),
                                      ("rectangular_window32_test", 3)])
def test_window_kats(name, wid):
    exp = kat(name)
    got = [orc.window_value(wid, 0.54, i, len(exp), np.float32) for i in range(len(exp))]
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-4)


def _conv_scan(fn, fid, rolloff, n, step, dtype):
    j0 = -(n // 2)
    return [fn(fid, rolloff, dtype(j0 + i) * dtype(step), dtype) for i in range(n)]


def test_conv_function_kats():
    exp = kat("raised_cosine_test")
    np.testing.assert_allclose(_conv_scan(orc.conv_time, 1, 0.35, len(exp), 0.2, np.float64), exp,
                               atol=1e-4)
    exp = kat("sinc_test")
    np.testing.assert_allclose(_conv_scan(orc.conv_time, 0, 0.0, len(exp), 0.5, np.float32), exp,
                               atol=1e-4)
    exp = kat("sinc_freq_test")
    np.testing.assert_allclose(_conv_scan(orc.conv_freq, 0, 0.0, len(exp), 0.5, np.float32), exp,
                               atol=1e-4)
    exp = kat("freq_test")
    np.testing.assert_allclose(_conv_scan(orc.conv_freq, 1, 0.5, len(exp), 0.4, np.float64), exp,
                               atol=0.1)


# ---------------------------------------------------------------- bit-exact index ops
def test_swap_halves_kats():
    for name, cplx, fwd in [("swap_halves_even_test", False, False),
                            ("swap_halves_odd_foward_test", False, True),
                            ("swap_halves_odd_inverse_test", False, False),
                            ("swap_halves_real_even_test", False, True),
                            ("swap_halves_real_odd_test", False, True),
                            ("swap_halves_complex_even_test", True, True),
                            ("swap_halves_complex_odd_test", True, True)]:
        inp, exp = kat(name, 0), kat(name, 1)
        assert np.array_equal(orc.swap_halves(inp, cplx, fwd), exp), name


def test_zero_pad_kats():
    cases = [("zero_pad_end_test", True, 9, 0, False), ("zero_pad_surround_test", True, 10, 1, False),
             ("zero_pad_center_test", True, 10, 2, False),
             ("zero_pad_b_center_test", True, 10, 2, True),
             ("zero_pad_surround_odd_signal_test", False, 20, 1, False),
             ("zero_pad_b_end_test", True, 9, 0, True),
             ("zero_pad_b_surround_test", True, 10, 1, True),
             ("zero_pad_b_surround_odd_signal_test", True, 10, 1, True),
             ("zero_pad_surround_overlap_test", True, 8, 1, False),
             ("zero_pad_center_overlap_test", True, 8, 2, False)]
    for name, cplx, points, opt, buffered in cases:
        code, got = orc.zero_pad(kat(name, 0), cplx, points, opt, buffered)
        assert code == 0 and np.array_equal(got, kat(name, 1)), name
    # data_reorganization.rs:315-317: len <= len_before -> InvalidArgumentLength (code 7)
    assert orc.zero_pad(np.arange(10.0), True, 5, 0)[0] == 7


def test_zero_interleave_kats():
    for name, cplx in [("zero_interleave_test", False), ("zero_interleave_even_test", False),
                       ("zero_interleave_b_test", False), ("zero_interleave_complex_test", True),
                       ("zero_interleave_b_complex_test", True)]:
        assert np.array_equal(orc.zero_interleave(kat(name, 0), cplx, 2), kat(name, 1)), name


def test_mirror_doc():
    # freq.rs:27-30 doc test
    got = orc.mirror(np.array([1.0, 2, 3, 4, 5, 6]))
    assert np.array_equal(got, [1, 2, 3, 4, 5, 6, 5, -6, 3, -4])


# ---------------------------------------------------------------- elementwise doc tests
def test_elementwise_doc_tests():
    # elementary.rs:31-33 offset doc: [1,2] + 2 = [3,4]; :58-60 scale doc: [1,2]*2 = [2,4]
    assert np.array_equal(orc.real_offset(np.array([1.0, 2.0], np.float32), 2.0), [3, 4])
    assert np.array_equal(orc.real_scale(np.array([1.0, 2.0], np.float32), 2.0), [2, 4])
    # complex_ops.rs:36-39: multiply_complex_exponential(2,3) on [1,2,3,4] (delta 1)
    got = orc.multiply_complex_exponential(np.array([1.0, 2, 3, 4]), 2.0, 3.0)
    exp = np.array([1 + 2j, 3 + 4j]) * np.exp(1j * (2.0 * np.arange(2) + 3.0))
    np.testing.assert_allclose(got, exp.view(np.float64), atol=1e-12)
    # complex_ops.rs:57-59 conj doc
    assert np.array_equal(orc.conj(np.array([1.0, 2, 3, 4])), [1, -2, 3, -4])


def test_magnitude_formulas():
    # tests/complex_test.rs:85-143: magnitude = sqrt(re^2+im^2); magnitude_squared; phase = atan2
    x = orc.fill_uniform(2000, 7, -10, 10, np.float32)
    z = x.astype(np.float64).view(np.complex128)
    np.testing.assert_allclose(orc.complex_to_real(x, 0), np.abs(z), rtol=1e-6)
    np.testing.assert_allclose(orc.complex_to_real(x, 1), np.abs(z) ** 2, rtol=1e-6)
    np.testing.assert_allclose(orc.complex_to_real(x, 4), np.angle(z), rtol=1e-6, atol=1e-6)
    assert np.array_equal(orc.complex_to_real(x, 2), x[0::2])
    assert np.array_equal(orc.complex_to_real(x, 3), x[1::2])


# ---------------------------------------------------------------- convolution KATs
def test_convolve_complex_vectors32():
    # convolution.rs:738-775: 11 complex zeros with scalar index 11 := 1.0 (imag of point 5),
    # taps = sinc(v*0.5), v = -5..5 (real parts), convolve_signal -> magnitude
    n = 11
    time = np.zeros(2 * n, np.float32)
    time[n] = 1.0
    real = np.array([orc.conv_time(0, 0.0, np.float32(v) * np.float32(0.5), np.float32)
                     for v in range(-5, 6)], np.float32)
    code, out, path = orc.convolve_signal(time, to_complex(real), True)
    assert code == 0
    np.testing.assert_allclose(orc.magnitude(out), kat("convolve_complex_vectors32"), atol=1e-4)


def test_shift_kats():
    # convolution.rs:819-842: a delta in tap position 4 of 10 leaves the signal in place;
    # taps [0,0,1] rotate it right by one with wrap-around ([9,0,1,...,8])
    for name in ("shift_left_by_1_as_conv", "shift_left_by_1_as_conv_shorter"):
        a = to_complex(kat(name, 0).astype(np.float32))
        b = to_complex(kat(name, 1).astype(np.float32))
        code, out, _ = orc.convolve_signal(a, b, True)
        assert code == 0
        np.testing.assert_allclose(orc.magnitude(out), kat(name, 2), atol=1e-4)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_conv_vs_freq_multiplication(dtype):
    # convolution.rs:803-816: conv(a,b) == swap_halves(reverse(ifft(fft(a)*fft(b))))
    a = kat("vector_conv_vs_freq_multiplication", 0).astype(dtype)
    b = kat("vector_conv_vs_freq_multiplication", 1).astype(dtype)
    code, conv, _ = orc.convolve_signal(a, b, True)
    assert code == 0
    code, prod = orc.binary(ref_fft(a), ref_fft(b), True, 2)
    mul = orc.swap_halves(orc.reverse(ref_ifft(prod), True), True, True)
    np.testing.assert_allclose(mul, conv, atol=1e-4 * 100)  # values are O(100); ref tol 1e-4 abs
    # pure real data, even and odd lengths (convolution.rs:845-882)
    for n in (10, 9):
        ar = np.arange(n, dtype=dtype)
        br = (15.0 - np.arange(n)).astype(dtype)
        a, b = to_complex(ar), to_complex(br)
        _, conv, _ = orc.convolve_signal(a, b, True)
        _, prod = orc.binary(ref_fft(a), ref_fft(b), True, 2)
        mul = orc.swap_halves(orc.reverse(orc.magnitude(ref_ifft(prod)), False), False, True)
        np.testing.assert_allclose(mul, orc.magnitude(conv), rtol=1e-4)


def test_overlap_discard_kat():
    # convolution.rs:885-898: 100 points x 6 taps, overlap_discard(fft_len arg 0) == convolve_signal
    a = to_complex(np.arange(100, dtype=np.float32))
    b = to_complex(np.array(kat("overlap_discard_test", 0), np.float32))
    code, conv, _ = orc.convolve_signal(a, b, True)
    assert code == 0
    code, od = orc.overlap_discard(a, b, 0)
    assert code == 0
    np.testing.assert_allclose(od, conv, atol=1e-4 * 50)
    code, fair = orc.overlap_discard(a, b, 0, fair=True)
    np.testing.assert_allclose(fair, conv, atol=1e-4 * 50)


@pytest.mark.parametrize("n,m", [(100, 6), (1000, 17), (5000, 64), (12288, 33), (4096, 1), (777, 2)])
def test_overlap_discard_matches_direct_f64(n, m):
    # restatement of the overlap_discard schedule vs the direct a9 sum
    x = orc.fill_uniform(2 * n, 1000 + n, -10, 10, np.float64)
    h = orc.fill_uniform(2 * m, 2000 + m, -1, 1, np.float64)
    ref = orc.convolve_direct(x, h, True)
    for fair in (False, True):
        code, got = orc.overlap_discard(x, h, 0, fair)
        assert code == 0
        assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)


def test_convolve_signal_dispatch_paths():
    # convolution.rs:499,530-534 thresholds: simd for 12..202-scalar taps, overlap_discard for
    # long complex vectors, scalar otherwise; all paths agree with the direct sum
    for n, m, cplx, want in [(600, 20, True, 1), (4000, 110, True, 4), (6000, 120, True, 3),
                             (12000, 300, False, 4), (40, 5, True, 4)]:
        e = 2 if cplx else 1
        x = orc.fill_uniform(e * n, n, -10, 10, np.float64)
        h = orc.fill_uniform(e * m, m, -1, 1, np.float64)
        code, out, path = orc.convolve_signal(x, h, cplx)
        assert code == 0 and path == want, (n, m, path)
        ref = orc.convolve_direct(x, h, cplx)
        assert np.linalg.norm(out - ref) <= 1e-12 * np.linalg.norm(ref)
    # points < impulse response points -> InvalidArgumentLength (convolution.rs:490-492)
    assert orc.convolve_signal(np.zeros(8), np.zeros(10), True)[0] == 7


def test_convolve_longer_taps_than_signal_uses_centre():
    # time_freq/mod.rs:284-288 (via convolve_vector_range/scalar): M > N keeps the centre taps
    x = orc.fill_uniform(2 * 8, 5, -1, 1, np.float64)
    h = orc.fill_uniform(2 * 20, 6, -1, 1, np.float64)
    got = orc.convolve_direct(x, h, True)
    hc = h[2 * (10 - 4):2 * (10 + 4)].view(np.complex128)
    xc = x.view(np.complex128)
    ref = np.array([sum(xc[(i + 4 - 1 - k) % 8] * hc[k] for k in range(8)) for i in range(8)])
    np.testing.assert_allclose(got.view(np.complex128), ref, atol=1e-12)


# ---------------------------------------------------------------- interpolatef KATs
def _impulse(n, dtype=np.float32):
    t = np.zeros(n, dtype)
    t[n // 2] = 1.0
    return to_complex(t)


def test_interpolatef_integer_even_odd():
    # interpolation.rs:753-796 (expected from Octave interpft, tolerance 0.1)
    for name, n in [("interpolatef_by_integer_sinc_even_test", 6),
                    ("interpolatef_by_integer_sinc_odd_test", 7)]:
        out, path = orc.interpolatef(_impulse(n), True, 0, 0.0, 2.0, 0.0, n)
        assert path == 0
        np.testing.assert_allclose(orc.complex_to_real(out, 2), kat(name), atol=0.1)


def test_interpolatef_fractional():
    # interpolation.rs:799-831: factor 13/6 on 6 points -> 13 points
    out, path = orc.interpolatef(_impulse(6), True, 0, 0.0, np.float32(13.0 / 6.0), 0.0, 6)
    assert out.size == 26 and path == 0
    np.testing.assert_allclose(orc.complex_to_real(out, 2),
                               kat("interpolatef_by_fractional_sinc_test"), atol=0.1)


def test_interpolatef_delayed():
    # interpolation.rs:900-919: 6 complex points, scalar index 6 := 1 (re of point 3), delay 1.0
    t = np.zeros(12, np.float32)
    t[6] = 1.0
    out, _ = orc.interpolatef(t, True, 0, 0.0, 2.0, 1.0, 6)
    np.testing.assert_allclose(orc.magnitude(out), kat("interpolatef_delayed_sinc_test"), atol=0.1)


@pytest.mark.parametrize("cplx", [True, False])
def test_interpolatef_simd_path_equals_scalar_path_without_delay(cplx):
    # with delay = 0 and a symmetric function both tap windows agree (to rounding)
    e = 2 if cplx else 1
    x = orc.fill_uniform(e * 600, 99, -10, 10, np.float64)
    simd, p1 = orc.interpolatef(x, cplx, 1, 0.35, 4.0, 0.0, 12)
    assert p1 == 1 and simd.size == e * 2400
    # Inner region (outputs (2L+1)*f .. new_points-(2L+1)*f): same window as the scalar path
    # y[i] = sum_{n=r-L}^{r+L} x[n mod N] h(n - i/f) (interpolation.rs:92-131) up to taps that sit
    # on zeros of the function.  Edges use interpolate_priv_simd_step's window r-L+1 .. r+L+1
    # (interpolation.rs:293-315), which differs by one end tap: a reference quirk we keep.
    pts, L, f = 600, 12, 4
    edge = (2 * L + 1) * f
    xs = x.view(np.complex128) if cplx else x
    got = simd.view(np.complex128) if cplx else simd
    for i in list(range(0, 2400, 7)):
        t = i / 4.0
        r = int(np.floor(t))
        lo = r - L if edge <= i < 2400 - edge else r - L + 1
        ref = sum(xs[n % pts] * orc.conv_time(1, 0.35, n - t, np.float64)
                  for n in range(lo, lo + 2 * L + 1))
        assert abs(got[i] - ref) <= 1e-10, (i, got[i], ref)


def test_interpolatef_new_len_is_even():
    # interpolation.rs:406-410
    assert orc.interpolatef_new_len(12, 13.0 / 6.0) == 26
    assert orc.interpolatef_new_len(10, 1.5) == 16  # round(15) = 15 -> made even


# ---------------------------------------------------------------- FFT-domain interpolation family (a14)
def test_multiply_frequency_response_kats():
    # convolution.rs:633-648: ones * raised-cosine(1.0) frequency response, ratio 2
    for name, n in (("convolve_complex_freq_and_freq32", 10), ("convolve_complex_freq_and_freq_even32", 12)):
        got = orc.multiply_frequency_response(np.ones(n, np.float32), True, 1, 1.0, 2.0, False)
        np.testing.assert_allclose(got, kat(name, -1), atol=1e-4)


def test_interpolatei_kats():
    # interpolation.rs:654-678 (sinc) and :726-750 (raised cosine 0.4): 6 complex points, impulse in re of point 3
    t = np.zeros(12, np.float32)
    t[6] = 1.0
    for name, fid, ro in (("interpolatei_sinc_test", 0, 0.0), ("interpolatei_rc_test", 1, 0.4)):
        code, out = orc.interpolatei(t, True, fid, ro, 2)
        assert code == 0
        np.testing.assert_allclose(orc.magnitude(out), kat(name), atol=1e-4)


def test_interpolate_kats():
    # interpolation.rs:681-723: interpolate(Some(sinc), 2*len, 0) == Octave interpft
    t = np.zeros(12, np.float32)
    t[6] = 1.0
    code, out, nd = orc.interpolate(t, True, 0, 0.0, 12)
    assert code == 0 and nd == pytest.approx(0.5)
    np.testing.assert_allclose(orc.complex_to_real(out, 2), kat("interpolate_sinc_even_test"), atol=1e-4)
    code, out, _ = orc.interpolate(_impulse(7), True, 0, 0.0, 14)
    np.testing.assert_allclose(orc.complex_to_real(out, 2), kat("interpolate_sinc_odd_test"), atol=1e-4)
    # :834-865 fractional 6 -> 13 points (tol 0.1)
    code, out, _ = orc.interpolate(_impulse(6), True, 0, 0.0, 13)
    np.testing.assert_allclose(orc.complex_to_real(out, 2), kat("interpolate_by_fractional_sinc_test"), atol=0.1)
    # :922-948 delayed by one sample (tol 0.1)
    fir = kat("interpolate_delayed_sinc_test", 0).astype(np.float32)
    code, out, _ = orc.interpolate(to_complex(fir), True, 0, 0.0, 12, delay=1.0)
    np.testing.assert_allclose(orc.magnitude(out), kat("interpolate_delayed_sinc_test", 1), atol=0.1)
    # :972-1007 downsampling 13 -> 6 points == Octave interpft(time, 6), tol 1e-4
    fir = kat("decimate_with_interpolate_test", 0).astype(np.float32)
    code, out, _ = orc.interpolate(to_complex(fir), True, 0, 0.0, 6)
    np.testing.assert_allclose(orc.magnitude(out), kat("decimate_with_interpolate_test", 1), atol=1e-4)


def test_decimatei_kat():
    # interpolation.rs:963-969
    got = orc.decimatei(kat("decimatei_test", 0), True, 2, 1)
    assert np.array_equal(got, kat("decimatei_test", 1))


# ---------------------------------------------------------------- convolve(function), correlate, real interpolation
def test_convolve_function_kats():
    # convolution.rs:651-669: real impulse, raised cosine 0.35, ratio 0.2, len 5
    x = np.zeros(10, np.float32)
    x[5] = 1.0
    got = orc.convolve_function(x, False, 1, 0.35, 0.2, 5)
    np.testing.assert_allclose(got, kat("convolve_real_time_and_time32"), atol=1e-4)
    # :672-702: complex impulse, sinc, ratio 0.5, len 11/2 -> magnitude
    t = np.zeros(22, np.float32)
    t[11] = 1.0  # data_mut(len): scalar index 11 = imaginary part of point 5
    got = orc.convolve_function(t, True, 0, 0.0, 0.5, 5)
    np.testing.assert_allclose(orc.magnitude(got), kat("convolve_complex_time_and_time32"), atol=1e-4)


def test_correlate_kats():
    # correlation.rs:171-215: prepare_argument_padded + correlate, tolerance 0.1
    for name in ("time_correlation_test", "time_correlation_test2"):
        a = kat(name, 0).astype(np.float32)
        b = kat(name, 1).astype(np.float32)
        code, arg = orc.prepare_argument(b, True)
        assert code == 0 and arg.size == 2 * (b.size - 1)
        code, res = orc.correlate(a, arg)
        assert code == 0
        np.testing.assert_allclose(res, kat(name, 2), atol=0.1)
    # the doc example (correlation.rs:52-62)
    a = np.array([1, 1, 2, 2, 3, 3], np.float32)
    b = np.array([3, 3, 2, 2, 1, 1], np.float32)
    _, arg = orc.prepare_argument(b, True)
    _, res = orc.correlate(a, arg)
    np.testing.assert_allclose(res, [2, 0, 8, 0, 20, 0, 24, 0, 18, 0], atol=1e-4)
    # an argument that is not longer than the vector makes zero_pad_b fail (code 7)
    _, arg = orc.prepare_argument(b, False)
    code, _ = orc.correlate(np.ones(8, np.float32), arg)
    assert code == 7


def test_real_interpolation_kats():
    # real_interpolation.rs:198-237
    x = kat("hermit_spline_test", 0).astype(np.float32)
    got = orc.interpolate_hermite(x, 4.0)
    exp = kat("hermit_spline_test", 1)
    assert got.size == exp.size
    np.testing.assert_allclose(got[4:-4], exp[4:-4], atol=6e-2)
    x = kat("hermit_spline_test_linear_increment", 0).astype(np.float32)
    np.testing.assert_allclose(orc.interpolate_hermite(x, 3.0), kat("hermit_spline_test_linear_increment", 1),
                               atol=5e-3)
    x = kat("linear_test", 0).astype(np.float32)
    np.testing.assert_allclose(orc.interpolate_lin(x, 4.0), kat("linear_test", 1), atol=0.1)


def test_statistics_sums_dot_products_kats():
    # statistics.rs:44-65 (doc example), :84-91 (split), :113-128 (sum, sum_sq), dot_products.rs:338-388
    z = np.array([1, 2, 3, 4, 5, 6], np.float32)
    s = orc.complex_statistics(z)
    assert s["sum"] == 9 + 12j and s["count"] == 3 and s["average"] == 3 + 4j
    assert abs(s["rms"] - (3.4027193 + 4.3102784j)) < 1e-4
    assert (s["min"], s["min_index"], s["max"], s["max_index"]) == (1 + 2j, 0, 5 + 6j, 2)
    assert orc.complex_statistics(z, 0, 2)["sum"] == 6 + 8j and orc.complex_statistics(z, 1, 2)["sum"] == 3 + 4j
    assert orc.vec_sum(z, True) == 9 + 12j
    assert orc.vec_sum(z.astype(np.float64), True, squared=True) == -21 + 88j
    assert orc.dot(np.array([1, 2, 3], np.float32), np.array([1, 2, 3], np.float32), False) == 14.0
    assert orc.dot(np.array([1, 0, 3, 0], np.float32), np.array([1, 0, 3, 0], np.float32), True) == 10 + 0j
    r = orc.real_statistics(np.array([3, -1, 4, -1, 5], np.float32))
    assert (r["sum"], r["count"], r["min"], r["min_index"], r["max"], r["max_index"]) == (10, 5, -1, 1, 5, 4)
    assert r["average"] == 2 and abs(r["rms"] - np.sqrt(52 / 5)) < 1e-6


def test_math_family_diff_sum_wrap_split_merge_kats():
    """Doc-test values of trigonometry_and_powers.rs:14-189, real_ops.rs:21-65, diff_sum.rs:18-53,
    data_reorganization.rs:185-212, plus identities that pin the complex (num-complex) formulas."""
    f = np.float32
    pi = np.float32(np.pi)
    assert list(orc.math(np.array([pi / 2, -pi / 2], f), False, "sin")) == [1.0, -1.0]
    assert list(orc.math(np.array([2 * pi, pi], f), False, "cos")) == [1.0, -1.0]
    assert list(orc.math(np.array([1, 4, 9, 16, 25], f), False, "sqrt")) == [1, 2, 3, 4, 5]
    assert np.isnan(orc.math(np.array([-1], f), False, "sqrt")[0])
    assert list(orc.math(np.array([1, 2, 3, 4, 5], f), False, "square")) == [1, 4, 9, 16, 25]
    np.testing.assert_allclose(orc.math(np.array([1, 8, 27], f), False, "powf", 1 / f(3)), [1, 2, 3], rtol=1e-6)
    assert list(orc.math(np.array([1, 2, 3], f), False, "powf", 3.0)) == [1, 8, 27]
    e = np.array([2.718281828459045, 7.389056, 20.085537])
    for name in ("ln", ):
        np.testing.assert_allclose(orc.math(e, False, name), [1, 2, 3], atol=1e-4)
    np.testing.assert_allclose(orc.math(np.array([1.0, 2, 3]), False, "exp"), e, atol=1e-4)
    np.testing.assert_allclose(orc.math(np.array([10.0, 100, 1000]), False, "log", 10.0), [1, 2, 3], atol=1e-4)
    np.testing.assert_allclose(orc.math(np.array([1, 2, 3], f), False, "expf", 10.0), [10, 100, 1000], rtol=1e-6)
    np.testing.assert_allclose(orc.math(np.array([1, 2, 3], f), False, "expf_approx", 10.0), [10, 100, 1000], rtol=1e-4)
    np.testing.assert_allclose(orc.math(np.array([1, 2, 3], f), False, "powf_approx", 3.0), [1, 8, 27], rtol=1e-4)
    assert list(orc.math(np.array([1, -2], f), False, "abs")) == [1, 2]
    assert list(orc.math(np.arange(1, 9, dtype=f), False, "wrap", 4.0)) == [1, 2, 3, 0, 1, 2, 3, 0]
    assert list(orc.unwrap(np.array([1, 2, 3, 0, 1, 2, 3, 0], f), 4.0)) == [1, 2, 3, 4, 5, 6, 7, 8]
    assert list(orc.diff(np.array([2, 3, 2, 6], f), False)) == [1, -1, 4]
    assert list(orc.diff(np.array([2, 2, 3, 3, 5, 5], f), True)) == [1, 1, 2, 2]
    assert list(orc.diff(np.array([2, 3, 2, 6], f), False, True)) == [2, 1, -1, 4]
    assert list(orc.diff(np.array([2, 2, 3, 3, 5, 5], f), True, True)) == [2, 2, 1, 1, 2, 2]
    assert list(orc.cum_sum(np.array([2, 1, -1, 4], f), False)) == [2, 3, 2, 6]
    assert list(orc.cum_sum(np.array([2, 2, 1, 1, 2, 2], f), True)) == [2, 2, 3, 3, 5, 5]
    code, parts = orc.split_into(np.arange(1, 11, dtype=f), False, 2)
    assert code == 0 and list(parts[0]) == [1, 3, 5, 7, 9] and list(parts[1]) == [2, 4, 6, 8, 10]
    assert orc.split_into(np.arange(1, 10, dtype=f), False, 2)[0] == 7
    assert list(orc.merge([np.array([1, 2], f), np.array([1, 2], f)], False)) == [1, 1, 2, 2]
    # complex family against numpy's complex functions (same principal branches) on a seeded vector
    x = orc.fill_uniform(2000, 77, -3, 3, np.float64)
    z = x[0::2] + 1j * x[1::2]
    ref = {"sqrt": np.sqrt, "square": lambda v: v * v, "ln": np.log, "exp": np.exp, "sin": np.sin, "cos": np.cos,
           "tan": np.tan, "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan, "sinh": np.sinh, "cosh": np.cosh,
           "tanh": np.tanh, "asinh": np.arcsinh, "acosh": np.arccosh, "atanh": np.arctanh}
    for name, fn in ref.items():
        got = orc.math(x, True, name)
        np.testing.assert_allclose(got[0::2] + 1j * got[1::2], fn(z), rtol=1e-9, atol=1e-9, err_msg=name)
    got = orc.math(x, True, "powf", 2.5)
    np.testing.assert_allclose(got[0::2] + 1j * got[1::2], z ** 2.5, rtol=1e-10, atol=1e-10)
    got = orc.math(x, True, "log", 7.0)
    np.testing.assert_allclose(got[0::2] + 1j * got[1::2], np.log(z) / np.log(7.0), rtol=1e-10, atol=1e-12)
    got = orc.math(x, True, "expf", 7.0)
    np.testing.assert_allclose(got[0::2] + 1j * got[1::2], 7.0 ** z, rtol=1e-10, atol=1e-12)
    mag, ph = orc.get_mag_phase(x)
    np.testing.assert_allclose(mag, np.abs(z), rtol=1e-14)
    np.testing.assert_allclose(ph, np.angle(z), rtol=1e-14)
    np.testing.assert_allclose(orc.set_mag_phase(mag, ph), x, atol=1e-13)


def test_reference_overlap_discard_schedule_leaves_a_gap_for_some_sizes():
    """A property of the REFERENCE algorithm, restated faithfully by orc_convolve_signal and documented as a deviation
    in DESIGN.md: when the scalar tail (remainder_len / 2 points, convolution.rs:337, 388-399) starts after the last
    block's results end (:453-458) the outputs in between are never written.  N = 14744, M = 1186: fft_len 8192, one
    block -> outputs [593, 7600), tail [10648, 14744), nothing in [7600, 10648).  The tail-free schedule and the
    direct form (the yardsticks of the GPU parity tests) agree everywhere."""
    n, m = 14744, 1186
    x = orc.fill_uniform(2 * n, 3, -10, 10, np.float64)
    h = orc.fill_uniform(2 * m, 9, -1, 1, np.float64) / m
    direct = orc.convolve_direct(x, h, True)
    code, y, path = orc.convolve_signal(x, h, True)
    assert code == 0 and path == 3            # the overlap_discard branch of the dispatcher
    bad = np.nonzero(np.abs(y - direct) > 1e-9)[0] // 2
    assert bad.min() == 7600 and bad.max() == 10647
    ok = np.ones(n, bool); ok[7600:10648] = False
    assert np.max(np.abs((y - direct).reshape(-1, 2)[ok])) < 1e-9
    code, fair = orc.overlap_discard(x, h, orc.next_power_of_two(m), fair=True)
    assert code == 0 and np.max(np.abs(fair - direct)) < 1e-9
    # ... and no gap at the headline shape of the same length
    h2 = orc.fill_uniform(2 * 1024, 9, -1, 1, np.float64) / 1024
    code, y2, _ = orc.convolve_signal(x, h2, True)
    assert code == 0 and np.max(np.abs(y2 - orc.convolve_direct(x, h2, True))) < 1e-9


def test_exact_weights_mode_is_the_literal_raised_cosine_where_that_is_well_conditioned():
    """orc_set_exact_weights(1) (oracle/bdsp_oracle.c) replaces the raised cosine's literal expression by a long-double,
    cancellation-free evaluation at the same argument: away from the second singularity the two agree to rounding, AT it
    both return the reference's limit value, and an ulp beside it the literal one has lost its digits (which is why the
    mode exists).  The golden vectors above pin the literal mode, the default."""
    import ctypes as C
    assert orc.lib.orc_get_exact_weights() == 0
    for dtype, tol in ((np.float32, 3e-6), (np.float64, 1e-13)):
        for beta in (0.2, 0.35, 0.5):
            xs = [x for x in np.linspace(-12.3, 12.3, 247) if abs(1 - abs(2 * beta * x)) > 0.05]
            lit = np.array([orc.conv_time(1, beta, float(x), dtype) for x in xs])
            with orc.exact_weights():
                assert orc.lib.orc_get_exact_weights() == 1
                ex = np.array([orc.conv_time(1, beta, float(x), dtype) for x in xs])
                at = orc.conv_time(1, beta, 1 / (2 * beta), dtype)
            assert orc.lib.orc_get_exact_weights() == 0
            assert np.max(np.abs(lit - ex)) < tol
            assert abs(at - orc.conv_time(1, beta, 1 / (2 * beta), dtype)) < tol
    beside = float(np.nextafter(2.5, 3.0))
    with orc.exact_weights():
        good = orc.conv_time(1, 0.2, beside, np.float64)
    assert abs(good - 0.1) < 1e-12 and abs(orc.conv_time(1, 0.2, beside, np.float64) - 0.1) > 1e-3
