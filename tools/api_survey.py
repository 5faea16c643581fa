#!/usr/bin/env python3
"""GPU box: one timing per public operation on a 4M-point complex f32 vector (facade calls, median of 5, host clock), next
to what the same bytes cost at the copy rate -- to find operations that still take extra trips through memory."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as orc, basic_dsp_amd as bd
from basic_dsp_amd import DspVec, vector as V
n = 1 << 22
x = orc.fill_uniform(2 * n, 1, -10, 10, np.float32)
xr = orc.fill_uniform(n, 2, -10, 10, np.float32)
sync = lambda: bd.lib.bdsp_hip_synchronize(None)


def t(make, fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        v = make(); sync(); t0 = time.perf_counter(); r = fn(v); sync(); ts.append(time.perf_counter() - t0)
        assert r in (0, None) or not isinstance(r, int), r
    return sorted(ts[1:])[reps // 2] * 1e6


cplx = lambda: DspVec(x, is_complex=True)
freq = lambda: DspVec(x, is_complex=True, domain=1)
real = lambda: DspVec(xr)
other = DspVec(x, is_complex=True)
copy_us = 2 * 8 * n / 6.3e6  # read + write of the vector at 6.3 TB/s
rows = [
    ("scale (1 trip)", cplx, lambda v: v.scale(1.5)),
    ("mul vector (1.5 trips)", cplx, lambda v: v.mul(other)),
    ("magnitude", cplx, lambda v: v.magnitude()),
    ("swap_halves", cplx, lambda v: v.swap_halves()),
    ("reverse", cplx, lambda v: v.reverse()),
    ("zero_pad x2 (End)", cplx, lambda v: v.zero_pad(2 * n, V.PAD_END)),
    ("zero_interleave x2", cplx, lambda v: v.zero_interleave(2)),
    ("apply_window Hamming", cplx, lambda v: v.apply_window(V.WINDOW_HAMMING)),
    ("apply_window Blackman-Harris", cplx, lambda v: v.apply_window(V.WINDOW_BLACKMAN_HARRIS)),
    ("multiply_complex_exponential", cplx, lambda v: v.multiply_complex_exponential(0.1, 0.2)),
    ("plain_fft (2 passes)", cplx, lambda v: v.plain_fft()),
    ("fft", cplx, lambda v: v.fft()),
    ("windowed_fft Hann", cplx, lambda v: v.windowed_fft(V.WINDOW_HANN)),
    ("plain_fft of a REAL vector", real, lambda v: v.plain_fft()),
    ("ifft", freq, lambda v: v.ifft()),
    ("windowed_ifft Hann", freq, lambda v: v.windowed_ifft(V.WINDOW_HANN)),
    ("plain_sfft (real, odd length n-1)", lambda: DspVec(xr[: n - 1]), lambda v: v.plain_sfft()),
    ("convolve_signal 1024 taps", cplx, lambda v: v.convolve_signal(DspVec(x[:2048] / 1024, is_complex=True))),
    ("correlate", cplx, lambda v: v.correlate(other.prepare_argument_padded() or other)),
    ("interpolatei x2 (sinc)", cplx, lambda v: v.interpolatei(0, 2)),
    ("interpolate to 1.5 n (sinc)", cplx, lambda v: v.interpolate(0, 3 * n // 2)),
    ("interpft to 2 n", cplx, lambda v: v.interpft(2 * n)),
    ("interpolatef x2 conv_len 12 (RC)", cplx, lambda v: v.interpolatef(1, 2.0, 0.0, 12, 0.35)),
    ("decimatei /2", cplx, lambda v: v.decimatei(2, 0)),
    ("multiply_frequency_response (sinc)", freq, lambda v: v.multiply_frequency_response(0, 0.5)),
    ("statistics", cplx, lambda v: v.statistics()),
]
print("# 4M-point complex f32 vector (32 MB); one read + one write at 6.3 TB/s = %.1f us" % copy_us)
for name, make, fn in rows:
    try:
        print("%-44s %8.1f us" % (name, t(make, fn)))
    except Exception as e:  # noqa: BLE001
        print("%-44s failed: %s" % (name, str(e)[:80]))
    sys.stdout.flush()
