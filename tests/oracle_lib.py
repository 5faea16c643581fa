"""ctypes binding of the CPU oracle (oracle/libbdsp_oracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "libbdsp_oracle.so")


def _build():
    src = [os.path.join(_DIR, f) for f in ("bdsp_oracle.c", "bdsp_oracle_impl.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src):
        return
    subprocess.check_call(["make", "-C", _DIR, "-s"])


_build()
lib = C.CDLL(_SO)

_T = {"f32": (C.c_float, np.float32), "f64": (C.c_double, np.float64)}


def sfx(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _fn(name, dtype):
    return getattr(lib, "%s_%s" % (name, sfx(dtype)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _real(dtype):
    return _T[sfx(dtype)][0]


def fill_uniform(n, seed, lo, hi, dtype=np.float32):
    x = np.empty(n, dtype=dtype)
    f = _fn("orc_fill_uniform", dtype)
    r = _real(dtype)
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, r, r]
    f.restype = None
    f(_p(x), n, seed, lo, hi)
    return x


def real_scale(x, f):
    y = np.array(x, copy=True); fn = _fn("orc_real_scale", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, _real(y.dtype)]; fn.restype = None
    fn(_p(y), y.size, f); return y


def real_offset(x, f, is_complex=False):
    y = np.array(x, copy=True); fn = _fn("orc_real_offset", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _real(y.dtype)]; fn.restype = None
    fn(_p(y), y.size, int(is_complex), f); return y


def complex_scale(x, re, im):
    y = np.array(x, copy=True); fn = _fn("orc_complex_scale", y.dtype)
    r = _real(y.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, r, r]; fn.restype = None
    fn(_p(y), y.size, re, im); return y


def complex_offset(x, re, im):
    y = np.array(x, copy=True); fn = _fn("orc_complex_offset", y.dtype)
    r = _real(y.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, r, r]; fn.restype = None
    fn(_p(y), y.size, re, im); return y


def binary(x, y, is_complex, op):
    """op: 0 add 1 sub 2 mul 3 div. returns (code, result)."""
    a = np.array(x, copy=True); b = np.ascontiguousarray(y, dtype=a.dtype)
    fn = _fn("orc_binary", a.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    fn.restype = C.c_int
    code = fn(_p(a), a.size, _p(b), b.size, int(is_complex), op)
    return code, a


def multiply_complex_exponential(x, a, b, delta=1.0):
    y = np.array(x, copy=True); fn = _fn("orc_multiply_complex_exponential", y.dtype)
    r = _real(y.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, r, r, r]; fn.restype = None
    fn(_p(y), y.size, a, b, delta); return y


def conj(x):
    y = np.array(x, copy=True); fn = _fn("orc_conj", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t]; fn.restype = None
    fn(_p(y), y.size); return y


def complex_to_real(x, kind):
    """kind: 0 magnitude 1 magnitude_squared 2 to_real 3 to_imag 4 phase"""
    x = np.ascontiguousarray(x); out = np.empty(x.size // 2, dtype=x.dtype)
    fn = _fn("orc_complex_to_real", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]; fn.restype = None
    fn(_p(x), x.size, _p(out), kind); return out


def magnitude(x):
    return complex_to_real(x, 0)


def window_value(wid, alpha, n, length, dtype=np.float32):
    fn = _fn("orc_window_value", dtype); r = _real(dtype)
    fn.argtypes = [C.c_int, r, C.c_size_t, C.c_size_t]; fn.restype = r
    return fn(wid, alpha, n, length)


def apply_window(x, is_complex, wid, alpha=0.54, unapply=False):
    y = np.array(x, copy=True); fn = _fn("orc_apply_window", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, _real(y.dtype), C.c_int]
    fn.restype = None
    fn(_p(y), y.size, int(is_complex), wid, alpha, int(unapply)); return y


def conv_time(fid, rolloff, x, dtype=np.float32):
    fn = _fn("orc_conv_time", dtype); r = _real(dtype)
    fn.argtypes = [C.c_int, r, r]; fn.restype = r
    return fn(fid, rolloff, x)


class exact_weights:
    """Context manager: inside it the oracle's raised cosine returns, for the same argument in T, the weight evaluated in
    long double through the cancellation-free form near its second singularity (oracle/bdsp_oracle.c: the reference's
    own expression has no correct digit left for a tap that lands NEXT to 1 / (2 beta)).  Everything built on
    orc_conv_time follows: interpolatef (both paths), convolve(function)."""

    def __enter__(self):
        lib.orc_set_exact_weights(1)
        return self

    def __exit__(self, *a):
        lib.orc_set_exact_weights(0)
        return False


def conv_freq(fid, rolloff, x, dtype=np.float32):
    fn = _fn("orc_conv_freq", dtype); r = _real(dtype)
    fn.argtypes = [C.c_int, r, r]; fn.restype = r
    return fn(fid, rolloff, x)


def swap_halves(x, is_complex, forward=True):
    y = np.array(x, copy=True); fn = _fn("orc_swap_halves", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]; fn.restype = None
    fn(_p(y), y.size, int(is_complex), int(forward)); return y


def reverse(x, is_complex):
    y = np.array(x, copy=True); fn = _fn("orc_reverse", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]; fn.restype = None
    fn(_p(y), y.size, int(is_complex)); return y


def zero_pad(x, is_complex, points, option, buffered=False):
    x = np.ascontiguousarray(x); step = 2 if is_complex else 1
    out = np.empty(points * step, dtype=x.dtype); fn = _fn("orc_zero_pad", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    fn.restype = C.c_int
    code = fn(_p(x), x.size, int(is_complex), points, option, int(buffered), _p(out))
    return code, out


def zero_interleave(x, is_complex, factor):
    x = np.ascontiguousarray(x); out = np.empty(x.size * factor, dtype=x.dtype)
    fn = _fn("orc_zero_interleave", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, int(is_complex), factor, _p(out)); return out


def mirror(x):
    x = np.ascontiguousarray(x); out = np.empty(2 * x.size - 2, dtype=x.dtype)
    fn = _fn("orc_mirror", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, _p(out)); return out


def fft(x, inverse=False):
    """x: interleaved complex. Unnormalised DFT, any length."""
    y = np.array(x, copy=True); fn = _fn("orc_fft", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]; fn.restype = None
    fn(_p(y), y.size // 2, int(inverse)); return y


def dft_naive(x, inverse=False):
    x = np.ascontiguousarray(x); out = np.empty_like(x); fn = _fn("orc_dft_naive", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; fn.restype = None
    fn(_p(x), _p(out), x.size // 2, int(inverse)); return out


def convolve_direct(x, h, is_complex, first=None, count=None):
    x = np.ascontiguousarray(x); h = np.ascontiguousarray(h, dtype=x.dtype)
    out = np.zeros_like(x)
    if first is None:
        fn = _fn("orc_convolve_direct", x.dtype)
        fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        fn.restype = None
        fn(_p(x), x.size, _p(h), h.size, int(is_complex), _p(out))
        return out
    fn = _fn("orc_convolve_direct_range", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                   C.c_size_t, C.c_size_t]
    fn.restype = None
    fn(_p(x), x.size, _p(h), h.size, int(is_complex), _p(out), first, count)
    step = 2 if is_complex else 1
    return out[first * step:(first + count) * step]


def overlap_discard(x, h, fft_len=0, fair=False):
    y = np.array(x, copy=True); h = np.ascontiguousarray(h, dtype=y.dtype)
    fn = _fn("orc_overlap_discard", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    fn.restype = C.c_int
    code = fn(_p(y), y.size, _p(h), h.size, fft_len, int(fair))
    return code, y


def overlap_save_mt(x, h, fft_len=0, threads=1):
    """Tail-free overlap-save with the blocks spread over `threads` OpenMP threads (bench.py's CPU baseline)."""
    y = np.array(x, copy=True); h = np.ascontiguousarray(h, dtype=y.dtype)
    fn = _fn("orc_overlap_save_mt", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    fn.restype = C.c_int
    code = fn(_p(y), y.size, _p(h), h.size, fft_len, int(threads))
    return code, y


def fft_pow2_mt(x, inverse=False, threads=1):
    """Power-of-two transform with each stage's butterflies spread over `threads` OpenMP threads."""
    y = np.array(x, copy=True); fn = _fn("orc_fft_pow2_mt", y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]; fn.restype = None
    fn(_p(y), y.size // 2, int(inverse), int(threads)); return y


def convolve_signal(x, h, is_complex):
    x = np.ascontiguousarray(x); h = np.ascontiguousarray(h, dtype=x.dtype)
    out = np.zeros_like(x); path = C.c_int(0); fn = _fn("orc_convolve_signal", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                   C.POINTER(C.c_int)]
    fn.restype = C.c_int
    code = fn(_p(x), x.size, _p(h), h.size, int(is_complex), _p(out), C.byref(path))
    return code, out, path.value


def interpolatef_new_len(length, factor, dtype=np.float32):
    fn = _fn("orc_interpolatef_new_len", dtype)
    fn.argtypes = [C.c_size_t, _real(dtype)]; fn.restype = C.c_size_t
    return fn(length, factor)


def interpolatef(x, is_complex, fid, rolloff, factor, delay, conv_len, delta=1.0):
    x = np.ascontiguousarray(x)
    new_len = interpolatef_new_len(x.size, factor, x.dtype)
    out = np.zeros(new_len, dtype=x.dtype); path = C.c_int(0)
    fn = _fn("orc_interpolatef", x.dtype); r = _real(x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, r, r, r, C.c_size_t, r, C.c_void_p,
                   C.POINTER(C.c_int)]
    fn.restype = None
    fn(_p(x), x.size, int(is_complex), fid, rolloff, factor, delay, conv_len, delta, _p(out),
       C.byref(path))
    return out, path.value


def next_power_of_two(v):
    fn = lib.orc_next_power_of_two_f32
    fn.argtypes = [C.c_size_t]; fn.restype = C.c_size_t
    return fn(v)


def multiply_frequency_response(x, is_complex, fid, rolloff, ratio, is_fft_shifted=False):
    y = np.array(x, copy=True); fn = _fn("orc_multiply_frequency_response", y.dtype); r = _real(y.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, r, r, C.c_int]; fn.restype = None
    fn(_p(y), y.size, int(is_complex), fid, rolloff, ratio, int(is_fft_shifted)); return y


def interpolatei(x, is_complex, fid, rolloff, factor):
    x = np.ascontiguousarray(x); out = np.zeros(x.size * max(factor, 1), dtype=x.dtype)
    fn = _fn("orc_interpolatei", x.dtype); r = _real(x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, r, C.c_uint, C.c_void_p]; fn.restype = C.c_int
    code = fn(_p(x), x.size, int(is_complex), fid, rolloff, factor, _p(out)); return code, out


def interpolate(x, is_complex, fid, rolloff, dest_points, delay=0.0, delta=1.0):
    """fid < 0: no frequency response (interpft). returns (code, out, new_delta)"""
    x = np.ascontiguousarray(x); e = 2 if is_complex else 1
    out = np.zeros(dest_points * e, dtype=x.dtype); r = _real(x.dtype); nd = r(0)
    fn = _fn("orc_interpolate", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, r, C.c_size_t, r, r, C.c_void_p, C.POINTER(r)]
    fn.restype = C.c_int
    code = fn(_p(x), x.size, int(is_complex), fid, rolloff, dest_points, delay, delta, _p(out), C.byref(nd))
    return code, out, nd.value


def decimatei(x, is_complex, factor, delay):
    x = np.ascontiguousarray(x); out = np.zeros_like(x); fn = _fn("orc_decimatei", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_void_p]; fn.restype = C.c_size_t
    n = fn(_p(x), x.size, int(is_complex), factor, delay, _p(out)); return out[:n]


def convolve_function(x, is_complex, fid, rolloff, ratio, conv_len):
    x = np.ascontiguousarray(x); out = np.zeros_like(x); fn = _fn("orc_convolve_function", x.dtype); r = _real(x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, r, r, C.c_size_t, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, int(is_complex), fid, rolloff, ratio, conv_len, _p(out)); return out


def prepare_argument(x, padded):
    x = np.ascontiguousarray(x); points = x.size // 2
    out = np.zeros(2 * (2 * points - 1 if padded else points), dtype=x.dtype)
    fn = _fn("orc_prepare_argument", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]; fn.restype = C.c_int
    code = fn(_p(x), x.size, int(padded), _p(out)); return code, out


def correlate(x, arg):
    x = np.ascontiguousarray(x); arg = np.ascontiguousarray(arg, dtype=x.dtype); out = np.zeros_like(arg)
    fn = _fn("orc_correlate", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]; fn.restype = C.c_int
    code = fn(_p(x), x.size, _p(arg), arg.size, _p(out)); return code, out


def interpolate_real_len(length, factor, dtype=np.float32):
    fn = _fn("orc_interpolate_real_len", np.dtype(dtype)); r = _real(np.dtype(dtype))
    fn.argtypes = [C.c_size_t, r]; fn.restype = C.c_size_t
    return fn(length, factor)


def _interp_real(name, x, factor, delay):
    x = np.ascontiguousarray(x); out = np.zeros(interpolate_real_len(x.size, factor, x.dtype), dtype=x.dtype)
    fn = _fn(name, x.dtype); r = _real(x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, r, r, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, factor, delay, _p(out)); return out


def interpolate_lin(x, factor, delay=0.0):
    return _interp_real("orc_interpolate_lin", x, factor, delay)


def interpolate_hermite(x, factor, delay=0.0):
    return _interp_real("orc_interpolate_hermite", x, factor, delay)


def real_statistics(x, first=0, step=1):
    x = np.ascontiguousarray(x); out = np.zeros(8); fn = _fn("orc_real_statistics", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, first, step, _p(out))
    return dict(sum=out[0], count=int(out[1]), average=out[2], rms=out[3], min=out[4], min_index=int(out[5]),
                max=out[6], max_index=int(out[7]))


def complex_statistics(x, first=0, step=1):
    x = np.ascontiguousarray(x); out = np.zeros(13); fn = _fn("orc_complex_statistics", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, first, step, _p(out))
    return dict(sum=complex(out[0], out[1]), count=int(out[2]), average=complex(out[3], out[4]),
                rms=complex(out[5], out[6]), min=complex(out[7], out[8]), min_index=int(out[9]),
                max=complex(out[10], out[11]), max_index=int(out[12]))


def vec_sum(x, is_complex, squared=False):
    x = np.ascontiguousarray(x); out = np.zeros(2); fn = _fn("orc_sum", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]; fn.restype = None
    fn(_p(x), x.size, int(is_complex), int(squared), _p(out))
    return complex(out[0], out[1]) if is_complex else out[0]


def dot(x, y, is_complex):
    x = np.ascontiguousarray(x); y = np.ascontiguousarray(y, dtype=x.dtype); out = np.zeros(2)
    fn = _fn("orc_dot", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]; fn.restype = None
    fn(_p(x), _p(y), min(x.size, y.size), int(is_complex), _p(out))
    return complex(out[0], out[1]) if is_complex else out[0]


# ---- per-element math family, differences / running sums, phase wrapping, split / merge ----------------------
MATH_IDS = {name: i for i, name in enumerate(
    ["sqrt", "square", "powf", "ln", "exp", "log", "expf", "sin", "cos", "tan", "asin", "acos", "atan", "sinh",
     "cosh", "tanh", "asinh", "acosh", "atanh", "abs", "wrap", "expf_approx", "powf_approx"])}


def _scalar(dtype):
    return C.c_float if np.dtype(dtype) == np.float32 else C.c_double


def math(x, is_complex, name, arg=0.0):
    out = np.array(x, copy=True); fn = _fn("orc_math", out.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, _scalar(out.dtype)]; fn.restype = None
    fn(_p(out), out.size, int(is_complex), MATH_IDS[name], arg)
    return out


def diff(x, is_complex, with_start=False):
    out = np.array(x, copy=True)
    if with_start:
        fn = _fn("orc_diff_with_start", out.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]; fn.restype = None
        fn(_p(out), out.size, int(is_complex))
        return out
    fn = _fn("orc_diff", out.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]; fn.restype = C.c_size_t
    return out[: fn(_p(out), out.size, int(is_complex))]


def cum_sum(x, is_complex):
    out = np.array(x, copy=True); fn = _fn("orc_cum_sum", out.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]; fn.restype = None
    fn(_p(out), out.size, int(is_complex))
    return out


def unwrap(x, divisor):
    out = np.array(x, copy=True); fn = _fn("orc_unwrap", out.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, _scalar(out.dtype)]; fn.restype = None
    fn(_p(out), out.size, divisor)
    return out


def get_mag_phase(x):
    x = np.ascontiguousarray(x); mag = np.zeros(x.size // 2, x.dtype); ph = np.zeros(x.size // 2, x.dtype)
    fn = _fn("orc_get_mag_phase", x.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(_p(x), x.size, _p(mag), _p(ph))
    return mag, ph


def set_mag_phase(mag, phase):
    mag = np.ascontiguousarray(mag); phase = np.ascontiguousarray(phase, dtype=mag.dtype)
    out = np.zeros(2 * mag.size, mag.dtype); fn = _fn("orc_set_mag_phase", mag.dtype)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]; fn.restype = None
    fn(_p(mag), _p(phase), mag.size, _p(out))
    return out


def split_into(x, is_complex, n):
    x = np.ascontiguousarray(x); out = np.zeros(x.size, x.dtype); fn = _fn("orc_split_into", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]; fn.restype = C.c_int
    code = fn(_p(x), x.size, int(is_complex), n, _p(out))
    return code, ([out[k * (x.size // n):(k + 1) * (x.size // n)] for k in range(n)] if code == 0 else [])


def merge(sources, is_complex):
    src = np.ascontiguousarray(np.concatenate(sources)); out = np.zeros(src.size, src.dtype)
    fn = _fn("orc_merge", src.dtype); fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]
    fn.restype = None
    fn(_p(src), sources[0].size, int(is_complex), len(sources), _p(out))
    return out
