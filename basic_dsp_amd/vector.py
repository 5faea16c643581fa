"""Host-side mirror of the reference's generic vector (GenDspVec<Vec<T>, T>) for the hot path.

`DspVec` wraps a device-resident handle of libbasic_dsp_hip.so and exposes the reference's method
names and argument meaning (vector crate traits ScaleOps/OffsetOps/ElementaryOps/ComplexOps/
TimeToFrequencyDomainOperations/FrequencyToTimeDomainOperations/ConvolutionOps/InterpolationOps,
SURVEY.md section 8a).  Error behaviour follows the C facade (interop/src/lib.rs:28-76): a method
returns the facade's result code -- 0 ok, -1 vector poisoned, 1..14 = ErrorReason -- and raises
BackendError only when the HIP backend itself fails.  Nothing is computed on the host.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib

TIME, FREQ = 0, 1
WINDOW_TRIANGULAR, WINDOW_HAMMING, WINDOW_BLACKMAN_HARRIS, WINDOW_RECTANGULAR, WINDOW_HANN = 0, 1, 2, 3, 4
CONV_SINC, CONV_RAISED_COSINE = 0, 1
PAD_END, PAD_SURROUND, PAD_CENTER = 0, 1, 2


class DspVec:
    """A real or complex, time- or frequency-domain vector living in HBM."""

    def __init__(self, data=None, is_complex=False, domain=TIME, delta=1.0, dtype=np.float32,
                 length=None, init=0.0, _handle=None, _sfx=None):
        if _handle is not None:
            self._h, self._sfx = _handle, _sfx
            self.dtype = np.float32 if _sfx == "32" else np.float64
            return
        _lib.require_gpu()
        if data is not None:
            data = np.ascontiguousarray(data)
            if data.dtype in (np.complex64, np.complex128):
                is_complex = True
                data = data.view(np.float32 if data.dtype == np.complex64 else np.float64)
            dtype = data.dtype
            length = data.size
        self.dtype = np.dtype(dtype).type
        self._sfx = "32" if self.dtype == np.float32 else "64"
        h = self._fn("new")(int(bool(is_complex)), int(domain), init, int(length), delta)
        if not h:
            raise _lib.BackendError("new%s failed: %s" % (self._sfx, _lib.last_error()))
        self._h = h
        if data is not None and length:
            self._call("overwrite_data", data.ctypes.data_as(C.c_void_p), int(length))

    # ------------------------------------------------------------------ plumbing
    def _fn(self, name):
        return getattr(lib, name + self._sfx)

    def _call(self, name, *args):
        res = self._fn(name)(self._h, *args)
        self._h = res.vector  # ownership moved in and comes back (facade convention)
        return _lib.check(res.result_code, name + self._sfx)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._fn("delete_vector")(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def clone(self):
        h = self._fn("bdsp_hip_vec_clone")(self._h)
        if not h:
            raise _lib.BackendError("clone failed: %s" % _lib.last_error())
        return DspVec(_handle=h, _sfx=self._sfx)

    # ------------------------------------------------------------------ metadata
    def __len__(self):
        return self._fn("get_len")(self._h)

    def len(self):
        return len(self)

    def points(self):
        return self._fn("get_points")(self._h)

    def delta(self):
        return self._fn("get_delta")(self._h)

    def is_complex(self):
        return bool(self._fn("is_complex")(self._h))

    def domain(self):
        return self._fn("get_domain")(self._h)

    def is_erroneous(self):
        return len(self) == 0 and np.isnan(self.delta())

    def device_ptr(self):
        return self._fn("bdsp_hip_vec_device_ptr")(self._h)

    def data(self):
        """Download the valid part as a numpy array of scalars (interleaved if complex)."""
        n = len(self)
        p = self._fn("data")(self._h)
        if not p:
            raise _lib.BackendError("data%s failed: %s" % (self._sfx, _lib.last_error()))
        ct = C.c_float if self._sfx == "32" else C.c_double
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(max(n, 1),))[:n].copy()

    def datac(self):
        d = self.data()
        return d.view(np.complex64 if self._sfx == "32" else np.complex128)

    def overwrite_data(self, data):
        data = np.ascontiguousarray(data, dtype=self.dtype)
        return self._call("overwrite_data", data.ctypes.data_as(C.c_void_p), data.size)

    # ------------------------------------------------------------------ elementwise (a2, a15, a16)
    def scale(self, factor):
        if isinstance(factor, complex):
            return self._call("complex_scale", factor.real, factor.imag)
        return self._call("real_scale", factor)

    def offset(self, value):
        if isinstance(value, complex):
            return self._call("complex_offset", value.real, value.imag)
        return self._call("real_offset", value)

    def add(self, other):
        return self._call("add", other._h)

    def sub(self, other):
        return self._call("sub", other._h)

    def mul(self, other):
        return self._call("mul", other._h)

    def div(self, other):
        return self._call("div", other._h)

    def conj(self):
        return self._call("conj")

    def multiply_complex_exponential(self, a, b):
        return self._call("multiply_complex_exponential", a, b)

    # ------------------------------------------------------------------ complex -> real (a8)
    def magnitude(self):
        return self._call("magnitude")

    def magnitude_squared(self):
        return self._call("magnitude_squared")

    def to_real(self):
        return self._call("to_real")

    def to_imag(self):
        return self._call("to_imag")

    def phase(self):
        return self._call("phase")

    def _get_into(self, name, destination):
        """The facade getters consume their source handle: they run on a clone, so `self` stays usable."""
        src = self.clone()
        code = self._fn(name)(src._h, destination._h)
        src._h = None  # consumed by the call
        return _lib.check(code, name + self._sfx)

    def get_real(self, destination):
        return self._get_into("get_real", destination)

    def get_imag(self, destination):
        return self._get_into("get_imag", destination)

    def get_magnitude(self, destination):
        return self._get_into("get_magnitude", destination)

    def get_magnitude_squared(self, destination):
        return self._get_into("get_magnitude_squared", destination)

    def get_phase(self, destination):
        return self._get_into("get_phase", destination)

    def to_complex(self):
        return self._call("to_complex")

    # ------------------------------------------------------------------ reorganisation (a6, a17)
    def reverse(self):
        return self._call("reverse")

    def swap_halves(self):
        return self._call("swap_halves")

    def fft_shift(self):
        return self._call("fft_shift")

    def ifft_shift(self):
        return self._call("ifft_shift")

    def mirror(self):
        return self._call("mirror")

    def zero_pad(self, points, option=PAD_END):
        return self._call("zero_pad", int(points), int(option))

    def zero_interleave(self, factor):
        return self._call("zero_interleave", int(factor))

    # ------------------------------------------------------------------ windows (a7)
    def apply_window(self, window):
        return self._call("apply_window", int(window))

    def unapply_window(self, window):
        return self._call("unapply_window", int(window))

    # ------------------------------------------------------------------ transforms (a3-a5)
    def plain_fft(self):
        return self._call("plain_fft")

    def plain_ifft(self):
        return self._call("plain_ifft")

    def fft(self):
        return self._call("fft")

    def ifft(self):
        return self._call("ifft")

    def windowed_fft(self, window):
        return self._call("windowed_fft", int(window))

    def windowed_ifft(self, window):
        return self._call("windowed_ifft", int(window))

    # ------------------------------------------------------------------ convolution / interpolation
    def convolve_signal(self, impulse_response):
        return self._call("convolve_signal", impulse_response._h)

    def interpolatei(self, function, interpolation_factor, rolloff=0.0):
        return self._call("interpolatei", int(function), rolloff, int(interpolation_factor))

    def interpolate(self, function, dest_points, delay=0.0, rolloff=0.0):
        if function is None:
            return self._call("interpft", int(dest_points))
        return self._call("interpolate", int(function), rolloff, int(dest_points), delay)

    def interpft(self, dest_points):
        return self._call("interpft", int(dest_points))

    def decimatei(self, decimation_factor, delay):
        return self._call("decimatei", int(decimation_factor), int(delay))

    def multiply_frequency_response(self, function, ratio, rolloff=0.0):
        return self._call("multiply_frequency_response", int(function), rolloff, ratio)

    def plain_sfft(self):
        return self._call("plain_sfft")

    def sfft(self):
        return self._call("sfft")

    def windowed_sfft(self, window):
        return self._call("windowed_sfft", int(window))

    def plain_sifft(self):
        return self._call("plain_sifft")

    def sifft(self):
        return self._call("sifft")

    def windowed_sifft(self, window):
        return self._call("windowed_sifft", int(window))

    # ------------------------------------------------------------------ per-element math family & co.
    def _math0(name):  # noqa: N805 -- method factory
        def method(self):
            return self._call(name)
        method.__name__ = name
        method.__doc__ = "TrigOps / PowerOps / RealOps `%s` in place (trigonometry_and_powers.rs, real_ops.rs)" % name
        return method

    def _math1(name):  # noqa: N805
        def method(self, value):
            return self._call(name, value)
        method.__name__ = name
        return method

    for _n in ("sqrt", "square", "ln", "exp", "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh",
               "asinh", "acosh", "atanh", "abs", "ln_approx", "exp_approx", "sin_approx", "cos_approx", "diff",
               "diff_with_start", "cum_sum"):
        locals()[_n] = _math0(_n)
    for _n in ("powf", "root", "log", "expf", "wrap", "unwrap", "log_approx", "expf_approx", "powf_approx"):
        locals()[_n] = _math1(_n)
    del _n, _math0, _math1

    def map_inplace(self, f):
        """f(value, index) -> value, called on the host for every element (mapping.rs:53-79, 163-190)"""
        if self.is_complex():
            def cb(re, im, i, out):
                r = complex(f(complex(re, im), i))
                out[0], out[1] = r.real, r.imag
            keep = getattr(_lib, "MAP_COMPLEX_PTR_FN" + self._sfx)(cb)
            self._fn("bdsp_hip_set_map_complex_bridge")(keep)
            import ctypes as C
            bridge = C.cast(self._fn("bdsp_hip_map_complex_bridge"), C.c_void_p)
            return self._call("map_inplace_complex", bridge)
        return self._call("map_inplace_real", getattr(_lib, "MAP_REAL_FN" + self._sfx)(lambda x, i: f(x, i)))

    def map_aggregate(self, map_fn, aggregate):
        """Folds map_fn(value, index) over the vector with aggregate(a, b); the C ABI carries the intermediate
        values as opaque pointers, here as keys into a table of Python objects.  Returns (code, result)."""
        table = [None]

        def put(obj):
            table.append(obj)
            return len(table) - 1
        agg = _lib.AGG_FN(lambda a, b: put(aggregate(table[a], table[b])))
        if self.is_complex():
            m = getattr(_lib, "AGG_MAP_COMPLEX_FN" + self._sfx)(lambda z, i: put(map_fn(complex(z.re, z.im), i)))
            r = self._fn("map_aggregate_complex")(self._h, m, agg)
        else:
            m = getattr(_lib, "AGG_MAP_REAL_FN" + self._sfx)(lambda x, i: put(map_fn(x, i)))
            r = self._fn("map_aggregate_real")(self._h, m, agg)
        _lib.check(r.result_code, "map_aggregate")
        return r.result_code, (table[r.result] if r.result_code == 0 and r.result else None)

    def _get_pair(self, name, a, b):
        src = self.clone()
        code = self._fn(name)(src._h, a._h, b._h)
        src._h = None  # consumed by the call
        return _lib.check(code, name + self._sfx)

    def get_real_imag(self, real, imag):
        return self._get_pair("get_real_imag", real, imag)

    def get_mag_phase(self, mag, phase):
        return self._get_pair("get_mag_phase", mag, phase)

    def set_real_imag(self, real, imag):
        return self._call("set_real_imag", real._h, imag._h)

    def set_mag_phase(self, mag, phase):
        return self._call("set_mag_phase", mag._h, phase._h)

    def split_into(self, targets):
        import ctypes as C
        arr = (C.c_void_p * len(targets))(*[t._h for t in targets])
        return _lib.check(self._fn("split_into")(self._h, arr, len(targets)), "split_into")

    def merge(self, sources):
        import ctypes as C
        arr = (C.c_void_p * len(sources))(*[t._h for t in sources])
        return self._call("merge", arr, len(sources))

    def _complex_fn(self, f):
        """(callback, data) for a complex-valued facade callback: the library's bridge + a pointer-style closure"""
        import ctypes as C

        def cb(_ctx, x, out):
            r = complex(f(x))
            out[0], out[1] = r.real, r.imag
        bridge = getattr(_lib, "ComplexBridge" + self._sfx)(getattr(_lib, "COMPLEX_PTR_FN" + self._sfx)(cb), None)
        return C.cast(self._fn("bdsp_hip_complex_fn_bridge"), C.c_void_p), bridge

    def convolve_complex(self, function, ratio, conv_len):
        import ctypes as C
        fn, bridge = self._complex_fn(function)
        return self._call("convolve_complex", fn, C.addressof(bridge), True, ratio, int(conv_len))

    def multiply_frequency_response_complex(self, function, ratio, is_symmetric=False):
        import ctypes as C
        fn, bridge = self._complex_fn(function)
        return self._call("multiply_frequency_response_complex", fn, C.addressof(bridge), bool(is_symmetric), ratio)

    def interpolatef_custom(self, function, interpolation_factor, delay, conv_len):
        cb = self._real_fn(lambda _data, x: function(x))
        return self._call("interpolatef_custom", cb, None, True, interpolation_factor, delay, int(conv_len))

    def interpolate_custom(self, function, dest_points, delay=0.0, is_symmetric=True):
        cb = self._real_fn(lambda _data, x: function(x))
        return self._call("interpolate_custom", cb, None, bool(is_symmetric), int(dest_points), delay)

    def interpolatei_custom(self, function, interpolation_factor, is_symmetric=True):
        cb = self._real_fn(lambda _data, x: function(x))
        return self._call("interpolatei_custom", cb, None, bool(is_symmetric), int(interpolation_factor))

    # ------------------------------------------------------------------ statistics, sums, dot products
    @staticmethod
    def _stats_dict(st, cplx):
        c = (lambda z: complex(z.re, z.im)) if cplx else (lambda z: z)
        return dict(sum=c(st.sum), count=st.count, average=c(st.average), rms=c(st.rms), min=c(st.min),
                    min_index=st.min_index, max=c(st.max), max_index=st.max_index)

    def statistics(self, prec=False):
        """StatisticsOps::statistics / PreciseStatisticsOps::statistics_prec as a dict."""
        cplx = self.is_complex()
        name = ("complex" if cplx else "real") + "_statistics" + ("_prec" if prec else "")
        return self._stats_dict(self._fn(name)(self._h), cplx)

    def statistics_split(self, length, prec=False):
        cplx = self.is_complex()
        kind = (_lib.ComplexStatistics64 if cplx else _lib.Statistics64) if (prec or self._sfx == "64") else \
            (_lib.ComplexStatistics32 if cplx else _lib.Statistics32)
        arr = (kind * max(length, 1))()
        name = ("complex" if cplx else "real") + "_statistics_split" + ("_prec" if prec else "")
        code = _lib.check(self._fn(name)(self._h, arr, int(length)), name)
        return code, [self._stats_dict(arr[i], cplx) for i in range(length if code == 0 else 0)]

    def sum(self, prec=False):
        if self.is_complex():
            z = self._fn("complex_sum" + ("_prec" if prec else ""))(self._h)
            return complex(z.re, z.im)
        return self._fn("real_sum" + ("_prec" if prec else ""))(self._h)

    def sum_sq(self, prec=False):
        if self.is_complex():
            z = self._fn("complex_sum_sq" + ("_prec" if prec else ""))(self._h)
            return complex(z.re, z.im)
        return self._fn("real_sum_sq" + ("_prec" if prec else ""))(self._h)

    def dot_product(self, other, prec=False):
        """returns (result_code, value): codes as in dot_products.rs (4 / 3 / 2), 0 ok, -1 poisoned"""
        cplx = self.is_complex()
        r = self._fn(("complex" if cplx else "real") + "_dot_product" + ("_prec" if prec else ""))(self._h, other._h)
        _lib.check(r.result_code, "dot_product")
        return r.result_code, (complex(r.result.re, r.result.im) if cplx else r.result)

    # ------------------------------------------------------------------ correlation, function convolution
    def prepare_argument(self):
        return self._call("prepare_argument")

    def prepare_argument_padded(self):
        return self._call("prepare_argument_padded")

    def correlate(self, other):
        return self._call("correlate", other._h)

    def convolve(self, function, ratio, conv_len, rolloff=0.0):
        """Convolution with a built-in impulse response id, or with a Python callable f(x) that the
        host samples into 2*conv_len+1 weights (the facade's convolve_real callback variant)."""
        if callable(function):
            cb = self._real_fn(lambda _data, x: function(x))
            return self._call("convolve_real", cb, None, True, ratio, int(conv_len))
        return self._call("convolve", int(function), rolloff, ratio, int(conv_len))

    def multiply_frequency_response_fn(self, function, ratio, is_symmetric=True):
        cb = self._real_fn(lambda _data, x: function(x))
        return self._call("multiply_frequency_response_real", cb, None, bool(is_symmetric), ratio)

    def _real_fn(self, f):
        return (_lib.REAL_FN32 if self._sfx == "32" else _lib.REAL_FN64)(f)

    def _window_fn(self, f):
        return (_lib.WINDOW_FN32 if self._sfx == "32" else _lib.WINDOW_FN64)(lambda _d, n, length: f(n, length))

    def apply_custom_window(self, window, is_symmetric=True):
        return self._call("apply_custom_window", self._window_fn(window), None, bool(is_symmetric))

    def unapply_custom_window(self, window, is_symmetric=True):
        return self._call("unapply_custom_window", self._window_fn(window), None, bool(is_symmetric))

    def windowed_custom_fft(self, window, is_symmetric=True):
        return self._call("windowed_custom_fft", self._window_fn(window), None, bool(is_symmetric))

    def windowed_custom_ifft(self, window, is_symmetric=True):
        return self._call("windowed_custom_ifft", self._window_fn(window), None, bool(is_symmetric))

    def windowed_custom_sfft(self, window, is_symmetric=True):
        return self._call("windowed_custom_sfft", self._window_fn(window), None, bool(is_symmetric))

    def windowed_custom_sifft(self, window, is_symmetric=True):
        return self._call("windowed_custom_sifft", self._window_fn(window), None, bool(is_symmetric))

    # ------------------------------------------------------------------ real interpolation, wrap-around ops
    def interpolate_lin(self, interpolation_factor, delay=0.0):
        return self._call("interpolate_lin", interpolation_factor, delay)

    def interpolate_hermite(self, interpolation_factor, delay=0.0):
        return self._call("interpolate_hermite", interpolation_factor, delay)

    def add_smaller(self, other):
        return self._call("add_smaller_vector", other._h)

    def sub_smaller(self, other):
        return self._call("sub_smaller_vector", other._h)

    def mul_smaller(self, other):
        return self._call("mul_smaller_vector", other._h)

    def div_smaller(self, other):
        return self._call("div_smaller_vector", other._h)

    def complex_divide(self, value):
        return self._call("complex_divide", value.real, value.imag)

    def set_value(self, index, value):
        self._fn("set_value")(self._h, int(index), value)

    def allocated_len(self):
        return self._fn("get_allocated_len")(self._h)

    def interpolatef(self, function, interpolation_factor, delay, conv_len, rolloff=0.0):
        return self._call("interpolatef", int(function), rolloff, interpolation_factor, delay,
                          int(conv_len))


# ---------------------------------------------------------------------- B1 helpers (host slices)
def _sfx_of(a):
    return "f32" if a.dtype == np.float32 else "f64"


def gpu_fft(signal, inverse=False):
    """GpuSupport::fft on a host slice (vector/src/gpu_support/mod.rs:35): in place, returns it."""
    signal = np.ascontiguousarray(signal)
    fn = getattr(lib, "bdsp_hip_fft_" + _sfx_of(signal))
    _lib.check(fn(1, signal.ctypes.data_as(C.c_void_p), signal.size, int(inverse)), "fft")
    return signal


def gpu_convolve_vector(source, imp_resp, is_complex=True):
    """GpuSupport::gpu_convolve_vector (mod.rs:24-29): returns (target, range) or (None, None)."""
    source = np.ascontiguousarray(source)
    imp = np.ascontiguousarray(imp_resp, dtype=source.dtype)
    target = np.zeros_like(source)
    rs, re = C.c_size_t(0), C.c_size_t(0)
    fn = getattr(lib, "bdsp_hip_convolve_vector_" + _sfx_of(source))
    code = fn(int(is_complex), source.ctypes.data_as(C.c_void_p), source.size,
              target.ctypes.data_as(C.c_void_p), target.size, imp.ctypes.data_as(C.c_void_p),
              imp.size, C.byref(rs), C.byref(re))
    _lib.check(code, "gpu_convolve_vector")
    if code == 0:
        return None, None
    return target, (rs.value, re.value)


def gpu_overlap_discard(x_time, tmp, h_freq, imp_len, step_size):
    """GpuSupport::overlap_discard (mod.rs:38-45); x_time and tmp are modified in place."""
    fn = getattr(lib, "bdsp_hip_overlap_discard_" + _sfx_of(x_time))
    x_freq = np.zeros_like(h_freq)
    pos = fn(x_time.ctypes.data_as(C.c_void_p), x_time.size, tmp.ctypes.data_as(C.c_void_p),
             tmp.size, x_freq.ctypes.data_as(C.c_void_p), x_freq.size,
             h_freq.ctypes.data_as(C.c_void_p), h_freq.size, int(imp_len), int(step_size))
    # the return value is a position; failure comes out of band (the call clears last_error on entry), as in shim/hip.rs
    err = _lib.last_error()
    if err:
        raise _lib.BackendError("overlap_discard failed: %s" % err)
    return pos
