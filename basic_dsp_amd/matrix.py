"""Host-side mirror of the reference's matrix crate for the hot path (matrix/src/lib.rs,
matrix/src/time_freq.rs): `DspMat` = a set of equally long row vectors that live in ONE HBM
allocation; every method is a batched launch over all rows (the reference loops over the rows on
one CPU thread, matrix/src/lib.rs:195-208).  Methods return the facade result codes like DspVec.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib
from .vector import DspVec, TIME, PAD_END


class DspMat:
    def __init__(self, rows_data=None, is_complex=False, domain=TIME, delta=1.0, dtype=np.float32,
                 rows=None, row_len=None):
        _lib.require_gpu()
        if rows_data is not None:
            a = np.ascontiguousarray(rows_data)
            if a.dtype in (np.complex64, np.complex128):
                is_complex = True
                a = a.view(np.float32 if a.dtype == np.complex64 else np.float64)
            assert a.ndim == 2, "rows_data must be [rows, row_len]"
            dtype, (rows, row_len) = a.dtype, a.shape
        self.dtype = np.dtype(dtype).type
        self._sfx = "32" if self.dtype == np.float32 else "64"
        h = self._fn("new")(int(bool(is_complex)), int(domain), int(rows), int(row_len), delta)
        if not h:
            raise _lib.BackendError("mat_new%s failed: %s" % (self._sfx, _lib.last_error()))
        self._h = h
        if rows_data is not None and a.size:
            _lib.check(self._fn("upload")(self._h, a.ctypes.data_as(C.c_void_p), a.size), "mat_upload")

    def _fn(self, name):
        return getattr(lib, "bdsp_hip_mat_" + name + self._sfx)

    def _call(self, name, *args):
        return _lib.check(self._fn(name)(self._h, *args), "mat_" + name + self._sfx)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._fn("delete")(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    # ------------------------------------------------------------------ metadata / transfer
    def rows(self):
        return self._fn("rows")(self._h)

    def row_len(self):
        return self._fn("row_len")(self._h)

    def row_points(self):
        return self._fn("row_points")(self._h)

    def is_complex(self):
        return bool(self._fn("is_complex")(self._h))

    def domain(self):
        return self._fn("get_domain")(self._h)

    def delta(self):
        return self._fn("get_delta")(self._h)

    def device_ptr(self):
        return self._fn("device_ptr")(self._h)

    def data(self):
        """Download as a [rows, row_len] array of scalars (interleaved if complex)."""
        out = np.empty((self.rows(), self.row_len()), dtype=self.dtype)
        if out.size:
            _lib.check(self._fn("download")(self._h, out.ctypes.data_as(C.c_void_p), out.size), "mat_download")
        return out

    def get_row(self, row):
        h = self._fn("get_row")(self._h, int(row))
        if not h:
            raise IndexError(row)
        return DspVec(_handle=h, _sfx=self._sfx)

    def set_row(self, row, vector):
        return self._call("set_row", int(row), vector._h)

    # ------------------------------------------------------------------ elementwise
    def scale(self, factor):
        if isinstance(factor, complex):
            return self._call("complex_scale", factor.real, factor.imag)
        return self._call("real_scale", factor)

    def offset(self, value):
        return self._call("real_offset", value)

    def conj(self):
        return self._call("conj")

    def _binary(self, name, other):
        if isinstance(other, DspVec):
            return self._call(name + "_vector", other._h)
        return self._call(name, other._h)

    def add(self, other):
        return self._binary("add", other)

    def sub(self, other):
        return self._binary("sub", other)

    def mul(self, other):
        return self._binary("mul", other)

    def div(self, other):
        return self._binary("div", other)

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

    # ------------------------------------------------------------------ transforms, windows, index moves
    def plain_fft(self):
        return self._call("plain_fft")

    def fft(self):
        return self._call("fft")

    def windowed_fft(self, window):
        return self._call("windowed_fft", int(window))

    def plain_ifft(self):
        return self._call("plain_ifft")

    def ifft(self):
        return self._call("ifft")

    def windowed_ifft(self, window):
        return self._call("windowed_ifft", int(window))

    def apply_window(self, window):
        return self._call("apply_window", int(window))

    def unapply_window(self, window):
        return self._call("unapply_window", int(window))

    def swap_halves(self):
        return self._call("swap_halves")

    def fft_shift(self):
        return self._call("fft_shift")

    def ifft_shift(self):
        return self._call("ifft_shift")

    def zero_pad(self, points, option=PAD_END):
        return self._call("zero_pad", int(points), int(option))

    # ------------------------------------------------------------------ convolution / interpolation
    def convolve_signal(self, impulse_response):
        """One DspVec shared by all rows, or a rows x rows nested list of DspVec (MIMO:
        out[n] = sum_r row[r] (*) h[n][r], matrix/src/time_freq.rs:439-483)."""
        if isinstance(impulse_response, DspVec):
            return self._call("convolve_signal", impulse_response._h)
        flat = [h for row in impulse_response for h in row]
        arr = (C.c_void_p * len(flat))(*[h._h for h in flat])
        return self._call("convolve_signal_mat", arr, len(flat))

    def interpolatef(self, function, interpolation_factor, delay, conv_len, rolloff=0.0):
        return self._call("interpolatef", int(function), rolloff, interpolation_factor, delay, int(conv_len))

    def multiply_frequency_response(self, function, ratio, rolloff=0.0):
        return self._call("multiply_frequency_response", int(function), rolloff, ratio)
