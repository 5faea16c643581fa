"""ctypes binding of libbasic_dsp_hip.so (include/basic_dsp_hip.h).

The shared library is the product; this module only loads it and declares prototypes.  There is no
CPU fallback: if the library (or a gfx950 device) is missing, every operation raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BDSP_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libbasic_dsp_hip.so")  # (override: A/B runs of two builds)


class BackendError(RuntimeError):
    """The HIP backend reported a failure (result code <= -100) or could not be loaded."""


class VectorInteropResult32(C.Structure):
    _fields_ = [("result_code", C.c_int32), ("vector", C.c_void_p)]


class VectorInteropResult64(C.Structure):
    _fields_ = [("result_code", C.c_int32), ("vector", C.c_void_p)]


class Complex32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class Complex64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class PointerInteropResult(C.Structure):
    _fields_ = [("result_code", C.c_int32), ("result", C.c_void_p)]


AGG_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p)


def _stats_struct(name, scalar):
    """#[repr(C)] Statistics<T> (vector/src/vector_types/general/statistics.rs:11-31)."""
    return type(name, (C.Structure,), {"_fields_": [
        ("sum", scalar), ("count", C.c_size_t), ("average", scalar), ("rms", scalar), ("min", scalar),
        ("min_index", C.c_size_t), ("max", scalar), ("max_index", C.c_size_t)]})


Statistics32 = _stats_struct("Statistics32", C.c_float)
Statistics64 = _stats_struct("Statistics64", C.c_double)
ComplexStatistics32 = _stats_struct("ComplexStatistics32", Complex32)
ComplexStatistics64 = _stats_struct("ComplexStatistics64", Complex64)


def _scalar_result(name, scalar):
    return type(name, (C.Structure,), {"_fields_": [("result_code", C.c_int32), ("result", scalar)]})


ScalarInteropResult32 = _scalar_result("ScalarInteropResult32", C.c_float)
ScalarInteropResult64 = _scalar_result("ScalarInteropResult64", C.c_double)
ComplexScalarInteropResult32 = _scalar_result("ComplexScalarInteropResult32", Complex32)
ComplexScalarInteropResult64 = _scalar_result("ComplexScalarInteropResult64", Complex64)


def _preload_shared_hip_runtime():
    """One process must hold ONE copy of the HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (file name without version, soname libamdhip64.so.7); if this library pulls in
    /opt/rocm's copy first and torch is imported later, torch loads a second runtime and whichever
    initialises second reports "no ROCm-capable device".  Opening torch's copy by path first makes
    both our NEEDED (matched by soname) and torch's own lookup (matched by file) resolve to it,
    whatever the import order.  Without torch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    _preload_shared_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            "libbasic_dsp_hip.so is not built (expected %s); run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `make -C basic_dsp_amd/csrc`" % LIB_PATH)
    return C.CDLL(LIB_PATH)


lib = _load()

# Every symbol include/basic_dsp_hip.h declares; tests/test_abi.py checks the header against it.
_P, _SZ, _I, _U = C.c_void_p, C.c_size_t, C.c_int, C.c_uint
_F, _D = C.c_float, C.c_double


def _proto(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_proto("bdsp_hip_last_error", C.c_char_p)
_proto("bdsp_hip_version", C.c_char_p)
for _s, _t in (("f32", _F), ("f64", _D)):
    _proto("bdsp_hip_has_gpu_support_" + _s, _I)
    _proto("bdsp_hip_is_supported_fft_len_" + _s, _I, _I, _SZ)
    _proto("bdsp_hip_fft_" + _s, _I, _I, _P, _SZ, _I)
    _proto("bdsp_hip_convolve_vector_" + _s, _I, _I, _P, _SZ, _P, _SZ, _P, _SZ,
           C.POINTER(_SZ), C.POINTER(_SZ))
    _proto("bdsp_hip_overlap_discard_" + _s, _SZ, _P, _SZ, _P, _SZ, _P, _SZ, _P, _SZ, _SZ, _SZ)

# the B1 size policy (include/basic_dsp_hip.h): keys and accessors
B1_FFT_MIN_LEN_F32, B1_FFT_MIN_LEN_F64, B1_CONV_MIN_WORK_F32, B1_CONV_MIN_WORK_F64 = 0, 1, 2, 3
_proto("bdsp_hip_b1_policy_get", _SZ, _I)
_proto("bdsp_hip_b1_policy_set", _I, _I, _SZ)

for _s, _t, _R in (("32", _F, VectorInteropResult32), ("64", _D, VectorInteropResult64)):
    _proto("new" + _s, _P, C.c_int32, C.c_int32, _t, _SZ, _t)
    _proto("new_with_performance_options" + _s, _P, C.c_int32, C.c_int32, _t, _SZ, _t, _SZ, _I)
    _proto("delete_vector" + _s, None, _P)
    _proto("clone" + _s, _P, _P)
    _proto("bdsp_hip_vec_clone" + _s, _P, _P)
    _proto("get_value" + _s, _t, _P, _SZ)
    _proto("get_len" + _s, _SZ, _P)
    _proto("get_points" + _s, _SZ, _P)
    _proto("get_delta" + _s, _t, _P)
    _proto("is_complex" + _s, C.c_int32, _P)
    _proto("get_domain" + _s, C.c_int32, _P)
    _proto("data" + _s, _P, _P)
    _proto("overwrite_data" + _s, _R, _P, _P, _SZ)
    _proto("set_len" + _s, None, _P, _SZ)
    _proto("bdsp_hip_vec_device_ptr" + _s, _P, _P)
    for _n in ("real_offset", "real_scale"):
        _proto(_n + _s, _R, _P, _t)
    for _n in ("complex_offset", "complex_scale", "multiply_complex_exponential"):
        _proto(_n + _s, _R, _P, _t, _t)
    for _n in ("add", "sub", "mul", "div", "convolve_signal"):
        _proto(_n + _s, _R, _P, _P)
    for _n in ("conj", "magnitude", "magnitude_squared", "to_real", "to_imag", "phase", "to_complex",
               "reverse", "swap_halves", "fft_shift", "ifft_shift", "mirror", "plain_fft",
               "plain_ifft", "fft", "ifft"):
        _proto(_n + _s, _R, _P)
    for _n in ("apply_window", "unapply_window", "windowed_fft", "windowed_ifft",
               "zero_interleave"):
        _proto(_n + _s, _R, _P, C.c_int32)
    _proto("zero_pad" + _s, _R, _P, _SZ, C.c_int32)
    _proto("interpolatef" + _s, _R, _P, C.c_int32, _t, _t, _t, _SZ)
    _proto("interpolatei" + _s, _R, _P, C.c_int32, _t, C.c_int32)
    _proto("interpolate" + _s, _R, _P, C.c_int32, _t, _SZ, _t)
    _proto("interpft" + _s, _R, _P, _SZ)
    _proto("decimatei" + _s, _R, _P, C.c_uint32, C.c_uint32)
    _proto("multiply_frequency_response" + _s, _R, _P, C.c_int32, _t, _t)
    for _n in ("plain_sfft", "sfft", "plain_sifft", "sifft"):
        _proto(_n + _s, _R, _P)
    for _n in ("windowed_sfft", "windowed_sifft"):
        _proto(_n + _s, _R, _P, C.c_int32)
    _proto("new_with_detailed_performance_options" + _s, _P, C.c_int32, C.c_int32, _t, _SZ, _t,
           _SZ, _SZ, _SZ, _SZ, _SZ)
    _proto("set_value" + _s, None, _P, _SZ, _t)
    _proto("get_allocated_len" + _s, _SZ, _P)
    _proto("complex_data" + _s, _P, _P)
    _proto("complex_divide" + _s, _R, _P, _t, _t)
    for _n in ("add_vector", "sub_vector", "mul_vector", "div_vector", "add_smaller_vector",
               "sub_smaller_vector", "mul_smaller_vector", "div_smaller_vector", "correlate"):
        _proto(_n + _s, _R, _P, _P)
    for _n in ("prepare_argument", "prepare_argument_padded"):
        _proto(_n + _s, _R, _P)
    _proto("convolve" + _s, _R, _P, C.c_int32, _t, _t, _SZ)
    for _n in ("get_real", "get_imag", "get_magnitude", "get_magnitude_squared", "get_phase"):
        _proto(_n + _s, C.c_int32, _P, _P)
    for _n in ("interpolate_lin", "interpolate_hermite"):
        _proto(_n + _s, _R, _P, _t, _t)
    # callback variants: the callbacks run on the host (sampled into a table), see the header
    _proto("convolve_real" + _s, _R, _P, _P, _P, C.c_bool, _t, _SZ)
    _proto("multiply_frequency_response_real" + _s, _R, _P, _P, _P, C.c_bool, _t)
    for _n in ("apply_custom_window", "unapply_custom_window", "windowed_custom_fft",
               "windowed_custom_sfft", "windowed_custom_ifft", "windowed_custom_sifft"):
        _proto(_n + _s, _R, _P, _P, _P, C.c_bool)

    # statistics, sums, dot products
    _ST = Statistics32 if _s == "32" else Statistics64
    _CST = ComplexStatistics32 if _s == "32" else ComplexStatistics64
    _CX = Complex32 if _s == "32" else Complex64
    _SR = ScalarInteropResult32 if _s == "32" else ScalarInteropResult64
    _CSR = ComplexScalarInteropResult32 if _s == "32" else ComplexScalarInteropResult64
    _proto("real_statistics" + _s, _ST, _P)
    _proto("complex_statistics" + _s, _CST, _P)
    _proto("real_statistics_prec" + _s, Statistics64, _P)
    _proto("complex_statistics_prec" + _s, ComplexStatistics64, _P)
    _proto("real_sum" + _s, _t, _P)
    _proto("real_sum_sq" + _s, _t, _P)
    _proto("real_sum_prec" + _s, _D, _P)
    _proto("real_sum_sq_prec" + _s, _D, _P)
    _proto("complex_sum" + _s, _CX, _P)
    _proto("complex_sum_sq" + _s, _CX, _P)
    _proto("complex_sum_prec" + _s, Complex64, _P)
    _proto("complex_sum_sq_prec" + _s, Complex64, _P)
    _proto("real_dot_product" + _s, _SR, _P, _P)
    _proto("complex_dot_product" + _s, _CSR, _P, _P)
    _proto("real_dot_product_prec" + _s, ScalarInteropResult64, _P, _P)
    _proto("complex_dot_product_prec" + _s, ComplexScalarInteropResult64, _P, _P)
    _proto("real_statistics_split" + _s, C.c_int32, _P, C.POINTER(_ST), _SZ)
    _proto("complex_statistics_split" + _s, C.c_int32, _P, C.POINTER(_CST), _SZ)
    _proto("real_statistics_split_prec" + _s, C.c_int32, _P, C.POINTER(Statistics64), _SZ)
    _proto("complex_statistics_split_prec" + _s, C.c_int32, _P, C.POINTER(ComplexStatistics64), _SZ)
    # per-element math family, differences / running sums, pairs, split / merge, callbacks
    _VR = VectorInteropResult32 if _s == "32" else VectorInteropResult64
    for _n in ("sqrt", "square", "ln", "exp", "sin", "cos", "tan", "asin", "acos", "atan", "sinh", "cosh", "tanh",
               "asinh", "acosh", "atanh", "abs", "ln_approx", "exp_approx", "sin_approx", "cos_approx", "diff",
               "diff_with_start", "cum_sum"):
        _proto(_n + _s, _VR, _P)
    for _n in ("powf", "root", "log", "expf", "wrap", "unwrap", "log_approx", "expf_approx", "powf_approx"):
        _proto(_n + _s, _VR, _P, _t)
    MAP_REAL_FN = C.CFUNCTYPE(_t, _t, _SZ)
    # ctypes cannot build callbacks that RETURN structs: the complex-valued callbacks go through the library's
    # bridges (pointer-style Python callback behind a C function with the facade's signature)
    COMPLEX_PTR_FN = C.CFUNCTYPE(None, _P, _t, C.POINTER(_t))
    MAP_COMPLEX_PTR_FN = C.CFUNCTYPE(None, _t, _t, _SZ, C.POINTER(_t))
    ComplexBridge = type("ComplexBridge" + _s, (C.Structure,), {"_fields_": [("fn", COMPLEX_PTR_FN), ("ctx", _P)]})
    MAP_COMPLEX_FN = _P
    COMPLEX_FN = _P
    _proto("bdsp_hip_set_map_complex_bridge" + _s, None, MAP_COMPLEX_PTR_FN)
    AGG_MAP_REAL_FN = C.CFUNCTYPE(_P, _t, _SZ)
    AGG_MAP_COMPLEX_FN = C.CFUNCTYPE(_P, _CX, _SZ)
    AGG_FN = C.CFUNCTYPE(_P, _P, _P)
    _REALFN = C.CFUNCTYPE(_t, _P, _t)
    globals().update({"MAP_REAL_FN" + _s: MAP_REAL_FN, "MAP_COMPLEX_FN" + _s: MAP_COMPLEX_FN,
                      "COMPLEX_PTR_FN" + _s: COMPLEX_PTR_FN, "MAP_COMPLEX_PTR_FN" + _s: MAP_COMPLEX_PTR_FN,
                      "ComplexBridge" + _s: ComplexBridge, "AGG_MAP_REAL_FN" + _s: AGG_MAP_REAL_FN,
                      "AGG_MAP_COMPLEX_FN" + _s: AGG_MAP_COMPLEX_FN})
    _proto("map_inplace_real" + _s, _VR, _P, MAP_REAL_FN)
    _proto("map_inplace_complex" + _s, _VR, _P, MAP_COMPLEX_FN)
    _proto("map_aggregate_real" + _s, PointerInteropResult, _P, AGG_MAP_REAL_FN, AGG_FN)
    _proto("map_aggregate_complex" + _s, PointerInteropResult, _P, AGG_MAP_COMPLEX_FN, AGG_FN)
    _proto("get_real_imag" + _s, C.c_int32, _P, _P, _P)
    _proto("get_mag_phase" + _s, C.c_int32, _P, _P, _P)
    _proto("set_real_imag" + _s, _VR, _P, _P, _P)
    _proto("set_mag_phase" + _s, _VR, _P, _P, _P)
    _proto("split_into" + _s, C.c_int32, _P, C.POINTER(_P), _SZ)
    _proto("merge" + _s, _VR, _P, C.POINTER(_P), _SZ)
    _proto("convolve_complex" + _s, _VR, _P, COMPLEX_FN, _P, C.c_bool, _t, _SZ)
    _proto("multiply_frequency_response_complex" + _s, _VR, _P, COMPLEX_FN, _P, C.c_bool, _t)
    _proto("interpolatef_custom" + _s, _VR, _P, _REALFN, _P, C.c_bool, _t, _t, _SZ)
    _proto("interpolate_custom" + _s, _VR, _P, _REALFN, _P, C.c_bool, _SZ, _t)
    _proto("interpolatei_custom" + _s, _VR, _P, _REALFN, _P, C.c_bool, C.c_int32)
    # matrix / batch API
    _m = "bdsp_hip_mat_"
    _proto(_m + "new" + _s, _P, C.c_int32, C.c_int32, _SZ, _SZ, _t)
    _proto(_m + "delete" + _s, None, _P)
    for _n in ("rows", "row_len", "row_points"):
        _proto(_m + _n + _s, _SZ, _P)
    for _n in ("is_complex", "get_domain"):
        _proto(_m + _n + _s, C.c_int32, _P)
    _proto(_m + "get_delta" + _s, _t, _P)
    _proto(_m + "device_ptr" + _s, _P, _P)
    _proto(_m + "upload" + _s, C.c_int32, _P, _P, _SZ)
    _proto(_m + "download" + _s, C.c_int32, _P, _P, _SZ)
    _proto(_m + "get_row" + _s, _P, _P, _SZ)
    _proto(_m + "set_row" + _s, C.c_int32, _P, _SZ, _P)
    for _n in ("real_scale", "real_offset"):
        _proto(_m + _n + _s, C.c_int32, _P, _t)
    _proto(_m + "complex_scale" + _s, C.c_int32, _P, _t, _t)
    for _n in ("conj", "magnitude", "magnitude_squared", "to_real", "to_imag", "phase", "plain_fft", "fft",
               "plain_ifft", "ifft", "swap_halves", "fft_shift", "ifft_shift"):
        _proto(_m + _n + _s, C.c_int32, _P)
    for _n in ("add", "sub", "mul", "div", "add_vector", "sub_vector", "mul_vector", "div_vector",
               "convolve_signal"):
        _proto(_m + _n + _s, C.c_int32, _P, _P)
    for _n in ("windowed_fft", "windowed_ifft", "apply_window", "unapply_window"):
        _proto(_m + _n + _s, C.c_int32, _P, C.c_int32)
    _proto(_m + "zero_pad" + _s, C.c_int32, _P, _SZ, C.c_int32)
    _proto(_m + "convolve_signal_mat" + _s, C.c_int32, _P, C.POINTER(_P), _SZ)
    _proto(_m + "interpolatef" + _s, C.c_int32, _P, C.c_int32, _t, _t, _t, _SZ)
    _proto(_m + "multiply_frequency_response" + _s, C.c_int32, _P, C.c_int32, _t, _t)

WINDOW_FN32 = C.CFUNCTYPE(_F, _P, _SZ, _SZ)
WINDOW_FN64 = C.CFUNCTYPE(_D, _P, _SZ, _SZ)
REAL_FN32 = C.CFUNCTYPE(_F, _P, _F)
REAL_FN64 = C.CFUNCTYPE(_D, _P, _D)

_proto("bdsp_hip_dev_fft", _I, _I, _P, _P, _SZ, _SZ, _U, _D, _I, _D, C.POINTER(_I), _P)
_proto("bdsp_hip_dev_convolve", _I, _I, _P, _P, _SZ, _SZ, _P, _SZ, _P)
_proto("bdsp_hip_dev_convolve_ex", _I, _I, _P, _P, _SZ, _SZ, _P, _SZ, _I, _I, _P)
_proto("bdsp_hip_conv_spectrum_points", _SZ)
_proto("bdsp_hip_fft_passes", _I, _I, _SZ)
_proto("bdsp_hip_capture_abort", _I, _P)
_proto("bdsp_hip_capture_reset", _I, _P)
_proto("bdsp_hip_compute_units", _I)
_proto("bdsp_hip_dev_conv_prepare", _I, _I, _P, _SZ, _P, _P)
_proto("bdsp_hip_dev_convolve_prepared", _I, _I, _P, _P, _SZ, _SZ, _P, _SZ, _P)
_proto("bdsp_hip_dev_real_scale", _I, _I, _P, _SZ, _D, _P)
_proto("bdsp_hip_dev_real_offset", _I, _I, _P, _SZ, _I, _D, _P)
_proto("bdsp_hip_interpolatef_new_len", _SZ, _I, _SZ, _D)
_proto("bdsp_hip_dev_interpolatef", _I, _I, _P, _P, _SZ, _I, _I, _D, _D, _D, _SZ, _D, _P)
_proto("bdsp_hip_synchronize", _I, _P)
_proto("bdsp_hip_set_device", _I, _I)
_proto("bdsp_hip_event_create", _P)
_proto("bdsp_hip_event_record", _I, _P, _P)
_proto("bdsp_hip_event_elapsed_ms", _I, _P, _P, C.POINTER(_F))
_proto("bdsp_hip_event_destroy", None, _P)
_proto("bdsp_hip_capture_begin", _I, _P)
_proto("bdsp_hip_capture_end", _I, _P, C.POINTER(_P))
_proto("bdsp_hip_graph_launch", _I, _P, _P)
_proto("bdsp_hip_graph_destroy", None, _P)

FFT_INVERSE, FFT_SHIFT_OUT, FFT_SHIFT_IN, FFT_MAGNITUDE = 1, 2, 4, 8

STREAM_DEFAULT = 1  # BDSP_HIP_STREAM_DEFAULT: HIP's null stream (NULL means the library's own stream)


def stream_arg(handle):
    """B3 `stream` argument for a framework stream handle (e.g. torch.cuda.current_stream().cuda_stream).
    A framework's default stream has handle 0, which the C ABI reads as "the library's own stream" -- work queued
    there would run unordered against the framework's.  0 is therefore forwarded as BDSP_HIP_STREAM_DEFAULT."""
    return C.c_void_p(int(handle) if handle else STREAM_DEFAULT)


def torch_stream_arg():
    """The B3 `stream` argument naming torch's current stream."""
    import torch
    return stream_arg(torch.cuda.current_stream().cuda_stream)



class Graph:
    """A captured sequence of library calls (HIP graph).  Usage:
        g = Graph.capture(lambda: (v.scale(2.5), v.offset(-1.25)))   # runs fn once to warm up, then captures
        g.launch()                                                   # replays with one launch
    Capture and replay happen on the library stream unless `stream` (a hipStream_t int) is given."""

    def __init__(self, handle, stream):
        self._h, self._stream = handle, stream

    @classmethod
    def capture(cls, fn, stream=None, warmup=True):
        sp = C.c_void_p(stream)
        if warmup:
            fn()
            check(lib.bdsp_hip_synchronize(sp), "synchronize")
        check(lib.bdsp_hip_capture_begin(sp), "capture_begin")
        try:
            fn()
        except BaseException:
            lib.bdsp_hip_capture_abort(sp)  # the sequence failed half way: leave capture mode, release what it pinned
            raise
        h = C.c_void_p(None)
        check(lib.bdsp_hip_capture_end(sp, C.byref(h)), "capture_end")
        return cls(h, sp)

    def launch(self):
        return check(lib.bdsp_hip_graph_launch(self._h, self._stream), "graph_launch")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib.bdsp_hip_graph_destroy(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None


def last_error():
    return lib.bdsp_hip_last_error().decode()


def check(code, what=""):
    """Raise on backend failures (<= -100); reference-style codes (-1, 1..14) are returned."""
    if code <= -100:
        raise BackendError("%s failed with code %d: %s" % (what or "backend call", code, last_error()))
    return code


def require_gpu():
    if not lib.bdsp_hip_has_gpu_support_f32():
        raise BackendError("no usable gfx950 device: %s" % last_error())
