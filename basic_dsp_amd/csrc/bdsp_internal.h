// bdsp_internal.h -- shared declarations of libbasic_dsp_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/basic_dsp_hip.h"
#include "fft_core.h"

namespace bdsp {

// ---------------------------------------------------------------- errors
void set_last_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define BDSP_HIP_TRY(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return ::bdsp::hip_fail(_e, #expr, __FILE__, __LINE__);              \
    } while (0)
#define BDSP_TRY(expr)                                                                             \
    do {                                                                                           \
        int _c = (expr);                                                                           \
        if (_c != BDSP_OK) return _c;                                                              \
    } while (0)
#define BDSP_LAUNCH_CHECK() BDSP_HIP_TRY(hipGetLastError())

// ---------------------------------------------------------------- runtime
// One context per (process, device).  B1/B2 work runs on the library's own non-blocking stream;
// B3 callers pass theirs.  Workspace blocks are cached per stream so that a block freed on one
// stream is never handed to work queued on another.
int device_ready();                       // BDSP_OK or BDSP_ERR_NO_DEVICE
hipStream_t lib_stream();                 // library stream of the current device
// B3 `stream` argument: NULL = the library's own (non-blocking) stream; BDSP_HIP_STREAM_DEFAULT = (void*)1, the value
// of hipStreamLegacy = HIP's null stream, which is what a framework's "default stream" handle 0 means -- a caller
// that forwards such a handle must map 0 to it, or its work would run unordered on the library stream.
inline hipStream_t pick_stream(void* s)
{
    if (!s) return lib_stream();
    if (s == reinterpret_cast<void*>(1)) return nullptr;
    return reinterpret_cast<hipStream_t>(s);
}
int ws_alloc(void** p, size_t bytes, hipStream_t stream);
void ws_free(void* p, hipStream_t stream);
// HIP-graph capture on `stream`: blocks freed between begin and end are not recycled; end hands them to the caller
// (the graph handle), which gives them back with ws_free when the graph is destroyed
int ws_capture_begin(hipStream_t stream);
void ws_capture_end(hipStream_t stream, std::vector<void*>* pinned);
// Streaming store of one complex value (8 or 16 bytes): bypasses the caches' allocation.  For outputs too large for the
// 256 MB Infinity Cache to hold until their reader comes (a 256 MB interpolation result ran 82 -> 69 us).
template <typename C>
__device__ __forceinline__ void nt_store(C* p, C v)
{
    using R = typename real_of<C>::type;
    typedef R vec2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(vec2{v.x, v.y}, reinterpret_cast<vec2*>(p));
}

// Streaming load (experiment, LAB builds with -DBDSP_FFT_NTLOAD: a pass's input is dead once read)
template <typename C>
__device__ __forceinline__ C nt_load(const C* p)
{
    using R = typename real_of<C>::type;
    typedef R vec2 __attribute__((ext_vector_type(2)));
    const vec2 v = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
    return C{v.x, v.y};
}

int num_cus();

// Experiment switches (tools/plan_probe.py, tools/chunk_probe.py, A/B runs) exist only in the LAB build of the library
// (`make -C basic_dsp_amd/csrc lab` -> lib/libbasic_dsp_hip_lab.so, -DBDSP_LAB); the product reads no environment
// variable and carries no kernel that only such a switch could reach.
#ifdef BDSP_LAB
inline const char* lab_env(const char* name) { return getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif
inline bool lab_flag(const char* name) { return lab_env(name) != nullptr; }

struct WsBlock { // RAII workspace
    void* p = nullptr;
    hipStream_t s = nullptr;
    int alloc(size_t bytes, hipStream_t stream) { s = stream; return ws_alloc(&p, bytes, stream); }
    ~WsBlock() { if (p) ws_free(p, s); }
    template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
};

// Forward twiddle table exp(-2*pi*i*m/n), m in [0, n), generated on the host in double precision,
// rounded once, cached per (device, n, precision).  n <= 8192.
template <typename T> int twiddle_table(int n, const cpx<T>** table);
template <typename T> bool twiddle_table_available(int n); // cached already, or the cache still has room

// ---------------------------------------------------------------- fused FFT prologue / epilogue
template <typename T>
struct FftIo {
    const void* in;
    void* out;
    size_t n;          // points per vector
    size_t in_stride;  // elements (of the input element type) between consecutive batch vectors
    size_t out_stride; // elements (of the output element type) between consecutive batch vectors
    unsigned flags;    // BDSP_FFT_* (SHIFT_IN, SHIFT_OUT, MAGNITUDE) + internal bits below
    T in_scale;        // applied to every input element (1 = none)
    int window_id;     // -1 none; else reference window id (4 = Hann)
    T window_alpha;
    size_t in_valid;   // 0 = all n points; else points >= in_valid read as zero (fused End zero-padding)
    // cos / sin of q * 2 pi (n/16) / (n-1), q = 0..7: a pass thread's sixteen rows are n/16 points apart, so a generalised
    // Hamming window costs it two sincospi and sixteen multiply-adds (filled by launch_pass; k_fft_pass, round 4)
    T win_c[8], win_s[8];
};
constexpr unsigned FFT_IN_REAL = 1u << 8;        // input is a real vector (zero imaginary parts)
constexpr unsigned FFT_WINDOW_OUT_DIV = 1u << 9; // divide OUTPUT by the window (windowed_ifft)
constexpr unsigned FFT_OUT_REAL = 1u << 10;      // write only the real parts

// ---------------------------------------------------------------- kernel launchers (per .hip TU)
// fft.hip
template <typename T>
int fft_pow2(const FftIo<T>& io, T* scratch_a, T* scratch_b, size_t batch, bool inverse,
             hipStream_t s);
template <typename T> int fft_pow2_passes(size_t n); // 1 (n <= 4096), 2 or 3 trips through memory
// trips of a PLAIN complex transform (no fused option): the same, except where fft_pow2 has a workgroup-resident kernel
// for that case only (f32 8192 points: k_fft_wg4); one predicate for the dispatch and for bdsp_hip_fft_passes
template <typename T> int fft_pow2_plain_trips(size_t n);
template <typename T>
int fft_any(const FftIo<T>& io, size_t batch, bool inverse, hipStream_t s);
bool is_pow2(size_t n);

// conv.hip
template <typename T>
int convolve_overlap_save(const T* in, T* out, size_t points, size_t batch, const T* taps_dev,
                          size_t taps, long long in_off, long long out_off, size_t nblocks_limit,
                          T* last_block_out, const T* h_freq_dev, hipStream_t s);
template <typename T>
int convolve_direct(const T* in, T* out, size_t points, size_t batch, const T* taps_dev,
                    size_t taps, bool is_complex, hipStream_t s);
size_t conv_fft_len(size_t taps);
// block step (valid outputs per 4096-point block) of the kernel conv_run_blocks<T> will use for these taps
template <typename T> size_t conv_block_step(size_t points, size_t taps, bool real_data);
// conv_v2.hip: the second-generation block kernel (complex f32 and f64)
size_t conv_v2_block_step(size_t taps);
bool conv_v2_applies(size_t points, size_t taps);
// shares of the block kernel's dispatch groups in percent (-1: the measured defaults) for the CALLING THREAD's next launches
// (bdsp_hip_dev_convolve_ex sets them around one call)
void conv_v2_set_shares(int first_pct, int second_pct);
template <typename T>
int conv_v2_run(const T* in, T* out, size_t points, size_t batch, const T* hs, size_t taps,
                size_t first_block, size_t nblocks, bool hs_is_taps, hipStream_t s, bool real = false);
template <typename T>
int conv_prepare_spectrum(const T* taps_dev, size_t taps, const T* h_freq_dev, T* hs, hipStream_t s);
template <typename T>
int conv_run_blocks(const T* in, T* out, size_t points, size_t batch, const T* hs, size_t taps,
                    long long in_off, long long out_off, size_t nblocks_limit, T* last_block_out,
                    hipStream_t s, bool real_data = false, bool hs_is_taps = false);
// real_data: in/out are REAL vectors of `points` samples and hs is the spectrum of REAL taps; two real
// blocks share one complex transform pair
// the spectrum handed to conv_run_blocks is UNSCALED; the kernel multiplies by 1/L while it loads it
template <typename T> int mul_bcast(T* z, const T* h, size_t l, size_t nb, T scale, hipStream_t s);
template <typename T>
int scatter_valid(const T* z, T* x, size_t l, size_t skip, size_t step, size_t dst_off, size_t nb,
                  size_t x_points, hipStream_t s);

// elementwise.hip
template <typename T> int ew_real_scale(T* x, size_t len, T f, hipStream_t s);
template <typename T> int ew_real_offset(T* x, size_t len, bool is_complex, T f, hipStream_t s);
template <typename T> int ew_complex_scale(T* x, size_t len, T re, T im, hipStream_t s);
template <typename T> int ew_complex_offset(T* x, size_t len, T re, T im, hipStream_t s);
template <typename T> int ew_binary(T* x, const T* y, size_t len, bool is_complex, int op, hipStream_t s);
template <typename T> int ew_binary_smaller(T* x, const T* y, size_t len, size_t ylen, bool is_complex, int op, hipStream_t s);
template <typename T> int ew_point_table(T* x, size_t len, bool is_complex, const T* table, bool divide, hipStream_t s);
template <typename T> int ew_conj(T* x, size_t len, hipStream_t s);
template <typename T> int ew_mul_cexp(T* x, size_t len, T a, T b, hipStream_t s);
template <typename T> int ew_complex_to_real(const T* x, T* out, size_t len, int kind, hipStream_t s);
template <typename T> int ew_window(T* x, size_t len, bool is_complex, int id, T alpha, bool unapply, hipStream_t s);
template <typename T> int ew_fill(T* x, size_t len, T value, hipStream_t s);
template <typename T> int ew_freq_response(T* x, size_t len, bool is_complex, int fid, T rolloff, T ratio, bool shifted, hipStream_t s);
template <typename T> int ew_linear_phase(T* x, size_t len, T delay, hipStream_t s);
// spectrum of N points -> dst_points, out of place, one trip (mode 0: periodic repetition = the transform of the
// zero-interleaved vector; mode 1: zero_pad(Center) with the linear phase on the source bin), x the frequency response
// on the destination axis (fid >= 0), x ratio (fid == -1) or nothing (fid == -2); elementwise.hip
template <typename T> int ew_spectrum_resample(const T* in, T* out, size_t src_points, size_t dst_points, int mode, int fid, T rolloff, T ratio, double phase_inc, hipStream_t s);

// reorg.hip
template <typename T> int rg_rotate(const T* in, T* out, size_t points, size_t elem, size_t shift, hipStream_t s);
template <typename T> int rg_wrap_copy(const T* in, T* out, size_t points, size_t elem, size_t total, long long start, hipStream_t s);
template <typename T> int rg_reverse(const T* in, T* out, size_t points, size_t elem, hipStream_t s);
template <typename T> int rg_zero_pad(const T* in, T* out, size_t len_before, bool is_complex, size_t points, int option, hipStream_t s);
template <typename T> int rg_zero_interleave(const T* in, T* out, size_t len, size_t elem, size_t factor, hipStream_t s);
template <typename T> int rg_mirror(const T* in, T* out, size_t len, hipStream_t s);
template <typename T> int rg_decimate(const T* in, T* out, size_t out_points, size_t elem, size_t factor, size_t delay, hipStream_t s);

// interp.hip
template <typename T>
int interpolatef_dev(const T* in, T* out, size_t len, bool is_complex, int fid, T rolloff, T factor,
                     T delay, size_t conv_len, T delta, hipStream_t s,
                     T (*host_fn)(const void*, T) = nullptr, const void* host_fn_data = nullptr);
template <typename T> size_t interpolatef_new_len(size_t len, T factor);
template <typename T> int conv_function_taps(T* taps, size_t conv_len, int fid, T rolloff, T ratio, int stride, bool reversed, hipStream_t s);
template <typename T> int conv_function_direct(const T* in, T* out, size_t points, bool is_complex, const T* taps, size_t conv_len, hipStream_t s,
                                                   bool complex_taps = false);
template <typename T> size_t interpolate_real_len(size_t len, T factor);
template <typename T> int interpolate_real_dev(const T* in, T* out, size_t len, T factor, T delay, bool hermite, hipStream_t s);

// reduce.hip: what one walk over a vector accumulates (sums in double; min / max with their keys and indices)
// mixed_radix.hip: lengths 2^a 3^b 5^c 7^d that are not powers of two
template <typename T> bool mr_supported(size_t n);
template <typename T> bool mr_resident(size_t n); // one workgroup-resident transform (may run in place)
template <typename T> int mr_passes(size_t n);    // 1 resident, 2 four-step (result in `out`), 3 Stockham passes (out must be `scratch`)
template <typename T> int mr_fft(const T* in, T* out, T* scratch, size_t n, size_t batch, bool inverse, unsigned flags, T in_scale,
                                 int window_id, T window_alpha, hipStream_t s);

// per-element math family ids (vecmath.hip); the oracle uses the same numbering
enum MathFn {
    MATH_SQRT = 0, MATH_SQUARE, MATH_POWF, MATH_LN, MATH_EXP, MATH_LOG, MATH_EXPF, MATH_SIN, MATH_COS, MATH_TAN,
    MATH_ASIN, MATH_ACOS, MATH_ATAN, MATH_SINH, MATH_COSH, MATH_TANH, MATH_ASINH, MATH_ACOSH, MATH_ATANH, MATH_ABS,
    MATH_WRAP, MATH_EXPF_APPROX, MATH_POWF_APPROX
};
template <typename T> int ew_math(T* x, size_t len, bool is_complex, int fn, T arg, hipStream_t s);
template <typename T> int vm_diff(const T* in, T* out, size_t n_out, size_t step, bool with_start, hipStream_t s);
template <typename T> size_t vm_cum_sum_scratch(size_t len, bool is_complex);
template <typename T> int vm_cum_sum(T* x, size_t len, bool is_complex, void* scratch, hipStream_t s);
template <typename T> int vm_unwrap(T* x, size_t len, T divisor, hipStream_t s);
template <typename T> int vm_complex_split(const T* x, T* a, T* b, size_t points, int kind, hipStream_t s);
template <typename T> int vm_complex_join(T* x, const T* a, const T* b, size_t points, int kind, hipStream_t s);
template <typename T> int vm_split_merge(T* whole, T* const* parts_dev, size_t len, bool is_complex, size_t n, bool merge, hipStream_t s);

struct StatPartial {
    double sr, si, qr, qi;       // sum, sum of squares (complex: z*z, not |z|^2 -- statistics.rs:344)
    double mn_key, mx_key;       // ordering keys: the value (real) or its norm (complex)
    double mnr, mni, mxr, mxi;   // the extreme elements themselves
    unsigned long long imn, imx, cnt;
};
template <typename T>
int red_stats(const T* x, size_t total, size_t buckets, bool is_complex, bool minmax, StatPartial* partials, StatPartial* out, hipStream_t s);
template <typename T>
int red_dot(const T* x, const T* y, size_t count, bool is_complex, StatPartial* partials, StatPartial* out, hipStream_t s);

// bluestein.hip
template <typename T> int bs_chirp(T* c, size_t n, bool inverse, hipStream_t s);
template <typename T> int bs_kernel(const T* c, T* b, size_t n, size_t m, hipStream_t s);
template <typename T> int bs_fused(const T* x, T* y, const T* c, const T* bspec, size_t n, size_t m, size_t batch, hipStream_t s);
template <typename T>
int bs_pre(const T* x, T* a, const T* c, size_t n, size_t m, size_t batch, bool in_real, T in_scale, size_t rot,
           int window_id, T alpha, hipStream_t s);
template <typename T>
int bs_post(const T* conv, T* out, const T* c, size_t n, size_t m, size_t batch, int out_kind, size_t rot,
            int div_window_id, T alpha, hipStream_t s);

// window value shared by fft.hip and elementwise.hip (device) -- defined inline in window.h
} // namespace bdsp
