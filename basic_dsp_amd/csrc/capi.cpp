// capi.cpp -- the C ABI of libbasic_dsp_hip.so (include/basic_dsp_hip.h):
//   B1  GpuSupport<T> entry points on host slices   (vector/src/gpu_support/mod.rs:18-46)
//   B2  the reference's C facade on HBM-resident vectors (interop/src/facade32.rs, lib.rs)
//   B3  the same kernels on caller-owned device pointers
// The host-side logic here mirrors the reference's operator layer for the hot path: type-state
// checks, result codes, delta/valid_len bookkeeping and the buffer "trade" of DspVec
// (vector/src/vector_types/mod.rs:125-229, support_std.rs:78-124).
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "bdsp_internal.h"

using namespace bdsp;

namespace {

// ----------------------------------------------------------------------------------------------
// FFT driver on two equally sized device buffers a (holds the input) and b (scratch).
// Both must hold batch * 2 * points scalars.  *in_b tells where the result ended up.
// ----------------------------------------------------------------------------------------------
template <typename T>
int fft_any_len(T* a, T* b, size_t points, size_t batch, bool inverse, unsigned flags, T in_scale,
                int window_id, T window_alpha, bool* in_b, hipStream_t s);

template <typename T>
int fft_two_buffers(T* a, T* b, size_t points, size_t batch, bool inverse, unsigned flags,
                    T in_scale, int window_id, T window_alpha, bool* in_b, hipStream_t s)
{
    *in_b = false;
    if (points == 0 || batch == 0) return BDSP_OK;
    if (!is_pow2(points))
        return fft_any_len<T>(a, b, points, batch, inverse, flags, in_scale, window_id, window_alpha,
                              in_b, s);
    FftIo<T> io{};
    io.n = points;
    io.flags = flags;
    io.in_scale = in_scale;
    io.window_id = window_id;
    io.window_alpha = window_alpha;
    io.in_stride = points;  // elements of the input type (real or complex) per vector
    io.out_stride = points; // elements of the output type per vector
    const bool reshaping = (flags & (FFT_IN_REAL | BDSP_FFT_MAGNITUDE | FFT_OUT_REAL)) != 0;
    if (points <= 4096) {
        io.in = a;
        if (reshaping && batch > 1) { io.out = b; *in_b = true; }
        else if (flags & FFT_IN_REAL) { io.out = b; *in_b = true; }
        else io.out = a;
        return fft_pow2<T>(io, nullptr, nullptr, batch, inverse, s);
    }
    const bool three = fft_pow2_passes<T>(points) == 3;
    io.in = a;
    if (!three) { // a -> b -> a
        io.out = a;
        // a -> b -> b from 2^19 points on: the last pass reads and writes the same index set per workgroup, so it may run in
        // place, and the working set of that pass halves.  *Measured* on valid data, input AND scratch cold / input in the
        // caches (tools/plan_probe.py, profiles/r05_plan_probe_valid.txt; rounds 2-4 had this for ONE 2^21-point f32 vector
        // only, chosen in loops that ran on inf / NaN): f32 2^19 15.2 -> 13.9 / 12.0 -> 10.8 us, 2^20 16.9 -> 16.6 / 15.8 -> 15.2,
        // 2^21 29.6 -> 24.3 / 25.1 -> 19.8, 2^22 43.3 -> 42.8 / 33.4 -> 32.9; f64 2^19 20.4 -> 17.5 / 18.5 -> 16.1, 2^20 22.6 -> 22.2 / =,
        // 2^21 37.3 -> 36.6 / =, 2^22 69.6 -> 66.9 / 46.4 -> 47.2, 16 x 2^20 f64 218.8 -> 214.0 / 213.5 -> 203.2, 64 x 2^20 f32 equal.
        // Below 2^19 nothing moves (2^14 ... 2^18: +-0.2 us) except batches, which LOSE (256 x 2^16 f32: 83.5 -> 97-100 us with the
        // input in the caches), and 2^24's third pass in place measured 135 against 128 us (below).
        static const bool force_inplace = lab_flag("BDSP_FFT_LAST_INPLACE"), no_inplace = lab_flag("BDSP_FFT_NO_LAST_INPLACE");
        if (!reshaping && !no_inplace && (force_inplace || points >= (size_t(1) << 19))) {
            io.out = b;
            *in_b = true;
            return fft_pow2<T>(io, b, nullptr, batch, inverse, s);
        }
        // (Rounds 2-4 sent a large batch through in Infinity-Cache-sized chunks of vectors -- 64 x 1M-point f32: "402 us in one
        // piece, 372 us in chunks of 16", measured in a loop that fed the transform its own output, i.e. on inf / NaN.  On
        // valid data, cold and cache-resident (tools/plan_probe.py, profiles/r05_plan_probe_valid.txt): one piece 421 / 420 us,
        // chunks of 16 423 / 438, of 32 441 / 439, of 8 508 / 500; with a magnitude output 402 / 409 against 404 / 402; 32 x 1M
        // f64 404 / 397 against 426 / 422.  The chunks are gone.)
        return fft_pow2<T>(io, b, nullptr, batch, inverse, s);
    }
    // a -> b -> a -> b.  (The last Stockham pass reads and writes the same index set per workgroup and could run in
    // place, a -> b -> a -> a, halving the working set for the 256 MiB Infinity Cache: *measured* 135 us against 128 us
    // for the 16M-point transform -- rewriting the lines just read is slower than ping-pong.)
    io.out = b;
    *in_b = true;
    return fft_pow2<T>(io, b, a, batch, inverse, s);
}

// Bluestein chirp-z for lengths that are not powers of two (any N, like rustfft); kernels and the
// algebra are in bluestein.hip.  The chirp and the spectrum of the convolution kernel depend only on
// (N, direction, precision): they are built once on the device and cached (LRU, bounded), so a
// steady-state call is pre -> FFT_m -> x B/m -> IFFT_m -> post with every option fused into pre/post.
struct BsPlan {
    void* chirp = nullptr; // N complex
    void* bspec = nullptr; // m complex: FFT_m of the wrapped chirp
    size_t n = 0, m = 0, bytes = 0;
    bool inverse = false;
    int esz = 0, dev = 0;
    unsigned long long stamp = 0;
    int pins = 0; // HIP graphs that recorded this plan's addresses: never evicted while > 0
};
static std::mutex g_bs_mu;
static std::vector<BsPlan> g_bs_plans;
static unsigned long long g_bs_clock = 0;
// capture bookkeeping (bdsp_hip_capture_begin/_end/_abort): ONE capture at a time per process, owned by the thread that
// opened it.  Plans that thread looks up while its capture is open are pinned to the graph; what OTHER threads do in the
// meantime (plan look-ups, buffer trades) is none of the capture's business -- the moves word below is per thread.
static int g_capture_open = 0;
static std::thread::id g_capture_thread;
static hipStream_t g_capture_stream = nullptr;
static std::vector<void*> g_capture_plans; // chirp pointers identify plans
static unsigned long long g_capture_moves = 0; // the capturing thread's t_buffer_moves when the capture was opened
static bool capturing_here() { return g_capture_open && g_capture_thread == std::this_thread::get_id(); }
struct GraphHandle {
    hipGraphExec_t exec = nullptr;
    hipStream_t stream = nullptr;
    std::vector<void*> pinned_blocks; // workspace blocks the captured calls released
    std::vector<void*> pinned_plans;  // Bluestein plans (by chirp pointer) the captured calls used
};
constexpr size_t BS_CACHE_BYTES = size_t(1) << 30;

template <typename T>
int bs_plan(size_t n, size_t m, bool inverse, hipStream_t s, const T** chirp, const T** bspec)
{
    int dev = 0;
    BDSP_HIP_TRY(hipGetDevice(&dev));
    for (auto& p : g_bs_plans)
        if (p.n == n && p.inverse == inverse && p.esz == (int)sizeof(T) && p.dev == dev) {
            p.stamp = ++g_bs_clock;
            if (capturing_here()) { ++p.pins; g_capture_plans.push_back(p.chirp); }
            *chirp = (const T*)p.chirp;
            *bspec = (const T*)p.bspec;
            return BDSP_OK;
        }
    if (capturing_here()) {
        // building a plan synchronises the stream (and may free memory): both would invalidate the open capture
        set_last_error("chirp-z plan missing while a capture is open: run the sequence once before capturing it (warm the plan)");
        return BDSP_ERR_UNSUPPORTED;
    }
    BsPlan p;
    p.n = n; p.m = m; p.inverse = inverse; p.esz = (int)sizeof(T); p.dev = dev;
    p.bytes = sizeof(T) * 2 * (n + m);
    // make room: least recently used plans go first (after the device has drained their users)
    size_t used = 0;
    for (auto& q : g_bs_plans) used += q.bytes;
    while (!g_bs_plans.empty() && used + p.bytes > BS_CACHE_BYTES) {
        size_t lru = g_bs_plans.size();
        for (size_t i = 0; i < g_bs_plans.size(); ++i)
            if (g_bs_plans[i].pins == 0 && (lru == g_bs_plans.size() || g_bs_plans[i].stamp < g_bs_plans[lru].stamp)) lru = i;
        if (lru == g_bs_plans.size()) break; // everything left is pinned by a graph
        BDSP_HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(g_bs_plans[lru].chirp);
        (void)hipFree(g_bs_plans[lru].bspec);
        used -= g_bs_plans[lru].bytes;
        g_bs_plans.erase(g_bs_plans.begin() + (long)lru);
    }
    BDSP_HIP_TRY(hipMalloc(&p.chirp, sizeof(T) * 2 * n));
    if (hipMalloc(&p.bspec, sizeof(T) * 2 * m) != hipSuccess) { (void)hipFree(p.chirp); set_last_error("hipMalloc failed"); return BDSP_ERR_HIP; }
    WsBlock scr;
    int c = scr.alloc(sizeof(T) * 2 * m, s);
    if (c == BDSP_OK) c = bs_chirp<T>((T*)p.chirp, n, inverse, s);
    if (c == BDSP_OK) c = bs_kernel<T>((const T*)p.chirp, (T*)p.bspec, n, m, s);
    bool rb = false;
    if (c == BDSP_OK) c = fft_two_buffers<T>((T*)p.bspec, scr.as<T>(), m, 1, false, 0, (T)1, -1, (T)0, &rb, s);
    if (c == BDSP_OK && rb &&
        hipMemcpyAsync(p.bspec, scr.p, sizeof(T) * 2 * m, hipMemcpyDeviceToDevice, s) != hipSuccess) c = BDSP_ERR_HIP;
    // the plan may be used from another stream next: finish building it first
    if (c == BDSP_OK && hipStreamSynchronize(s) != hipSuccess) c = BDSP_ERR_HIP;
    if (c != BDSP_OK) { (void)hipFree(p.chirp); (void)hipFree(p.bspec); return c; }
    p.stamp = ++g_bs_clock;
    g_bs_plans.push_back(p);
    *chirp = (const T*)p.chirp;
    *bspec = (const T*)p.bspec;
    return BDSP_OK;
}

template <typename T>
int fft_any_len(T* a, T* b, size_t n, size_t batch, bool inverse, unsigned flags, T in_scale,
                int window_id, T window_alpha, bool* in_b, hipStream_t s)
{
    *in_b = false;
    static const bool no_mixed = lab_flag("BDSP_FFT_NO_MIXED_RADIX");
    if (!no_mixed && mr_supported<T>(n) && (batch <= 65535 || mr_resident<T>(n))) {
        // 2,3,5,7-smooth lengths: mixed-radix Stockham (mixed_radix.hip).  The four-step form goes a -> b -> a; the
        // workgroup-resident form runs in place unless the output has another shape than the input.
        const bool reshaping = (flags & (FFT_IN_REAL | BDSP_FFT_MAGNITUDE | FFT_OUT_REAL)) != 0;
        const bool resident = mr_resident<T>(n);
        // (three global passes -- lengths whose four-step tile would be narrow -- end in the scratch buffer: a -> b -> a -> b)
        T* out = ((resident && reshaping) || mr_passes<T>(n) == 3) ? b : a;
        *in_b = out == b;
        return mr_fft<T>(a, out, b, n, batch, inverse, flags, in_scale, window_id, window_alpha, s);
    }
    // the Bluestein epilogue writes the result back over the (consumed) input buffer
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    if (m < 512) m = 512; // the single-kernel path starts at 512-point workgroup transforms
    if (m > (size_t(1) << 30)) { set_last_error("Bluestein length above 2^30"); return BDSP_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lk(g_bs_mu); // plans cannot be evicted between lookup and launch
    const T *chirp = nullptr, *bspec = nullptr;
    BDSP_TRY(bs_plan<T>(n, m, inverse, s, &chirp, &bspec));
    if (m <= 4096 && flags == 0 && window_id < 0 && in_scale == (T)1 && n >= 2)
        return bs_fused<T>(a, a, chirp, bspec, n, m, batch, s); // one kernel, in place
    WsBlock wa, wt;
    BDSP_TRY(wa.alloc(sizeof(T) * 2 * m * batch, s));
    BDSP_TRY(wt.alloc(sizeof(T) * 2 * m * batch, s));
    const bool out_div = (flags & FFT_WINDOW_OUT_DIV) != 0;
    BDSP_TRY(bs_pre<T>(a, wa.as<T>(), chirp, n, m, batch, (flags & FFT_IN_REAL) != 0, in_scale,
                       (flags & BDSP_FFT_SHIFT_IN) ? n / 2 : 0, out_div ? -1 : window_id, window_alpha, s));
    bool ra = false;
    BDSP_TRY(fft_two_buffers<T>(wa.as<T>(), wt.as<T>(), m, batch, false, 0, (T)1, -1, (T)0, &ra, s));
    T* spec = ra ? wt.as<T>() : wa.as<T>();
    T* scr = ra ? wa.as<T>() : wt.as<T>();
    BDSP_TRY(mul_bcast<T>(spec, bspec, m, batch, (T)1 / (T)m, s));
    bool rc = false;
    BDSP_TRY(fft_two_buffers<T>(spec, scr, m, batch, true, 0, (T)1, -1, (T)0, &rc, s));
    const T* conv = rc ? scr : spec;
    const int out_kind = (flags & BDSP_FFT_MAGNITUDE) ? 2 : ((flags & FFT_OUT_REAL) ? 1 : 0);
    return bs_post<T>(conv, a, chirp, n, m, batch, out_kind, (flags & BDSP_FFT_SHIFT_OUT) ? n - n / 2 : 0,
                      out_div ? window_id : -1, window_alpha, s);
}

template <typename T> constexpr int elem_of() { return sizeof(T) == 8; }

int check_device()
{
    int c = device_ready();
    return c;
}

// ----------------------------------------------------------------------------------------------
// B1 implementations
// ----------------------------------------------------------------------------------------------
// Small host slices travel through a per-thread PINNED staging buffer (hipHostMalloc, grown geometrically; signals of at
// most B1_STAGE_MAX bytes), and where the kernels allow it they are not copied by the GPU at all: the first kernel reads the
// staging buffer over PCIe and the last one writes it.  *Measured* (tools/b1_crossover.py, profiles/r06_b1_crossover.txt; wall
// time of one call, f32; "stage k" = LAB build with BDSP_B1_STAGE=k):
//   * stage 0, pageable copies (rounds 1-5): a floor of 34-47 us from 4096 to 16384 points -- launch, two copy packets and the
//     completion wait, not the pinning of the caller's pages;
//   * stage 1, pinned copies: 0-5 us better on a complex vector, 15-40 us on real / f64 ones up to 1 MiB, and WORSE from
//     2 MiB on, where the runtime pins the caller's pages itself and two host memcpys cost more than that (2^18 points
//     122 -> 272 us);
//   * stage 2, kernels read the stage, the result is copied down: another 2-7 us;
//   * stage 3, no copy packet at all, is what moves the floor: fft 4096 points 36 -> 20 us, 8192 41 -> 26, 16384 47 -> 29,
//     65536 81 -> 58, 131072 (1 MiB) 147 -> 137; convolve_vector 5001 x 5 taps 45 -> 29, 16384 x 1024 56 -> 39, 131072 x 1024
//     166 -> 122; real data 131072 x 5 taps 96 -> 71.  Above 1 MiB every staged form loses to the pageable path.
// The product: stage 3 for power-of-two and smooth (mixed-radix, up to two passes) transforms and for the fused block kernel,
// stage 1 for the rest (chirp-z lengths, long or direct-form filters), stage 0 above B1_STAGE_MAX.
namespace {
constexpr size_t B1_STAGE_MAX = size_t(1) << 20;        // pinned staging up to here (bytes per call), pageable copies above
struct B1Stage {
    char* p = nullptr;
    size_t cap = 0;
    int dev = -1;
    // no destructor work: at thread exit the HIP runtime may already be gone
    char* get(size_t bytes)
    {
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) return nullptr;
        if (p && dev == device && bytes <= cap) return p;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } // every B1 call ends synchronised: nothing reads it any more
        size_t ncap = size_t(64) << 10;
        while (ncap < bytes) ncap <<= 1;
        if (hipHostMalloc((void**)&p, ncap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return nullptr; }
        cap = ncap;
        dev = device;
        return p;
    }
};
thread_local B1Stage t_b1stage;

// The size policy of the B1 boundary (bdsp_hip_b1_policy_get/_set, include/basic_dsp_hip.h).  Defaults: the crossovers
// measured on the MI355X box against one host core (profiles/r06_b1_crossover.txt); 0 = accept everything.
// profiles/r06_b1_crossover.txt, one host core against the MI355X round trip:
//   fft: the shortest power of two whose B1 call takes at most HALF the faster CPU row's time (numpy / pocketfft; the factor
//   of two is the allowance for rustfft's SIMD butterflies, which this image cannot run) -- f32 8192 points (25.5 against 92 us;
//   4096: 19.9 against 22.8), f64 16384 points (38 against 88 us; 8192: 30 against 43.5);
//   convolution: where the reference's direct form (the port's scalar loop: 0.45-0.95 ns per point and tap) ties with the round
//   trip (27-34 us + 0.3-0.8 ns per point) -- complex f32 16384 points x 3 taps (37.6 against 36.2 us) or 8000 x 5, real f32
//   20000 x 3; complex f64 21000 x 3 (work 63k), real f64 45000 x 3 (135k): one figure per precision, between the two.
constexpr size_t B1_DEFAULT_FFT_MIN_LEN_F32 = 2 * 8192, B1_DEFAULT_FFT_MIN_LEN_F64 = 2 * 16384; // scalars
constexpr size_t B1_DEFAULT_CONV_MIN_WORK_F32 = 65536, B1_DEFAULT_CONV_MIN_WORK_F64 = 98304;    // points x taps
constexpr int B1_POLICY_KEYS = 4;
std::atomic<size_t> g_b1_policy[B1_POLICY_KEYS] = {
    {B1_DEFAULT_FFT_MIN_LEN_F32}, {B1_DEFAULT_FFT_MIN_LEN_F64}, {B1_DEFAULT_CONV_MIN_WORK_F32}, {B1_DEFAULT_CONV_MIN_WORK_F64}};

int b1_stage_mode(size_t bytes, bool zero_copy_ok)
{
    static const char* force = lab_env("BDSP_B1_STAGE");
    if (force) return bytes <= (size_t(64) << 20) ? atoi(force) : 0;
    if (bytes > B1_STAGE_MAX) return 0;
    return zero_copy_ok ? 3 : 1;
}
} // namespace

template <typename T>
int b1_fft(int is_complex, T* signal, size_t len, int inverse)
{
    if (!is_complex) { set_last_error("real fft isn't supported, call is_supported_fft_len first"); return BDSP_ERR_UNSUPPORTED; }
    BDSP_TRY(check_device());
    size_t points = len / 2;
    if (points == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    const size_t bytes = sizeof(T) * len;
    const bool pow2 = is_pow2(points);
    const int trips = pow2 ? fft_pow2_plain_trips<T>(points) : 0; // 1 = one workgroup-resident kernel
    // smooth lengths (mixed radix, resident or four-step) take separate input / output pointers too; chirp-z lengths and
    // the three-pass smooth ones work in their device buffers
    const bool smooth = !pow2 && mr_supported<T>(points) && mr_passes<T>(points) < 3;
    int mode = b1_stage_mode(bytes, pow2 || smooth);
    char* stage = mode ? t_b1stage.get(bytes) : nullptr;
    if (!stage) mode = 0;
    if (mode >= 2 && !(pow2 || smooth)) mode = 1;
    WsBlock a, b;
    if (mode >= 2) {
        // the first pass reads the pinned stage over PCIe; with mode 3 the last pass writes it: stage -> a [-> b] -> stage
        memcpy(stage, signal, bytes);
        if (smooth) {
            if (mode == 2 || !mr_resident<T>(points)) BDSP_TRY(a.alloc(bytes, s));
            T* dst = mode == 3 ? (T*)stage : a.as<T>();
            if (mode == 2 && !mr_resident<T>(points)) BDSP_TRY(b.alloc(bytes, s));
            BDSP_TRY(mr_fft<T>((const T*)stage, dst, mode == 3 ? a.as<T>() : b.as<T>(), points, 1, inverse != 0, 0, (T)1, -1, (T)0, s));
            if (mode == 2) BDSP_HIP_TRY(hipMemcpyAsync(stage, a.p, bytes, hipMemcpyDeviceToHost, s));
        } else {
            if (trips >= 2 || points > 4096) BDSP_TRY(a.alloc(bytes, s)); // (8192 f32: one kernel, but the two-pass plan is its fallback)
            if (trips >= 3 || mode == 2) BDSP_TRY(b.alloc(bytes, s));
            FftIo<T> io{};
            io.n = points; io.flags = 0; io.in_scale = (T)1; io.window_id = -1; io.window_alpha = (T)0;
            io.in_stride = points; io.out_stride = points;
            io.in = stage;
            io.out = mode == 3 ? (void*)stage : b.p;
            BDSP_TRY(fft_pow2<T>(io, a.as<T>(), trips >= 3 ? b.as<T>() : nullptr, 1, inverse != 0, s));
            if (mode == 2) BDSP_HIP_TRY(hipMemcpyAsync(stage, b.p, bytes, hipMemcpyDeviceToHost, s));
        }
        BDSP_HIP_TRY(hipStreamSynchronize(s));
        memcpy(signal, stage, bytes);
        return BDSP_OK;
    }
    BDSP_TRY(a.alloc(bytes, s));
    if (trips != 1) BDSP_TRY(b.alloc(bytes, s)); // (a single-kernel length runs in place: no second block)
    bool in_b = false;
    if (mode == 1) memcpy(stage, signal, bytes);
    BDSP_HIP_TRY(hipMemcpyAsync(a.p, mode == 1 ? (const void*)stage : (const void*)signal, bytes, hipMemcpyHostToDevice, s));
    BDSP_TRY(fft_two_buffers<T>(a.as<T>(), b.as<T>(), points, 1, inverse != 0, 0, (T)1, -1, (T)0, &in_b, s));
    BDSP_HIP_TRY(hipMemcpyAsync(mode ? (void*)stage : (void*)signal, in_b ? b.p : a.p, bytes, hipMemcpyDeviceToHost, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    if (mode) memcpy(signal, stage, bytes);
    return BDSP_OK;
}

// Long filters (more than 1025 taps): overlap-save with blocks of L = next_pow2(4 (M-1)) points -- the
// reference's own block length rule (convolution.rs:296-302) -- on the batched multi-pass FFT:
//   circular extension of the signal -> FFT_L of the overlapping windows (row stride V = L-(M-1)) ->
//   x H / L -> IFFT_L -> valid parts to the output,     O(N log L) instead of the direct form's O(N M).
// The windows of one chunk (at most ~1 GiB of spectra) are transformed together.
template <typename T>
int conv_long_dev(const T* in, T* out, size_t points, const T* taps, size_t ntaps, hipStream_t s)
{
    size_t L = 8192;
    while (L < 4 * (ntaps - 1)) L <<= 1;
    const size_t ov = ntaps - 1, V = L - ov;
    const size_t nb_total = (points + V - 1) / V;
    size_t nb_chunk = (size_t(1) << 27) / L; // 2^27 complex points of spectra per chunk
    if (nb_chunk < 1) nb_chunk = 1;
    if (nb_chunk > nb_total) nb_chunk = nb_total;
    WsBlock hb, hs2, xe, z1, z2;
    BDSP_TRY(hb.alloc(sizeof(T) * 2 * L, s));
    BDSP_TRY(hs2.alloc(sizeof(T) * 2 * L, s));
    BDSP_TRY(xe.alloc(sizeof(T) * 2 * ((nb_chunk - 1) * V + L), s));
    BDSP_TRY(z1.alloc(sizeof(T) * 2 * L * nb_chunk, s));
    BDSP_TRY(z2.alloc(sizeof(T) * 2 * L * nb_chunk, s));
    // H = FFT_L(zero-padded taps)
    BDSP_HIP_TRY(hipMemsetAsync(hb.p, 0, sizeof(T) * 2 * L, s));
    BDSP_HIP_TRY(hipMemcpyAsync(hb.p, taps, sizeof(T) * 2 * ntaps, hipMemcpyDeviceToDevice, s));
    bool rh = false;
    BDSP_TRY(fft_two_buffers<T>(hb.as<T>(), hs2.as<T>(), L, 1, false, 0, (T)1, -1, (T)0, &rh, s));
    const T* H = rh ? hs2.as<T>() : hb.as<T>();
    for (size_t b0 = 0; b0 < nb_total; b0 += nb_chunk) {
        const size_t nb = nb_total - b0 < nb_chunk ? nb_total - b0 : nb_chunk;
        // block b reads x[(b V - floor(M/2) + i) mod N], i < L, and yields outputs b V .. b V + V - 1
        BDSP_TRY(rg_wrap_copy<T>(in, xe.as<T>(), points, 2, (nb - 1) * V + L,
                                 (long long)(b0 * V) - (long long)(ntaps / 2), s));
        FftIo<T> io{};
        io.n = L; io.in = xe.p; io.in_stride = V; io.out_stride = L; io.flags = 0;
        io.in_scale = (T)1; io.window_id = -1; io.window_alpha = (T)0;
        bool r1 = false;
        if (fft_pow2_passes<T>(L) < 3) { io.out = z1.p; BDSP_TRY(fft_pow2<T>(io, z2.as<T>(), nullptr, nb, false, s)); }
        else { io.out = z2.p; r1 = true; BDSP_TRY(fft_pow2<T>(io, z2.as<T>(), z1.as<T>(), nb, false, s)); }
        T* spec = r1 ? z2.as<T>() : z1.as<T>();
        T* scr = r1 ? z1.as<T>() : z2.as<T>();
        BDSP_TRY(mul_bcast<T>(spec, H, L, nb, (T)1 / (T)L, s));
        bool r2 = false;
        BDSP_TRY(fft_two_buffers<T>(spec, scr, L, nb, true, 0, (T)1, -1, (T)0, &r2, s));
        BDSP_TRY(scatter_valid<T>(r2 ? scr : spec, out, L, ov, V, b0 * V, nb, points, s));
    }
    return BDSP_OK;
}

// The fused 4096-point block kernel takes filters of up to FUSED_MAX_TAPS taps: with M-1 discarded points
// per block its cost grows like 4096/(4096-(M-1)) -- 2x at 2049 taps, 4x at 3073 -- and stays below the
// generic long-filter path (about 6x the 1024-tap time) up to there.
constexpr size_t FUSED_MAX_TAPS = 3073;

// complex convolution of device vectors; picks the block kernel whenever it applies
template <typename T>
int conv_complex_dev(const T* in, T* out, size_t points, size_t batch, const T* taps, size_t ntaps,
                     hipStream_t s)
{
    if (ntaps >= 1 && ntaps <= FUSED_MAX_TAPS && ntaps <= points && points >= 1)
        return convolve_overlap_save<T>(in, out, points, batch, taps, ntaps, -(long long)(ntaps / 2), 0, 0,
                                        nullptr, nullptr, s);
    if (ntaps > FUSED_MAX_TAPS && ntaps <= points && 4 * (ntaps - 1) <= (size_t(1) << 24)) {
        for (size_t v = 0; v < batch; ++v)
            BDSP_TRY(conv_long_dev<T>(in + 2 * points * v, out + 2 * points * v, points, taps, ntaps, s));
        return BDSP_OK;
    }
    return convolve_direct<T>(in, out, points, batch, taps, ntaps, true, s);
}

// real data with real taps: the block kernel packs two real blocks into one complex transform pair
template <typename T>
int conv_real_dev(const T* in, T* out, size_t points, const T* taps, size_t ntaps, hipStream_t s, size_t batch = 1)
{
    if (ntaps > FUSED_MAX_TAPS && ntaps <= points && 4 * (ntaps - 1) <= (size_t(1) << 24)) {
        // long real filters: complexify, run the long-filter path, keep the real parts
        WsBlock xc, yc, hc2;
        BDSP_TRY(xc.alloc(sizeof(T) * 2 * points * batch, s));
        BDSP_TRY(yc.alloc(sizeof(T) * 2 * points * batch, s));
        BDSP_TRY(hc2.alloc(sizeof(T) * 2 * ntaps, s));
        BDSP_TRY(rg_zero_interleave<T>(in, xc.as<T>(), points * batch, 1, 2, s));
        BDSP_TRY(rg_zero_interleave<T>(taps, hc2.as<T>(), ntaps, 1, 2, s));
        BDSP_TRY(conv_complex_dev<T>(xc.as<T>(), yc.as<T>(), points, batch, hc2.as<T>(), ntaps, s));
        return ew_complex_to_real<T>(yc.as<T>(), out, 2 * points * batch, 2, s);
    }
    if (!(ntaps >= 1 && ntaps <= FUSED_MAX_TAPS && ntaps <= points))
        return convolve_direct<T>(in, out, points, batch, taps, ntaps, false, s);
    static const bool real_prep = lab_flag("BDSP_CONV_REAL_PREP"); // (LAB: spectrum in its own launch, round 5)
    if (!real_prep) // one launch: the block kernel reads the real taps and transforms them itself
        return conv_run_blocks<T>(in, out, points, batch, taps, ntaps, -(long long)(ntaps / 2), 0, 0, nullptr, s, true, true);
    WsBlock hc, hsb;
    BDSP_TRY(hc.alloc(sizeof(T) * 2 * ntaps, s));
    BDSP_TRY(hsb.alloc(sizeof(T) * 2 * conv_fft_len(ntaps), s));
    BDSP_TRY(rg_zero_interleave<T>(taps, hc.as<T>(), ntaps, 1, 2, s));
    BDSP_TRY(conv_prepare_spectrum<T>(hc.as<T>(), ntaps, nullptr, hsb.as<T>(), s));
    return conv_run_blocks<T>(in, out, points, batch, hsb.as<T>(), ntaps, -(long long)(ntaps / 2), 0, 0, nullptr, s, true);
}

// gpu_convolve_vector on a long complex vector: the PCIe transfers dominate (128 MiB each way for 16M f32 points
// against 0.1 ms of kernel time), so they are pipelined.  The input goes up in chunks on its own stream; as soon as
// a chunk has landed the blocks whose 4096-point windows it completes run on the compute stream; a helper thread
// brings their outputs down on a third stream while the next chunk is still going up -- PCIe is full duplex.
// Block 0 and the last blocks read across the wrap-around point and run when the whole vector is resident.
// *Measured* 16M f32 points x 1024 taps: 4.88 -> 3.94 -> 3.36 ms (the two directions share 68-80 GB/s on this host;
// registering the caller's pages first changed nothing).
//
// The two transfer streams and the stage events are created once per calling thread and device and reused by
// every later call (the OpenCL backend this replaces built a context, a queue and a plan per call,
// ocl/mod.rs:301-357); the downloader waits on a condition variable, not a spin.
namespace {
constexpr int B1_STAGES = 8;
struct B1Pipe {
    int dev = -1;
    hipStream_t up = nullptr, down = nullptr;
    hipEvent_t landed[B1_STAGES] = {}, computed[B1_STAGES + 1] = {}, ready = nullptr;
    bool ok = false;
    int init(int device)
    {
        if (ok && dev == device) return BDSP_OK;
        release();
        dev = device;
        BDSP_HIP_TRY(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        BDSP_HIP_TRY(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
        for (auto& e : landed) BDSP_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : computed) BDSP_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        BDSP_HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        ok = true;
        return BDSP_OK;
    }
    void release()
    {
        for (auto& e : landed) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (auto& e : computed) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (ready) { (void)hipEventDestroy(ready); ready = nullptr; }
        if (up) { (void)hipStreamDestroy(up); up = nullptr; }
        if (down) { (void)hipStreamDestroy(down); down = nullptr; }
        ok = false;
    }
    // no destructor work: at thread exit the HIP runtime may already be gone, and the handles die with the process
};
thread_local B1Pipe t_b1pipe;
} // namespace

template <typename T>
int b1_convolve_pipelined(const T* src, T* dst, size_t points, const T* imp, size_t ntaps)
{
    hipStream_t s = lib_stream();
    const size_t L = conv_fft_len(ntaps);
    const size_t V = conv_block_step<T>(points, ntaps, false); // the block kernel's step (conv.hip / conv_v2.hip)
    const size_t nb = (points + V - 1) / V;
    const long long in_off = -(long long)(ntaps / 2);
    constexpr int K = B1_STAGES;
    int dev = 0;
    BDSP_HIP_TRY(hipGetDevice(&dev));
    B1Pipe& pipe = t_b1pipe;
    {
        int c = pipe.init(dev);
        if (c != BDSP_OK) { pipe.release(); return c; }
    }
    WsBlock dx, dy, dh;
    BDSP_TRY(dx.alloc(sizeof(T) * 2 * points, s));
    BDSP_TRY(dy.alloc(sizeof(T) * 2 * points, s));
    BDSP_TRY(dh.alloc(sizeof(T) * 2 * ntaps, s));
    BDSP_HIP_TRY(hipMemcpyAsync(dh.p, imp, sizeof(T) * 2 * ntaps, hipMemcpyHostToDevice, s));
    // the workspace blocks come from the library stream's cache: earlier (asynchronous) work on that stream may still
    // be using them, so the transfer streams start behind everything queued on it so far
    BDSP_HIP_TRY(hipEventRecord(pipe.ready, s));
    BDSP_HIP_TRY(hipStreamWaitEvent(pipe.up, pipe.ready, 0));
    BDSP_HIP_TRY(hipStreamWaitEvent(pipe.down, pipe.ready, 0));
    // the block kernel transforms the taps itself (as convolve_signal on a device vector does, so the two paths stay
    // bit-identical)
    const T* hsp = dh.as<T>();
    const size_t ch = ((points + K - 1) / K + 1023) & ~(size_t)1023;
    struct Piece { size_t first, count; bool valid; };
    Piece pieces[K], tail[2] = {Piece{0, 0, false}, Piece{0, 0, false}}; // output ranges per stage
    for (int k = 0; k < K; ++k) pieces[k] = Piece{0, 0, false};
    std::mutex mu;
    std::condition_variable cv;
    int stages_recorded = 0; // stage k may be waited for once its event has been recorded (guarded by mu)
    int drc = BDSP_OK;
    auto publish = [&](int n) {
        { std::lock_guard<std::mutex> lk(mu); stages_recorded = n; }
        cv.notify_one();
    };
    // downloads run on their own thread and stream, in step with the compute stream (a device-to-host copy into
    // pageable memory blocks its caller, so it cannot share the thread that feeds the uploads)
    std::thread downloader([&] {
        if (hipSetDevice(dev) != hipSuccess) { drc = BDSP_ERR_HIP; return; }
        auto fetch = [&](const Piece& p) {
            if (!p.valid || p.count == 0) return;
            if (hipMemcpyAsync(dst + 2 * p.first, dy.as<T>() + 2 * p.first, sizeof(T) * 2 * p.count, hipMemcpyDeviceToHost, pipe.down) != hipSuccess) drc = BDSP_ERR_HIP;
        };
        for (int k = 0; k <= K; ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stages_recorded > k; });
            }
            if (hipStreamWaitEvent(pipe.down, pipe.computed[k], 0) != hipSuccess) drc = BDSP_ERR_HIP;
            if (k < K) fetch(pieces[k]);
            else { fetch(tail[0]); fetch(tail[1]); }
        }
        if (hipStreamSynchronize(pipe.down) != hipSuccess) drc = BDSP_ERR_HIP;
    });
    int rc = BDSP_OK;
    auto hip_ok = [&](hipError_t e) { if (e != hipSuccess && rc == BDSP_OK) { set_last_error(hipGetErrorString(e)); rc = BDSP_ERR_HIP; } };
    // blocks whose window starts before x[0] (block 0 always; more of them when M/2 exceeds the block step) read the
    // END of the vector through the wrap-around: deferred until everything is resident
    size_t head = (size_t)((-in_off + (long long)V - 1) / (long long)V);
    if (head > nb) head = nb;
    const size_t nhead = head < 1 ? (nb < 1 ? nb : 1) : head; // the deferred head blocks are [0, nhead)
    size_t next_block = nhead;
    for (int k = 0; k < K; ++k) {
        const size_t c0 = (size_t)k * ch, c1 = c0 + ch < points ? c0 + ch : points;
        if (c0 < points && rc == BDSP_OK) {
            hip_ok(hipMemcpyAsync(dx.as<T>() + 2 * c0, src + 2 * c0, sizeof(T) * 2 * (c1 - c0), hipMemcpyHostToDevice, pipe.up));
            hip_ok(hipEventRecord(pipe.landed[k], pipe.up));
            hip_ok(hipStreamWaitEvent(s, pipe.landed[k], 0));
            // blocks whose window [b V + in_off, + L) lies inside the uploaded prefix [0, c1)
            size_t bend = next_block;
            while (bend < nb && (long long)(bend * V) + in_off + (long long)L <= (long long)c1) ++bend;
            if (bend > next_block && rc == BDSP_OK) {
                rc = conv_run_blocks<T>(dx.as<T>(), dy.as<T>(), points, 1, hsp, ntaps, in_off + (long long)(next_block * V),
                                        (long long)(next_block * V), bend - next_block, nullptr, s, false, true);
                const size_t o1 = bend * V < points ? bend * V : points;
                if (rc == BDSP_OK) pieces[k] = Piece{next_block * V, o1 - next_block * V, true};
                next_block = bend;
            }
        }
        hip_ok(hipEventRecord(pipe.computed[k], s));
        publish(k + 1);
    }
    // the wrap-around blocks: the deferred head [0, nhead) and everything from next_block on
    if (rc == BDSP_OK) {
        rc = conv_run_blocks<T>(dx.as<T>(), dy.as<T>(), points, 1, hsp, ntaps, in_off, 0, nhead, nullptr, s, false, true);
        if (rc == BDSP_OK) tail[0] = Piece{0, nhead * V < points ? nhead * V : points, true};
        if (rc == BDSP_OK && next_block < nb) {
            rc = conv_run_blocks<T>(dx.as<T>(), dy.as<T>(), points, 1, hsp, ntaps, in_off + (long long)(next_block * V),
                                    (long long)(next_block * V), nb - next_block, nullptr, s, false, true);
            if (rc == BDSP_OK) tail[1] = Piece{next_block * V, points - next_block * V, true};
        }
    }
    hip_ok(hipEventRecord(pipe.computed[K], s));
    publish(K + 1);
    downloader.join();
    hip_ok(hipStreamSynchronize(pipe.up));
    hip_ok(hipStreamSynchronize(s));
    return rc != BDSP_OK ? rc : drc;
}

template <typename T>
int b1_convolve(int is_complex, const T* src, size_t src_len, T* dst, size_t dst_len, const T* imp,
                size_t imp_len, size_t* range_start, size_t* range_end)
{
    const size_t elem = is_complex ? 2 : 1;
    const size_t points = src_len / elem, ntaps = imp_len / elem;
    if (points == 0 || ntaps == 0 || ntaps > points || dst_len < src_len) return 0; // None
    // Size policy: None where the host round trip loses to what the reference runs next.  That is only ever its direct
    // form, convolve_signal_scalar -- real data, at most 15 tap scalars, or a vector no longer than ten impulse responses
    // (convolution.rs:530-541) -- and only below the measured points x taps product.  A complex vector the reference
    // would send into its own overlap_discard (with the O(N M / 2) scalar tail, :388-399) is never declined.
    const bool falls_to_scalar = !is_complex || imp_len <= 15 || src_len <= 10 * imp_len;
    if (falls_to_scalar && points * ntaps < g_b1_policy[sizeof(T) == 8 ? BDSP_B1_CONV_MIN_WORK_F64 : BDSP_B1_CONV_MIN_WORK_F32].load())
        return 0;
    int c = check_device();
    if (c != BDSP_OK) return c;
    hipStream_t s = lib_stream();
    static const bool no_pipeline = lab_flag("BDSP_B1_NO_PIPELINE");
    if (is_complex && !no_pipeline && points >= (size_t(1) << 20) && ntaps <= FUSED_MAX_TAPS && points < (size_t(1) << 31)) {
        BDSP_TRY(b1_convolve_pipelined<T>(src, dst, points, imp, ntaps));
        if (range_start) *range_start = 0;
        if (range_end) *range_end = src_len;
        return 1;
    }
    // signal and taps travel as ONE upload: [src | pad to 256 B | taps] in the staging buffer and in the device block
    const size_t sbytes = sizeof(T) * src_len, ibytes = sizeof(T) * imp_len;
    const size_t ioff = (sbytes + 255) & ~(size_t)255;
    const bool block_kernel = ntaps <= FUSED_MAX_TAPS; // the fused kernel reads every input point ~1.3 times: fit for PCIe reads
    int mode = b1_stage_mode(sbytes, block_kernel); // (the limit counts the signal; the taps ride along)
    if (mode >= 2 && !block_kernel) mode = 1;
    char* stage = mode ? t_b1stage.get(mode == 3 ? 2 * ioff + ibytes + 256 : ioff + ibytes) : nullptr;
    if (!stage) mode = 0;
    WsBlock dx, dy;
    if (mode < 2) BDSP_TRY(dx.alloc(ioff + ibytes, s));
    if (mode < 3) BDSP_TRY(dy.alloc(sbytes, s));
    if (mode) {
        memcpy(stage, src, sbytes);
        memcpy(stage + ioff, imp, ibytes);
    }
    if (mode == 1) BDSP_HIP_TRY(hipMemcpyAsync(dx.p, stage, ioff + ibytes, hipMemcpyHostToDevice, s));
    else if (mode == 0) {
        BDSP_HIP_TRY(hipMemcpyAsync(dx.p, src, sbytes, hipMemcpyHostToDevice, s));
        BDSP_HIP_TRY(hipMemcpyAsync((char*)dx.p + ioff, imp, ibytes, hipMemcpyHostToDevice, s));
    }
    const T* din = mode >= 2 ? (const T*)stage : dx.as<T>();
    const T* dtaps = (const T*)((const char*)din + ioff);
    T* dout = mode == 3 ? (T*)(stage + ioff + ((ibytes + 255) & ~(size_t)255)) : dy.as<T>();
    if (is_complex) BDSP_TRY(conv_complex_dev<T>(din, dout, points, 1, dtaps, ntaps, s));
    else BDSP_TRY(conv_real_dev<T>(din, dout, points, dtaps, ntaps, s));
    if (mode < 3) BDSP_HIP_TRY(hipMemcpyAsync(mode ? (void*)stage : (void*)dst, dy.p, sbytes, hipMemcpyDeviceToHost, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    if (mode) memcpy(dst, mode == 3 ? (const char*)dout : stage, sbytes);
    if (range_start) *range_start = 0;
    if (range_end) *range_end = src_len;
    return 1; // Some(0..src_len)
}

// GpuSupport::overlap_discard, any power-of-two fft_len (ocl/mod.rs:361-520): batched block FFTs
// over overlapping windows of the uploaded signal (in_stride = step), one broadcast spectrum
// product, batched inverse FFTs, one scatter of the valid parts.
template <typename T>
size_t b1_overlap_discard(T* x_time, size_t x_len, T* tmp, size_t tmp_len, const T* h_freq,
                          size_t h_len, size_t imp_len, size_t step_size)
{
    const size_t l = h_len / 2, xp = x_len / 2, m = imp_len / 2, step = step_size / 2;
    set_last_error(""); // the error channel of this call: the return value is a position, never an error code
    if (l == 0 || !is_pow2(l) || step == 0 || xp < l || tmp_len < h_len || m == 0 || m > l) {
        set_last_error("overlap_discard: unsupported argument combination");
        return 0;
    }
    // (the return value is a position: a failure is only visible through last_error, so every failing path leaves a
    // message there -- also the ones whose callee returned a bare code)
    auto failed = [](int code) -> size_t {
        if (bdsp_hip_last_error()[0] == '\0') set_last_error("overlap_discard: backend failure, code " + std::to_string(code));
        return 0;
    };
    if (const int c = check_device(); c != BDSP_OK) return failed(c);
    // blocks at positions 0, step, 2*step, ... : the first one always, then while pos + l < xp
    size_t nb = 1;
    while (nb * step + l < xp) ++nb;
    hipStream_t s = lib_stream();
    auto run = [&]() -> int {
        WsBlock dx, dz, dz2, dh;
        BDSP_TRY(dx.alloc(sizeof(T) * x_len, s));
        BDSP_TRY(dz.alloc(sizeof(T) * 2 * l * nb, s));
        BDSP_TRY(dz2.alloc(sizeof(T) * 2 * l * nb, s));
        BDSP_TRY(dh.alloc(sizeof(T) * h_len, s));
        BDSP_HIP_TRY(hipMemcpyAsync(dx.p, x_time, sizeof(T) * x_len, hipMemcpyHostToDevice, s));
        BDSP_HIP_TRY(hipMemcpyAsync(dh.p, h_freq, sizeof(T) * h_len, hipMemcpyHostToDevice, s));
        // forward FFT of the overlapping windows straight from the signal (row stride = step
        // points), never touching the signal buffer itself: in -> sa [-> sb] -> out
        bool r1 = false;
        {
            FftIo<T> io{};
            io.n = l; io.in = dx.p; io.in_stride = step; io.out_stride = l; io.flags = 0;
            io.in_scale = (T)1; io.window_id = -1; io.window_alpha = (T)0;
            if (l <= 4096) { io.out = dz.p; BDSP_TRY(fft_pow2<T>(io, nullptr, nullptr, nb, false, s)); }
            else if (fft_pow2_passes<T>(l) < 3) { io.out = dz.p; BDSP_TRY(fft_pow2<T>(io, dz2.as<T>(), nullptr, nb, false, s)); }
            else { io.out = dz2.p; r1 = true; BDSP_TRY(fft_pow2<T>(io, dz2.as<T>(), dz.as<T>(), nb, false, s)); }
        }
        T* spec = r1 ? dz2.as<T>() : dz.as<T>();
        T* scr = r1 ? dz.as<T>() : dz2.as<T>();
        BDSP_TRY(mul_bcast<T>(spec, dh.as<T>(), l, nb, (T)1 / (T)l, s));
        bool r2 = false;
        BDSP_TRY(fft_two_buffers<T>(spec, scr, l, nb, true, 0, (T)1, -1, (T)0, &r2, s));
        T* z = r2 ? scr : spec;
        // head computed by the caller: tmp[0 .. imp_len/2) -> x_time[0 .. imp_len/2)
        BDSP_HIP_TRY(hipMemcpyAsync(dx.p, tmp, sizeof(T) * (imp_len / 2), hipMemcpyHostToDevice, s));
        // valid parts of all blocks but the last: z[b][m-1 .. l) -> x[b*step + m/2 ..]
        if (nb > 1) BDSP_TRY(scatter_valid<T>(z, dx.as<T>(), l, m - 1, step, m / 2, nb - 1, xp, s));
        BDSP_HIP_TRY(hipMemcpyAsync(x_time, dx.p, sizeof(T) * x_len, hipMemcpyDeviceToHost, s));
        BDSP_HIP_TRY(hipMemcpyAsync(tmp, z + 2 * l * (nb - 1), sizeof(T) * 2 * l, hipMemcpyDeviceToHost, s));
        BDSP_HIP_TRY(hipStreamSynchronize(s));
        return BDSP_OK;
    };
    if (const int c = run(); c != BDSP_OK) return failed(c);
    return nb * step_size;
}

// ----------------------------------------------------------------------------------------------
// B2: HBM-resident vector behind the facade handle
// ----------------------------------------------------------------------------------------------
// Every buffer trade (and every reallocation) of a handle XORs a token into this word; two trades of the same
// vector cancel.  A captured HIP graph records device addresses, so a capture may only contain sequences that
// leave every vector's (live, trade) pair as they found it: bdsp_hip_capture_end compares the word.
static thread_local unsigned long long t_buffer_moves = 0; // (per thread: a capture only answers for its own thread's calls)
static std::atomic<unsigned long long> g_realloc_ticket{1};

template <typename T>
struct DevVec {
    T* data = nullptr; // live buffer
    T* buf = nullptr;  // trade buffer (reference: SingleBuffer, support_std.rs:78-124)
    size_t cap = 0;    // scalars each buffer can hold
    size_t valid_len = 0;
    T delta = (T)1;
    bool complex_ = false;
    bool freq = false;
    std::vector<T> mirror; // host copy handed out by data32/data64

    size_t points() const { return complex_ ? valid_len / 2 : valid_len; }
    bool erroneous() const { return valid_len == 0 && std::isnan((double)delta); }
    void poison() { valid_len = 0; delta = std::numeric_limits<T>::quiet_NaN(); } // mod.rs:226-229
    void trade() { T* t = data; data = buf; buf = t; t_buffer_moves ^= (unsigned long long)(uintptr_t)this * 0x9E3779B97F4A7C15ull; }

    int reserve(size_t scalars)
    {
        if (scalars <= cap) return BDSP_OK;
        hipStream_t s = lib_stream();
        size_t ncap = scalars + scalars / 8 + 64;
        void *nd = nullptr, *nb = nullptr;
        BDSP_TRY(ws_alloc(&nd, sizeof(T) * ncap, s));
        int c = ws_alloc(&nb, sizeof(T) * ncap, s);
        if (c != BDSP_OK) { ws_free(nd, s); return c; }
        if (data && valid_len)
            BDSP_HIP_TRY(hipMemcpyAsync(nd, data, sizeof(T) * valid_len, hipMemcpyDeviceToDevice, s));
        if (data) ws_free(data, s);
        if (buf) ws_free(buf, s);
        data = (T*)nd;
        buf = (T*)nb;
        cap = ncap;
        t_buffer_moves ^= g_realloc_ticket.fetch_add(1) * 0xD1B54A32D192ED03ull; // never cancels: no capture across a reallocation
        return BDSP_OK;
    }
    ~DevVec()
    {
        hipStream_t s = lib_stream();
        if (data) ws_free(data, s);
        if (buf) ws_free(buf, s);
    }
};

template <typename T> struct ResultOf;
template <> struct ResultOf<float> { using type = VectorInteropResult32; using handle = VecBuf32; };
template <> struct ResultOf<double> { using type = VectorInteropResult64; using handle = VecBuf64; };

template <typename T> DevVec<T>* H(typename ResultOf<T>::handle* h) { return reinterpret_cast<DevVec<T>*>(h); }
template <typename T> const DevVec<T>* H(const typename ResultOf<T>::handle* h) { return reinterpret_cast<const DevVec<T>*>(h); }

// convert_vec / trans_vec result convention (interop/src/lib.rs:28-76): an Err(reason) maps to its
// code; otherwise -1 if the vector is poisoned, else 0.  Backend failures use the <= -100 range.
template <typename T>
typename ResultOf<T>::type finish(DevVec<T>* v, int code)
{
    typename ResultOf<T>::type r;
    r.vector = reinterpret_cast<typename ResultOf<T>::handle*>(v);
    if (code == BDSP_OK) code = v->erroneous() ? BDSP_ERR_POISONED : BDSP_OK;
    r.result_code = code;
    return r;
}

template <typename T>
DevVec<T>* vec_new(int is_complex, int domain, T init, size_t length, T delta)
{
    if (device_ready() != BDSP_OK) return nullptr;
    DevVec<T>* v = new DevVec<T>();
    v->complex_ = is_complex != 0;
    v->freq = domain != 0;
    v->delta = delta;
    if (v->reserve(length ? length : 1) != BDSP_OK) { delete v; return nullptr; }
    v->valid_len = length;
    // to_gen_dsp_vec on an odd-length complex vector yields valid_len 0 (support_std.rs:288-291)
    if (v->complex_ && length % 2 != 0) v->valid_len = 0;
    if (length) {
        if (init == (T)0) (void)hipMemsetAsync(v->data, 0, sizeof(T) * length, lib_stream());
        else (void)ew_fill<T>(v->data, length, init, lib_stream());
    }
    return v;
}

template <typename T>
DevVec<T>* vec_clone(const DevVec<T>* o)
{
    DevVec<T>* v = new DevVec<T>();
    v->complex_ = o->complex_;
    v->freq = o->freq;
    v->delta = o->delta;
    if (v->reserve(o->valid_len ? o->valid_len : 1) != BDSP_OK) { delete v; return nullptr; }
    v->valid_len = o->valid_len;
    if (o->valid_len)
        (void)hipMemcpyAsync(v->data, o->data, sizeof(T) * o->valid_len, hipMemcpyDeviceToDevice, lib_stream());
    return v;
}

// assert_meta_data! (elementary.rs:370-381, convolution.rs:257-268)
template <typename T>
bool meta_agrees(const DevVec<T>* a, const DevVec<T>* b)
{
    T ratio = a->delta / b->delta;
    return a->complex_ == b->complex_ && a->freq == b->freq && !(ratio > (T)1.1) && !(ratio < (T)0.9);
}

template <typename T>
int op_binary(DevVec<T>* v, const DevVec<T>* o, int op)
{
    if (v->valid_len != o->valid_len) return BDSP_ERR_SAME_SIZE; // elementary.rs:392
    if (!meta_agrees(v, o)) return BDSP_ERR_META_DATA;
    return ew_binary<T>(v->data, o->data, v->valid_len, v->complex_, op, lib_stream());
}

template <typename T>
int op_complex_to_real(DevVec<T>* v, int kind)
{
    if (!v->complex_) { v->poison(); return BDSP_OK; } // assert_complex!, complex_to_real.rs:352-362
    BDSP_TRY(ew_complex_to_real<T>(v->data, v->buf, v->valid_len, kind, lib_stream()));
    v->trade();
    v->valid_len /= 2;
    v->complex_ = false;
    return BDSP_OK;
}

// window ids: facade ids 0..3 (interop/src/lib.rs:153-164, unknown -> rectangular) + 4 = Hann
template <typename T>
void map_window(int id, int* wid, T* alpha)
{
    *alpha = (T)0.54;
    if (id == 4) { *wid = 1; *alpha = (T)0.5; }
    else if (id >= 0 && id <= 3) *wid = id;
    else *wid = 3;
}

// plain_fft / fft / windowed_fft (time_to_freq.rs:136-176) and plain_ifft / ifft / windowed_ifft
// (freq_to_time.rs:136-177) with the shift, window and 1/N scale fused into the transform.
template <typename T>
int op_fft(DevVec<T>* v, bool inverse, bool shift, int window /* -1 none */, size_t rows = 1)
{
    // rows > 1: the vector holds `rows` equally long rows back to back (the matrix API); every row is
    // transformed by the same batched launches
    hipStream_t s = lib_stream();
    if (!inverse) {
        if (v->freq) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_OK; } // :140-145
    } else {
        if (!v->freq) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_OK; } // :142-147
    }
    unsigned flags = 0;
    size_t points = v->points() / rows;
    if (!v->complex_) { // real input is zero-interleaved to complex first (:147-150)
        flags |= FFT_IN_REAL;
        BDSP_TRY(v->reserve(2 * v->valid_len));
    }
    int wid = -1;
    T alpha = 0;
    T in_scale = (T)1;
    if (!inverse) {
        if (window >= 0) map_window<T>(window, &wid, &alpha);
        if (shift) flags |= BDSP_FFT_SHIFT_OUT;
    } else if (shift) {
        // ifft = scale(1/points) -> ifft_shift -> plain_ifft (freq_to_time.rs:160-168)
        in_scale = (T)1 / (T)points;
        flags |= BDSP_FFT_SHIFT_IN;
        if (window >= 0) { map_window<T>(window, &wid, &alpha); flags |= FFT_WINDOW_OUT_DIV; }
    }
    if (points == 0) { v->complex_ = true; v->freq = !inverse; return BDSP_OK; }
    bool in_b = false;
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, points, rows, inverse, flags, in_scale, wid, alpha, &in_b, s));
    if (in_b) v->trade();
    v->valid_len = 2 * points * rows;
    v->complex_ = true;
    // fft(): delta <- points * delta (time_freq/mod.rs:54-55; the reference's own GPU branch
    // forgets this, SURVEY.md section 3.1 -- the CPU behaviour is the contract)
    v->delta = (T)points * v->delta;
    // Deviation, documented in DESIGN.md: the reference leaves a GenDspVec in the FREQUENCY domain
    // after plain_ifft (freq_to_time.rs:153 calls to_freq()); we report Time.
    v->freq = !inverse;
    return BDSP_OK;
}

template <typename T>
int op_convolve_signal(DevVec<T>* v, const DevVec<T>* h)
{
    if (!meta_agrees(v, h)) return BDSP_ERR_META_DATA;  // convolution.rs:485
    if (v->freq) return BDSP_ERR_MUST_BE_TIME;           // :486-488
    if (v->points() < h->points()) return BDSP_ERR_ARG_LENGTH; // :490-492
    if (v->points() == 0 || h->points() == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    if (v->complex_) BDSP_TRY(conv_complex_dev<T>(v->data, v->buf, v->points(), 1, h->data, h->points(), s));
    else BDSP_TRY(conv_real_dev<T>(v->data, v->buf, v->points(), h->data, h->points(), s));
    v->trade();
    return BDSP_OK;
}

template <typename T>
int op_interpolatef(DevVec<T>* v, int fid, T rolloff, T factor, T delay, size_t conv_len,
                    T (*host_fn)(const void*, T) = nullptr, const void* host_fn_data = nullptr)
{
    size_t new_len = interpolatef_new_len<T>(v->valid_len, factor);
    BDSP_TRY(v->reserve(new_len > v->valid_len ? new_len : v->valid_len));
    if (v->valid_len == 0) return BDSP_OK;
    BDSP_TRY(interpolatef_dev<T>(v->data, v->buf, v->valid_len, v->complex_, fid, rolloff, factor, delay,
                                 conv_len, v->delta, lib_stream(), host_fn, host_fn_data));
    v->trade();
    v->valid_len = new_len; // interpolation.rs:481; delta is left alone by interpolatef
    return BDSP_OK;
}

template <typename T>
int op_swap(DevVec<T>* v, bool forward)
{
    size_t p = v->points();
    if (p == 0) return BDSP_OK;
    size_t shift = forward ? p - p / 2 : p / 2;
    BDSP_TRY(rg_rotate<T>(v->data, v->buf, p, v->complex_ ? 2 : 1, shift, lib_stream()));
    v->trade();
    return BDSP_OK;
}

template <typename T>
int op_zero_pad(DevVec<T>* v, size_t points, int option)
{
    size_t step = v->complex_ ? 2 : 1, len = points * step;
    if (len <= v->valid_len) return BDSP_ERR_ARG_LENGTH; // data_reorganization.rs:415-417
    BDSP_TRY(v->reserve(len));
    int opt = option == 0 ? 0 : (option == 1 ? 1 : 2); // interop/src/lib.rs:194-200
    BDSP_TRY(rg_zero_pad<T>(v->data, v->buf, v->valid_len, v->complex_, points, opt, lib_stream()));
    v->trade();
    v->valid_len = len;
    return BDSP_OK;
}

template <typename T>
int op_zero_interleave(DevVec<T>* v, int factor)
{
    if (factor <= 1) return BDSP_OK; // data_reorganization.rs:256-258
    size_t nl = v->valid_len * (size_t)factor;
    BDSP_TRY(v->reserve(nl));
    BDSP_TRY(rg_zero_interleave<T>(v->data, v->buf, v->valid_len, v->complex_ ? 2 : 1, (size_t)factor, lib_stream()));
    v->trade();
    v->valid_len = nl;
    return BDSP_OK;
}

template <typename T>
int op_to_complex(DevVec<T>* v)
{
    if (v->complex_) { v->poison(); return BDSP_OK; } // real_to_complex.rs: assert_real!
    size_t nl = 2 * v->valid_len;
    BDSP_TRY(v->reserve(nl));
    BDSP_TRY(rg_zero_interleave<T>(v->data, v->buf, v->valid_len, 1, 2, lib_stream()));
    v->trade();
    v->valid_len = nl;
    v->complex_ = true;
    return BDSP_OK;
}

template <typename T>
int op_mirror(DevVec<T>* v)
{
    if (!v->freq && !v->complex_) { v->poison(); return BDSP_OK; } // freq.rs:56-59
    if (v->valid_len < 2) return BDSP_OK;
    size_t nl = 2 * v->valid_len - 2;
    BDSP_TRY(v->reserve(nl));
    BDSP_TRY(rg_mirror<T>(v->data, v->buf, v->valid_len, lib_stream()));
    v->trade();
    v->valid_len = nl;
    return BDSP_OK;
}

template <typename T>
int op_window(DevVec<T>* v, int window, bool unapply)
{
    int wid;
    T alpha;
    map_window<T>(window, &wid, &alpha);
    return ew_window<T>(v->data, v->valid_len, v->complex_, wid, alpha, unapply, lib_stream());
}


// A host callback standing in for a built-in frequency response / impulse response (the *_custom and *_complex
// facade variants, interop/src/lib.rs:245-377): sampled on the host into a table, applied on the device.
template <typename T> struct CRet { T re, im; }; // #[repr(C)] Complex<T> returned by value
template <typename T> struct Sampler {
    T (*rfn)(const void*, T) = nullptr;
    CRet<T> (*cfn)(const void*, T) = nullptr;
    const void* data = nullptr;
    bool symmetric = false;
};

template <typename T>
int upload_table(WsBlock& tb, const std::vector<T>& h, hipStream_t s);

// multiply_function_priv (time_freq/mod.rs:612-723) with a sampled function: element i is multiplied by
// ratio * f(fft_swap_x(shifted, j, max) * ratio), j = -max + i (a symmetric function is only evaluated for j <= 0)
template <typename T>
int apply_sampled_response(T* x, size_t len, bool is_complex, const Sampler<T>& f, T ratio, bool shifted, hipStream_t s)
{
    const size_t points = is_complex ? len / 2 : len;
    if (points == 0) return BDSP_OK;
    const T maxv = (T)(points - points % 2) / (T)2;
    auto axis = [&](size_t i) {
        T j = -maxv + (T)i;
        if (f.symmetric && j > (T)0) j = -j;
        T xs;
        if (!shifted) xs = j / maxv;                       // fft_swap_x, mod.rs:67-77
        else if (j <= (T)0) xs = (T)1 + j / maxv;
        else xs = -(maxv - j + (T)1) / maxv;
        return xs * ratio;
    };
    WsBlock tb;
    if (f.cfn) {
        if (!is_complex) return BDSP_ERR_MUST_BE_COMPLEX;
        std::vector<T> h(2 * points);
        for (size_t i = 0; i < points; ++i) {
            const CRet<T> v = f.cfn(f.data, axis(i));
            h[2 * i] = ratio * v.re;
            h[2 * i + 1] = ratio * v.im;
        }
        BDSP_TRY(upload_table<T>(tb, h, s));
        return ew_binary<T>(x, tb.as<T>(), len, true, 2, s);
    }
    std::vector<T> h(points);
    for (size_t i = 0; i < points; ++i) h[i] = ratio * f.rfn(f.data, axis(i));
    BDSP_TRY(upload_table<T>(tb, h, s));
    return ew_point_table<T>(x, len, is_complex, tb.as<T>(), false, s);
}

// ---- FFT-domain interpolation family, composed from the fused kernels -----------
// interpolatei (interpolation.rs:484-532): zero_interleave -> plain_fft -> x frequency response on the
// fft-shifted axis (scaled by the factor) -> plain_ifft -> scale(1/points) [-> real parts]
template <typename T>
int op_interpolatei(DevVec<T>* v, int fid, T rolloff, int factor, const Sampler<T>* custom = nullptr)
{
    if (factor <= 1) return BDSP_OK;
    hipStream_t s = lib_stream();
    const bool was_complex = v->complex_;
    const size_t points = v->points(), np = points * (size_t)factor;
    if (points == 0) return BDSP_OK;
    BDSP_TRY(v->reserve(2 * np));
    // Round 4: the transform of the vector interleaved with factor - 1 zeros is the factor-fold periodic repetition of
    // the transform of the vector itself, so the reference's zero_interleave -> plain_fft of factor * N points
    // (interpolation.rs:484-532) is an N-point transform (real input read directly: zero imaginary parts) whose
    // spectrum is read `factor` times under the frequency response -- one resampling trip instead of memset +
    // interleave + response, and the forward transform at 1/factor of the size (*measured*, 4M points f32, factor 2:
    // 221 -> see DESIGN.md 4.4).
    bool in_b = false;
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, points, 1, false, was_complex ? 0u : FFT_IN_REAL, (T)1, -1, (T)0, &in_b, s));
    if (in_b) v->trade();
    BDSP_TRY(ew_spectrum_resample<T>(v->data, v->buf, points, np, 0, custom ? -2 : fid, rolloff, (T)factor, 0.0, s));
    v->trade();
    if (custom) BDSP_TRY(apply_sampled_response<T>(v->data, 2 * np, true, *custom, (T)factor, true, s));
    // plain_ifft then scale(1/points): the scale rides on the inverse transform's input
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, np, 1, true, was_complex ? 0u : FFT_OUT_REAL, (T)1 / (T)np, -1, (T)0, &in_b, s));
    if (in_b) v->trade();
    v->valid_len = was_complex ? 2 * np : np; // (a real vector's result: the real parts straight from the last pass)
    return BDSP_OK; // the reference does not touch delta here (interpolation.rs:484-532)
}

// interpolate / interpft (interpolation.rs:534-605).  fid < 0 = no frequency response.
template <typename T>
int op_interpolate(DevVec<T>* v, int fid, T rolloff, size_t dest_points, T delay, const Sampler<T>* custom = nullptr)
{
    hipStream_t s = lib_stream();
    const bool was_complex = v->complex_;
    const size_t points = v->points();
    if (points == 0 || dest_points == 0) return BDSP_ERR_ARG_LENGTH;
    const T delta_t = v->delta;
    const T factorf = (T)dest_points / (T)points;
    const size_t maxp = points > dest_points ? points : dest_points;
    BDSP_TRY(v->reserve(2 * maxp));
    if (!was_complex) {
        BDSP_TRY(rg_zero_interleave<T>(v->data, v->buf, points, 1, 2, s));
        v->trade();
    }
    bool in_b = false;
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, points, 1, false, 0, (T)1, -1, (T)0, &in_b, s));
    if (in_b) v->trade();
    if (dest_points > points) {
        // linear phase (on the source bins) + zero_pad(Center) + frequency response / scale: ONE resampling trip
        // (round 4; before: three, and the padding's own memset)
        const T dly = delay / delta_t;
        const T phase_inc = (T)2 * (T)3.14159265358979323846 * dly / (T)points; // as ew_linear_phase computes it
        BDSP_TRY(ew_spectrum_resample<T>(v->data, v->buf, points, dest_points, 1, custom ? -2 : (fid < 0 ? -1 : fid), rolloff, factorf,
                                         delay != (T)0 ? (double)phase_inc : 0.0, s));
        v->trade();
        if (custom) BDSP_TRY(apply_sampled_response<T>(v->data, 2 * dest_points, true, *custom, factorf, true, s));
    } else if (dest_points < points) {
        // interpolate_downsample (:362-376): linear phase + the crop to the first pos and the last neg bins + the
        // scale in one resampling trip
        const T dly = delay / delta_t;
        const T phase_inc = (T)2 * (T)3.14159265358979323846 * dly / (T)points;
        BDSP_TRY(ew_spectrum_resample<T>(v->data, v->buf, points, dest_points, 2, -1, (T)0, (T)(2 * dest_points) / (T)(2 * points),
                                         delay != (T)0 ? (double)phase_inc : 0.0, s));
        v->trade();
    } else if (delay != (T)0) {
        BDSP_TRY(ew_linear_phase<T>(v->data, 2 * points, delay / delta_t, s));
    }
    // (a real vector's result: the real parts straight from the inverse transform's last pass, FFT_OUT_REAL)
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, dest_points, 1, true, was_complex ? 0u : FFT_OUT_REAL, (T)1 / (T)dest_points, -1, (T)0, &in_b, s));
    if (in_b) v->trade();
    v->delta = delta_t / factorf;
    v->valid_len = was_complex ? 2 * dest_points : dest_points;
    return BDSP_OK;
}

template <typename T>
int op_decimatei(DevVec<T>* v, unsigned factor, unsigned delay)
{
    if (factor == 0) return BDSP_ERR_ARG_LENGTH;
    const size_t elem = v->complex_ ? 2 : 1, points = v->points();
    const size_t outp = delay < points ? (points - delay + factor - 1) / factor : 0;
    if (outp) {
        BDSP_TRY(rg_decimate<T>(v->data, v->buf, outp, elem, factor, delay, lib_stream()));
        v->trade();
    }
    v->valid_len = outp * elem;
    return BDSP_OK;
}

template <typename T>
int op_multiply_frequency_response(DevVec<T>* v, int fid, T rolloff, T ratio)
{
    if (!v->freq) { v->poison(); return BDSP_OK; } // convolution.rs:590-593
    return ew_freq_response<T>(v->data, v->valid_len, v->complex_, fid, rolloff, ratio, false, lib_stream());
}

// Symmetric real FFT family (time_to_freq.rs:188-298, freq_to_time.rs:180-248): a real vector of odd
// length -> the non-redundant half spectrum (points/2 + 1 bins), and back.
template <typename T>
int op_sfft(DevVec<T>* v, bool shift, int window)
{
    if (v->freq || v->complex_) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_ERR_MUST_BE_TIME; }
    const size_t points = v->valid_len;
    if (points % 2 == 0) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_ERR_ODD_LENGTH; }
    BDSP_TRY(op_fft<T>(v, false, shift, window));
    if (!shift) v->valid_len = 2 * (points / 2 + 1); // unmirror! (time_to_freq.rs:178-186)
    else {
        // fft() is shifted: the non-negative frequencies are the LAST points/2+1 bins; the reference
        // truncates the shifted vector to its first points/2+1 bins (unmirror! after fft_shift), i.e.
        // it keeps the negative half plus DC -- we do the same
        v->valid_len = 2 * (points / 2 + 1);
    }
    return BDSP_OK;
}

template <typename T>
int op_sifft(DevVec<T>* v, bool shift, int window)
{
    if (!v->freq || !v->complex_) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_ERR_MUST_BE_FREQ; }
    hipStream_t s = lib_stream();
    if (shift) {
        // sifft: scale(1/points) and ifft_shift come first (freq_to_time.rs:226-236), so the symmetry test of
        // plain_sifft below sees the first bin of the SHIFTED half spectrum, as in the reference
        const size_t p = v->points();
        if (p) BDSP_TRY(ew_real_scale<T>(v->data, v->valid_len, (T)1 / (T)p, s));
        BDSP_TRY(op_swap<T>(v, false));
    }
    if (v->points() > 0) {
        // The first bin must be real (freq_to_time.rs:203-211 tests |im| > 1e-10).  A spectrum that
        // was COMPUTED (e.g. by plain_sfft through Bluestein) carries rounding noise of a few
        // eps * |X| there, so the absolute test is paired with a relative one against the first bins.
        T h[4] = {0, 0, 0, 0};
        size_t nh = v->valid_len < 4 ? v->valid_len : 4;
        BDSP_HIP_TRY(hipMemcpyAsync(h, v->data, sizeof(T) * nh, hipMemcpyDeviceToHost, s));
        BDSP_HIP_TRY(hipStreamSynchronize(s));
        double im0 = std::fabs((double)h[1]);
        double scale = std::fabs((double)h[0]) + std::fabs((double)h[2]) + std::fabs((double)h[3]);
        if (im0 > 1e-10 && im0 > 1e-3 * scale) {
            v->poison(); v->complex_ = true; v->freq = true;
            return BDSP_ERR_CONJ_SYMMETRIC;
        }
    }
    BDSP_TRY(op_mirror<T>(v));
    BDSP_TRY(op_fft<T>(v, true, false, -1));
    BDSP_TRY(op_complex_to_real<T>(v, 2));
    if (shift && window >= 0) BDSP_TRY(op_window<T>(v, window, true));
    return BDSP_OK;
}

// convolve(function, ratio, len) in the time domain (convolution.rs:136-192 -> convolve_function_priv,
// time_freq/mod.rs:174-213): y[i] = sum_{m=-L}^{L} x[(i+m) mod N] * f(-m*ratio).  The 2L+1 weights are
// tabulated once; when the table fits the vector it is the tap vector of the centred convolution and the
// fused overlap-save kernel does the work.  host_taps (window order, 2L+1 values) replaces the built-in
// functions for the callback variants (convolve_real32).  The reference's alternative branch for long
// vectors (:148-172) builds its tap vector with 1/ratio instead of ratio and fills every other tap of a
// real vector only; the convolve_function_priv semantics are the documented ones and are used throughout.
template <typename T>
int op_convolve_function(DevVec<T>* v, int fid, T rolloff, T ratio, size_t conv_len, const T* host_taps,
                         bool complex_taps = false)
{
    if (v->freq) { v->poison(); return BDSP_OK; } // assert_time! (convolution.rs:95-102)
    const size_t points = v->points();
    if (points == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    if (conv_len > points) conv_len = points; // mod.rs:197
    const size_t ntaps = 2 * conv_len + 1;
    const bool as_taps = ntaps <= points;
    const int stride = ((as_taps && v->complex_) || complex_taps) ? 2 : 1;
    WsBlock tb;
    BDSP_TRY(tb.alloc(sizeof(T) * ntaps * stride, s));
    if (host_taps) {
        std::vector<T> h(ntaps * stride, (T)0);
        for (size_t k = 0; k < ntaps; ++k) {
            const size_t at = (as_taps ? ntaps - 1 - k : k) * stride;
            if (complex_taps) { h[at] = host_taps[2 * k]; h[at + 1] = host_taps[2 * k + 1]; }
            else h[at] = host_taps[k];
        }
        BDSP_HIP_TRY(hipMemcpyAsync(tb.p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s));
        BDSP_HIP_TRY(hipStreamSynchronize(s));
    } else {
        BDSP_TRY(conv_function_taps<T>(tb.as<T>(), conv_len, fid, rolloff, ratio, stride, as_taps, s));
    }
    if (!as_taps) BDSP_TRY(conv_function_direct<T>(v->data, v->buf, points, v->complex_, tb.as<T>(), conv_len, s, complex_taps));
    else if (v->complex_) BDSP_TRY(conv_complex_dev<T>(v->data, v->buf, points, 1, tb.as<T>(), ntaps, s));
    else BDSP_TRY(conv_real_dev<T>(v->data, v->buf, points, tb.as<T>(), ntaps, s));
    v->trade();
    return BDSP_OK;
}

// Cross correlation (correlation.rs:96-160)
template <typename T>
int op_prepare_argument(DevVec<T>* v, bool padded)
{
    if (padded) {
        const size_t points = v->points();
        // zero_pad_b(2*points-1, Surround).expect(..): the reference PANICS for points <= 1; code 7 here
        if (points <= 1) return BDSP_ERR_ARG_LENGTH;
        BDSP_TRY(op_zero_pad<T>(v, 2 * points - 1, 1));
    }
    BDSP_TRY(op_fft<T>(v, false, false, -1));
    if (v->erroneous()) return BDSP_OK;
    return ew_conj<T>(v->data, v->valid_len, lib_stream());
}

template <typename T>
int op_correlate(DevVec<T>* v, const DevVec<T>* other)
{
    // both failures report InputMustBeInTimeDomain (correlation.rs:134-146)
    if (v->freq || !v->complex_ || !other->freq || !other->complex_) {
        v->poison(); v->complex_ = true; v->freq = true;
        return BDSP_ERR_MUST_BE_TIME;
    }
    hipStream_t s = lib_stream();
    const size_t points = other->points();
    BDSP_TRY(op_zero_pad<T>(v, points, 1)); // 7 unless the argument is longer than self
    bool in_b = false;
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, points, 1, false, 0, (T)1, -1, (T)0, &in_b, s));
    if (in_b) v->trade();
    BDSP_TRY(ew_binary<T>(v->data, other->data, v->valid_len, true, 2, s));
    // plain_ifft -> scale(1/points) -> swap_halves: scale and shift ride on the inverse transform
    BDSP_TRY(fft_two_buffers<T>(v->data, v->buf, points, 1, true, BDSP_FFT_SHIFT_OUT, (T)1 / (T)points, -1, (T)0, &in_b, s));
    if (in_b) v->trade();
    return BDSP_OK; // delta is untouched: the transforms ran on a view (correlation.rs:150-153)
}

// interpolate_lin / interpolate_hermite (real_interpolation.rs:33-176)
template <typename T>
int op_interpolate_real(DevVec<T>* v, T factor, T delay, bool hermite)
{
    if (v->complex_) { v->poison(); return BDSP_OK; } // :47-50, :89-92
    if (v->valid_len == 0) return BDSP_OK;             // (the reference underflows len-1 and panics)
    const size_t dest_len = interpolate_real_len<T>(v->valid_len, factor);
    BDSP_TRY(v->reserve(dest_len > v->valid_len ? dest_len : v->valid_len));
    BDSP_TRY(interpolate_real_dev<T>(v->data, v->buf, v->valid_len, factor, delay, hermite, lib_stream()));
    v->trade();
    v->valid_len = dest_len;
    return BDSP_OK;
}

// get_real / get_imag / get_magnitude / get_magnitude_squared / get_phase (complex_to_real.rs:620-700): the
// result goes into `dst` (resized to `points` reals); a real source or a complex destination empties dst.
// The facade passes the source by value (it is consumed) and returns convert_void(Ok(())) = 9, whatever
// happened (interop/src/lib.rs:100-105) -- kept for link compatibility.
template <typename T>
int op_get_complex_to_real(DevVec<T>* v, DevVec<T>* dst, int kind)
{
    int rc = BDSP_OK;
    if (!v->complex_ || dst->complex_) {
        dst->valid_len = 0;
    } else {
        const size_t points = v->points();
        rc = dst->reserve(points ? points : 1);
        if (rc == BDSP_OK && points) rc = ew_complex_to_real<T>(v->data, dst->data, v->valid_len, kind, lib_stream());
        if (rc == BDSP_OK) dst->valid_len = points;
    }
    delete v;
    return rc == BDSP_OK ? 9 : rc;
}

template <typename T>
int op_binary_smaller(DevVec<T>* v, const DevVec<T>* o, int op)
{
    if (o->valid_len == 0 || v->valid_len % o->valid_len != 0) return BDSP_ERR_ARG_LENGTH; // elementary.rs:613-617
    if (!meta_agrees(v, o)) return BDSP_ERR_META_DATA;
    return ew_binary_smaller<T>(v->data, o->data, v->valid_len, o->valid_len, v->complex_, op, lib_stream());
}

// Host-sampled callbacks (interop/src/lib.rs:245-377).  A window callback is sampled for every point
// (or for the first ceil(P/2) points and mirrored when is_symmetric, vector_types/mod.rs:567-594).
template <typename T>
int upload_table(WsBlock& tb, const std::vector<T>& h, hipStream_t s)
{
    BDSP_TRY(tb.alloc(sizeof(T) * (h.size() ? h.size() : 1), s));
    if (h.empty()) return BDSP_OK;
    BDSP_HIP_TRY(hipMemcpyAsync(tb.p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    return BDSP_OK;
}

template <typename T>
int op_custom_window(DevVec<T>* v, T (*window)(const void*, size_t, size_t), const void* data, bool symmetric,
                     bool unapply)
{
    const size_t points = v->points();
    if (points == 0) return BDSP_OK;
    std::vector<T> h(points);
    const size_t half = points - points / 2;
    for (size_t i = 0; i < points; ++i) {
        if (symmetric && i >= half) h[i] = h[points - 1 - i];
        else h[i] = window(data, i, points);
    }
    WsBlock tb;
    hipStream_t s = lib_stream();
    BDSP_TRY(upload_table<T>(tb, h, s));
    return ew_point_table<T>(v->data, v->valid_len, v->complex_, tb.as<T>(), unapply, s);
}

// windowed_custom_{fft,ifft,sfft,sifft}: the table multiply cannot be fused; the transform still is
template <typename T>
int op_custom_windowed(DevVec<T>* v, T (*window)(const void*, size_t, size_t), const void* data, bool symmetric,
                       int kind /* 0 fft, 1 ifft, 2 sfft, 3 sifft */)
{
    switch (kind) {
    case 0:
        if (v->freq) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_OK; }
        BDSP_TRY(op_custom_window<T>(v, window, data, symmetric, false));
        return op_fft<T>(v, false, true, -1);
    case 1:
        BDSP_TRY(op_fft<T>(v, true, true, -1));
        if (v->erroneous()) return BDSP_OK;
        return op_custom_window<T>(v, window, data, symmetric, true);
    case 2:
        if (v->freq || v->complex_) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_ERR_MUST_BE_TIME; }
        if (v->valid_len % 2 == 0) { v->poison(); v->complex_ = true; v->freq = true; return BDSP_ERR_ODD_LENGTH; }
        BDSP_TRY(op_custom_window<T>(v, window, data, symmetric, false));
        return op_sfft<T>(v, true, -1);
    default: {
        int c = op_sifft<T>(v, true, -1);
        if (c != BDSP_OK || v->erroneous()) return c;
        return op_custom_window<T>(v, window, data, symmetric, true);
    }
    }
}

// multiply_frequency_response with a sampled real callback (multiply_function_priv, mod.rs:612-723):
// element i of a natural-order spectrum uses x = fft_swap_x(false, j, max) * ratio = j/max*ratio,
// j = -max + i, max = (points - points%2)/2; a symmetric function is evaluated on j <= 0 only.
template <typename T>
int op_custom_frequency_response(DevVec<T>* v, T (*fun)(const void*, T), const void* data, bool symmetric, T ratio)
{
    if (!v->freq) { v->poison(); return BDSP_OK; }
    const size_t points = v->points();
    if (points == 0) return BDSP_OK;
    const T maxv = (T)(points - points % 2) / (T)2;
    std::vector<T> h(points);
    for (size_t i = 0; i < points; ++i) {
        T j = -maxv + (T)i;
        if (symmetric && j > (T)0) j = -j;
        h[i] = ratio * fun(data, j / maxv * ratio);
    }
    WsBlock tb;
    hipStream_t s = lib_stream();
    BDSP_TRY(upload_table<T>(tb, h, s));
    return ew_point_table<T>(v->data, v->valid_len, v->complex_, tb.as<T>(), false, s);
}

template <typename T>
int op_convolve_callback(DevVec<T>* v, T (*fun)(const void*, T), const void* data, T ratio, size_t conv_len)
{
    const size_t points = v->points();
    if (conv_len > points) conv_len = points;
    std::vector<T> h(2 * conv_len + 1);
    T j = -(T)conv_len;
    for (size_t k = 0; k < h.size(); ++k) { h[k] = fun(data, -j * ratio); j = j + (T)1; }
    return op_convolve_function<T>(v, 0, (T)0, ratio, conv_len, h.data());
}

// convolve_complex (convolution.rs:204-254): complex vector, complex impulse response callback
template <typename T>
int op_convolve_callback_complex(DevVec<T>* v, CRet<T> (*fun)(const void*, T), const void* data, T ratio, size_t conv_len)
{
    if (!v->complex_) { v->poison(); return BDSP_OK; } // assert_complex!
    const size_t points = v->points();
    if (conv_len > points) conv_len = points;
    std::vector<T> h(2 * (2 * conv_len + 1));
    T j = -(T)conv_len;
    for (size_t k = 0; k < 2 * conv_len + 1; ++k) {
        const CRet<T> w = fun(data, -j * ratio);
        h[2 * k] = w.re; h[2 * k + 1] = w.im;
        j = j + (T)1;
    }
    return op_convolve_function<T>(v, 0, (T)0, ratio, conv_len, h.data(), true);
}

// multiply_frequency_response_complex (convolution.rs:578-610): complex frequency-domain vector
template <typename T>
int op_custom_frequency_response_complex(DevVec<T>* v, CRet<T> (*fun)(const void*, T), const void* data, bool symmetric, T ratio)
{
    if (!v->complex_ || !v->freq) { v->poison(); return BDSP_OK; }
    Sampler<T> sm;
    sm.cfn = fun; sm.data = data; sm.symmetric = symmetric;
    return apply_sampled_response<T>(v->data, v->valid_len, true, sm, ratio, false, lib_stream());
}

template <typename T>
const T* vec_download(DevVec<T>* v)
{
    v->mirror.resize(v->valid_len ? v->valid_len : 1);
    hipStream_t s = lib_stream();
    if (v->valid_len) {
        if (hipMemcpyAsync(v->mirror.data(), v->data, sizeof(T) * v->valid_len, hipMemcpyDeviceToHost, s) != hipSuccess)
            return nullptr;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
    return v->mirror.data();
}

// ----------------------------------------------------------------------------------------------
// Per-element math family, differences / running sums, phase wrapping, real<->complex pairs, split / merge
// (vecmath.hip).  Number-space rules follow the reference: the RealOps family (abs, wrap, unwrap, *_approx)
// poisons a complex vector (real_ops.rs:222-233), TrigOps / PowerOps accept both.
// ----------------------------------------------------------------------------------------------
template <typename T>
int op_math(DevVec<T>* v, int fn, T arg, bool real_only)
{
    if (real_only && v->complex_) { v->poison(); return BDSP_OK; }
    return ew_math<T>(v->data, v->valid_len, v->complex_, fn, arg, lib_stream());
}

// diff (diff_sum.rs:65-82) drops the first element; diff_with_start (:84-108) keeps it.  An empty vector stays
// empty (the reference underflows valid_len there).
template <typename T>
int op_diff(DevVec<T>* v, bool with_start)
{
    const size_t step = v->complex_ ? 2 : 1;
    if (v->valid_len < step) return BDSP_OK;
    const size_t n_out = with_start ? v->valid_len : v->valid_len - step;
    BDSP_TRY(vm_diff<T>(v->data, v->buf, n_out, step, with_start, lib_stream()));
    v->trade();
    v->valid_len = n_out;
    return BDSP_OK;
}

template <typename T>
int op_cum_sum(DevVec<T>* v)
{
    if (v->valid_len == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    WsBlock sc;
    BDSP_TRY(sc.alloc(vm_cum_sum_scratch<T>(v->valid_len, v->complex_), s));
    return vm_cum_sum<T>(v->data, v->valid_len, v->complex_, sc.p, s);
}

template <typename T>
int op_unwrap(DevVec<T>* v, T divisor)
{
    if (v->complex_) { v->poison(); return BDSP_OK; }
    return vm_unwrap<T>(v->data, v->valid_len, divisor, lib_stream());
}

// get_real_imag / get_mag_phase (complex_to_real.rs:674-712): both targets are resized to `points` reals; a real
// source or a complex target empties both.  The source is consumed; the facade answers convert_void(Ok) = 9.
template <typename T>
int op_get_pair(DevVec<T>* v, DevVec<T>* a, DevVec<T>* b, int kind)
{
    int rc = BDSP_OK;
    if (!v->complex_ || a->complex_ || b->complex_) {
        a->valid_len = 0; b->valid_len = 0;
    } else {
        const size_t points = v->points();
        rc = a->reserve(points ? points : 1);
        if (rc == BDSP_OK) rc = b->reserve(points ? points : 1);
        if (rc == BDSP_OK) rc = vm_complex_split<T>(v->data, a->data, b->data, points, kind, lib_stream());
        if (rc == BDSP_OK) { a->valid_len = points; b->valid_len = points; }
    }
    delete v;
    return rc == BDSP_OK ? 9 : rc;
}

// set_real_imag / set_mag_phase (complex_to_real.rs:726-770): code 7 unless both arguments have the same length
template <typename T>
int op_set_pair(DevVec<T>* v, const DevVec<T>* a, const DevVec<T>* b, int kind)
{
    if (a->valid_len != b->valid_len) return BDSP_ERR_ARG_LENGTH;
    const size_t points = a->valid_len;
    BDSP_TRY(v->reserve(2 * points ? 2 * points : 1));
    BDSP_TRY(vm_complex_join<T>(v->data, a->data, b->data, points, kind, lib_stream()));
    v->valid_len = 2 * points;
    return BDSP_OK;
}

template <typename T>
int upload_parts(WsBlock& tb, DevVec<T>* const* parts, size_t n, hipStream_t s)
{
    std::vector<T*> h(n);
    for (size_t k = 0; k < n; ++k) h[k] = parts[k]->data;
    BDSP_TRY(tb.alloc(sizeof(T*) * n, s));
    BDSP_HIP_TRY(hipMemcpyAsync(tb.p, h.data(), sizeof(T*) * n, hipMemcpyHostToDevice, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    return BDSP_OK;
}

// split_into (data_reorganization.rs:484-512): every target is resized to len / n (13 for an odd length on a
// complex target), element i goes to target i % n, position i / n.  convert_void: 9 on success.
template <typename T>
int op_split_into(const DevVec<T>* v, DevVec<T>* const* targets, size_t n)
{
    if (n == 0 || v->valid_len % n != 0) return BDSP_ERR_ARG_LENGTH;
    const size_t tlen = v->valid_len / n;
    for (size_t k = 0; k < n; ++k) {
        if (targets[k]->complex_ && tlen % 2 != 0) return BDSP_ERR_EVEN_LENGTH;
        BDSP_TRY(targets[k]->reserve(tlen ? tlen : 1));
        targets[k]->valid_len = tlen;
    }
    if (tlen == 0) return 9;
    if (v->complex_ && tlen % 2 != 0) return BDSP_ERR_ARG_LENGTH; // complex points do not divide evenly
    hipStream_t s = lib_stream();
    WsBlock tb;
    BDSP_TRY(upload_parts<T>(tb, targets, n, s));
    BDSP_TRY(vm_split_merge<T>(v->data, tb.as<T*>(), v->valid_len, v->complex_, n, false, s));
    return 9;
}

// merge (data_reorganization.rs:522-555): the inverse; all sources must have the same length
template <typename T>
int op_merge(DevVec<T>* v, DevVec<T>* const* sources, size_t n)
{
    if (n == 0) return BDSP_ERR_ARG_LENGTH;
    for (size_t k = 1; k < n; ++k)
        if (sources[k]->valid_len != sources[0]->valid_len) return BDSP_ERR_ARG_LENGTH;
    const size_t len = sources[0]->valid_len * n;
    if (v->complex_ && len % 2 != 0) return BDSP_ERR_EVEN_LENGTH;
    BDSP_TRY(v->reserve(len ? len : 1));
    v->valid_len = len;
    if (len == 0) return BDSP_OK;
    if (v->complex_ && sources[0]->valid_len % 2 != 0) return BDSP_ERR_ARG_LENGTH;
    hipStream_t s = lib_stream();
    WsBlock tb;
    BDSP_TRY(upload_parts<T>(tb, sources, n, s));
    return vm_split_merge<T>(v->data, tb.as<T*>(), len, v->complex_, n, true, s);
}

// map_inplace / map_aggregate (mapping.rs:53-79, 92-140, 163-190, 203-250): the callback is host code, so the
// vector makes one round trip -- download, one call per element in index order, upload.
template <typename T, typename F>
int op_map_inplace(DevVec<T>* v, bool want_complex, F&& per_element)
{
    if (v->complex_ != want_complex) { v->poison(); return BDSP_OK; }
    if (v->valid_len == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    std::vector<T> h(v->valid_len);
    BDSP_HIP_TRY(hipMemcpyAsync(h.data(), v->data, sizeof(T) * h.size(), hipMemcpyDeviceToHost, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    per_element(h);
    BDSP_HIP_TRY(hipMemcpyAsync(v->data, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    return BDSP_OK;
}

template <typename T, typename F>
int op_map_aggregate(const DevVec<T>* v, bool want_complex, const void** result, F&& fold)
{
    *result = nullptr;
    if (v->complex_ != want_complex) return want_complex ? BDSP_ERR_MUST_BE_COMPLEX : BDSP_ERR_MUST_BE_REAL;
    if (v->valid_len == 0) return BDSP_ERR_NOT_EMPTY;
    hipStream_t s = lib_stream();
    std::vector<T> h(v->valid_len);
    BDSP_HIP_TRY(hipMemcpyAsync(h.data(), v->data, sizeof(T) * h.size(), hipMemcpyDeviceToHost, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    *result = fold(h);
    return v->erroneous() ? BDSP_ERR_POISONED : BDSP_OK;
}

// ----------------------------------------------------------------------------------------------
// Statistics, sums, dot products: the device folds the vector into one StatPartial (reduce.hip), the host
// turns it into the reference's result structs.
// ----------------------------------------------------------------------------------------------
static void stat_empty(StatPartial& p, bool cplx)
{
    std::memset(&p, 0, sizeof(p));
    const double inf = std::numeric_limits<double>::infinity();
    if (cplx) { p.mnr = p.mni = inf; p.mn_key = inf; }
    else { p.mnr = inf; p.mn_key = inf; p.mxr = -inf; p.mx_key = -inf; }
}

// 16 results in pinned host memory per calling thread: the folding kernel writes them directly, no copy is queued
static StatPartial* stat_pinned()
{
    static thread_local StatPartial* pin = nullptr;
    if (!pin && hipHostMalloc((void**)&pin, sizeof(StatPartial) * 16, hipHostMallocDefault) != hipSuccess) pin = nullptr;
    return pin;
}

// statistics of `buckets` interleaved sub-sequences (1 = the whole vector) into host[0 .. buckets)
template <typename T>
int stats_run(const DevVec<T>* v, bool cplx, bool minmax, size_t buckets, StatPartial* host)
{
    const size_t units = cplx ? v->valid_len / 2 : v->valid_len;
    for (size_t b = 0; b < buckets; ++b) stat_empty(host[b], cplx);
    if (units == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    WsBlock pb;
    BDSP_TRY(pb.alloc(sizeof(StatPartial) * 1024 * buckets, s));
    StatPartial* pin = stat_pinned();
    if (!pin) return BDSP_ERR_HIP;
    BDSP_TRY(red_stats<T>(v->data, units, buckets, cplx, minmax, pb.as<StatPartial>(), pin, s));
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(host, pin, sizeof(StatPartial) * buckets);
    return BDSP_OK;
}

template <typename S, typename R>
void stat_fill_real(S* out, const StatPartial& p)
{
    const double n = (double)p.cnt;
    out->sum = (R)p.sr; out->count = (size_t)p.cnt;
    out->average = (R)(p.sr / n); out->rms = (R)std::sqrt(p.qr / n);
    out->min = (R)p.mnr; out->min_index = (size_t)p.imn; out->max = (R)p.mxr; out->max_index = (size_t)p.imx;
}

template <typename S, typename R>
void stat_fill_complex(S* out, const StatPartial& p)
{
    const double n = (double)p.cnt;
    // sqrt of the COMPLEX mean of z*z (principal branch), like (sum_squared / count).sqrt() in the reference
    const double qr = p.qr / n, qi = p.qi / n;
    const double r = std::hypot(qr, qi);
    double rr, ri;
    if (r != r) { rr = ri = std::numeric_limits<double>::quiet_NaN(); }
    else if (qi == 0.0 && qr >= 0.0) { rr = std::sqrt(qr); ri = qi; }
    else if (qi == 0.0) { rr = 0.0; ri = std::signbit(qi) ? -std::sqrt(-qr) : std::sqrt(-qr); }
    else { const double th = std::atan2(qi, qr) / 2; rr = std::sqrt(r) * std::cos(th); ri = std::sqrt(r) * std::sin(th); }
    out->sum.re = (R)p.sr; out->sum.im = (R)p.si; out->count = (size_t)p.cnt;
    out->average.re = (R)(p.sr / n); out->average.im = (R)(p.si / n);
    out->rms.re = (R)rr; out->rms.im = (R)ri;
    out->min.re = (R)p.mnr; out->min.im = (R)p.mni; out->min_index = (size_t)p.imn;
    out->max.re = (R)p.mxr; out->max.im = (R)p.mxi; out->max_index = (size_t)p.imx;
}

template <typename T, typename S, typename R>
S stats_real(const DevVec<T>* v)
{
    StatPartial p;
    S out;
    if (stats_run<T>(v, false, true, 1, &p) != BDSP_OK) stat_empty(p, false);
    stat_fill_real<S, R>(&out, p);
    return out;
}
template <typename T, typename S, typename R>
S stats_complex(const DevVec<T>* v)
{
    StatPartial p;
    S out;
    if (stats_run<T>(v, true, true, 1, &p) != BDSP_OK) stat_empty(p, true);
    stat_fill_complex<S, R>(&out, p);
    return out;
}
template <typename T, typename S, typename R, bool CPLX>
int stats_split(const DevVec<T>* v, S* data, size_t len)
{
    if (len == 0) return BDSP_OK;
    if (len > 16) return BDSP_ERR_ARG_LENGTH; // STATS_VEC_CAPACTIY (statistics.rs:34, 403-405)
    StatPartial p[16];
    BDSP_TRY(stats_run<T>(v, CPLX, true, len, p));
    for (size_t b = 0; b < len; ++b) {
        if constexpr (CPLX) stat_fill_complex<S, R>(&data[b], p[b]); else stat_fill_real<S, R>(&data[b], p[b]);
    }
    return BDSP_OK;
}
// sums: which = 0 sum, 1 sum of squares; returns (re, im) in double
template <typename T>
void sums(const DevVec<T>* v, bool cplx, int which, double* re, double* im)
{
    StatPartial p;
    if (stats_run<T>(v, cplx, false, 1, &p) != BDSP_OK) stat_empty(p, cplx);
    *re = which ? p.qr : p.sr;
    *im = which ? p.qi : p.si;
}
// dot products (dot_products.rs:67-165): code 4 / 3 / 2 for the number-space and metadata errors, else the
// vector's error marker (0 or -1)
template <typename T>
int dot(const DevVec<T>* v, const DevVec<T>* o, bool cplx, double* re, double* im)
{
    *re = *im = 0.0;
    if (!cplx && v->complex_) return BDSP_ERR_MUST_BE_REAL;
    if (cplx && !v->complex_) return BDSP_ERR_MUST_BE_COMPLEX;
    if (cplx && (!o->complex_ || o->freq != v->freq)) return BDSP_ERR_META_DATA;
    const size_t len = v->valid_len < o->valid_len ? v->valid_len : o->valid_len;
    const size_t count = cplx ? len / 2 : len;
    if (count) {
        hipStream_t s = lib_stream();
        WsBlock pb;
        BDSP_TRY(pb.alloc(sizeof(StatPartial) * 1024, s));
        StatPartial* pin = stat_pinned();
        if (!pin) return BDSP_ERR_HIP;
        BDSP_TRY(red_dot<T>(v->data, o->data, count, cplx, pb.as<StatPartial>(), pin, s));
        BDSP_HIP_TRY(hipStreamSynchronize(s));
        *re = pin->sr; *im = pin->si;
    }
    return v->erroneous() ? BDSP_ERR_POISONED : BDSP_OK;
}

// ----------------------------------------------------------------------------------------------
// Matrix / batch API: `rows` equally long vectors back to back in ONE allocation, every operation a
// batched launch over all rows.  Mirrors the matrix crate (matrix/src/lib.rs:195-208 applies an
// operation to the rows one after the other; matrix/src/time_freq.rs:49-530 forwards the
// time/frequency traits row by row) -- here the row loop is the grid's batch dimension.
// ----------------------------------------------------------------------------------------------
template <typename T>
struct DevMat {
    DevVec<T> v;       // flat storage: rows * row_len scalars, shared metadata
    size_t rows = 0;
    size_t row_len() const { return rows ? v.valid_len / rows : 0; }
    size_t row_points() const { return v.complex_ ? row_len() / 2 : row_len(); }
};

template <typename T>
DevMat<T>* mat_new(int is_complex, int domain, size_t rows, size_t row_len, T delta)
{
    if (device_ready() != BDSP_OK) return nullptr;
    if (is_complex && row_len % 2 != 0) { set_last_error("complex rows need an even scalar length"); return nullptr; }
    DevMat<T>* m = new DevMat<T>();
    m->v.complex_ = is_complex != 0;
    m->v.freq = domain != 0;
    m->v.delta = delta;
    m->rows = rows;
    const size_t total = rows * row_len;
    if (m->v.reserve(total ? total : 1) != BDSP_OK) { delete m; return nullptr; }
    m->v.valid_len = total;
    if (total) (void)hipMemsetAsync(m->v.data, 0, sizeof(T) * total, lib_stream());
    return m;
}

template <typename T> int mat_code(DevMat<T>* m, int code)
{
    if (code == BDSP_OK && m->v.erroneous()) return BDSP_ERR_POISONED;
    return code;
}

// run a single-vector kernel launcher on every row (index-dependent maps): fn(row_ptr, row_len)
template <typename T, class F>
int mat_each_row(DevMat<T>* m, F fn)
{
    const size_t rl = m->row_len();
    for (size_t r = 0; r < m->rows; ++r) BDSP_TRY(fn(m->v.data + r * rl, rl));
    return BDSP_OK;
}

// rows change length: fn(in_row, out_row) writes new_len scalars per row into the trade buffer
template <typename T, class F>
int mat_resize_rows(DevMat<T>* m, size_t new_len, F fn)
{
    const size_t rl = m->row_len();
    const size_t need = m->rows * (new_len > rl ? new_len : rl);
    BDSP_TRY(m->v.reserve(need));
    for (size_t r = 0; r < m->rows; ++r) BDSP_TRY(fn(m->v.data + r * rl, m->v.buf + r * new_len));
    m->v.trade();
    m->v.valid_len = m->rows * new_len;
    return BDSP_OK;
}

template <typename T>
int mat_binary(DevMat<T>* m, const DevMat<T>* o, int op)
{
    if (m->rows != o->rows) return BDSP_ERR_ARG_LENGTH; // matrix/src/general/elementary.rs: row counts must agree
    return op_binary<T>(&m->v, &o->v, op);
}

template <typename T>
int mat_binary_vector(DevMat<T>* m, const DevVec<T>* o, int op)
{
    // every row (.)= the same vector: the wrap-around kernel with period = one row
    if (o->valid_len != m->row_len()) return BDSP_ERR_SAME_SIZE;
    if (!meta_agrees(&m->v, o)) return BDSP_ERR_META_DATA;
    return ew_binary_smaller<T>(m->v.data, o->data, m->v.valid_len, o->valid_len, m->v.complex_, op, lib_stream());
}

template <typename T>
int mat_complex_to_real(DevMat<T>* m, int kind) { return op_complex_to_real<T>(&m->v, kind); }

template <typename T>
int mat_window(DevMat<T>* m, int window, bool unapply)
{
    int wid;
    T alpha;
    map_window<T>(window, &wid, &alpha);
    const bool c = m->v.complex_;
    return mat_each_row<T>(m, [&](T* row, size_t len) { return ew_window<T>(row, len, c, wid, alpha, unapply, lib_stream()); });
}

template <typename T>
int mat_swap(DevMat<T>* m, bool forward)
{
    const size_t p = m->row_points(), e = m->v.complex_ ? 2 : 1;
    if (p == 0) return BDSP_OK;
    const size_t shift = forward ? p - p / 2 : p / 2;
    return mat_resize_rows<T>(m, m->row_len(), [&](const T* in, T* out) { return rg_rotate<T>(in, out, p, e, shift, lib_stream()); });
}

template <typename T>
int mat_zero_pad(DevMat<T>* m, size_t points, int option)
{
    const size_t step = m->v.complex_ ? 2 : 1, len = points * step, rl = m->row_len();
    if (len <= rl) return BDSP_ERR_ARG_LENGTH;
    const int opt = option == 0 ? 0 : (option == 1 ? 1 : 2);
    const bool c = m->v.complex_;
    return mat_resize_rows<T>(m, len, [&](const T* in, T* out) { return rg_zero_pad<T>(in, out, rl, c, points, opt, lib_stream()); });
}

// convolve_signal with ONE impulse response shared by all rows (matrix/src/time_freq.rs:421-431)
template <typename T>
int mat_convolve_signal(DevMat<T>* m, const DevVec<T>* h)
{
    if (!meta_agrees(&m->v, h)) return BDSP_ERR_META_DATA;
    if (m->v.freq) return BDSP_ERR_MUST_BE_TIME;
    const size_t p = m->row_points();
    if (p < h->points()) return BDSP_ERR_ARG_LENGTH;
    if (p == 0 || h->points() == 0 || m->rows == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    if (m->v.complex_) BDSP_TRY(conv_complex_dev<T>(m->v.data, m->v.buf, p, m->rows, h->data, h->points(), s));
    else BDSP_TRY(conv_real_dev<T>(m->v.data, m->v.buf, p, h->data, h->points(), s, m->rows));
    m->v.trade();
    return BDSP_OK;
}

// convolve_signal with a rows x rows matrix of impulse responses (convolve_mat, time_freq/mod.rs:365-453):
//   out_row[n] = sum_r  row[r] (*) h[n][r]          (MIMO filtering; h row-major [n][r])
template <typename T>
int mat_convolve_mimo(DevMat<T>* m, const DevVec<T>* const* h, size_t count)
{
    const size_t R = m->rows;
    if (count != R * R || R == 0) return BDSP_ERR_ARG_LENGTH; // mod.rs:373-375
    for (size_t i = 0; i < count; ++i) {
        if (h[i]->valid_len != h[0]->valid_len) return BDSP_ERR_ARG_LENGTH; // :384-389
        if (!meta_agrees(&m->v, h[i])) return BDSP_ERR_META_DATA;
    }
    const size_t p = m->row_points(), rl = m->row_len(), hp = h[0]->points();
    if (p == 0 || hp == 0) return BDSP_OK;
    hipStream_t s = lib_stream();
    BDSP_TRY(m->v.reserve(m->v.valid_len));
    WsBlock tmp;
    BDSP_TRY(tmp.alloc(sizeof(T) * rl, s));
    for (size_t n = 0; n < R; ++n) {
        T* out = m->v.buf + n * rl;
        for (size_t r = 0; r < R; ++r) {
            T* dst = r == 0 ? out : tmp.as<T>();
            const T* row = m->v.data + r * rl;
            const DevVec<T>* hh = h[n * R + r];
            if (m->v.complex_) BDSP_TRY(conv_complex_dev<T>(row, dst, p, 1, hh->data, hp, s));
            else BDSP_TRY(conv_real_dev<T>(row, dst, p, hh->data, hp, s));
            if (r) BDSP_TRY(ew_binary<T>(out, tmp.as<T>(), rl, m->v.complex_, 0, s));
        }
    }
    m->v.trade();
    return BDSP_OK;
}

template <typename T>
int mat_interpolatef(DevMat<T>* m, int fid, T rolloff, T factor, T delay, size_t conv_len)
{
    const size_t rl = m->row_len();
    if (rl == 0) return BDSP_OK;
    const size_t new_len = interpolatef_new_len<T>(rl, factor);
    const bool c = m->v.complex_;
    const T delta = m->v.delta;
    return mat_resize_rows<T>(m, new_len, [&](const T* in, T* out) {
        return interpolatef_dev<T>(in, out, rl, c, fid, rolloff, factor, delay, conv_len, delta, lib_stream());
    });
}

template <typename T>
int mat_multiply_frequency_response(DevMat<T>* m, int fid, T rolloff, T ratio)
{
    if (!m->v.freq) { m->v.poison(); return BDSP_OK; }
    const bool c = m->v.complex_;
    return mat_each_row<T>(m, [&](T* row, size_t len) { return ew_freq_response<T>(row, len, c, fid, rolloff, ratio, false, lib_stream()); });
}

template <typename T>
DevVec<T>* mat_get_row(const DevMat<T>* m, size_t row)
{
    if (row >= m->rows) return nullptr;
    DevVec<T>* v = new DevVec<T>();
    v->complex_ = m->v.complex_;
    v->freq = m->v.freq;
    v->delta = m->v.delta;
    const size_t rl = m->row_len();
    if (v->reserve(rl ? rl : 1) != BDSP_OK) { delete v; return nullptr; }
    v->valid_len = rl;
    if (rl) (void)hipMemcpyAsync(v->data, m->v.data + row * rl, sizeof(T) * rl, hipMemcpyDeviceToDevice, lib_stream());
    return v;
}

template <typename T>
int mat_set_row(DevMat<T>* m, size_t row, const DevVec<T>* v)
{
    if (row >= m->rows || v->valid_len != m->row_len()) return BDSP_ERR_ARG_LENGTH;
    if (v->valid_len)
        BDSP_HIP_TRY(hipMemcpyAsync(m->v.data + row * v->valid_len, v->data, sizeof(T) * v->valid_len,
                                    hipMemcpyDeviceToDevice, lib_stream()));
    return BDSP_OK;
}

template <typename T>
int mat_transfer(DevMat<T>* m, T* host, const T* src, size_t len)
{
    if (len != m->v.valid_len) return BDSP_ERR_ARG_LENGTH;
    hipStream_t s = lib_stream();
    if (len) {
        if (src) BDSP_HIP_TRY(hipMemcpyAsync(m->v.data, src, sizeof(T) * len, hipMemcpyHostToDevice, s));
        else BDSP_HIP_TRY(hipMemcpyAsync(host, m->v.data, sizeof(T) * len, hipMemcpyDeviceToHost, s));
    }
    BDSP_HIP_TRY(hipStreamSynchronize(s));
    return BDSP_OK;
}

} // namespace

// ==============================================================================================
// extern "C"
// ==============================================================================================
extern "C" {

int bdsp_hip_has_gpu_support_f32(void) { return device_ready() == BDSP_OK; }
int bdsp_hip_has_gpu_support_f64(void) { return device_ready() == BDSP_OK; }

static int supported_len(int is_complex, size_t len, int f64)
{
    // below the measured crossover the caller's rustfft is faster than a host round trip: the trait's own mechanism
    // (time_freq/mod.rs:41-44 falls back to rustfft when this answers false)
    if (len < g_b1_policy[f64 ? BDSP_B1_FFT_MIN_LEN_F64 : BDSP_B1_FFT_MIN_LEN_F32].load()) return 0;
    // complex only (like ocl/mod.rs:277-281), at least one point, interleaved length even
    return is_complex && len >= 2 && len % 2 == 0 && (len / 2) <= (size_t(1) << 29);
}
int bdsp_hip_is_supported_fft_len_f32(int is_complex, size_t len) { return supported_len(is_complex, len, 0); }
int bdsp_hip_is_supported_fft_len_f64(int is_complex, size_t len) { return supported_len(is_complex, len, 1); }
size_t bdsp_hip_b1_policy_get(int key) { return key >= 0 && key < B1_POLICY_KEYS ? g_b1_policy[key].load() : 0; }
int bdsp_hip_b1_policy_set(int key, size_t value)
{
    if (key < 0 || key >= B1_POLICY_KEYS) return BDSP_ERR_UNSUPPORTED;
    g_b1_policy[key].store(value);
    return BDSP_OK;
}

int bdsp_hip_fft_f32(int is_complex, float* signal, size_t len, int inverse) { return b1_fft<float>(is_complex, signal, len, inverse); }
int bdsp_hip_fft_f64(int is_complex, double* signal, size_t len, int inverse) { return b1_fft<double>(is_complex, signal, len, inverse); }

int bdsp_hip_convolve_vector_f32(int is_complex, const float* src, size_t src_len, float* dst, size_t dst_len,
                                 const float* imp, size_t imp_len, size_t* range_start, size_t* range_end)
{ return b1_convolve<float>(is_complex, src, src_len, dst, dst_len, imp, imp_len, range_start, range_end); }
int bdsp_hip_convolve_vector_f64(int is_complex, const double* src, size_t src_len, double* dst, size_t dst_len,
                                 const double* imp, size_t imp_len, size_t* range_start, size_t* range_end)
{ return b1_convolve<double>(is_complex, src, src_len, dst, dst_len, imp, imp_len, range_start, range_end); }

size_t bdsp_hip_overlap_discard_f32(float* x_time, size_t x_len, float* tmp, size_t tmp_len, float*, size_t,
                                    const float* h_freq, size_t h_len, size_t imp_len, size_t step_size)
{ return b1_overlap_discard<float>(x_time, x_len, tmp, tmp_len, h_freq, h_len, imp_len, step_size); }
size_t bdsp_hip_overlap_discard_f64(double* x_time, size_t x_len, double* tmp, size_t tmp_len, double*, size_t,
                                    const double* h_freq, size_t h_len, size_t imp_len, size_t step_size)
{ return b1_overlap_discard<double>(x_time, x_len, tmp, tmp_len, h_freq, h_len, imp_len, step_size); }

// ---------------------------------------------------------------------------------------------- B2
#define BDSP_FACADE(SFX, T, VB, RES)                                                                        \
    VB* new##SFX(int32_t is_complex, int32_t domain, T init_value, size_t length, T delta)                  \
    { return reinterpret_cast<VB*>(vec_new<T>(is_complex, domain, init_value, length, delta)); }           \
    VB* new_with_performance_options##SFX(int32_t is_complex, int32_t domain, T init_value, size_t length, \
                                          T delta, size_t, int)                                             \
    { return reinterpret_cast<VB*>(vec_new<T>(is_complex, domain, init_value, length, delta)); }           \
    void delete_vector##SFX(VB* vector) { delete H<T>(vector); }                                            \
    VB* bdsp_hip_vec_clone##SFX(const VB* vector) { return reinterpret_cast<VB*>(vec_clone<T>(H<T>(vector))); } \
    VB* clone##SFX(VB* vector)                                                                              \
    { VB* c = reinterpret_cast<VB*>(vec_clone<T>(H<T>(vector))); delete H<T>(vector); return c; }          \
    T get_value##SFX(const VB* vector, size_t index)                                                        \
    {                                                                                                       \
        T out = 0;                                                                                          \
        const DevVec<T>* v = H<T>(vector);                                                                  \
        if (index >= v->valid_len) return std::numeric_limits<T>::quiet_NaN();                             \
        (void)hipMemcpyAsync(&out, v->data + index, sizeof(T), hipMemcpyDeviceToHost, lib_stream());        \
        (void)hipStreamSynchronize(lib_stream());                                                           \
        return out;                                                                                         \
    }                                                                                                       \
    size_t get_len##SFX(const VB* vector) { return H<T>(vector)->valid_len; }                               \
    size_t get_points##SFX(const VB* vector) { return H<T>(vector)->points(); }                             \
    T get_delta##SFX(const VB* vector) { return H<T>(vector)->delta; }                                      \
    int32_t is_complex##SFX(const VB* vector) { return H<T>(vector)->complex_ ? 1 : 0; }                    \
    int32_t get_domain##SFX(const VB* vector) { return H<T>(vector)->freq ? 1 : 0; }                        \
    const T* data##SFX(VB* vector) { return vec_download<T>(H<T>(vector)); }                                \
    void* bdsp_hip_vec_device_ptr##SFX(VB* vector) { return H<T>(vector)->data; }                           \
    RES overwrite_data##SFX(VB* vector, const T* data, size_t len)                                          \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (len > v->valid_len) return finish<T>(v, BDSP_ERR_ARG_LENGTH);                                   \
        hipStream_t s = lib_stream();                                                                       \
        if (len && hipMemcpyAsync(v->data, data, sizeof(T) * len, hipMemcpyHostToDevice, s) != hipSuccess)  \
            return finish<T>(v, BDSP_ERR_HIP);                                                              \
        if (hipStreamSynchronize(s) != hipSuccess) return finish<T>(v, BDSP_ERR_HIP);                       \
        return finish<T>(v, BDSP_OK);                                                                       \
    }                                                                                                       \
    void set_len##SFX(VB* vector, size_t len)                                                               \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector); /* resize (vec_impl_and_indexers.rs): grow within the allocation */    \
        if (v->complex_ && len % 2 != 0) return;                                                            \
        if (v->reserve(len) == BDSP_OK) v->valid_len = len;                                                 \
    }                                                                                                       \
    RES real_offset##SFX(VB* vector, T value)                                                               \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, ew_real_offset<T>(v->data, v->valid_len, v->complex_, value, lib_stream())); } \
    RES real_scale##SFX(VB* vector, T value)                                                                \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, ew_real_scale<T>(v->data, v->valid_len, value, lib_stream())); } \
    RES complex_offset##SFX(VB* vector, T re, T im)                                                         \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (!v->complex_) { v->poison(); return finish<T>(v, BDSP_OK); } /* elementary.rs:273-279 */        \
        return finish<T>(v, ew_complex_offset<T>(v->data, v->valid_len, re, im, lib_stream()));             \
    }                                                                                                       \
    RES complex_scale##SFX(VB* vector, T re, T im)                                                          \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (!v->complex_) { v->poison(); return finish<T>(v, BDSP_OK); }                                    \
        return finish<T>(v, ew_complex_scale<T>(v->data, v->valid_len, re, im, lib_stream()));              \
    }                                                                                                       \
    RES add##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 0)); } \
    RES sub##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 1)); } \
    RES mul##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 2)); } \
    RES div##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 3)); } \
    RES conj##SFX(VB* vector)                                                                               \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (!v->complex_) { v->complex_ = false; v->poison(); return finish<T>(v, BDSP_OK); } /* complex_ops.rs:65-72 */ \
        return finish<T>(v, ew_conj<T>(v->data, v->valid_len, lib_stream()));                               \
    }                                                                                                       \
    RES multiply_complex_exponential##SFX(VB* vector, T a, T b)                                             \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (!v->complex_) { v->poison(); return finish<T>(v, BDSP_OK); }                                    \
        return finish<T>(v, ew_mul_cexp<T>(v->data, v->valid_len, a * v->delta, b * v->delta, lib_stream())); \
    }                                                                                                       \
    RES magnitude##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_complex_to_real<T>(v, 0)); } \
    RES magnitude_squared##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_complex_to_real<T>(v, 1)); } \
    RES to_real##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_complex_to_real<T>(v, 2)); } \
    RES to_imag##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_complex_to_real<T>(v, 3)); } \
    RES phase##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_complex_to_real<T>(v, 4)); } \
    RES to_complex##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_to_complex<T>(v)); } \
    RES reverse##SFX(VB* vector)                                                                            \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        int c = rg_reverse<T>(v->data, v->buf, v->points(), v->complex_ ? 2 : 1, lib_stream());             \
        if (c == BDSP_OK && v->points()) v->trade();                                                        \
        return finish<T>(v, c);                                                                             \
    }                                                                                                       \
    RES swap_halves##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_swap<T>(v, true)); } \
    RES fft_shift##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_swap<T>(v, true)); } \
    RES ifft_shift##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_swap<T>(v, false)); } \
    RES zero_pad##SFX(VB* vector, size_t points, int32_t option) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_zero_pad<T>(v, points, option)); } \
    RES zero_interleave##SFX(VB* vector, int32_t factor) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_zero_interleave<T>(v, factor)); } \
    RES mirror##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_mirror<T>(v)); }     \
    RES apply_window##SFX(VB* vector, int32_t window) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_window<T>(v, window, false)); } \
    RES unapply_window##SFX(VB* vector, int32_t window) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_window<T>(v, window, true)); } \
    RES plain_fft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_fft<T>(v, false, false, -1)); } \
    RES plain_ifft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_fft<T>(v, true, false, -1)); } \
    RES fft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_fft<T>(v, false, true, -1)); } \
    RES ifft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_fft<T>(v, true, true, -1)); } \
    RES windowed_fft##SFX(VB* vector, int32_t window) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_fft<T>(v, false, true, window < 0 ? 3 : window)); } \
    RES windowed_ifft##SFX(VB* vector, int32_t window) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_fft<T>(v, true, true, window < 0 ? 3 : window)); } \
    RES convolve_signal##SFX(VB* vector, const VB* impulse_response)                                        \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_convolve_signal<T>(v, H<T>(impulse_response))); } \
    RES interpolatef##SFX(VB* vector, int32_t impulse_response, T rolloff, T interpolation_factor, T delay, size_t conv_len) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_interpolatef<T>(v, impulse_response == 0 ? 0 : 1, rolloff, interpolation_factor, delay, conv_len)); } \
    RES interpolatei##SFX(VB* vector, int32_t frequency_response, T rolloff, int32_t interpolation_factor)  \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_interpolatei<T>(v, frequency_response == 0 ? 0 : 1, rolloff, interpolation_factor)); } \
    RES interpolate##SFX(VB* vector, int32_t frequency_response, T rolloff, size_t dest_points, T delay)    \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_interpolate<T>(v, frequency_response == 0 ? 0 : 1, rolloff, dest_points, delay)); } \
    RES interpft##SFX(VB* vector, size_t dest_points)                                                       \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_interpolate<T>(v, -1, (T)0, dest_points, (T)0)); } \
    RES decimatei##SFX(VB* vector, uint32_t decimation_factor, uint32_t delay)                              \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_decimatei<T>(v, decimation_factor, delay)); }     \
    RES multiply_frequency_response##SFX(VB* vector, int32_t frequency_response, T rolloff, T ratio)        \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_multiply_frequency_response<T>(v, frequency_response == 0 ? 0 : 1, rolloff, ratio)); } \
    RES plain_sfft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_sfft<T>(v, false, -1)); } \
    RES sfft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_sfft<T>(v, true, -1)); } \
    RES windowed_sfft##SFX(VB* vector, int32_t window) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_sfft<T>(v, true, window < 0 ? 3 : window)); } \
    RES plain_sifft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_sifft<T>(v, false, -1)); } \
    RES sifft##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_sifft<T>(v, true, -1)); } \
    RES windowed_sifft##SFX(VB* vector, int32_t window) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_sifft<T>(v, true, window < 0 ? 3 : window)); } \
    VB* new_with_detailed_performance_options##SFX(int32_t is_complex, int32_t domain, T init_value, size_t length, \
                                                   T delta, size_t, size_t, size_t, size_t, size_t)        \
    { return reinterpret_cast<VB*>(vec_new<T>(is_complex, domain, init_value, length, delta)); }           \
    void set_value##SFX(VB* vector, size_t index, T value)                                                  \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (index >= v->valid_len) return;                                                                  \
        (void)hipMemcpyAsync(v->data + index, &value, sizeof(T), hipMemcpyHostToDevice, lib_stream());      \
        (void)hipStreamSynchronize(lib_stream());                                                           \
    }                                                                                                       \
    size_t get_allocated_len##SFX(const VB* vector) { return H<T>(vector)->cap; }                           \
    const T* complex_data##SFX(VB* vector) { return vec_download<T>(H<T>(vector)); }                        \
    RES complex_divide##SFX(VB* vector, T re, T im)                                                         \
    {                                                                                                       \
        DevVec<T>* v = H<T>(vector);                                                                        \
        if (!v->complex_) { v->poison(); return finish<T>(v, BDSP_OK); }                                    \
        T nn = re * re + im * im; /* Complex::new(1, 0) / Complex::new(re, im), facade32.rs:550-556 */      \
        return finish<T>(v, ew_complex_scale<T>(v->data, v->valid_len, ((T)1 * re + (T)0 * im) / nn, ((T)0 * re - (T)1 * im) / nn, lib_stream())); \
    }                                                                                                       \
    RES add_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 0)); } \
    RES sub_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 1)); } \
    RES mul_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 2)); } \
    RES div_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary<T>(v, H<T>(operand), 3)); } \
    RES add_smaller_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary_smaller<T>(v, H<T>(operand), 0)); } \
    RES sub_smaller_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary_smaller<T>(v, H<T>(operand), 1)); } \
    RES mul_smaller_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary_smaller<T>(v, H<T>(operand), 2)); } \
    RES div_smaller_vector##SFX(VB* vector, const VB* operand) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_binary_smaller<T>(v, H<T>(operand), 3)); } \
    RES prepare_argument##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_prepare_argument<T>(v, false)); } \
    RES prepare_argument_padded##SFX(VB* vector) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_prepare_argument<T>(v, true)); } \
    RES correlate##SFX(VB* vector, const VB* other) { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_correlate<T>(v, H<T>(other))); } \
    RES convolve##SFX(VB* vector, int32_t impulse_response, T rolloff, T ratio, size_t len)                 \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_convolve_function<T>(v, impulse_response == 0 ? 0 : 1, rolloff, ratio, len, nullptr)); } \
    RES convolve_real##SFX(VB* vector, T (*impulse_response)(const void*, T), const void* impulse_response_data, \
                           bool, T ratio, size_t len)                                                       \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_convolve_callback<T>(v, impulse_response, impulse_response_data, ratio, len)); } \
    RES multiply_frequency_response_real##SFX(VB* vector, T (*frequency_response)(const void*, T),          \
                                              const void* frequency_response_data, bool is_symmetric, T ratio) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_frequency_response<T>(v, frequency_response, frequency_response_data, is_symmetric, ratio)); } \
    int32_t get_real##SFX(VB* vector, VB* destination) { return op_get_complex_to_real<T>(H<T>(vector), H<T>(destination), 2); } \
    int32_t get_imag##SFX(VB* vector, VB* destination) { return op_get_complex_to_real<T>(H<T>(vector), H<T>(destination), 3); } \
    int32_t get_magnitude##SFX(VB* vector, VB* destination) { return op_get_complex_to_real<T>(H<T>(vector), H<T>(destination), 0); } \
    int32_t get_magnitude_squared##SFX(VB* vector, VB* destination) { return op_get_complex_to_real<T>(H<T>(vector), H<T>(destination), 1); } \
    int32_t get_phase##SFX(VB* vector, VB* destination) { return op_get_complex_to_real<T>(H<T>(vector), H<T>(destination), 4); } \
    RES interpolate_lin##SFX(VB* vector, T interpolation_factor, T delay)                                   \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_interpolate_real<T>(v, interpolation_factor, delay, false)); } \
    RES interpolate_hermite##SFX(VB* vector, T interpolation_factor, T delay)                               \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_interpolate_real<T>(v, interpolation_factor, delay, true)); } \
    RES apply_custom_window##SFX(VB* vector, T (*window)(const void*, size_t, size_t), const void* window_data, bool is_symmetric) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_window<T>(v, window, window_data, is_symmetric, false)); } \
    RES unapply_custom_window##SFX(VB* vector, T (*window)(const void*, size_t, size_t), const void* window_data, bool is_symmetric) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_window<T>(v, window, window_data, is_symmetric, true)); } \
    RES windowed_custom_fft##SFX(VB* vector, T (*window)(const void*, size_t, size_t), const void* window_data, bool is_symmetric) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_windowed<T>(v, window, window_data, is_symmetric, 0)); } \
    RES windowed_custom_ifft##SFX(VB* vector, T (*window)(const void*, size_t, size_t), const void* window_data, bool is_symmetric) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_windowed<T>(v, window, window_data, is_symmetric, 1)); } \
    RES windowed_custom_sfft##SFX(VB* vector, T (*window)(const void*, size_t, size_t), const void* window_data, bool is_symmetric) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_windowed<T>(v, window, window_data, is_symmetric, 2)); } \
    RES windowed_custom_sifft##SFX(VB* vector, T (*window)(const void*, size_t, size_t), const void* window_data, bool is_symmetric) \
    { DevVec<T>* v = H<T>(vector); return finish<T>(v, op_custom_windowed<T>(v, window, window_data, is_symmetric, 3)); }

BDSP_FACADE(32, float, VecBuf32, VectorInteropResult32)
BDSP_FACADE(64, double, VecBuf64, VectorInteropResult64)
#undef BDSP_FACADE

// ---------------------------------------------------------------------------------------------- math family, pairs, split / merge, callbacks
// (one block per precision, generated from the same template)
VectorInteropResult32 sqrt32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_SQRT, (float)0, false)); }
VectorInteropResult32 square32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_SQUARE, (float)0, false)); }
VectorInteropResult32 ln32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_LN, (float)0, false)); }
VectorInteropResult32 exp32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_EXP, (float)0, false)); }
VectorInteropResult32 sin32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_SIN, (float)0, false)); }
VectorInteropResult32 cos32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_COS, (float)0, false)); }
VectorInteropResult32 tan32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_TAN, (float)0, false)); }
VectorInteropResult32 asin32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ASIN, (float)0, false)); }
VectorInteropResult32 acos32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ACOS, (float)0, false)); }
VectorInteropResult32 atan32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ATAN, (float)0, false)); }
VectorInteropResult32 sinh32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_SINH, (float)0, false)); }
VectorInteropResult32 cosh32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_COSH, (float)0, false)); }
VectorInteropResult32 tanh32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_TANH, (float)0, false)); }
VectorInteropResult32 asinh32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ASINH, (float)0, false)); }
VectorInteropResult32 acosh32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ACOSH, (float)0, false)); }
VectorInteropResult32 atanh32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ATANH, (float)0, false)); }
VectorInteropResult32 abs32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_ABS, (float)0, true)); }
VectorInteropResult32 ln_approx32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_LN, (float)0, true)); }
VectorInteropResult32 exp_approx32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_EXP, (float)0, true)); }
VectorInteropResult32 sin_approx32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_SIN, (float)0, true)); }
VectorInteropResult32 cos_approx32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_COS, (float)0, true)); }
VectorInteropResult32 bdsp_powf32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_POWF, value, false)); }
VectorInteropResult32 log32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_LOG, value, false)); }
VectorInteropResult32 bdsp_expf32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_EXPF, value, false)); }
VectorInteropResult32 wrap32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_WRAP, value, true)); }
VectorInteropResult32 log_approx32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_LOG, value, true)); }
VectorInteropResult32 expf_approx32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_EXPF_APPROX, value, true)); }
VectorInteropResult32 powf_approx32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_POWF_APPROX, value, true)); }
VectorInteropResult32 root32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_math<float>(v, MATH_POWF, (float)1 / value, false)); } // powf(1/degree), trigonometry_and_powers.rs:384-386
VectorInteropResult32 diff32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_diff<float>(v, false)); }
VectorInteropResult32 diff_with_start32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_diff<float>(v, true)); }
VectorInteropResult32 cum_sum32(VecBuf32* vector) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_cum_sum<float>(v)); }
VectorInteropResult32 unwrap32(VecBuf32* vector, float value) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_unwrap<float>(v, value)); }
VectorInteropResult32 map_inplace_real32(VecBuf32* vector, float (*map)(float, size_t))
{
    DevVec<float>* v = H<float>(vector);
    return finish<float>(v, op_map_inplace<float>(v, false, [&](std::vector<float>& h) { for (size_t i = 0; i < h.size(); ++i) h[i] = map(h[i], i); }));
}
VectorInteropResult32 map_inplace_complex32(VecBuf32* vector, bdsp_complex32 (*map)(bdsp_complex32, size_t))
{
    DevVec<float>* v = H<float>(vector);
    return finish<float>(v, op_map_inplace<float>(v, true, [&](std::vector<float>& h) {
        for (size_t i = 0; i + 1 < h.size(); i += 2) { const bdsp_complex32 r = map(bdsp_complex32{h[i], h[i + 1]}, i / 2); h[i] = r.re; h[i + 1] = r.im; }
    }));
}
PointerInteropResult map_aggregate_real32(const VecBuf32* vector, const void* (*map)(float, size_t), const void* (*aggregate)(const void*, const void*))
{
    PointerInteropResult r;
    r.result_code = op_map_aggregate<float>(H<float>(vector), false, &r.result, [&](const std::vector<float>& h) {
        const void* acc = map(h[0], 0);
        for (size_t i = 1; i < h.size(); ++i) acc = aggregate(acc, map(h[i], i));
        return acc;
    });
    return r;
}
PointerInteropResult map_aggregate_complex32(const VecBuf32* vector, const void* (*map)(bdsp_complex32, size_t), const void* (*aggregate)(const void*, const void*))
{
    PointerInteropResult r;
    r.result_code = op_map_aggregate<float>(H<float>(vector), true, &r.result, [&](const std::vector<float>& h) {
        const void* acc = map(bdsp_complex32{h[0], h[1]}, 0);
        for (size_t i = 2; i + 1 < h.size(); i += 2) acc = aggregate(acc, map(bdsp_complex32{h[i], h[i + 1]}, i / 2));
        return acc;
    });
    return r;
}
int32_t get_real_imag32(VecBuf32* vector, VecBuf32* real, VecBuf32* imag) { return op_get_pair<float>(H<float>(vector), H<float>(real), H<float>(imag), 0); }
int32_t get_mag_phase32(VecBuf32* vector, VecBuf32* mag, VecBuf32* phase) { return op_get_pair<float>(H<float>(vector), H<float>(mag), H<float>(phase), 1); }
VectorInteropResult32 set_real_imag32(VecBuf32* vector, const VecBuf32* real, const VecBuf32* imag) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_set_pair<float>(v, H<float>(real), H<float>(imag), 0)); }
VectorInteropResult32 set_mag_phase32(VecBuf32* vector, const VecBuf32* mag, const VecBuf32* phase) { DevVec<float>* v = H<float>(vector); return finish<float>(v, op_set_pair<float>(v, H<float>(mag), H<float>(phase), 1)); }
int32_t split_into32(const VecBuf32* vector, VecBuf32** targets, size_t len)
{ return op_split_into<float>(H<float>(vector), reinterpret_cast<DevVec<float>* const*>(targets), len); }
VectorInteropResult32 merge32(VecBuf32* vector, VecBuf32* const* sources, size_t len)
{ DevVec<float>* v = H<float>(vector); return finish<float>(v, op_merge<float>(v, reinterpret_cast<DevVec<float>* const*>(sources), len)); }
VectorInteropResult32 convolve_complex32(VecBuf32* vector, bdsp_complex32 (*impulse_response)(const void*, float), const void* impulse_response_data, bool, float ratio, size_t len)
{
    DevVec<float>* v = H<float>(vector);
    return finish<float>(v, op_convolve_callback_complex<float>(v, reinterpret_cast<CRet<float> (*)(const void*, float)>(impulse_response), impulse_response_data, ratio, len));
}
VectorInteropResult32 multiply_frequency_response_complex32(VecBuf32* vector, bdsp_complex32 (*frequency_response)(const void*, float), const void* frequency_response_data, bool is_symmetric, float ratio)
{
    DevVec<float>* v = H<float>(vector);
    return finish<float>(v, op_custom_frequency_response_complex<float>(v, reinterpret_cast<CRet<float> (*)(const void*, float)>(frequency_response), frequency_response_data, is_symmetric, ratio));
}
VectorInteropResult32 interpolatef_custom32(VecBuf32* vector, float (*impulse_response)(const void*, float), const void* impulse_response_data, bool, float interpolation_factor, float delay, size_t len)
{ DevVec<float>* v = H<float>(vector); return finish<float>(v, op_interpolatef<float>(v, 0, (float)0, interpolation_factor, delay, len, impulse_response, impulse_response_data)); }
VectorInteropResult32 interpolate_custom32(VecBuf32* vector, float (*frequency_response)(const void*, float), const void* frequency_response_data, bool is_symmetric, size_t dest_points, float delay)
{
    DevVec<float>* v = H<float>(vector);
    Sampler<float> sm; sm.rfn = frequency_response; sm.data = frequency_response_data; sm.symmetric = is_symmetric;
    return finish<float>(v, op_interpolate<float>(v, 0, (float)0, dest_points, delay, &sm));
}
VectorInteropResult32 interpolatei_custom32(VecBuf32* vector, float (*frequency_response)(const void*, float), const void* frequency_response_data, bool is_symmetric, int32_t interpolation_factor)
{
    DevVec<float>* v = H<float>(vector);
    Sampler<float> sm; sm.rfn = frequency_response; sm.data = frequency_response_data; sm.symmetric = is_symmetric;
    return finish<float>(v, op_interpolatei<float>(v, 0, (float)0, interpolation_factor, &sm));
}

VectorInteropResult64 sqrt64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_SQRT, (double)0, false)); }
VectorInteropResult64 square64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_SQUARE, (double)0, false)); }
VectorInteropResult64 ln64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_LN, (double)0, false)); }
VectorInteropResult64 exp64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_EXP, (double)0, false)); }
VectorInteropResult64 sin64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_SIN, (double)0, false)); }
VectorInteropResult64 cos64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_COS, (double)0, false)); }
VectorInteropResult64 tan64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_TAN, (double)0, false)); }
VectorInteropResult64 asin64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ASIN, (double)0, false)); }
VectorInteropResult64 acos64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ACOS, (double)0, false)); }
VectorInteropResult64 atan64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ATAN, (double)0, false)); }
VectorInteropResult64 sinh64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_SINH, (double)0, false)); }
VectorInteropResult64 cosh64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_COSH, (double)0, false)); }
VectorInteropResult64 tanh64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_TANH, (double)0, false)); }
VectorInteropResult64 asinh64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ASINH, (double)0, false)); }
VectorInteropResult64 acosh64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ACOSH, (double)0, false)); }
VectorInteropResult64 atanh64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ATANH, (double)0, false)); }
VectorInteropResult64 abs64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_ABS, (double)0, true)); }
VectorInteropResult64 ln_approx64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_LN, (double)0, true)); }
VectorInteropResult64 exp_approx64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_EXP, (double)0, true)); }
VectorInteropResult64 sin_approx64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_SIN, (double)0, true)); }
VectorInteropResult64 cos_approx64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_COS, (double)0, true)); }
VectorInteropResult64 bdsp_powf64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_POWF, value, false)); }
VectorInteropResult64 log64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_LOG, value, false)); }
VectorInteropResult64 bdsp_expf64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_EXPF, value, false)); }
VectorInteropResult64 wrap64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_WRAP, value, true)); }
VectorInteropResult64 log_approx64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_LOG, value, true)); }
VectorInteropResult64 expf_approx64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_EXPF_APPROX, value, true)); }
VectorInteropResult64 powf_approx64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_POWF_APPROX, value, true)); }
VectorInteropResult64 root64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_math<double>(v, MATH_POWF, (double)1 / value, false)); } // powf(1/degree), trigonometry_and_powers.rs:384-386
VectorInteropResult64 diff64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_diff<double>(v, false)); }
VectorInteropResult64 diff_with_start64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_diff<double>(v, true)); }
VectorInteropResult64 cum_sum64(VecBuf64* vector) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_cum_sum<double>(v)); }
VectorInteropResult64 unwrap64(VecBuf64* vector, double value) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_unwrap<double>(v, value)); }
VectorInteropResult64 map_inplace_real64(VecBuf64* vector, double (*map)(double, size_t))
{
    DevVec<double>* v = H<double>(vector);
    return finish<double>(v, op_map_inplace<double>(v, false, [&](std::vector<double>& h) { for (size_t i = 0; i < h.size(); ++i) h[i] = map(h[i], i); }));
}
VectorInteropResult64 map_inplace_complex64(VecBuf64* vector, bdsp_complex64 (*map)(bdsp_complex64, size_t))
{
    DevVec<double>* v = H<double>(vector);
    return finish<double>(v, op_map_inplace<double>(v, true, [&](std::vector<double>& h) {
        for (size_t i = 0; i + 1 < h.size(); i += 2) { const bdsp_complex64 r = map(bdsp_complex64{h[i], h[i + 1]}, i / 2); h[i] = r.re; h[i + 1] = r.im; }
    }));
}
PointerInteropResult map_aggregate_real64(const VecBuf64* vector, const void* (*map)(double, size_t), const void* (*aggregate)(const void*, const void*))
{
    PointerInteropResult r;
    r.result_code = op_map_aggregate<double>(H<double>(vector), false, &r.result, [&](const std::vector<double>& h) {
        const void* acc = map(h[0], 0);
        for (size_t i = 1; i < h.size(); ++i) acc = aggregate(acc, map(h[i], i));
        return acc;
    });
    return r;
}
PointerInteropResult map_aggregate_complex64(const VecBuf64* vector, const void* (*map)(bdsp_complex64, size_t), const void* (*aggregate)(const void*, const void*))
{
    PointerInteropResult r;
    r.result_code = op_map_aggregate<double>(H<double>(vector), true, &r.result, [&](const std::vector<double>& h) {
        const void* acc = map(bdsp_complex64{h[0], h[1]}, 0);
        for (size_t i = 2; i + 1 < h.size(); i += 2) acc = aggregate(acc, map(bdsp_complex64{h[i], h[i + 1]}, i / 2));
        return acc;
    });
    return r;
}
int32_t get_real_imag64(VecBuf64* vector, VecBuf64* real, VecBuf64* imag) { return op_get_pair<double>(H<double>(vector), H<double>(real), H<double>(imag), 0); }
int32_t get_mag_phase64(VecBuf64* vector, VecBuf64* mag, VecBuf64* phase) { return op_get_pair<double>(H<double>(vector), H<double>(mag), H<double>(phase), 1); }
VectorInteropResult64 set_real_imag64(VecBuf64* vector, const VecBuf64* real, const VecBuf64* imag) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_set_pair<double>(v, H<double>(real), H<double>(imag), 0)); }
VectorInteropResult64 set_mag_phase64(VecBuf64* vector, const VecBuf64* mag, const VecBuf64* phase) { DevVec<double>* v = H<double>(vector); return finish<double>(v, op_set_pair<double>(v, H<double>(mag), H<double>(phase), 1)); }
int32_t split_into64(const VecBuf64* vector, VecBuf64** targets, size_t len)
{ return op_split_into<double>(H<double>(vector), reinterpret_cast<DevVec<double>* const*>(targets), len); }
VectorInteropResult64 merge64(VecBuf64* vector, VecBuf64* const* sources, size_t len)
{ DevVec<double>* v = H<double>(vector); return finish<double>(v, op_merge<double>(v, reinterpret_cast<DevVec<double>* const*>(sources), len)); }
VectorInteropResult64 convolve_complex64(VecBuf64* vector, bdsp_complex64 (*impulse_response)(const void*, double), const void* impulse_response_data, bool, double ratio, size_t len)
{
    DevVec<double>* v = H<double>(vector);
    return finish<double>(v, op_convolve_callback_complex<double>(v, reinterpret_cast<CRet<double> (*)(const void*, double)>(impulse_response), impulse_response_data, ratio, len));
}
VectorInteropResult64 multiply_frequency_response_complex64(VecBuf64* vector, bdsp_complex64 (*frequency_response)(const void*, double), const void* frequency_response_data, bool is_symmetric, double ratio)
{
    DevVec<double>* v = H<double>(vector);
    return finish<double>(v, op_custom_frequency_response_complex<double>(v, reinterpret_cast<CRet<double> (*)(const void*, double)>(frequency_response), frequency_response_data, is_symmetric, ratio));
}
VectorInteropResult64 interpolatef_custom64(VecBuf64* vector, double (*impulse_response)(const void*, double), const void* impulse_response_data, bool, double interpolation_factor, double delay, size_t len)
{ DevVec<double>* v = H<double>(vector); return finish<double>(v, op_interpolatef<double>(v, 0, (double)0, interpolation_factor, delay, len, impulse_response, impulse_response_data)); }
VectorInteropResult64 interpolate_custom64(VecBuf64* vector, double (*frequency_response)(const void*, double), const void* frequency_response_data, bool is_symmetric, size_t dest_points, double delay)
{
    DevVec<double>* v = H<double>(vector);
    Sampler<double> sm; sm.rfn = frequency_response; sm.data = frequency_response_data; sm.symmetric = is_symmetric;
    return finish<double>(v, op_interpolate<double>(v, 0, (double)0, dest_points, delay, &sm));
}
VectorInteropResult64 interpolatei_custom64(VecBuf64* vector, double (*frequency_response)(const void*, double), const void* frequency_response_data, bool is_symmetric, int32_t interpolation_factor)
{
    DevVec<double>* v = H<double>(vector);
    Sampler<double> sm; sm.rfn = frequency_response; sm.data = frequency_response_data; sm.symmetric = is_symmetric;
    return finish<double>(v, op_interpolatei<double>(v, 0, (double)0, interpolation_factor, &sm));
}

// ---------------------------------------------------------------------------------------------- callback bridges
#define BDSP_BRIDGE(SFX, T)                                                                                  \
    bdsp_complex##SFX bdsp_hip_complex_fn_bridge##SFX(const void* bridge, T x)                               \
    {                                                                                                        \
        const bdsp_complex_bridge##SFX* b = static_cast<const bdsp_complex_bridge##SFX*>(bridge);            \
        T out[2] = {0, 0};                                                                                   \
        b->fn(b->ctx, x, out);                                                                               \
        return bdsp_complex##SFX{out[0], out[1]};                                                            \
    }                                                                                                        \
    static thread_local void (*g_map_bridge##SFX)(T, T, size_t, T*) = nullptr;                               \
    void bdsp_hip_set_map_complex_bridge##SFX(void (*fn)(T, T, size_t, T*)) { g_map_bridge##SFX = fn; }      \
    bdsp_complex##SFX bdsp_hip_map_complex_bridge##SFX(bdsp_complex##SFX value, size_t index)                \
    {                                                                                                        \
        T out[2] = {value.re, value.im};                                                                     \
        if (g_map_bridge##SFX) g_map_bridge##SFX(value.re, value.im, index, out);                            \
        return bdsp_complex##SFX{out[0], out[1]};                                                            \
    }
BDSP_BRIDGE(32, float)
BDSP_BRIDGE(64, double)
#undef BDSP_BRIDGE

// ---------------------------------------------------------------------------------------------- statistics
#define BDSP_STATS(SFX, T, VB)                                                                              \
    Statistics##SFX real_statistics##SFX(const VB* vector) { return stats_real<T, Statistics##SFX, T>(H<T>(vector)); } \
    ComplexStatistics##SFX complex_statistics##SFX(const VB* vector) { return stats_complex<T, ComplexStatistics##SFX, T>(H<T>(vector)); } \
    Statistics64 real_statistics_prec##SFX(const VB* vector) { return stats_real<T, Statistics64, double>(H<T>(vector)); } \
    ComplexStatistics64 complex_statistics_prec##SFX(const VB* vector) { return stats_complex<T, ComplexStatistics64, double>(H<T>(vector)); } \
    T real_sum##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), false, 0, &a, &b); return (T)a; } \
    T real_sum_sq##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), false, 1, &a, &b); return (T)a; } \
    double real_sum_prec##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), false, 0, &a, &b); return a; } \
    double real_sum_sq_prec##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), false, 1, &a, &b); return a; } \
    bdsp_complex##SFX complex_sum##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), true, 0, &a, &b); return bdsp_complex##SFX{(T)a, (T)b}; } \
    bdsp_complex##SFX complex_sum_sq##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), true, 1, &a, &b); return bdsp_complex##SFX{(T)a, (T)b}; } \
    bdsp_complex64 complex_sum_prec##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), true, 0, &a, &b); return bdsp_complex64{a, b}; } \
    bdsp_complex64 complex_sum_sq_prec##SFX(const VB* vector) { double a, b; sums<T>(H<T>(vector), true, 1, &a, &b); return bdsp_complex64{a, b}; } \
    ScalarInteropResult##SFX real_dot_product##SFX(const VB* vector, const VB* operand)                     \
    { double a, b; int c = dot<T>(H<T>(vector), H<T>(operand), false, &a, &b); return ScalarInteropResult##SFX{c, (T)a}; } \
    ComplexScalarInteropResult##SFX complex_dot_product##SFX(const VB* vector, const VB* operand)           \
    { double a, b; int c = dot<T>(H<T>(vector), H<T>(operand), true, &a, &b); return ComplexScalarInteropResult##SFX{c, bdsp_complex##SFX{(T)a, (T)b}}; } \
    ScalarInteropResult64 real_dot_product_prec##SFX(const VB* vector, const VB* operand)                   \
    { double a, b; int c = dot<T>(H<T>(vector), H<T>(operand), false, &a, &b); return ScalarInteropResult64{c, a}; } \
    ComplexScalarInteropResult64 complex_dot_product_prec##SFX(const VB* vector, const VB* operand)         \
    { double a, b; int c = dot<T>(H<T>(vector), H<T>(operand), true, &a, &b); return ComplexScalarInteropResult64{c, bdsp_complex64{a, b}}; } \
    int32_t real_statistics_split##SFX(const VB* vector, Statistics##SFX* data, size_t len)                 \
    { return stats_split<T, Statistics##SFX, T, false>(H<T>(vector), data, len); }                         \
    int32_t complex_statistics_split##SFX(const VB* vector, ComplexStatistics##SFX* data, size_t len)       \
    { return stats_split<T, ComplexStatistics##SFX, T, true>(H<T>(vector), data, len); }                   \
    int32_t real_statistics_split_prec##SFX(const VB* vector, Statistics64* data, size_t len)               \
    { return stats_split<T, Statistics64, double, false>(H<T>(vector), data, len); }                       \
    int32_t complex_statistics_split_prec##SFX(const VB* vector, ComplexStatistics64* data, size_t len)     \
    { return stats_split<T, ComplexStatistics64, double, true>(H<T>(vector), data, len); }

BDSP_STATS(32, float, VecBuf32)
BDSP_STATS(64, double, VecBuf64)
#undef BDSP_STATS

// ---------------------------------------------------------------------------------------------- matrix / batch
#define BDSP_MAT(SFX, T, MB, VB)                                                                            \
    static inline DevMat<T>* M##SFX(MB* m) { return reinterpret_cast<DevMat<T>*>(m); }                      \
    static inline const DevMat<T>* MC##SFX(const MB* m) { return reinterpret_cast<const DevMat<T>*>(m); }   \
    MB* bdsp_hip_mat_new##SFX(int32_t is_complex, int32_t domain, size_t rows, size_t row_len, T delta)     \
    { return reinterpret_cast<MB*>(mat_new<T>(is_complex, domain, rows, row_len, delta)); }                \
    void bdsp_hip_mat_delete##SFX(MB* m) { delete M##SFX(m); }                                              \
    size_t bdsp_hip_mat_rows##SFX(const MB* m) { return MC##SFX(m)->rows; }                                 \
    size_t bdsp_hip_mat_row_len##SFX(const MB* m) { return MC##SFX(m)->row_len(); }                         \
    size_t bdsp_hip_mat_row_points##SFX(const MB* m) { return MC##SFX(m)->row_points(); }                   \
    int32_t bdsp_hip_mat_is_complex##SFX(const MB* m) { return MC##SFX(m)->v.complex_ ? 1 : 0; }            \
    int32_t bdsp_hip_mat_get_domain##SFX(const MB* m) { return MC##SFX(m)->v.freq ? 1 : 0; }                \
    T bdsp_hip_mat_get_delta##SFX(const MB* m) { return MC##SFX(m)->v.delta; }                              \
    void* bdsp_hip_mat_device_ptr##SFX(MB* m) { return M##SFX(m)->v.data; }                                 \
    int32_t bdsp_hip_mat_upload##SFX(MB* m, const T* data, size_t len) { return mat_transfer<T>(M##SFX(m), nullptr, data, len); } \
    int32_t bdsp_hip_mat_download##SFX(MB* m, T* out, size_t len) { return mat_transfer<T>(M##SFX(m), out, nullptr, len); } \
    VB* bdsp_hip_mat_get_row##SFX(const MB* m, size_t row) { return reinterpret_cast<VB*>(mat_get_row<T>(MC##SFX(m), row)); } \
    int32_t bdsp_hip_mat_set_row##SFX(MB* m, size_t row, const VB* vector) { return mat_set_row<T>(M##SFX(m), row, H<T>(vector)); } \
    int32_t bdsp_hip_mat_real_scale##SFX(MB* m, T f) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, ew_real_scale<T>(a->v.data, a->v.valid_len, f, lib_stream())); } \
    int32_t bdsp_hip_mat_real_offset##SFX(MB* m, T f) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, ew_real_offset<T>(a->v.data, a->v.valid_len, a->v.complex_, f, lib_stream())); } \
    int32_t bdsp_hip_mat_complex_scale##SFX(MB* m, T re, T im)                                              \
    {                                                                                                       \
        DevMat<T>* a = M##SFX(m);                                                                           \
        if (!a->v.complex_) { a->v.poison(); return BDSP_ERR_POISONED; }                                    \
        return mat_code<T>(a, ew_complex_scale<T>(a->v.data, a->v.valid_len, re, im, lib_stream()));        \
    }                                                                                                       \
    int32_t bdsp_hip_mat_conj##SFX(MB* m)                                                                   \
    {                                                                                                       \
        DevMat<T>* a = M##SFX(m);                                                                           \
        if (!a->v.complex_) { a->v.poison(); return BDSP_ERR_POISONED; }                                    \
        return mat_code<T>(a, ew_conj<T>(a->v.data, a->v.valid_len, lib_stream()));                         \
    }                                                                                                       \
    int32_t bdsp_hip_mat_add##SFX(MB* m, const MB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary<T>(a, MC##SFX(o), 0)); } \
    int32_t bdsp_hip_mat_sub##SFX(MB* m, const MB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary<T>(a, MC##SFX(o), 1)); } \
    int32_t bdsp_hip_mat_mul##SFX(MB* m, const MB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary<T>(a, MC##SFX(o), 2)); } \
    int32_t bdsp_hip_mat_div##SFX(MB* m, const MB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary<T>(a, MC##SFX(o), 3)); } \
    int32_t bdsp_hip_mat_add_vector##SFX(MB* m, const VB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary_vector<T>(a, H<T>(o), 0)); } \
    int32_t bdsp_hip_mat_sub_vector##SFX(MB* m, const VB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary_vector<T>(a, H<T>(o), 1)); } \
    int32_t bdsp_hip_mat_mul_vector##SFX(MB* m, const VB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary_vector<T>(a, H<T>(o), 2)); } \
    int32_t bdsp_hip_mat_div_vector##SFX(MB* m, const VB* o) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_binary_vector<T>(a, H<T>(o), 3)); } \
    int32_t bdsp_hip_mat_magnitude##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_complex_to_real<T>(a, 0)); } \
    int32_t bdsp_hip_mat_magnitude_squared##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_complex_to_real<T>(a, 1)); } \
    int32_t bdsp_hip_mat_to_real##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_complex_to_real<T>(a, 2)); } \
    int32_t bdsp_hip_mat_to_imag##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_complex_to_real<T>(a, 3)); } \
    int32_t bdsp_hip_mat_phase##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_complex_to_real<T>(a, 4)); } \
    int32_t bdsp_hip_mat_plain_fft##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, op_fft<T>(&a->v, false, false, -1, a->rows)); } \
    int32_t bdsp_hip_mat_fft##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, op_fft<T>(&a->v, false, true, -1, a->rows)); } \
    int32_t bdsp_hip_mat_windowed_fft##SFX(MB* m, int32_t window) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, op_fft<T>(&a->v, false, true, window < 0 ? 3 : window, a->rows)); } \
    int32_t bdsp_hip_mat_plain_ifft##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, op_fft<T>(&a->v, true, false, -1, a->rows)); } \
    int32_t bdsp_hip_mat_ifft##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, op_fft<T>(&a->v, true, true, -1, a->rows)); } \
    int32_t bdsp_hip_mat_windowed_ifft##SFX(MB* m, int32_t window) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, op_fft<T>(&a->v, true, true, window < 0 ? 3 : window, a->rows)); } \
    int32_t bdsp_hip_mat_apply_window##SFX(MB* m, int32_t window) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_window<T>(a, window, false)); } \
    int32_t bdsp_hip_mat_unapply_window##SFX(MB* m, int32_t window) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_window<T>(a, window, true)); } \
    int32_t bdsp_hip_mat_swap_halves##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_swap<T>(a, true)); } \
    int32_t bdsp_hip_mat_fft_shift##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_swap<T>(a, true)); } \
    int32_t bdsp_hip_mat_ifft_shift##SFX(MB* m) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_swap<T>(a, false)); } \
    int32_t bdsp_hip_mat_zero_pad##SFX(MB* m, size_t points, int32_t option) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_zero_pad<T>(a, points, option)); } \
    int32_t bdsp_hip_mat_convolve_signal##SFX(MB* m, const VB* impulse_response) { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_convolve_signal<T>(a, H<T>(impulse_response))); } \
    int32_t bdsp_hip_mat_convolve_signal_mat##SFX(MB* m, const VB* const* impulse_responses, size_t count)  \
    { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_convolve_mimo<T>(a, reinterpret_cast<const DevVec<T>* const*>(impulse_responses), count)); } \
    int32_t bdsp_hip_mat_interpolatef##SFX(MB* m, int32_t impulse_response, T rolloff, T interpolation_factor, T delay, size_t conv_len) \
    { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_interpolatef<T>(a, impulse_response == 0 ? 0 : 1, rolloff, interpolation_factor, delay, conv_len)); } \
    int32_t bdsp_hip_mat_multiply_frequency_response##SFX(MB* m, int32_t frequency_response, T rolloff, T ratio) \
    { DevMat<T>* a = M##SFX(m); return mat_code<T>(a, mat_multiply_frequency_response<T>(a, frequency_response == 0 ? 0 : 1, rolloff, ratio)); }

BDSP_MAT(32, float, MatBuf32, VecBuf32)
BDSP_MAT(64, double, MatBuf64, VecBuf64)
#undef BDSP_MAT

// ---------------------------------------------------------------------------------------------- B3
int bdsp_hip_dev_fft(int elem, void* data, void* scratch, size_t points, size_t batch, unsigned flags,
                     double in_scale, int window_id, double window_alpha, int* result_in_scratch, void* stream)
{
    BDSP_TRY(check_device());
    hipStream_t s = pick_stream(stream);
    bool inverse = (flags & BDSP_FFT_INVERSE) != 0;
    unsigned f = flags & (BDSP_FFT_SHIFT_OUT | BDSP_FFT_SHIFT_IN | BDSP_FFT_MAGNITUDE);
    bool in_b = false;
    int c;
    if (elem == 0) {
        int wid = -1; float alpha = (float)window_alpha;
        if (window_id >= 0) { map_window<float>(window_id, &wid, &alpha); if (window_id == 1) alpha = (float)window_alpha; }
        c = fft_two_buffers<float>((float*)data, (float*)scratch, points, batch, inverse, f, (float)in_scale, wid, alpha, &in_b, s);
    } else {
        int wid = -1; double alpha = window_alpha;
        if (window_id >= 0) { map_window<double>(window_id, &wid, &alpha); if (window_id == 1) alpha = window_alpha; }
        c = fft_two_buffers<double>((double*)data, (double*)scratch, points, batch, inverse, f, in_scale, wid, alpha, &in_b, s);
    }
    if (result_in_scratch) *result_in_scratch = in_b ? 1 : 0;
    return c;
}

int bdsp_hip_dev_convolve(int elem, const void* in, void* out, size_t points, size_t batch, const void* taps_dev,
                          size_t taps, void* stream)
{
    BDSP_TRY(check_device());
    if (in == out) { set_last_error("dev_convolve: out must not alias in"); return BDSP_ERR_UNSUPPORTED; }
    if (taps > points) return BDSP_ERR_ARG_LENGTH;
    hipStream_t s = pick_stream(stream);
    if (elem == 0) return conv_complex_dev<float>((const float*)in, (float*)out, points, batch, (const float*)taps_dev, taps, s);
    return conv_complex_dev<double>((const double*)in, (double*)out, points, batch, (const double*)taps_dev, taps, s);
}

int bdsp_hip_dev_convolve_ex(int elem, const void* in, void* out, size_t points, size_t batch, const void* taps_dev,
                             size_t taps, int first_pct, int second_pct, void* stream)
{
    const bool defaults = first_pct < 0 && second_pct < 0;
    if (!defaults && (first_pct < 1 || second_pct < 0 || first_pct + second_pct > 99)) {
        set_last_error("dev_convolve_ex: the shares leave the last dispatch group nothing to do");
        return BDSP_ERR_ARG_LENGTH;
    }
    struct Scope { // the override lives for this call on this thread only
        Scope(int a, int b) { conv_v2_set_shares(a, b); }
        ~Scope() { conv_v2_set_shares(-1, -1); }
    } scope(defaults ? -1 : first_pct, defaults ? -1 : second_pct);
    return bdsp_hip_dev_convolve(elem, in, out, points, batch, taps_dev, taps, stream);
}

size_t bdsp_hip_conv_spectrum_points(void) { return conv_fft_len(0); }

int bdsp_hip_fft_passes(int elem, size_t points)
{
    if (points == 0 || (points & (points - 1)) != 0 || points > (size_t(1) << 30)) return 0;
    // (a PLAIN transform -- what bdsp_hip_dev_fft launches without flags, scale or window; the predicate fft_pow2 itself
    // dispatches on, fft_impl.h: f32 8192 points are one workgroup-resident trip, two passes once an option is fused)
    return elem == 0 ? fft_pow2_plain_trips<float>(points) : fft_pow2_plain_trips<double>(points);
}

int bdsp_hip_dev_conv_prepare(int elem, const void* taps_dev, size_t taps, void* spectrum_dev, void* stream)
{
    BDSP_TRY(check_device());
    hipStream_t s = pick_stream(stream);
    if (elem == 0) return conv_prepare_spectrum<float>((const float*)taps_dev, taps, nullptr, (float*)spectrum_dev, s);
    return conv_prepare_spectrum<double>((const double*)taps_dev, taps, nullptr, (double*)spectrum_dev, s);
}

int bdsp_hip_dev_convolve_prepared(int elem, const void* in, void* out, size_t points, size_t batch,
                                   const void* spectrum_dev, size_t taps, void* stream)
{
    BDSP_TRY(check_device());
    if (in == out) { set_last_error("dev_convolve: out must not alias in"); return BDSP_ERR_UNSUPPORTED; }
    if (taps > points || taps == 0) return BDSP_ERR_ARG_LENGTH;
    hipStream_t s = pick_stream(stream);
    if (elem == 0)
        return conv_run_blocks<float>((const float*)in, (float*)out, points, batch, (const float*)spectrum_dev, taps,
                                      -(long long)(taps / 2), 0, 0, nullptr, s);
    return conv_run_blocks<double>((const double*)in, (double*)out, points, batch, (const double*)spectrum_dev, taps,
                                   -(long long)(taps / 2), 0, 0, nullptr, s);
}

int bdsp_hip_dev_real_scale(int elem, void* data, size_t len, double factor, void* stream)
{
    BDSP_TRY(check_device());
    hipStream_t s = pick_stream(stream);
    if (elem == 0) return ew_real_scale<float>((float*)data, len, (float)factor, s);
    return ew_real_scale<double>((double*)data, len, factor, s);
}

int bdsp_hip_dev_real_offset(int elem, void* data, size_t len, int is_complex, double offset, void* stream)
{
    BDSP_TRY(check_device());
    hipStream_t s = pick_stream(stream);
    if (elem == 0) return ew_real_offset<float>((float*)data, len, is_complex != 0, (float)offset, s);
    return ew_real_offset<double>((double*)data, len, is_complex != 0, offset, s);
}

size_t bdsp_hip_interpolatef_new_len(int elem, size_t len, double factor)
{
    return elem == 0 ? interpolatef_new_len<float>(len, (float)factor) : interpolatef_new_len<double>(len, factor);
}

int bdsp_hip_dev_interpolatef(int elem, const void* in, void* out, size_t len, int is_complex, int function_id,
                              double rolloff, double factor, double delay, size_t conv_len, double delta, void* stream)
{
    BDSP_TRY(check_device());
    hipStream_t s = pick_stream(stream);
    if (elem == 0)
        return interpolatef_dev<float>((const float*)in, (float*)out, len, is_complex != 0, function_id == 0 ? 0 : 1,
                                       (float)rolloff, (float)factor, (float)delay, conv_len, (float)delta, s);
    return interpolatef_dev<double>((const double*)in, (double*)out, len, is_complex != 0, function_id == 0 ? 0 : 1,
                                    rolloff, factor, delay, conv_len, delta, s);
}

int bdsp_hip_synchronize(void* stream)
{
    BDSP_TRY(check_device());
    BDSP_HIP_TRY(hipStreamSynchronize(pick_stream(stream)));
    return BDSP_OK;
}

int bdsp_hip_set_device(int ordinal)
{
    BDSP_HIP_TRY(hipSetDevice(ordinal));
    return device_ready();
}

int bdsp_hip_compute_units(void)
{
    if (check_device() != BDSP_OK) return 0;
    return num_cus();
}


void* bdsp_hip_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
int bdsp_hip_event_record(void* event, void* stream)
{
    BDSP_HIP_TRY(hipEventRecord((hipEvent_t)event, pick_stream(stream)));
    return BDSP_OK;
}
int bdsp_hip_event_elapsed_ms(void* start, void* stop, float* ms)
{
    BDSP_HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    BDSP_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return BDSP_OK;
}
void bdsp_hip_event_destroy(void* event) { if (event) (void)hipEventDestroy((hipEvent_t)event); }

// ---- HIP graphs: capture a sequence of calls once, replay it with one launch -------------------------
// Small vectors are launch-bound (config C1: scale + offset of 65 536 samples is two ~3 us launches
// around ~1 us of work).  Everything this library enqueues after warm-up is plain kernel launches and
// device-to-device copies on one stream -- workspace comes from the per-stream cache, twiddle tables and
// Bluestein plans are built on first use -- so a sequence of B2/B3 calls can be stream-captured.
// Rules: run the sequence once before capturing it (tables, plans, workspace), replay on the stream it
// was captured on (workspace reuse is ordered per stream), and do not capture calls that talk to the
// host (data32, get_value32, overwrite_data32, the B1 entry points, plain_sifft32's symmetry check).
int bdsp_hip_capture_begin(void* stream)
{
    BDSP_TRY(check_device());
    hipStream_t st = pick_stream(stream);
    {
        std::lock_guard<std::mutex> lk(g_bs_mu);
        if (g_capture_open) {
            set_last_error("capture_begin: a capture is already open (finish it with bdsp_hip_capture_end or drop it with bdsp_hip_capture_abort)");
            return BDSP_ERR_UNSUPPORTED;
        }
        BDSP_TRY(ws_capture_begin(st));
        // the stream capture starts UNDER the lock and the bookkeeping is published only once it has: another thread's
        // capture_abort between the two would otherwise see "open" on a stream that is not recording yet, tear the
        // bookkeeping down, and leave the owner's stream in capture mode with nobody able to end it
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
        if (e != hipSuccess) {
            ws_capture_end(st, nullptr);
            set_last_error(hipGetErrorString(e));
            return BDSP_ERR_HIP;
        }
        g_capture_open = 1;
        g_capture_thread = std::this_thread::get_id();
        g_capture_stream = st;
        g_capture_plans.clear();
        g_capture_moves = t_buffer_moves;
    }
    return BDSP_OK;
}
// Drops an open capture without building a graph: ends the stream capture, releases what the captured calls pinned and
// re-opens the library for the next capture_begin.  For a caller whose captured sequence failed half way.  No-op (0)
// when no capture is open.
static int capture_drop(hipStream_t st, bool force)
{
    std::vector<void*> blocks, plans;
    {
        std::lock_guard<std::mutex> lk(g_bs_mu);
        if (!g_capture_open) return BDSP_OK;
        if (g_capture_stream != st) { set_last_error("capture_abort: the open capture is on another stream"); return BDSP_ERR_UNSUPPORTED; }
        // a capture belongs to the thread that opened it (capture_end enforces the same): another thread must not tear it
        // down, and unpin its plans and blocks, while the owner is still recording -- unless the stream is no longer
        // recording anything (the capture was invalidated or ended behind the library's back), or the caller says the
        // owner is gone (bdsp_hip_capture_reset: a thread that exits or dies mid-capture would otherwise leave every
        // later capture_begin of the process refused)
        if (!force && g_capture_thread != std::this_thread::get_id()) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusActive;
            const bool query_ok = hipStreamIsCapturing(st, &cap) == hipSuccess;
            if (!query_ok) (void)hipGetLastError();
            if (!query_ok || cap == hipStreamCaptureStatusActive) {
                set_last_error("capture_abort: the open capture belongs to another thread (bdsp_hip_capture_reset drops it if that thread is gone)");
                return BDSP_ERR_UNSUPPORTED;
            }
        }
        ws_capture_end(st, &blocks);
        plans.swap(g_capture_plans);
        for (void* c : plans)
            for (auto& p : g_bs_plans)
                if (p.chirp == c && p.pins > 0) { --p.pins; break; }
        g_capture_open = 0;
    }
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(st, &g); // (fails harmlessly if the capture was already invalidated)
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    for (void* p : blocks) ws_free(p, st);
    return BDSP_OK;
}
int bdsp_hip_capture_abort(void* stream) { return capture_drop(pick_stream(stream), false); }
int bdsp_hip_capture_reset(void* stream) { return capture_drop(pick_stream(stream), true); }
int bdsp_hip_capture_end(void* stream, void** graph_exec)
{
    if (!graph_exec) return BDSP_ERR_ARG_LENGTH;
    *graph_exec = nullptr;
    hipStream_t st = pick_stream(stream);
    {
        std::lock_guard<std::mutex> lk(g_bs_mu);
        if (!g_capture_open || g_capture_stream != st || g_capture_thread != std::this_thread::get_id()) {
            set_last_error("capture_end: no capture opened by this thread on this stream");
            return BDSP_ERR_UNSUPPORTED;
        }
    }
    hipGraph_t g = nullptr;
    hipError_t ec = hipStreamEndCapture(st, &g);
    GraphHandle* h = new GraphHandle;
    h->stream = st;
    bool moved;
    {
        std::lock_guard<std::mutex> lk(g_bs_mu);
        ws_capture_end(st, &h->pinned_blocks);
        h->pinned_plans.swap(g_capture_plans);
        moved = t_buffer_moves != g_capture_moves;
        g_capture_open = 0;
    }
    auto fail = [&](const char* msg, int code) {
        if (g) (void)hipGraphDestroy(g);
        bdsp_hip_graph_destroy(h);
        set_last_error(msg);
        return code;
    };
    if (ec != hipSuccess) return fail(hipGetErrorString(ec), BDSP_ERR_HIP);
    // a graph replays ADDRESSES: the captured sequence must leave every vector's (live, trade) buffer pair where it
    // found it -- an even number of trades per vector (fft followed by ifft is fine, a lone fft of 2^21 points is not)
    // and no reallocation
    if (moved)
        return fail("capture_end: the captured calls traded or reallocated a vector's buffers an odd number of times; "
                    "a replay would run on stale addresses", BDSP_ERR_UNSUPPORTED);
    hipError_t rc = hipGraphInstantiate(&h->exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    g = nullptr;
    if (rc != hipSuccess) return fail("hipGraphInstantiate failed", BDSP_ERR_HIP);
    *graph_exec = h;
    return BDSP_OK;
}
int bdsp_hip_graph_launch(void* graph_exec, void* stream)
{
    if (!graph_exec) return BDSP_ERR_ARG_LENGTH;
    BDSP_HIP_TRY(hipGraphLaunch(static_cast<GraphHandle*>(graph_exec)->exec, pick_stream(stream)));
    return BDSP_OK;
}
// Destroys the executable graph and releases what it pinned: the workspace blocks the captured calls used (they go
// back to the stream's cache) and its Bluestein plans (evictable again).  The caller makes sure no replay is in flight.
void bdsp_hip_graph_destroy(void* graph_exec)
{
    if (!graph_exec) return;
    GraphHandle* h = static_cast<GraphHandle*>(graph_exec);
    if (h->exec) (void)hipGraphExecDestroy(h->exec);
    for (void* p : h->pinned_blocks) ws_free(p, h->stream);
    {
        std::lock_guard<std::mutex> lk(g_bs_mu);
        for (void* c : h->pinned_plans)
            for (auto& p : g_bs_plans)
                if (p.chirp == c && p.pins > 0) { --p.pins; break; }
    }
    delete h;
}

} // extern "C"
