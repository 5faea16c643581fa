// runtime.cpp -- device context, stream, workspace cache and twiddle-table cache.
//
// The reference's OpenCL backend builds a context, a queue and a plan on EVERY call
// (vector/src/gpu_support/ocl/mod.rs:301-357, 335-349); here they are created once per device and
// reused, and scratch memory is recycled instead of hipMalloc'ed per call.
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "bdsp_internal.h"

namespace bdsp {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    g_last_error = std::string(hipGetErrorString(e)) + " in " + what + " at " + file + ":" +
                   std::to_string(line);
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? BDSP_ERR_NO_DEVICE : BDSP_ERR_HIP;
}

namespace {

struct DeviceCtx {
    hipStream_t stream = nullptr;
    int cus = 0;
    // free workspace blocks per stream, keyed by capacity
    std::map<hipStream_t, std::multimap<size_t, void*>> free_blocks;
    std::map<void*, size_t> block_size;
    // streams whose work is being captured into a HIP graph: blocks released meanwhile are PINNED to the graph
    // (a replay writes to the addresses it recorded, so they must not be handed to anybody else before the
    // graph is destroyed)
    std::map<hipStream_t, std::vector<void*>> capturing;
    std::map<std::pair<int, int>, void*> twiddles; // (n, sizeof(T)) -> device table
    size_t cached_bytes = 0; // sum of the free blocks' capacities
    size_t cache_limit = 0;  // a quarter of the device memory: beyond it blocks go back to the driver
};

std::mutex g_mu;
std::map<int, DeviceCtx> g_ctx;
int g_probe = -2; // -2 unknown, BDSP_OK, or error
std::string g_probe_msg; // why the probe failed: repeated to EVERY later caller (last_error is per thread and per call)

int probe_locked()
{
    if (g_probe != -2) {
        if (g_probe != BDSP_OK) g_last_error = g_probe_msg;
        return g_probe;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_probe_msg = std::string("no HIP device: ") + hipGetErrorString(e);
        g_last_error = g_probe_msg;
        (void)hipGetLastError();
        g_probe = BDSP_ERR_NO_DEVICE;
        return g_probe;
    }
    g_probe = BDSP_OK;
    return g_probe;
}

// caller holds g_mu
int ctx_locked(DeviceCtx** out)
{
    int c = probe_locked();
    if (c != BDSP_OK) return c;
    int dev = 0;
    BDSP_HIP_TRY(hipGetDevice(&dev));
    auto it = g_ctx.find(dev);
    if (it == g_ctx.end()) {
        DeviceCtx ctx;
        hipDeviceProp_t prop;
        BDSP_HIP_TRY(hipGetDeviceProperties(&prop, dev));
        std::string arch = prop.gcnArchName;
        if (arch.rfind("gfx950", 0) != 0) {
            g_last_error = "device " + std::to_string(dev) + " is " + arch +
                           "; this library carries gfx950 code objects only";
            return BDSP_ERR_NO_DEVICE;
        }
        ctx.cus = prop.multiProcessorCount;
        ctx.cache_limit = (size_t)prop.totalGlobalMem / 4;
        BDSP_HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
        it = g_ctx.emplace(dev, ctx).first;
    }
    *out = &it->second;
    return BDSP_OK;
}

} // namespace

int device_ready()
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    return ctx_locked(&c);
}

hipStream_t lib_stream()
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    if (ctx_locked(&c) != BDSP_OK) return nullptr;
    return c->stream;
}

int num_cus()
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    if (ctx_locked(&c) != BDSP_OK) return 256;
    return c->cus > 0 ? c->cus : 256;
}

int ws_alloc(void** p, size_t bytes, hipStream_t stream)
{
    *p = nullptr;
    if (bytes == 0) bytes = 256;
    // round up: 256 B granules below 1 MiB, 1 MiB granules above (keeps reuse likely)
    size_t g = bytes < (1u << 20) ? 256 : (1u << 20);
    size_t cap = (bytes + g - 1) / g * g;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    BDSP_TRY(ctx_locked(&c));
    auto& fl = c->free_blocks[stream];
    auto it = fl.lower_bound(cap);
    if (it != fl.end() && it->first <= cap + cap / 2 + (1u << 16)) {
        *p = it->second;
        c->cached_bytes -= it->first;
        fl.erase(it);
        return BDSP_OK;
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, cap);
    if (e != hipSuccess) {
        // drop every cached block on this device and retry once
        (void)hipGetLastError();
        for (auto& kv : c->free_blocks) {
            for (auto& b : kv.second) {
                (void)hipFree(b.second);
                c->block_size.erase(b.second);
            }
            kv.second.clear();
        }
        c->cached_bytes = 0;
        e = hipMalloc(&q, cap);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(workspace)", __FILE__, __LINE__);
    }
    c->block_size[q] = cap;
    *p = q;
    return BDSP_OK;
}

void ws_free(void* p, hipStream_t stream)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    if (ctx_locked(&c) != BDSP_OK) return;
    auto it = c->block_size.find(p);
    if (it == c->block_size.end()) return;
    auto cap = c->capturing.find(stream);
    if (cap != c->capturing.end()) { cap->second.push_back(p); return; }
    if (c->cache_limit && c->cached_bytes + it->second > c->cache_limit) {
        // the cache is full: hand the block back (hipFree waits for the device, so work that still uses
        // the block has finished) instead of letting the cache grow without bound
        (void)hipFree(p);
        c->block_size.erase(it);
        return;
    }
    c->cached_bytes += it->second;
    c->free_blocks[stream].emplace(it->second, p);
}

int ws_capture_begin(hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    BDSP_TRY(ctx_locked(&c));
    if (c->capturing.count(stream)) { g_last_error = "a capture is already open on this stream"; return BDSP_ERR_UNSUPPORTED; }
    c->capturing[stream];
    return BDSP_OK;
}

void ws_capture_end(hipStream_t stream, std::vector<void*>* pinned)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    if (ctx_locked(&c) != BDSP_OK) return;
    auto it = c->capturing.find(stream);
    if (it == c->capturing.end()) return;
    if (pinned) pinned->swap(it->second);
    c->capturing.erase(it);
}

template <typename T>
int twiddle_table(int n, const cpx<T>** table)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    BDSP_TRY(ctx_locked(&c));
    auto key = std::make_pair(n, (int)sizeof(T));
    auto it = c->twiddles.find(key);
    if (it != c->twiddles.end()) {
        *table = reinterpret_cast<const cpx<T>*>(it->second);
        return BDSP_OK;
    }
    std::vector<cpx<T>> host((size_t)n);
    for (int m = 0; m < n; ++m) {
        // exact octant symmetries keep cos/sin consistent: evaluate through long double
        long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
        host[m].x = (T)cosl(a);
        host[m].y = (T)sinl(a);
    }
    void* d = nullptr;
    BDSP_HIP_TRY(hipMalloc(&d, sizeof(cpx<T>) * (size_t)n));
    BDSP_HIP_TRY(hipMemcpy(d, host.data(), sizeof(cpx<T>) * (size_t)n, hipMemcpyHostToDevice));
    c->twiddles[key] = d;
    *table = reinterpret_cast<const cpx<T>*>(d);
    return BDSP_OK;
}

// The tables are never freed (kernels in flight may be reading them), so callers that can live without one
// (mixed_radix.hip falls back to Bluestein, whose plan cache is LRU-bounded) ask first: a program that walks
// through thousands of distinct lengths must not grow the cache without bound.
template <typename T>
bool twiddle_table_available(int n)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* c;
    if (ctx_locked(&c) != BDSP_OK) return false;
    return c->twiddles.size() < 1024 || c->twiddles.count(std::make_pair(n, (int)sizeof(T))) != 0;
}
template bool twiddle_table_available<float>(int);
template bool twiddle_table_available<double>(int);

template int twiddle_table<float>(int, const cpx<float>**);
template int twiddle_table<double>(int, const cpx<double>**);

bool is_pow2(size_t n) { return n != 0 && (n & (n - 1)) == 0; }

} // namespace bdsp

extern "C" const char* bdsp_hip_last_error(void) { return bdsp::g_last_error.c_str(); }
extern "C" const char* bdsp_hip_version(void) { return "basic_dsp_hip 0.1.0 (gfx950)"; }
