// Micro-benchmark: can one pass of a TWO-pass 2^24-point FFT (4096-point columns, 16-column tiles = 128-byte
// runs) be run by CLUSTERS of 16 workgroups that exchange through their XCD's L2 inside one launch?
//
// The data movement of such a pass without the butterflies: matrix in[row][col], 4096 x 4096 complex f32.
// A cluster owns 16 adjacent columns.  Member b loads rows {b + 16 a : a < 256} (what the inner 256-point
// sub-transform over a needs), writes its 4096 values into the cluster's OUTPUT tile at rows {a + 256 b}
// (plain stores: they stop in the XCD's L2), the cluster meets at a counter barrier, then member m re-reads rows
// {a + 256 b : 16 m <= a < 16 m + 16, all b} -- the 16 values per (a, column) a radix-16 butterfly over b
// needs, written by all 16 members -- and stores them back to the same rows (in place: every row set is read and
// rewritten by the same member).  HBM sees one read of `in` and one write of `out`; the exchange is L2 traffic.
// Correctness of the hand-off is checked on the host: out[(a + 256 b)][c] must equal in[(b + 16 a)][c].
//
// Clusters are formed at run time from workgroups that read the same HW_REG_XCC_ID (a ticket per XCD), so that
// "same L2" is a fact the kernel observed, not an assumption about dispatch order.  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

struct Ctl {
    unsigned tickets[8];
    unsigned pad0[8];
    unsigned bar[128];  // one monotonic counter per cluster slot
    unsigned error;
    unsigned maxwait;
};

constexpr unsigned NR = 4096, NC = 4096, CL = 16; // rows, columns, workgroups per cluster
constexpr unsigned SPIN_LIMIT = 1u << 22;

// MODE 0: cluster exchange through L2, re-read with sc1 (agent-scope relaxed atomic) loads
// MODE 1: the same, re-read with plain loads after an agent-scope acquire fence
// MODE 2: no exchange -- every workgroup just copies its 256 rows (the plain-pass baseline)
template <int MODE>
__global__ __launch_bounds__(256) void k_cluster(const f2* __restrict__ in, f2* __restrict__ out, Ctl* ctl,
                                                 unsigned clusters_per_xcd, unsigned base_iter)
{
    __shared__ unsigned s_xcc, s_ticket;
    const unsigned tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        s_xcc = xcc;
        s_ticket = atomicAdd(&ctl->tickets[xcc], 1u) - base_iter * (gridDim.x / 8);
    }
    __syncthreads();
    const unsigned xcc = s_xcc, ticket = s_ticket;
    const unsigned cid = ticket / CL, member = ticket % CL;
    if (cid >= clusters_per_xcd) { // more workgroups on this XCD than the launch assumed
        if (tid == 0) atomicOr(&ctl->error, 1u);
        return;
    }
    const unsigned slot = xcc * clusters_per_xcd + cid, nslots = 8 * clusters_per_xcd;
    const unsigned c = tid & 15, ti = tid >> 4;
    unsigned iter = 0;
    const unsigned ipl = (NC / 16 - slot + nslots - 1) / nslots; // tiles this slot takes per launch
    for (unsigned cg = slot; cg < NC / 16; cg += nslots, ++iter) {
        const f2* ib = in + cg * 16 + c;
        f2* ob = out + cg * 16 + c;
        f2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ib[(size_t)(member + 16 * (ti + 16 * r)) * NC];
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ob[(size_t)((ti + 16 * r) + 256 * member) * NC] = v[r];
            continue;
        }
        // intermediate: value (a, b = member) to row a + 256 b
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(size_t)((ti + 16 * r) + 256 * member) * NC] = v[r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores have reached L2
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (base_iter * ipl + iter + 1) * CL;
            __hip_atomic_fetch_add(&ctl->bar[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(&ctl->bar[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT) { atomicOr(&ctl->error, 2u); break; }
            }
            atomicMax(&ctl->maxwait, spins);
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // member m takes a in [16 m, 16 m + 16): thread (a_l = ti, c) reads b = r
        const unsigned a = 16 * member + ti;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f2* p = ob + (size_t)(a + 256 * r) * NC;
            if (MODE == 0) {
                unsigned long long raw = __hip_atomic_load(reinterpret_cast<unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                v[r] = __builtin_bit_cast(f2, raw);
            } else {
                v[r] = *p;
            }
        }
        // (radix-16 butterfly over r would sit here) final values back to the same rows
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(size_t)(a + 256 * r) * NC] = v[r] + f2{1.0f, 0.0f};
    }
}

int main(int argc, char** argv)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t n = (size_t)NR * NC;
    f2 *in[2], *out;
    std::vector<f2> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2{(float)(i & 0xffffff), (float)(i >> 12)};
    for (auto& p : in) { hipMalloc(&p, sizeof(f2) * n); hipMemcpy(p, h.data(), sizeof(f2) * n, hipMemcpyHostToDevice); }
    hipMalloc(&out, sizeof(f2) * n);
    Ctl* ctl;
    hipMalloc(&ctl, sizeof(Ctl));
    std::vector<f2> ho(n);
    for (int mode = 0; mode < 3; ++mode) {
        for (int per_cu = 2; per_cu <= 4; ++per_cu) {
            const unsigned grid = (unsigned)cus * per_cu;
            if (grid % 128) continue;
            const unsigned cpx = grid / 8 / CL;
            hipMemset(ctl, 0, sizeof(Ctl));
            hipMemset(out, 0, sizeof(f2) * n);
            unsigned launches = 0;
            auto launch = [&](int i) {
                if (mode == 0) hipLaunchKernelGGL(k_cluster<0>, dim3(grid), dim3(256), 0, 0, in[i & 1], out, ctl, cpx, launches);
                else if (mode == 1) hipLaunchKernelGGL(k_cluster<1>, dim3(grid), dim3(256), 0, 0, in[i & 1], out, ctl, cpx, launches);
                else hipLaunchKernelGGL(k_cluster<2>, dim3(grid), dim3(256), 0, 0, in[i & 1], out, ctl, cpx, launches);
                ++launches;
            };
            launch(0);
            hipDeviceSynchronize();
            Ctl hc;
            hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
            hipMemcpy(ho.data(), out, sizeof(f2) * n, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (unsigned a = 0; a < 256; ++a)
                for (unsigned b = 0; b < 16; ++b)
                    for (unsigned col = 0; col < NC; col += 1) {
                        f2 got = ho[(size_t)(a + 256 * b) * NC + col], want = h[(size_t)(b + 16 * a) * NC + col];
                        if (mode != 2) want.x += 1.0f;
                        if (got.x != want.x || got.y != want.y) ++bad;
                    }
            printf("mode %d wg/CU %d: first launch error=%u tickets=%u,%u,%u,%u,%u,%u,%u,%u maxwait=%u mismatches=%zu\n", mode,
                   per_cu, hc.error, hc.tickets[0], hc.tickets[1], hc.tickets[2], hc.tickets[3], hc.tickets[4],
                   hc.tickets[5], hc.tickets[6], hc.tickets[7], hc.maxwait, bad);
            if (hc.error) continue;
            for (int i = 0; i < 5; ++i) launch(i);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) launch(i);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
            printf("mode %d wg/CU %d: %.1f us per pass (%.2f TB/s of HBM-side traffic), error=%u maxwait=%u\n", mode, per_cu,
                   ms * 1e3 / reps, 2.0 * n * 8 / (ms * 1e-3 / reps) / 1e12, hc.error, hc.maxwait);
        }
    }
    return 0;
}
