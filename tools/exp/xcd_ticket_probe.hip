// Probe for a split-K reduction inside the GEMM launch ("Next" 1): can the LAST-arriving workgroup of a group of
// S workgroups read what the others stored, without a kernel boundary and without device-scope cache maintenance, when all S
// sit on ONE XCD (same L2)?
//   group g = S consecutive "work items"; item -> workgroup id so that all items of a group share  wgid % 8  (the XCD the
//   dispatcher deals that id to).  Every workgroup: checks HW_REG_XCC_ID against wgid % 8 (counts mismatches), stores its
//   64 KB slab (value = f(group, member, element, round)), drains its stores (s_waitcnt vmcnt(0)), takes a ticket
//   (relaxed agent-scope atomic add); the one that draws S - 1 loads the other members' slabs with sc0 loads (past the L1),
//   checks every element and resets the ticket.  Reports: XCC mismatches, wrong elements, time per launch.
//   Variant 1 puts the members of a group on DIFFERENT XCDs (wgid = item): the stale-read rate there is the control.
//   Round 5: coherence modes for the ACROSS-XCD placement (argv 4): 0 = as above (plain stores, sc0 loads); 1 = write-through
//   stores and L2-bypassing loads (sc0 sc1 on both: system scope per access, no cache-wide maintenance); 2 = plain stores, an
//   agent-scope release fence before the ticket and an acquire fence behind it (__threadfence: buffer_wbl2 / buffer_inv sc1 —
//   the whole L2 of the XCD), plain loads.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_ticket_probe xcd_ticket_probe.hip ; run: ./xcd_ticket_probe [rounds] [variant] [load] [coherence]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int S = 4;            // members per group (split-K factor)
constexpr int SLAB = 16384;     // floats per member (64 KB: one 128 x 128 fp32 tile)

__device__ __forceinline__ float pattern(int g, int m, int e, int round) {
    return (float)((g * 131 + m * 17 + e + round * 7) & 0xffff);
}

__device__ __forceinline__ float load_sc0(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ float load_sys(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sys(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void probe(float* slabs, unsigned* tickets, unsigned long long* stats, int ngroups, int round,
                                             int variant, int coh) {
    const int wg = blockIdx.x;
    int g, m;
    if (variant == 0) {   // same XCD: xcd = wg & 7 owns groups xcd, xcd + 8, ...; inside an XCD items run group-major
        const int xcd = wg & 7, pos = wg >> 3;
        g = (pos / S) * 8 + xcd;
        m = pos % S;
    } else {              // control: consecutive workgroup ids (= consecutive XCDs) form a group
        g = wg / S;
        m = wg % S;
    }
    if (g >= ngroups) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0 && (int)xcc != (wg & 7)) atomicAdd(&stats[0], 1ull);
    float* mine = slabs + ((size_t)g * S + m) * SLAB;
    if (coh == 1) {
        for (int e = threadIdx.x; e < SLAB; e += 256) store_sys(mine + e, pattern(g, m, e, round));
    } else {
        for (int e = threadIdx.x; e < SLAB; e += 256) mine[e] = pattern(g, m, e, round);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (coh == 2) __threadfence();
    __syncthreads();   // every wave's stores are acknowledged
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(&tickets[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)(S - 1));
    }
    __syncthreads();
    if (!last) return;
    if (coh == 2) __threadfence();
    unsigned long long bad = 0;
    for (int mm = 0; mm < S; ++mm) {
        if (mm == m) continue;
        const float* o = slabs + ((size_t)g * S + mm) * SLAB;
        for (int e = threadIdx.x; e < SLAB; e += 256) {
            const float v = coh == 1 ? load_sys(o + e) : coh == 2 ? __builtin_nontemporal_load(o + e) : load_sc0(o + e);
            if (v != pattern(g, mm, e, round)) ++bad;
        }
    }
    if (bad) atomicAdd(&stats[1], bad);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&tickets[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&stats[2], 1ull);
    }
}

// something else on the chip that dirties the L2s: a streaming copy on a second stream
__global__ void churn(float* a, const float* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i] + 1.f;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    const int variant = argc > 2 ? atoi(argv[2]) : 0;
    const int load = argc > 3 ? atoi(argv[3]) : 1;
    const int coh = argc > 4 ? atoi(argv[4]) : 0;
    const int ngroups = 48;   // 48 output tiles x 4 splits = 192 workgroups, the forward product's launch
    float* slabs; unsigned* tickets; unsigned long long* stats;
    hipMalloc(&slabs, (size_t)ngroups * S * SLAB * 4);
    hipMalloc(&tickets, ngroups * 4);
    hipMalloc(&stats, 3 * 8);
    hipMemset(tickets, 0, ngroups * 4);
    hipMemset(stats, 0, 24);
    float *ca, *cb;
    const size_t cn = 64u << 20;
    hipMalloc(&ca, cn * 4); hipMalloc(&cb, cn * 4);
    hipMemset(cb, 0, cn * 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int nwg = (ngroups + 7) / 8 * 8 * S;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s1);
    for (int r = 0; r < rounds; ++r) {
        if (load && (r % 4) == 0) hipLaunchKernelGGL(churn, dim3(512), dim3(256), 0, s2, ca, cb, cn);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, s1, slabs, tickets, stats, ngroups, r, variant, coh);
    }
    hipEventRecord(e1, s1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    hipMemcpy(h, stats, 24, hipMemcpyDeviceToHost);
    printf("coherence %d, variant %d (%s), %d rounds, load %d: XCC_ID != wgid %% 8: %llu | wrong elements: %llu | groups finished: %llu of %d | %.2f us per launch\n",
           coh, variant, variant == 0 ? "group on one XCD" : "group across XCDs", rounds, load, h[0], h[1], h[2], rounds * ngroups,
           ms * 1e3 / rounds);
    return 0;
}
