// Does a 12-byte load (global_load_dwordx3) issued among 16-byte loads deliver its data after the counted
// s_waitcnt vmcnt(N) that in-order return says covers it?  (profiles/r03_experiments.txt 7: bps_mfma_kernel's exact stage.)
//
// Every lane runs ITER rounds.  A round issues, in THIS order (inline asm, so neither order nor waits are the
// compiler's): 2 16-byte loads from a small, cache-resident array, ONE 3-float load from a large array (a different row
// every round, so it tends to miss the vector L1), 10 more 16-byte loads — then s_waitcnt vmcnt(10): if loads return in
// order, the first three have landed.  The 3-float row holds (k, k + 0.25, k + 0.5) for row k and its destination
// registers are poisoned (-1) before the round: a round that reads anything else after the counted wait is counted as
// "early"; the same registers after vmcnt(0) are checked too ("final").
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/late_load_probe tools/exp/late_load_probe.hip && /tmp/late_load_probe [iters] [lds bytes]
//   -DX3=0: the 12-byte row as dwordx2 + dword.   -DCLOBBER=1: the row's address registers are overwritten by a vector
//   move right after the next load's issue (is the address read protected?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef X3
#define X3 1
#endif
#ifndef CLOBBER   // 1: a vector move overwrites the 12-byte load's ADDRESS registers one load after its issue
#define CLOBBER 0
#endif
#if CLOBBER
#define CLOBBER_ASM "v_mov_b64 %[cp], %[hp]\n\t"
#else
#define CLOBBER_ASM
#endif

__global__ __launch_bounds__(512) void probe(const float* __restrict__ hot, int hot_rows, const float* __restrict__ cold,
                                             int cold_rows, int iters, unsigned long long* bad, float* sink) {
    extern __shared__ float lds[];  // occupancy control + some LDS traffic beside the loads, as in the BPS kernel
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned state = (blockIdx.x * 512 + tid) * 2654435761u + 12345u;
    float acc = 0.f;
    unsigned long long early = 0, fin = 0, early_hi = 0;
    for (int it = 0; it < iters; ++it) {
        state = state * 1664525u + 1013904223u;
        const int hr = (state >> 8) % (unsigned)(hot_rows - 16);
        const int cr = (state >> 4) % (unsigned)cold_rows;
        const float* hp = hot + (size_t)hr * 12;
        const float* cp = cold + (size_t)cr * 3;
        f32x4 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11;
#if X3
        f32x3 xyz = {-1.f, -1.f, -1.f};
        asm volatile(
            "global_load_dwordx4 %[p0], %[hp], off offset:16\n\t"
            "global_load_dwordx4 %[p1], %[hp], off\n\t"
            "global_load_dwordx3 %[xyz], %[cp], off\n\t"
#else
        f32x2 xy = {-1.f, -1.f};
        float zz = -1.f;
        asm volatile(
            "global_load_dwordx4 %[p0], %[hp], off offset:16\n\t"
            "global_load_dwordx4 %[p1], %[hp], off\n\t"
            "global_load_dwordx2 %[xy], %[cp], off\n\t"
            "global_load_dword %[zz], %[cp], off offset:8\n\t"
#endif
            "global_load_dwordx4 %[p2], %[hp], off offset:32\n\t"
            CLOBBER_ASM
            "global_load_dwordx4 %[p3], %[hp], off offset:48\n\t"
            "global_load_dwordx4 %[p4], %[hp], off offset:64\n\t"
            "global_load_dwordx4 %[p5], %[hp], off offset:80\n\t"
            "global_load_dwordx4 %[p6], %[hp], off offset:96\n\t"
            "global_load_dwordx4 %[p7], %[hp], off offset:112\n\t"
            "global_load_dwordx4 %[p8], %[hp], off offset:128\n\t"
            "global_load_dwordx4 %[p9], %[hp], off offset:144\n\t"
            "global_load_dwordx4 %[p10], %[hp], off offset:160\n\t"
            "global_load_dwordx4 %[p11], %[hp], off offset:176\n\t"
            "s_waitcnt vmcnt(10)"
            : [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4), [p5] "=&v"(p5),
              [p6] "=&v"(p6), [p7] "=&v"(p7), [p8] "=&v"(p8), [p9] "=&v"(p9), [p10] "=&v"(p10), [p11] "=&v"(p11),
#if X3
              [xyz] "+&v"(xyz)
#else
              [xy] "+&v"(xy), [zz] "+&v"(zz)
#endif
#if CLOBBER
              , [cp] "+&v"(cp)
            : [hp] "v"(hp)
#else
            : [hp] "v"(hp), [cp] "v"(cp)
#endif
            : "memory");
        // after the counted wait: the first three (four) loads have landed IF loads return in order
#if X3
        float ex = xyz[0], ey = xyz[1], ez = xyz[2];
#else
        float ex = xy[0], ey = xy[1], ez = zz;
#endif
        asm volatile("" : "+v"(ex), "+v"(ey), "+v"(ez));  // read them NOW
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(p8), "+v"(p9),
                       "+v"(p10), "+v"(p11),
#if X3
                       "+v"(xyz)
#else
                       "+v"(xy), "+v"(zz)
#endif
                     :: "memory");
#if X3
        const float fx = xyz[0], fy = xyz[1], fz = xyz[2];
#else
        const float fx = xy[0], fy = xy[1], fz = zz;
#endif
        const float s = p0[0] + p1[1] + p2[2] + p3[3] + p4[0] + p5[1] + p6[2] + p7[3] + p8[0] + p9[1] + p10[2] + p11[3];
        acc += s;
        lds[(tid * 33 + it) & 8191] = s;
        const float e = (float)cr;
        const bool eb = ex != e || ey != e + 0.25f || ez != e + 0.5f;
        early += eb;
        early_hi += eb && lane >= 32;
        fin += fx != e || fy != e + 0.25f || fz != e + 0.5f;
    }
    if (early) atomicAdd(bad, early);
    if (early_hi) atomicAdd(bad + 1, early_hi);
    if (fin) atomicAdd(bad + 2, fin);
    if (acc == 1234.5f) sink[0] = acc + lds[lane];
}

int main(int argc, char** argv) {
    const int hot_rows = 2048, cold_rows = 4 << 20, iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int lds_bytes = argc > 2 ? atoi(argv[2]) : 32768;  // 32 KB: several workgroups per CU
    std::vector<float> h(hot_rows * 12), c((size_t)cold_rows * 3);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 0.001f;
    for (int k = 0; k < cold_rows; ++k) c[3 * (size_t)k] = (float)k, c[3 * (size_t)k + 1] = k + 0.25f, c[3 * (size_t)k + 2] = k + 0.5f;
    float *dh, *dc, *sink;
    unsigned long long* bad;
    hipMalloc(&dh, h.size() * 4);
    hipMalloc(&dc, c.size() * 4);
    hipMalloc(&sink, 4);
    hipMalloc(&bad, 24);
    hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(bad, 0, 24);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(512), lds_bytes, 0, dh, hot_rows, dc, cold_rows, iters, bad, sink);
        unsigned long long nb[3] = {0, 0, 0};
        hipMemcpy(nb, bad, 24, hipMemcpyDeviceToHost);
        printf("X3=%d CLOBBER=%d rep %d: after vmcnt(10) %llu rounds read a row that had not landed (%llu of them in lanes 32-63), "
               "after vmcnt(0) %llu, of %.3g rounds\n", X3, CLOBBER, rep, nb[0], nb[1], nb[2], 4096.0 * 512 * iters);
    }
    return 0;
}
