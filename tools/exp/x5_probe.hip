// Round 5 probe: what does a k-step cost when ONE wave owns a 128 x 128 accumulator block (16 MFMA tiles, 256 accumulator
// registers) and splits its 8 fp32 fragments itself?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -w -fno-slp-vectorize tools/exp/x5_probe.hip -o tools/exp/x5_probe && tools/exp/x5_probe
// gemm_x3_kernel gives every wave a 64 x 64 block: 4 fragments split per 24 MFMAs (7.3 vector instructions per MFMA) and
// measures 57 cycles per MFMA against the pipe's 32.  With a 128 x 128 block per wave — the workgroup's four waves take
// different k-steps of the same output tile and add up at the end — it is 8 fragments per 96 MFMAs: 3.7 per MFMA, under the
// ~5 single-issue instructions an MFMA gap hides (MI355X_MICROARCH.md, cycle constants).  This probe runs that instruction
// mix per k-step and wave: 96 MFMAs, 8 fragment splits (44 vector instructions each) cut into 88 slices of 4, 16
// ds_read_b128, 16 LDS-DMA pieces of 1 KiB, one barrier; timing only (the operands are not a real GEMM's).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <type_traits>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f4v lds_f4;

__device__ __forceinline__ unsigned pack(float a, float b) {
    f2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

struct Split {   // one fragment in flight: 8 raw values -> 12 packed registers, 11 slices of 4 vector instructions
    float x[8], r[8], e[4];
    unsigned h[4], m[4], l[4];
};

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): no unroll pragma, no size threshold
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int k>
__device__ __forceinline__ void slice(Split& s) {
    switch (k) {
    case 0: for (int p = 0; p < 4; ++p) s.h[p] = pack(s.x[2 * p], s.x[2 * p + 1]); break;
    case 1: for (int p = 0; p < 2; ++p) { s.e[2 * p] = __builtin_bit_cast(float, s.h[p] << 16); s.e[2 * p + 1] = __builtin_bit_cast(float, s.h[p] & 0xffff0000u); } break;
    case 2: for (int j = 0; j < 4; ++j) s.r[j] = s.x[j] - s.e[j]; break;
    case 3: for (int p = 2; p < 4; ++p) { s.e[2 * p - 4] = __builtin_bit_cast(float, s.h[p] << 16); s.e[2 * p - 3] = __builtin_bit_cast(float, s.h[p] & 0xffff0000u); } break;
    case 4: for (int j = 0; j < 4; ++j) s.r[4 + j] = s.x[4 + j] - s.e[j]; break;
    case 5: for (int p = 0; p < 4; ++p) s.m[p] = pack(s.r[2 * p], s.r[2 * p + 1]); break;
    case 6: for (int p = 0; p < 2; ++p) { s.e[2 * p] = __builtin_bit_cast(float, s.m[p] << 16); s.e[2 * p + 1] = __builtin_bit_cast(float, s.m[p] & 0xffff0000u); } break;
    case 7: for (int j = 0; j < 4; ++j) s.r[j] -= s.e[j]; break;
    case 8: for (int p = 2; p < 4; ++p) { s.e[2 * p - 4] = __builtin_bit_cast(float, s.m[p] << 16); s.e[2 * p - 3] = __builtin_bit_cast(float, s.m[p] & 0xffff0000u); } break;
    case 9: for (int j = 0; j < 4; ++j) s.r[4 + j] -= s.e[j]; break;
    case 10: for (int p = 0; p < 4; ++p) s.l[p] = pack(s.r[2 * p], s.r[2 * p + 1]); break;
    default: break;
    }
}

// The same split with v_dot2c_f32_bf16 forming the residuals: r = x - hi is ONE instruction, x + (-1) * hi_half + 0 * other_half
// (exact: the products are exact and the sum is representable), instead of widening the bf16 half back to f32 (shift / and)
// and subtracting: 28 vector instructions per fragment instead of 44, 7 slices of 4.
__device__ __forceinline__ float resid(unsigned pk, int half, float x) {
    typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
    const bf2v m0 = {(__bf16)-1.0f, (__bf16)0.0f}, m1 = {(__bf16)0.0f, (__bf16)-1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, pk), half ? m1 : m0, x, false);
}
template <int k>
__device__ __forceinline__ void slice2(Split& s) {
    switch (k) {
    case 0: for (int p = 0; p < 4; ++p) s.h[p] = pack(s.x[2 * p], s.x[2 * p + 1]); break;
    case 1: for (int j = 0; j < 4; ++j) s.r[j] = resid(s.h[j >> 1], j & 1, s.x[j]); break;
    case 2: for (int j = 4; j < 8; ++j) s.r[j] = resid(s.h[j >> 1], j & 1, s.x[j]); break;
    case 3: for (int p = 0; p < 4; ++p) s.m[p] = pack(s.r[2 * p], s.r[2 * p + 1]); break;
    case 4: for (int j = 0; j < 4; ++j) s.r[j] = resid(s.m[j >> 1], j & 1, s.r[j]); break;
    case 5: for (int j = 4; j < 8; ++j) s.r[j] = resid(s.m[j >> 1], j & 1, s.r[j]); break;
    case 6: for (int p = 0; p < 4; ++p) s.l[p] = pack(s.r[2 * p], s.r[2 * p + 1]); break;
    default: break;
    }
}

// both splits of n values: bitwise comparison of the three planes (host checks)
__global__ void split_check(const float* x, unsigned* out, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 > n) return;
    Split a, b;
    for (int j = 0; j < 8; ++j) { a.x[j] = b.x[j] = x[i + j]; a.r[j] = b.r[j] = 0.f; }
    for (int j = 0; j < 4; ++j) { a.e[j] = b.e[j] = 0.f; a.h[j] = a.m[j] = a.l[j] = b.h[j] = b.m[j] = b.l[j] = 0u; }
    slice<0>(a); slice<1>(a); slice<2>(a); slice<3>(a); slice<4>(a); slice<5>(a); slice<6>(a); slice<7>(a); slice<8>(a); slice<9>(a); slice<10>(a);
    slice2<0>(b); slice2<1>(b); slice2<2>(b); slice2<3>(b); slice2<4>(b); slice2<5>(b); slice2<6>(b);
    unsigned bad = 0;
    for (int p = 0; p < 4; ++p) bad += (a.h[p] != b.h[p]) + (a.m[p] != b.m[p]) + (a.l[p] != b.l[p]);
    out[i / 8] = bad;
}

constexpr int STAGE = 64 * 1024;

// MF: MFMAs, SP: splits, RD: LDS fragment reads, DM: LDS-DMA staging (+ counted wait and barrier)
template <int T, bool MF, bool SP, bool RD, bool DM, bool D2 = false>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, long long* cyc, int iters, size_t wg_bytes) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    asm volatile("" : "+s"(lds0));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const char* base = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * wg_bytes;
    const unsigned nst = (unsigned)(wg_bytes / STAGE);

    constexpr int NM = T * T * 6;          // MFMAs per k-step
    constexpr int PF = NM / (2 * T);       // MFMAs per fragment split (12 or 6)
    constexpr int NP = 4 * T;              // DMA pieces per wave and k-step
    f16v acc[T * T];
    for (int t = 0; t < T * T; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    bf8 pl[2 * T][3];   // planes of the 2 T fragments (A blocks, B blocks)
    for (int f = 0; f < 2 * T; ++f)
        for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 8; ++i) pl[f][p][i] = (__bf16)(float)((lane * 7 + i + f + p) % 13 - 6);
    Split s;
    for (int i = 0; i < 8; ++i) { s.x[i] = lane * 0.37f + i * 1.7f; s.r[i] = 0.f; }
    for (int i = 0; i < 4; ++i) { s.e[i] = 0.f; s.h[i] = s.m[i] = s.l[i] = 0u; }
    f4v raw[2];
    raw[0] = raw[1] = (f4v){1.f, 2.f, 3.f, 4.f};

    auto dma = [&](int it, int q) {   // piece q (0..NP-1) of this wave for stage it
        const unsigned st = lds0 + (unsigned)(it & 1) * STAGE + (unsigned)(wave * NP + q) * 1024u;
        const char* g = base + (size_t)((unsigned)it % nst) * STAGE + (size_t)(wave * NP + q) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(g), (lds_f32*)(size_t)st, 16, 0, 0);
    };
    if (DM) {
        for (int q = 0; q < NP; ++q) dma(0, q);
        for (int q = 0; q < NP; ++q) dma(1, q);
    }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (DM) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");   // this stage landed, the next one stays in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const unsigned st = lds0 + (unsigned)(it & 1) * STAGE + (unsigned)wave * 16384u;
        static_for<NM>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int ta = g / (6 * T), tb = (g / 6) % T, pr = g % 6;
            if (MF) {
                constexpr int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
                constexpr int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
                acc[ta * T + tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl[ta][ia], pl[T + tb][ib], acc[ta * T + tb], 0, 0, 0);
            }
            constexpr int f = g / PF, q = g % PF;   // fragment f is split during MFMAs PF f .. PF f + PF - 1
            constexpr int SPM = 12 / PF;             // slices per MFMA (1 or 2)
            if (RD && q == PF - 1) {                 // its successor's two 16-byte reads
                const unsigned a = st + (unsigned)((f + 1) % (2 * T)) * 2048u + (unsigned)lane * 16u;
                raw[0] = *(lds_f4*)(size_t)a;
                raw[1] = *(lds_f4*)(size_t)(a + 1024u);
            }
            if (SP) {
                if (q == 0) {
                    for (int j = 0; j < 4; ++j) { s.x[j] = raw[0][j]; s.x[4 + j] = raw[1][j]; }
                }
                if (D2) {
                    if (SPM == 1) slice2<q>(s);                                   // 7 slices over the first 7 of 12 MFMAs
                    else { slice2<SPM * q>(s); slice2<SPM * q + 1>(s); }
                } else {
                    slice<SPM * q>(s);
                    if (SPM == 2) slice<SPM * q + 1>(s);
                }
                if (q == PF - 1) {
                    pl[f][0] = __builtin_bit_cast(bf8, (u4v){s.h[0], s.h[1], s.h[2], s.h[3]});
                    pl[f][1] = __builtin_bit_cast(bf8, (u4v){s.m[0], s.m[1], s.m[2], s.m[3]});
                    pl[f][2] = __builtin_bit_cast(bf8, (u4v){s.l[0], s.l[1], s.l[2], s.l[3]});
                }
            }
            if (DM && g >= 4 && (g - 4) % (T == 4 ? 5 : 4) == 0 && (g - 4) / (T == 4 ? 5 : 4) < NP) dma(it + 2, (g - 4) / (T == 4 ? 5 : 4));   // NP pieces, spread
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float a = 0.f;
    for (int t = 0; t < T * T; ++t)
        for (int i = 0; i < 16; ++i) a += acc[t][i];
    for (int f = 0; f < 2 * T; ++f)
        for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 8; ++i) a += (float)pl[f][p][i];
    for (int i = 0; i < 8; ++i) a += s.r[i] + s.x[i];
    a += raw[0][0] + raw[1][3];
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}


// The same work per wave with v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md, DVFS give-back item 7: the chip may hold a higher
// clock on this shape): a k-step is 32 deep — 8 A fragments (16 rows x 32 k) x 8 B fragments x 6 products = 384 MFMAs of 16
// cycles, 16 fragment splits (1.83 vector instructions per MFMA), 32 ds_read_b128, 32 DMA pieces: the FLOPs, splits, reads and
// bytes of TWO k-steps of the 4x4 mix above.  Live planes as an A-outer schedule would keep them: 2 A fragments + 8 B fragments.
template <bool MF, bool SP, bool RD, bool DM>
__global__ __launch_bounds__(256) void k16(const float* __restrict__ src, float* out, long long* cyc, int iters, size_t wg_bytes) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    asm volatile("" : "+s"(lds0));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const char* base = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * wg_bytes;
    const unsigned nst = (unsigned)(wg_bytes / STAGE);
    constexpr int NM = 384, PF = 24, NP = 16;   // (NP: a 64 KB stage holds one 32-deep step of the four waves: 16 KB each)
    f4v acc[64];
    static_for<64>([&](auto tc) { acc[decltype(tc)::value] = (f4v){0.f, 0.f, 0.f, 0.f}; });
    bf8 pl[10][3];
    for (int f = 0; f < 10; ++f)
        for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 8; ++i) pl[f][p][i] = (__bf16)(float)((lane * 7 + i + f + p) % 13 - 6);
    Split s;
    for (int i = 0; i < 8; ++i) { s.x[i] = lane * 0.37f + i * 1.7f; s.r[i] = 0.f; }
    for (int i = 0; i < 4; ++i) { s.e[i] = 0.f; s.h[i] = s.m[i] = s.l[i] = 0u; }
    f4v raw[2];
    raw[0] = raw[1] = (f4v){1.f, 2.f, 3.f, 4.f};
    auto dma = [&](int it, int q) {
        const unsigned st = lds0 + (unsigned)(it & 1) * STAGE + (unsigned)(wave * NP + q) * 1024u;
        const char* g = base + (size_t)((unsigned)it % nst) * STAGE + (size_t)(wave * NP + q) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(g), (lds_f32*)(size_t)st, 16, 0, 0);
    };
    if (DM) {
        for (int q = 0; q < NP; ++q) dma(0, q);
        for (int q = 0; q < NP; ++q) dma(1, q);
    }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (DM) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const unsigned st = lds0 + (unsigned)(it & 1) * STAGE + (unsigned)wave * 16384u;
        static_for<NM>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            // A fragment ta (48 MFMAs each); inside it product-major, so that an accumulator tile returns after 8 MFMAs
            constexpr int ta = g / 48, pr = (g / 8) % 6, tb = g % 8;
            if (MF) {
                constexpr int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
                constexpr int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
                acc[ta * 8 + tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[ta & 1][ia], pl[2 + tb][ib], acc[ta * 8 + tb], 0, 0, 0);
            }
            constexpr int f = g / PF, q = g % PF;   // 16 fragments split per step, one per 24 MFMAs: a slice after every other MFMA
            if (RD && q == PF - 1) {
                const unsigned a = st + (unsigned)((f + 1) % 16) * 1024u + (unsigned)lane * 16u;
                raw[0] = *(lds_f4*)(size_t)a;
                raw[1] = *(lds_f4*)(size_t)(a + 8192u);
            }
            if (SP) {
                if (q == 0) {
                    for (int j = 0; j < 4; ++j) { s.x[j] = raw[0][j]; s.x[4 + j] = raw[1][j]; }
                }
                if (q % 2 == 0 && q / 2 < 11) slice<q / 2>(s);
                if (q == PF - 1) {
                    constexpr int d = (f * 7 + 3) % 10;   // any plane set: the probe's operands are not a real product's
                    pl[d][0] = __builtin_bit_cast(bf8, (u4v){s.h[0], s.h[1], s.h[2], s.h[3]});
                    pl[d][1] = __builtin_bit_cast(bf8, (u4v){s.m[0], s.m[1], s.m[2], s.m[3]});
                    pl[d][2] = __builtin_bit_cast(bf8, (u4v){s.l[0], s.l[1], s.l[2], s.l[3]});
                }
            }
            if (DM && g >= 8 && (g - 8) % 20 == 0 && (g - 8) / 20 < NP) dma(it + 2, (g - 8) / 20);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float a = 0.f;
    static_for<64>([&](auto tc) { constexpr int t = decltype(tc)::value; a += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]); });
    for (int f = 0; f < 10; ++f)
        for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 8; ++i) a += (float)pl[f][p][i];
    for (int i = 0; i < 8; ++i) a += s.r[i] + s.x[i];
    a += raw[0][0] + raw[1][3];
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool MF, bool SP, bool RD, bool DM>
void run16(const char* name, const float* src, size_t wg_bytes, int nb = 256) {
    const int iters = 400;   // 400 x 384 MFMAs of 16 cycles = the 800 x 96 of 32 cycles above
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k16<MF, SP, RD, DM><<<nb, 256>>>(src, out, cyc, 50, wg_bytes);
    hipEventRecord(e0); k16<MF, SP, RD, DM><<<nb, 256>>>(src, out, cyc, iters, wg_bytes); hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < nb * 4; ++i) c += (double)h[i];
    c /= nb * 4.0 * iters;
    printf("%-62s %8.1f us | %6.0f cycles per 32-deep step and wave (384 MFMA = %d) = %.1f per 32-cycle MFMA equivalent | %.2f GHz | %.2f ns per equivalent\n",
           name, ms * 1e3, c, 384 * 16, c / 192, c * iters / (ms * 1e3) * 1e-3, ms * 1e6 / ((double)iters * 192));
    hipFree(out); hipFree(cyc);
}

template <int T, bool MF, bool SP, bool RD, bool DM, bool D2 = false>
void run(const char* name, const float* src, size_t wg_bytes, int nb = 256) {
    const int iters = T == 4 ? 800 : 3200;   // the same number of MFMAs either way
    float* out; long long* cyc;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<T, MF, SP, RD, DM, D2><<<nb, 256>>>(src, out, cyc, 50, wg_bytes);
    hipEventRecord(e0); k<T, MF, SP, RD, DM, D2><<<nb, 256>>>(src, out, cyc, iters, wg_bytes); hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < nb * 4; ++i) c += (double)h[i];
    c /= nb * 4.0 * iters;
    const int nm = T * T * 6;
    printf("%-62s %8.1f us | %6.0f cycles per k-step and wave (%d MFMA = %d) = %.1f per MFMA | %.2f GHz | %.2f ns per MFMA\n", name, ms * 1e3, c,
           nm, nm * 32, c / nm, c * iters / (ms * 1e3) * 1e-3, ms * 1e6 / ((double)iters * nm));
    hipFree(out); hipFree(cyc);
}

int main() {
    const size_t wg_bytes = 4 * STAGE;   // 256 KB per workgroup, 64 MB in all: L2 / Infinity Cache resident like a GEMM's operands
    float* src;
    hipMalloc(&src, 256 * wg_bytes);
    {
        float* h = (float*)malloc(256 * wg_bytes);
        for (size_t i = 0; i < 256 * wg_bytes / 4; ++i) h[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.f;
        hipMemcpy(src, h, 256 * wg_bytes, hipMemcpyHostToDevice);
        free(h);
    }
    {   // the two splits agree bit for bit? (values of many magnitudes, both signs, denormal residuals included)
        const int n = 1 << 22;
        float* hx = (float*)malloc(n * 4);
        unsigned long long st = 88172645463325252ull;
        for (int i = 0; i < n; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const unsigned bits = (unsigned)(st >> 16);
            float f; unsigned b = bits;
            if (i % 3 == 0) b = (bits & 0x807fffffu) | ((100u + (bits >> 23) % 56u) << 23);   // exponents 2^-27 .. 2^28
            memcpy(&f, &b, 4);
            if (!(f == f) || f - f != 0.f) f = 1.5f;   // no NaN / inf
            hx[i] = f;
        }
        float* dx; unsigned* dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, n / 8 * 4);
        hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
        split_check<<<n / 8 / 256, 256>>>(dx, dout, n);
        unsigned* ho = (unsigned*)malloc(n / 8 * 4);
        hipMemcpy(ho, dout, n / 8 * 4, hipMemcpyDeviceToHost);
        long bad = 0; int first = -1;
        for (int i = 0; i < n / 8; ++i) { if (ho[i] && first < 0) first = i; bad += ho[i]; }
        printf("split by v_dot2c_f32_bf16 vs shift/and/sub: %ld differing plane registers of %d fragments", bad, n / 8);
        if (first >= 0) { printf(" (first: fragment %d, values", first); for (int j = 0; j < 8; ++j) printf(" %a", hx[first * 8 + j]); printf(")"); }
        printf("\n");
        hipFree(dx); hipFree(dout); free(hx); free(ho);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<4, true, false, false, false>("4x4 tiles/wave: MFMA only", src, wg_bytes);
        run<4, true, true, true, false>("4x4: MFMA + splits + LDS reads", src, wg_bytes);
        run<4, true, false, true, true>("4x4: MFMA + LDS reads + DMA, barrier", src, wg_bytes);
        run<4, true, true, true, true>("4x4: all (MFMA + splits + LDS reads + DMA, barrier)", src, wg_bytes);
        run<4, false, true, true, true>("4x4: no MFMA (splits + LDS reads + DMA, barrier)", src, wg_bytes);
        run<2, true, false, false, false>("2x2 tiles/wave (gemm_x3's mix): MFMA only", src, wg_bytes);
        run<2, true, true, true, false>("2x2: MFMA + splits + LDS reads", src, wg_bytes);
        run<2, true, false, true, true>("2x2: MFMA + LDS reads + DMA, barrier", src, wg_bytes);
        run<2, true, true, true, true>("2x2: all", src, wg_bytes);
        run<2, false, true, true, true>("2x2: no MFMA", src, wg_bytes);
        run16<true, false, false, false>("16x16x32, 8x8 tiles/wave: MFMA only", src, wg_bytes);
        run16<true, true, true, false>("16x16x32: MFMA + splits + LDS reads", src, wg_bytes);
        run16<true, false, true, true>("16x16x32: MFMA + LDS reads + DMA, barrier", src, wg_bytes);
        run16<true, true, true, true>("16x16x32: all", src, wg_bytes);
        // 198 workgroups: the weight gradient's grid (58 CUs idle)
        run<4, true, true, true, true>("4x4: all, 198 workgroups", src, wg_bytes, 198);
        run16<true, true, true, true>("16x16x32: all, 198 workgroups", src, wg_bytes, 198);
        // residuals by v_dot2c_f32_bf16: 28 instead of 44 vector instructions per fragment
        run<4, true, true, true, false, true>("4x4: MFMA + dot2 splits + LDS reads", src, wg_bytes);
        run<4, true, true, true, true, true>("4x4: all, dot2 splits", src, wg_bytes);
        run<4, true, true, true, true, true>("4x4: all, dot2 splits, 198 workgroups", src, wg_bytes, 198);
        run<4, true, true, true, true>("4x4: all, 198 workgroups (again)", src, wg_bytes, 198);
        run<4, true, true, true, true, true>("4x4: all, dot2 splits, 198 workgroups (again)", src, wg_bytes, 198);
    }
    return 0;
}
