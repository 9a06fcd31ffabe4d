// Can a SECOND wave of a SIMD split fp32 -> 3 x bf16 in the shadow of the first wave's MFMAs?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -w tools/exp/split_probe.hip -o tools/exp/split_probe && tools/exp/split_probe
// 512-thread workgroups, one per CU (100 KB of LDS declared): waves 0-3 issue 48 v_mfma_f32_32x32x16_bf16 per "k-tile",
// waves 4-7 the vector instructions of NF fragment splits per k-tile (one fragment = 8 fp32 -> 12 packed bf16 registers:
// 12 v_cvt_pk_bf16_f32 + 16 v_sub_f32 + 8 v_lshlrev_b32 + 8 v_and_b32 = 44), free running, no barrier between the roles.
// Cycles per k-tile of each role (s_memtime around its loop), roles alone and together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_frag(float (&x)[8], unsigned (&o)[12]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hi, mid, lo;
        float r0, r1, q0, q1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(x[2 * p]), "v"(x[2 * p + 1]));
        unsigned h0, h1;
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(h0) : "v"(hi));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h1) : "v"(hi));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x[2 * p]), "v"(h0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x[2 * p + 1]), "v"(h1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(mid) : "v"(r0), "v"(r1));
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(h0) : "v"(mid));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h1) : "v"(mid));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q0) : "v"(r0), "v"(h0));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q1) : "v"(r1), "v"(h1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(q0), "v"(q1));
        o[p] = hi; o[4 + p] = mid; o[8 + p] = lo;
    }
}

template <int NF, bool MF, bool VA>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    __shared__ char pad[100 * 1024];
    if (threadIdx.x == 9999) pad[0] = 1;
    const int wave = threadIdx.x >> 6;
    float acc = 0.f;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        f16v a0 = {}, a1 = {}, a2 = {}, a3 = {};
        bf8 fa, fb;
        for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)((threadIdx.x * 7 + i) % 13 - 6); fb[i] = (__bf16)(float)((threadIdx.x + i) % 5 - 2); }
        t0 = __builtin_amdgcn_s_memtime();
        if (MF)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a3, 0, 0, 0);
                }
            }
        for (int i = 0; i < 16; ++i) acc += a0[i] + a1[i] + a2[i] + a3[i];
        asm volatile("s_nop 0" ::"v"(acc));
        t1 = __builtin_amdgcn_s_memtime();
    } else {
        float x[8];
        unsigned o[12];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.37f + i * 1.7f;
        for (int i = 0; i < 12; ++i) o[i] = 0;
        t0 = __builtin_amdgcn_s_memtime();
        if (VA)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    split_frag(x, o);
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
                }
            }
        for (int i = 0; i < 12; ++i) acc += (float)o[i];
        asm volatile("s_nop 0" ::"v"(acc));
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NF, bool MF, bool VA>
void run(const char* name) {
    const int nb = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NF, MF, VA><<<nb, 512>>>(out, cyc, 100);
    hipEventRecord(e0); k<NF, MF, VA><<<nb, 512>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += (double)h[b * 8 + w];
    cm /= nb * 4.0 * iters; cv /= nb * 4.0 * iters;
    printf("%-44s kernel %.1f us | per k-tile: MFMA waves %.0f cycles (48 MFMA = 1536), split waves %.0f cycles (%d VALU)\n", name, ms * 1e3 / 1.0,
           cm, cv, NF * 44);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<4, true, false>("MFMA waves alone");
    run<4, false, true>("split waves alone, 4 fragments / k-tile");
    run<8, false, true>("split waves alone, 8 fragments / k-tile");
    run<2, true, true>("both, 2 fragments / k-tile per split wave");
    run<4, true, true>("both, 4 fragments / k-tile per split wave");
    run<6, true, true>("both, 6 fragments / k-tile per split wave");
    run<8, true, true>("both, 8 fragments / k-tile per split wave");
    return 0;
}
