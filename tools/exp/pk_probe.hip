// Issue-rate probe for packed fp32 on gfx950: cycles per wave-instruction of v_pk_fma_f32 (VGPR and SGPR-broadcast
// operands), v_fma_f32, v_min3_f32, measured with s_memtime inside one wave per SIMD and against wall time with the chip full.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/pk_probe.hip -o tools/exp/pk_probe && tools/exp/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, float s, int iters) {
    f2 a[8]; float m[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{(float)threadIdx.x + i, 1.f + i}; m[i] = threadIdx.x * 0.5f + i; }
    f2 b = {s, s * 0.5f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], b, a[i]);          // pk_fma, VGPR operands
                if (MODE == 1) a[i] = __builtin_elementwise_fma(a[i], f2{s, s}, a[i]);   // pk_fma, SGPR broadcast
                if (MODE == 2) m[i] = __builtin_fmaf(m[i], s, m[i]);                     // v_fma_f32
                if (MODE == 3) a[i] = f2{s, s} - a[i];                                   // pk_add with SGPR + neg
                if (MODE == 4) m[i] = __builtin_fminf(__builtin_fminf(m[i], a[i][0]), a[i][1]);  // v_min3
                if (MODE == 5) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(m[i]) : "s"(s));
                if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(m[i]));
                if (MODE == 7) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(m[i]) : "v"(a[i][0]));
                if (MODE == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 9) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a[i]) : "s"(b));
                if (MODE == 10) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(b));
                if (MODE == 11) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[i]));
                if (MODE == 12) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(m[i]) : "v"(a[i][1]));
                if (MODE == 13) asm volatile("v_min_f32 %0, %1, %0" : "+v"(m[i]) : "v"(a[i][1]));
                if (MODE == 14) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(m[i]) : "v"(a[i][1]) : "vcc");
            }
        }
    }
    long long t1 = clock64();
    float acc = 0; for (int i = 0; i < 8; ++i) acc += a[i][0] + a[i][1] + m[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
    int nb = 256 * waves_per_simd;  // 256-thread blocks: one wave per SIMD each
    float* out; long long* cyc; hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nb, 256>>>(out, cyc, 1.0001f, 100);
    hipEventRecord(e0); k<MODE><<<nb, 256>>>(out, cyc, 1.0001f, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    double instr = 64.0 * iters;  // per wave
    printf("%-28s waves/SIMD %d: %.2f s_memtime ticks per instr per wave, wall %.3f ms -> %.2f ns per instr per SIMD-slot, chip %.1f G wave-instr/s = %.2f cycles @2.4GHz\n", name, waves_per_simd,
           h[0] / instr, ms, ms * 1e6 / (instr * waves_per_simd), instr * nb * 4 / (ms * 1e-3) / 1e9, 2.4e9 * 1024 / (instr * nb * 4 / (ms * 1e-3)));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {4}) {
        run<0>("v_pk_fma_f32 vgpr", w); run<1>("v_pk_fma_f32 sgpr-bcast", w); run<2>("v_fma_f32", w);
        run<3>("v_pk_add_f32 sgpr neg", w); run<4>("v_min3_f32", w);
        run<5>("v_sub_f32 s,v", w); run<6>("v_mul_f32 v,v", w); run<7>("v_fmac_f32", w); run<8>("v_pk_add_f32 v,v", w);
        run<9>("v_pk_add_f32 s(bcast),v", w); run<10>("v_pk_add_f32 v,-v", w); run<11>("v_pk_mul_f32", w);
        run<12>("v_sub_f32 v,v", w); run<13>("v_min_f32", w); run<14>("v_cmp+v_cndmask (2 instr)", w);
    }
    return 0;
}
