// Issue cost of the instructions the in-register bf16x3 split is made of, alone and in the shadow of an MFMA (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -w tools/exp/valu_probe.hip -o tools/exp/valu_probe && tools/exp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, float s, int iters) {
    float m[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { m[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x * 77u + i; }
    f16v acc0 = {}, acc1 = {};
    bf8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x + i); fb[i] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE >= 100) {  // one MFMA, then NV VALU ops of kind MODE-100 in its shadow
                if (r & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int j = i & 7;
                const int M = MODE % 100;
                if (M == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(m[j]), "v"(m[(j + 1) & 7]));
                if (M == 1) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[j]));
                if (M == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[j]));
                if (M == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(m[j]) : "v"(m[(j + 1) & 7]));
                if (M == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
                if (M == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(f2*)&m[2 * (j & 3)]) : "v"(*(f2*)&m[2 * ((j + 1) & 3)]));
                if (M == 6) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
                if (M == 7) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
                if (M == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 7]));
                if (M == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(f2*)&m[2 * (j & 3)]) : "v"(*(f2*)&m[2 * ((j + 1) & 3)]));
                if (M == 10) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[j]) : "s"(0xffff0000u));
            }
        }
    }
    float acc = 0; for (int i = 0; i < 8; ++i) acc += m[i] + (float)u[i];
    for (int i = 0; i < 16; ++i) acc += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE, int NV> void run(const char* name) {
    const int w = 2, nb = 256 * w, iters = 4000;
    float* out; hipMalloc(&out, nb * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NV><<<nb, 256>>>(out, 1.0001f, 100);
    hipEventRecord(e0); k<MODE, NV><<<nb, 256>>>(out, 1.0001f, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = 8.0 * iters * w;  // (MFMA + NV VALU) groups per SIMD
    printf("%-34s NV %2d: %.1f cycles per group per SIMD @2.4GHz", name, NV, ms * 1e-3 * 2.4e9 / groups);
    if (MODE < 100) printf("  = %.2f per instr\n", ms * 1e-3 * 2.4e9 / groups / NV); else printf("\n");
    hipFree(out);
}
int main() {
    run<0, 8>("v_cvt_pk_bf16_f32"); run<1, 8>("v_lshlrev_b32"); run<2, 8>("v_and_b32 literal"); run<10, 8>("v_and_b32 sgpr");
    run<3, 8>("v_sub_f32"); run<4, 8>("v_perm_b32"); run<5, 8>("v_pk_add_f32"); run<6, 8>("v_and_or_b32"); run<7, 8>("v_bfi_b32");
    run<8, 8>("v_mov_b32"); run<9, 8>("v_pk_mul_f32");
    run<100, 0>("mfma alone");
    run<103, 4>("mfma + v_sub_f32"); run<103, 8>("mfma + v_sub_f32"); run<103, 12>("mfma + v_sub_f32"); run<103, 16>("mfma + v_sub_f32");
    run<100, 4>("mfma + v_cvt_pk"); run<100, 8>("mfma + v_cvt_pk");
    run<101, 8>("mfma + v_lshlrev"); run<101, 16>("mfma + v_lshlrev");
    run<105, 4>("mfma + v_pk_add_f32"); run<105, 8>("mfma + v_pk_add_f32");
    return 0;
}
