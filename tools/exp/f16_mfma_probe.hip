// Does v_mfma_f32_32x32x16_f16 keep fp16 DENORMAL inputs (|x| < 2^-14)?  A[i][k] = a for k == 0 else 0, B[k][j] = b.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f16_mfma_probe tools/exp/f16_mfma_probe.hip && /tmp/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float a, float b, float* out) {
    const int lane = threadIdx.x;
    h8 A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) {  // k = 0..7 live in lanes 0-31
        A[0] = (_Float16)a;
        B[0] = (_Float16)b;
    }
    f16v c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (lane == 0) {
        out[0] = c[0];
        out[1] = (float)(_Float16)a;
        out[2] = (float)(_Float16)b;
    }
}
int main() {
    float* d;
    hipMalloc(&d, 12);
    const float as[] = {1.0f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 2.0e-5f};
    const float bs[] = {2.0f, 1024.0f, 1024.0f, 1024.0f, 3.0e-5f};
    for (int i = 0; i < 5; ++i) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, as[i], bs[i], d);
        float h[3];
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a %.3g (as fp16 %.9g) x b %.3g (as fp16 %.9g): mfma %.9g   exact product of the fp16 values %.9g\n", as[i], h[1], bs[i], h[2],
               h[0], (double)h[1] * h[2]);
    }
    return 0;
}
