#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a0 + i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) { a = lds[(it * 64 + lane) & 8191]; b = lds[(it * 64 + lane + 4096) & 8191]; }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void run(const char* name, F launch, double flop_per_iter_per_wave, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double fl = flop_per_iter_per_wave * iters * blocks * 4;
    printf("%-28s blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    for (int blocks : {240, 256, 512}) {
        run("32x32x2 4acc pure", [&] { hipLaunchKernelGGL((k<4, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 4096.0, blocks, iters);
        run("32x32x2 1acc pure", [&] { hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 1 * 4096.0, blocks, iters);
        run("32x32x2 4acc + lds operands", [&] { hipLaunchKernelGGL((k<4, 1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 4096.0, blocks, iters);
        run("16x16x4 4acc pure", [&] { hipLaunchKernelGGL((k16<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 2048.0, blocks, iters);
        run("16x16x4 8acc pure", [&] { hipLaunchKernelGGL((k16<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 8 * 2048.0, blocks, iters);
    }
    return 0;
}
