#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
// MODE 0: 2x2 blocking, constant operands.  1: operands from LDS, read one group ahead (like the GEMM).
// 2: like 1 plus a __syncthreads() every 16 groups.  3: like 2 plus 32 ds_write_b32 per 16 groups.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0v, float b0v) {
    __shared__ float lds[2 * 32 * 258];
    for (int i = threadIdx.x; i < 2 * 32 * 258; i += 256) lds[i] = a0v + i * 1e-6f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    float a[2][2], b[2][2];
    a[0][0] = a0v; a[0][1] = a0v + 1; b[0][0] = b0v; b[0][1] = b0v + 1;
    float w0 = a0v, w1 = b0v;
    for (int it = 0; it < iters; ++it) {
        const float* As = lds + (it & 1) * 32 * 258;
        const float* Bs = As + 32 * 129;
        if (MODE >= 1) { a[0][0] = As[lh * 129 + wm0 + li]; a[0][1] = As[lh * 129 + wm0 + 32 + li]; b[0][0] = Bs[lh * 129 + wn0 + li]; b[0][1] = Bs[lh * 129 + wn0 + 32 + li]; }
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) {
            const int cur = kp & 1, nxt = cur ^ 1;
            if (kp + 1 < 16) {
                const int kk = 2 * (kp + 1) + lh;
                if (MODE >= 1) { a[nxt][0] = As[kk * 129 + wm0 + li]; a[nxt][1] = As[kk * 129 + wm0 + 32 + li]; b[nxt][0] = Bs[kk * 129 + wn0 + li]; b[nxt][1] = Bs[kk * 129 + wn0 + 32 + li]; }
                else { a[nxt][0] = a[cur][0]; a[nxt][1] = a[cur][1]; b[nxt][0] = b[cur][0]; b[nxt][1] = b[cur][1]; }
            }
            if (MODE >= 3 && (kp & 1) == 0) { float* S = lds + ((it + 1) & 1) * 32 * 258; S[(threadIdx.x & 31) * 129 + (threadIdx.x >> 5) + 8 * kp] = w0; S[32 * 129 + (threadIdx.x & 31) * 129 + (threadIdx.x >> 5) + 8 * kp] = w1; S[(threadIdx.x & 31) * 129 + (threadIdx.x >> 5) + 8 * (kp + 1)] = w0; S[32 * 129 + (threadIdx.x & 31) * 129 + (threadIdx.x >> 5) + 8 * (kp + 1)] = w1; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void run(const char* name, F launch, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double fl = 64.0 * 4096.0 * iters * blocks * 4;
    printf("%-40s blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 600;
    for (int blocks : {256, 512}) {
        run("2x2 const operands", [&] { hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, blocks, iters);
        run("2x2 LDS operands, read-ahead", [&] { hipLaunchKernelGGL((k<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, blocks, iters);
        run("  + barrier per k-tile", [&] { hipLaunchKernelGGL((k<2>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, blocks, iters);
        run("  + 32 ds_write per k-tile (interleaved)", [&] { hipLaunchKernelGGL((k<3>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, blocks, iters);
    }
    return 0;
}
