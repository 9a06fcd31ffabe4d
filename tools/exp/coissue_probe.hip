// Does the bf16x3 split VALU hide beside v_mfma_f32_32x32x16_bf16 on gfx950?  Register-only loop (no LDS, no memory):
// per iteration 24 MFMAs (4 accumulators) and/or the split of 4 fragments (176 VALU), interleaved one slice per MFMA.
// MODE 0: MFMA only, 1: split only, 2: both interleaved (sched_barrier per MFMA), 3: both, compiler-scheduled.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
struct Frag { bf16x8 p[3]; };
__device__ __forceinline__ void slice(int sl, float (&x)[8], unsigned (&sh)[4], unsigned (&sm)[4], float (&sr)[8], float (&sx)[8], Frag& out) {
    if (sl == 0) {
        for (int p = 0; p < 4; ++p) sh[p] = pk(x[2 * p], x[2 * p + 1]);
        for (int p = 0; p < 2; ++p) { sr[2 * p] = __builtin_bit_cast(float, sh[p] << 16); sr[2 * p + 1] = __builtin_bit_cast(float, sh[p] & 0xffff0000u); }
    } else if (sl == 1) {
        for (int p = 2; p < 4; ++p) { sr[2 * p] = __builtin_bit_cast(float, sh[p] << 16); sr[2 * p + 1] = __builtin_bit_cast(float, sh[p] & 0xffff0000u); }
        for (int j = 0; j < 4; ++j) sr[j] = x[j] - sr[j];
    } else if (sl == 2) {
        for (int j = 4; j < 8; ++j) sr[j] = x[j] - sr[j];
        for (int p = 0; p < 4; ++p) sm[p] = pk(sr[2 * p], sr[2 * p + 1]);
    } else if (sl == 3) {
        for (int p = 0; p < 4; ++p) { sx[2 * p] = __builtin_bit_cast(float, sm[p] << 16); sx[2 * p + 1] = __builtin_bit_cast(float, sm[p] & 0xffff0000u); }
    } else if (sl == 4) {
        for (int j = 0; j < 8; ++j) sr[j] -= sx[j];
    } else {
        unsigned l[4];
        for (int p = 0; p < 4; ++p) l[p] = pk(sr[2 * p], sr[2 * p + 1]);
        out.p[0] = __builtin_bit_cast(bf16x8, (u32x4){sh[0], sh[1], sh[2], sh[3]});
        out.p[1] = __builtin_bit_cast(bf16x8, (u32x4){sm[0], sm[1], sm[2], sm[3]});
        out.p[2] = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
    }
}
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    float raw[4][8];
    for (int f = 0; f < 4; ++f) for (int i = 0; i < 8; ++i) raw[f][i] = in[(threadIdx.x * 32 + f * 8 + i) & 4095];
    Frag pl[2][4];
    unsigned sh[4], sm[4]; float sr[8], sx[8];
    for (int f = 0; f < 4; ++f) { for (int s = 0; s < 6; ++s) slice(s, raw[f], sh, sm, sr, sx, pl[0][f]); pl[1][f] = pl[0][f]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            // new raw values depend on the loop (keeps the split live and varying): perturb with a cheap op per fragment
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                const int ta = g / 12, tb = (g / 6) & 1, pr = g % 6;
                const int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
                const int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
                if (MODE != 1) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl[par][ta].p[ia], pl[par][2 + tb].p[ib], acc[ta][tb], 0, 0, 0);
                if (MODE != 0) slice(g % 6, raw[g / 6], sh, sm, sr, sx, pl[par ^ 1][g / 6]);
                if (MODE == 2 || MODE == 0 || MODE == 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE != 0) for (int f = 0; f < 4; ++f) raw[f][par] = sr[f] * 1.5f;  // loop-carried: no hoisting
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
    for (int f = 0; f < 4; ++f) { s += raw[f][0]; u32x4 q = __builtin_bit_cast(u32x4, pl[0][f].p[2]); s += (float)q[0]; }
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 4095] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const float* in, float* out, unsigned long long* cyc, int threads) {
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, in, out, cyc, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> c(256);
    hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : c) avg += (double)v; avg /= 256;
    const char* nm[] = {"MFMA only (24)", "split only (176 VALU)", "both, one slice per MFMA (pinned)", "both, compiler order"};
    printf("waves/SIMD %d  %-36s: %.0f cycles per k-step per wave (%.1f per MFMA)\n", threads / 256, nm[MODE], avg / (iters * 2.0), avg / (iters * 48.0));
}
int main() {
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 16384); hipMalloc(&out, 16384); hipMalloc(&cyc, 2048);
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), 16384, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) { run<0>(in, out, cyc, threads); run<1>(in, out, cyc, threads); run<2>(in, out, cyc, threads); run<3>(in, out, cyc, threads); }
    return 0;
}
