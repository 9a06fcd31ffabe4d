// Basis-Point-Set encode (A1): replaces bps_torch.encode -> chamfer_distance NN kernel as reached
// from DexGanGrasp/data/bps_encoder.py:33 and dexgangrasp_real_robot.py:139.
//   id_i = argmin_j |b_i - x_j|^2   (strict '<': the lowest j wins a tie),   dist_i = sqrt(min)
// Arithmetic contract (identical to oracle/bps_oracle.c, so ids are bit-exact against it):
//   dx = x - bx ...;  d2 = fmaf(dz, dz, fmaf(dy, dy, dx*dx));  dist = sqrtf(d2) correctly rounded.
//
// This kernel is fp32-VALU bound (arithmetic intensity ~1e3 flop/B: 4096*N pair evaluations against
// 12*N + 32 KiB bytes per cloud), not HBM bound.  Mapping for CDNA4:
//  - a lane owns basis points (registers: bx,by,bz,best,id per point); the cloud point index j is
//    wave-uniform, so x_j,y_j,z_j arrive through the SCALAR cache (s_load) and are broadcast as SGPR
//    operands — no LDS traffic, no per-lane loads in the inner loop;
//  - batched clouds (B >= 64): 256-thread workgroups, 4 basis points per lane -> grid (P/1024, B);
//  - few clouds (robot path, one 15k..73k-point cloud): 1024-thread workgroups in which the 16 waves
//    scan 16 disjoint ranges of the cloud for the same 64 basis points and merge through LDS in
//    range order (keeps the lowest-index tie rule) -> grid (P/64, B).
#include "dgg_internal.h"

namespace dgg {

__device__ __forceinline__ float dist2(float x, float y, float z, float bx, float by, float bz) {
    const float dx = x - bx, dy = y - by, dz = z - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

template <int BPT>
__global__ __launch_bounds__(256) void bps_batched_kernel(const float* __restrict__ clouds, int N,
                                                          const float* __restrict__ basis, int P,
                                                          float* __restrict__ dists, int32_t* __restrict__ ids) {
    const int b = blockIdx.y;
    const float* __restrict__ c = clouds + (size_t)b * N * 3;
    float bx[BPT], by[BPT], bz[BPT], best[BPT];
    int bid[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        int i = blockIdx.x * (256 * BPT) + q * 256 + threadIdx.x;
        i = i < P ? i : P - 1;
        bx[q] = basis[3 * i]; by[q] = basis[3 * i + 1]; bz[q] = basis[3 * i + 2];
        best[q] = __builtin_inff();
        bid[q] = 0;
    }
    int j = 0;
    for (; j + 4 <= N; j += 4) {
        float p[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) p[e] = c[3 * j + e];  // uniform address -> scalar loads
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const float d2 = dist2(p[3 * u], p[3 * u + 1], p[3 * u + 2], bx[q], by[q], bz[q]);
                const bool lt = d2 < best[q];
                best[q] = lt ? d2 : best[q];
                bid[q] = lt ? j + u : bid[q];
            }
        }
    }
    for (; j < N; ++j) {
        const float x = c[3 * j], y = c[3 * j + 1], z = c[3 * j + 2];
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            const float d2 = dist2(x, y, z, bx[q], by[q], bz[q]);
            const bool lt = d2 < best[q];
            best[q] = lt ? d2 : best[q];
            bid[q] = lt ? j : bid[q];
        }
    }
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int i = blockIdx.x * (256 * BPT) + q * 256 + threadIdx.x;
        if (i < P) {
            dists[(size_t)b * P + i] = __builtin_sqrtf(best[q]);
            if (ids) ids[(size_t)b * P + i] = bid[q];
        }
    }
}

// 1024 threads: wave w scans points [w*chunk, (w+1)*chunk) for basis points blockIdx.x*64 + lane.
__global__ __launch_bounds__(1024) void bps_single_kernel(const float* __restrict__ clouds, int N,
                                                          const float* __restrict__ basis, int P,
                                                          float* __restrict__ dists, int32_t* __restrict__ ids) {
    __shared__ float s_best[16][64];
    __shared__ int s_id[16][64];
    const int b = blockIdx.y;
    const float* __restrict__ c = clouds + (size_t)b * N * 3;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int i = blockIdx.x * 64 + lane;
    const bool ok = i < P;
    i = ok ? i : P - 1;
    const float bx = basis[3 * i], by = basis[3 * i + 1], bz = basis[3 * i + 2];
    const int chunk = (N + 15) / 16;
    const int j0 = w * chunk;
    const int j1 = min(N, j0 + chunk);
    float best = __builtin_inff();
    int bid = 0;
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        float p[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) p[e] = c[3 * j + e];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d2 = dist2(p[3 * u], p[3 * u + 1], p[3 * u + 2], bx, by, bz);
            const bool lt = d2 < best;
            best = lt ? d2 : best;
            bid = lt ? j + u : bid;
        }
    }
    for (; j < j1; ++j) {
        const float d2 = dist2(c[3 * j], c[3 * j + 1], c[3 * j + 2], bx, by, bz);
        const bool lt = d2 < best;
        best = lt ? d2 : best;
        bid = lt ? j : bid;
    }
    s_best[w][lane] = best;
    s_id[w][lane] = bid;
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {  // ascending ranges + strict '<' keep the lowest index on ties
            const float d2 = s_best[k][lane];
            const bool lt = d2 < best;
            best = lt ? d2 : best;
            bid = lt ? s_id[k][lane] : bid;
        }
        if (ok) {
            dists[(size_t)b * P + i] = __builtin_sqrtf(best);
            if (ids) ids[(size_t)b * P + i] = bid;
        }
    }
}

int bps_encode_launch(const float* clouds, int B, int N, const float* basis, int P, float* dists, int32_t* ids,
                      hipStream_t st) {
    if (!clouds || !basis || !dists || B <= 0 || N <= 0 || P <= 0) return DGG_ERR_INVALID;
    if (B >= 64) {
        dim3 grid((P + 1023) / 1024, B);
        hipLaunchKernelGGL((bps_batched_kernel<4>), grid, dim3(256), 0, st, clouds, N, basis, P, dists, ids);
    } else {
        dim3 grid((P + 63) / 64, B);
        hipLaunchKernelGGL(bps_single_kernel, grid, dim3(1024), 0, st, clouds, N, basis, P, dists, ids);
    }
    return check_launch("bps_encode");
}

}  // namespace dgg
