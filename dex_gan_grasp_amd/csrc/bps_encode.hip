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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Batched clouds.  Two things keep the vector ALU's issue slots for the distance arithmetic itself:
//  - two basis points of a lane sit side by side in 64-bit register pairs, so the subtract, the square and the two fused
//    multiply-adds are packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two IEEE results per lane and
//    issue slot, each bit-identical to dist2() above);
//  - the running argmin is kept per BLOCK of U consecutive cloud points: the block's minimum (v_min3_f32 chains) is
//    compared once against the best so far, and only the block's first index is remembered.  Strict '<' between blocks
//    keeps the earliest block on ties; after the scan the winning block is re-evaluated once per basis point and the
//    first point whose squared distance EQUALS the minimum (same instructions, same bits) is the answer — exactly the
//    sequential lowest-index rule.
//  3 packed + ~0.7 other vector instructions per pair instead of 9 (U = 16).  Measured issue costs on gfx950
//  (tools/exp/pk_probe.hip, cycles per wave instruction): v_pk_{add,mul,fma}_f32 4.2-4.7 for two results per lane;
//  v_fma / v_mul / v_sub on VGPR operands 2.2-2.7 (double rate) but 4.2 as soon as one operand is an SGPR, which the
//  wave-uniform cloud point is; v_min_f32 / v_min3_f32 4.2; v_cmp + v_cndmask 6.4 — hence packed arithmetic (the SGPR
//  pair is broadcast by op_sel at no cost) and as few compare / select instructions as the tie rule allows.
template <int BPT, int U>
__global__ __launch_bounds__(256) void bps_batched_kernel(const float* __restrict__ clouds, int N,
                                                          const float* __restrict__ basis, int P,
                                                          float* __restrict__ dists, int32_t* __restrict__ ids,
                                                          int B, int gx) {
    static_assert(BPT % 2 == 0 && (U == 8 || U == 16), "pairs of basis points, blocks of 8 / 16 cloud points");
    constexpr int H = BPT / 2;
    // 1-D grid of gx * B workgroups (gx basis slices per cloud).  Consecutive workgroup ids go round the 8 XCDs: the gx
    // slices of one cloud take ids 8 apart, so they share an XCD's L2 and the cloud is fetched from memory once instead
    // of gx times (the ragged last group of < 8 clouds keeps the plain order).
    const int L = blockIdx.x, per = 8 * gx;
    int b, slice;
    if (L < (B / 8) * per) {
        b = (L / per) * 8 + (L & 7);
        slice = (L % per) >> 3;
    } else {
        b = L / gx;
        slice = L % gx;
    }
    const float* __restrict__ c = clouds + (size_t)b * N * 3;
    f32x2 bx[H], by[H], bz[H];
    float best[BPT];
    int blk[BPT];  // first point of the block that set `best`; -1: nothing was ever smaller than +inf
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        int i = slice * (256 * BPT) + q * 256 + threadIdx.x;
        i = i < P ? i : P - 1;
        bx[q >> 1][q & 1] = basis[3 * i];
        by[q >> 1][q & 1] = basis[3 * i + 1];
        bz[q >> 1][q & 1] = basis[3 * i + 2];
        best[q] = __builtin_inff();
        blk[q] = -1;
    }
    auto pair_d2 = [&](int h, float x, float y, float z) {
        const f32x2 dx = f32x2{x, x} - bx[h], dy = f32x2{y, y} - by[h], dz = f32x2{z, z} - bz[h];
        return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    };
    auto offer = [&](int q, float m, int j0) {  // m: minimum of a block starting at point j0
        const bool lt = m < best[q];
        best[q] = lt ? m : best[q];
        blk[q] = lt ? j0 : blk[q];
    };
    int j = 0;
    for (; j + U <= N; j += U) {
        float p[3 * U];
#pragma unroll
        for (int e = 0; e < 3 * U; ++e) p[e] = c[3 * j + e];  // uniform address -> scalar loads
#pragma unroll
        for (int h = 0; h < H; ++h) {
            f32x2 d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = pair_d2(h, p[3 * u], p[3 * u + 1], p[3 * u + 2]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                auto min3 = [](float a, float b2, float c2) { return __builtin_fminf(__builtin_fminf(a, b2), c2); };
                float m = min3(min3(d[0][e], d[1][e], d[2][e]), min3(d[3][e], d[4][e], d[5][e]), d[6][e]);
                if (U == 16) {
                    const float n0 = min3(d[8][e], d[9][e], d[10][e]), n1 = min3(d[11][e], d[12][e], d[13][e]);
                    m = min3(m, min3(n0, n1, d[14][e]), d[U - 1][e]);
                }
                offer(2 * h + e, __builtin_fminf(m, d[7][e]), j);
            }
        }
    }
    for (; j < N; ++j) {  // ragged tail: blocks of one point
        const float x = c[3 * j], y = c[3 * j + 1], z = c[3 * j + 2];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x2 d = pair_d2(h, x, y, z);
            offer(2 * h, d[0], j);
            offer(2 * h + 1, d[1], j);
        }
    }
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int i = slice * (256 * BPT) + q * 256 + threadIdx.x;
        // the winning block again, this lane's own addresses: the first point that reproduces the minimum
        int id = 0;
        if (blk[q] >= 0) {
            const int j0 = blk[q];
            const int n = (j0 + U <= N && j0 % U == 0) ? U : 1;  // full blocks start at multiples of U, tail blocks hold 1
            id = j0;
            bool found = false;
            for (int u = 0; u < n; ++u) {
                const float d2 = dist2(c[3 * (j0 + u)], c[3 * (j0 + u) + 1], c[3 * (j0 + u) + 2], bx[q >> 1][q & 1],
                                       by[q >> 1][q & 1], bz[q >> 1][q & 1]);
                if (!found && d2 == best[q]) {
                    id = j0 + u;
                    found = true;
                }
            }
        }
        if (i < P) {
            dists[(size_t)b * P + i] = __builtin_sqrtf(best[q]);
            if (ids) ids[(size_t)b * P + i] = id;
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
        const int gx = (P + 1023) / 1024;
        hipLaunchKernelGGL((bps_batched_kernel<4, 16>), dim3((unsigned)gx * B), dim3(256), 0, st, clouds, N, basis, P,
                           dists, ids, B, gx);
    } else {
        dim3 grid((P + 63) / 64, B);
        hipLaunchKernelGGL(bps_single_kernel, grid, dim3(1024), 0, st, clouds, N, basis, P, dists, ids);
    }
    return check_launch("bps_encode");
}

}  // namespace dgg
