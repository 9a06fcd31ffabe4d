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
//  - 256-thread workgroups, 4 basis points per lane (two packed pairs), P/1024 basis slices per cloud;
//  - few clouds (robot path, one 15k..73k-point cloud): the same kernel over RANGES of the cloud, merged through the
//    output arrays with atomicMin in two passes (see the kernel) -> the whole chip works on one cloud.
#include "dgg_internal.h"
#include <cstdlib>
#include <cstring>

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
//
// MODE 0: one workgroup scans a whole cloud for its 1024 basis points (many clouds: the training / sweep front end).
// MODE 1 / 2: FEW clouds with many points (the robot path: one 15k..73k-point cloud, dexgangrasp_real_robot.py:139):
// the cloud is cut into `nchunk` ranges of `chunk_len` points (a multiple of U) so that the launch fills the chip, and
// the ranges meet in the output arrays themselves — no workspace in the ABI:
//   pass 1 (MODE 1): atomicMin of the range's best d^2 (non-negative floats order like their bit patterns) into dists,
//                    pre-set to +inf;
//   pass 2 (MODE 2): the range that holds the global minimum finds the first point reproducing it (same block rule) and
//                    atomicMin's its index into ids, pre-set to INT_MAX — the lowest index over all ranges wins, the
//                    sequential rule again;
//   bps_finish_kernel: dist = sqrt(d^2), untouched ids (nothing was below +inf) -> 0.
template <int BPT, int U, int MODE>
__global__ __launch_bounds__(256) void bps_batched_kernel(const float* __restrict__ clouds, int N,
                                                          const float* __restrict__ basis, int P,
                                                          float* __restrict__ dists, int32_t* __restrict__ ids,
                                                          int B, int gx, int nchunk, int chunk_len) {
    static_assert(BPT % 2 == 0 && (U == 8 || U == 16), "pairs of basis points, blocks of 8 / 16 cloud points");
    constexpr int H = BPT / 2;
    // 1-D grid of gx * B workgroups (gx basis slices per cloud).  Consecutive workgroup ids go round the 8 XCDs: the gx
    // slices of one cloud take ids 8 apart, so they share an XCD's L2 and the cloud is fetched from memory once instead
    // of gx times (the ragged last group of < 8 clouds keeps the plain order).
    const int L = blockIdx.x, per = 8 * gx;
    int b, slice, jlo = 0, jhi = N;
    if (MODE != 0) {  // (cloud, range, slice): slices fastest
        slice = L % gx;
        const int rest = L / gx;
        b = rest / nchunk;
        jlo = (rest % nchunk) * chunk_len;
        jhi = min(N, jlo + chunk_len);
    } else if (L < (B / 8) * per) {
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
        // MODE 2: `best` is the TARGET, the global minimum pass 1 left in dists (+inf: nothing was smaller, leave ids)
        best[q] = MODE == 2 ? __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)dists + (size_t)b * P + i,
                                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                            : __builtin_inff();
        blk[q] = -1;
    }
    auto pair_d2 = [&](int h, float x, float y, float z) {
        const f32x2 dx = f32x2{x, x} - bx[h], dy = f32x2{y, y} - by[h], dz = f32x2{z, z} - bz[h];
        return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    };
    auto offer = [&](int q, float m, int j0) {  // m: minimum of a block starting at point j0
        if (MODE == 2) {  // the first block that reaches the target
            const bool hit = (m == best[q]) & (blk[q] < 0);
            blk[q] = hit ? j0 : blk[q];
        } else {
            const bool lt = m < best[q];
            best[q] = lt ? m : best[q];
            blk[q] = lt ? j0 : blk[q];
        }
    };
    int j = jlo;
    const int N_ = N;
    N = jhi;  // the scan below covers [jlo, jhi)
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
        if (MODE == 1) {
            if (i < P && blk[q] >= 0)
                atomicMin((unsigned*)dists + (size_t)b * P + i, __builtin_bit_cast(unsigned, best[q]));
            continue;
        }
        // the winning block again, this lane's own addresses: the first point that reproduces the minimum
        int id = 0;
        if (blk[q] >= 0) {
            const int j0 = blk[q];
            const int n = (j0 + U <= N && (j0 - jlo) % U == 0) ? U : 1;  // full blocks: multiples of U from jlo; tail blocks hold 1
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
        if (MODE == 2) {
            if (i < P && blk[q] >= 0 && best[q] < __builtin_inff()) atomicMin(ids + (size_t)b * P + i, id);
            continue;
        }
        if (i < P) {
            dists[(size_t)b * P + i] = __builtin_sqrtf(best[q]);
            if (ids) ids[(size_t)b * P + i] = id;
        }
    }
    (void)N_;
}

// ---------------------------------------------------------------------------------------------------------------------
// Many clouds, coarse pass on the matrix cores (round 3).  The scan above spends ~3.7 vector instructions per PAIR of
// (basis point, cloud point); the fp32 matrix instruction evaluates 32 x 32 pairs per 128 cycles with no vector work:
//     v_ij = x_j.(-2 b_i) + |x_j|^2  =  |x_j - b_i|^2 - |b_i|^2                         (K = 4: x, y, z, |x|^2  against  -2bx, -2by, -2bz, 1)
// as two v_mfma_f32_32x32x2_f32 per 32 x 32 tile.  v orders the cloud points of one basis point like the squared
// distance does, but NOT bit for bit like the contract's  d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)).  So v only SELECTS:
//   - a lane sees 16 of a tile's 32 points for its basis point (the accumulator rows of its half-wave) = one block; it
//     keeps the three blocks with the smallest block minimum (values by v_med3 / v_min, indices by compare-select) and
//     the fourth smallest minimum;
//   - |v_ij + |b_i|^2 - d2_ij| <= eps_i := 2^-20 (R + |b_i|)^2, R = the cloud's largest |x_j|: the K = 4 chain rounds
//     4 times on terms whose magnitudes sum to at most (R + |b|)^2 (4u), |x|^2 itself carries 3u, the contract's d2 is
//     within 6u of the true squared distance (u = 2^-24): 13u < 16u.  Hence every point whose d2 can be the minimum or
//     tie with it has v <= v_min + 2 eps, and sits in a block whose minimum is <= v_min + 2 eps;
//   - those blocks (1 always, 2 or 3 when their minima are that close, a few % of basis points) are re-evaluated with the
//     contract's own instructions and the answer is the lexicographic minimum of (d2, j) — the sequential scan's
//     lowest-index rule.  A fourth block within 2 eps, a cloud or basis point with a coordinate that is not below 1e3 in
//     magnitude (NaN, inf, the 3e19 cases of the tests) or a degenerate scale send the lane through ALL its points with
//     the exact formula: always right, never fast.
// The two half-waves of a basis point meet through one lane exchange at the end.  Ids and distances are bit-identical to
// the scan's (same tests); the matrix pipe is the bound: 2 x 64 cycles per 1024 pairs and SIMD.
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((aligned(16))) Pt4 { float x, y, z, n2; };

template <int NT>  // 32-wide basis tiles per wave
__global__ __launch_bounds__(256) void bps_mfma_kernel(const float* __restrict__ clouds, int N,
                                                       const float* __restrict__ basis, int P,
                                                       float* __restrict__ dists, int32_t* __restrict__ ids, int B,
                                                       int gx) {
    extern __shared__ Pt4 pts[];  // the cloud, padded to a multiple of 32 points with (0, 0, 0, +inf)
    __shared__ int s_r2, s_bad;
    const int L = blockIdx.x, per = 8 * gx;
    int b, slice;
    if (L < (B / 8) * per) {  // the gx slices of a cloud on one XCD (see bps_batched_kernel)
        b = (L / per) * 8 + (L & 7);
        slice = (L % per) >> 3;
    } else {
        b = L / gx;
        slice = L % gx;
    }
    const float* __restrict__ c = clouds + (size_t)b * N * 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 31, h = lane >> 5;
    const int ntile = (N + 31) >> 5;
    if (tid == 0) {
        s_r2 = 0;
        s_bad = 0;
    }
    __syncthreads();
    {
        float r2 = 0.f;
        bool bad = false;
        for (int j = tid; j < ntile * 32; j += 256) {
            Pt4 v{0.f, 0.f, 0.f, __builtin_inff()};
            if (j < N) {
                v.x = c[3 * j];
                v.y = c[3 * j + 1];
                v.z = c[3 * j + 2];
                v.n2 = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
                bad |= !(__builtin_fabsf(v.x) < 1e3f && __builtin_fabsf(v.y) < 1e3f && __builtin_fabsf(v.z) < 1e3f);
                r2 = __builtin_fmaxf(r2, v.n2);
            }
            pts[j] = v;
        }
        if (bad) atomicOr(&s_bad, 1);
        else atomicMax(&s_r2, __builtin_bit_cast(int, r2));  // non-negative floats order like their bits
    }
    __syncthreads();
    const bool cloud_bad = s_bad != 0;
    const float R = __builtin_sqrtf(__builtin_bit_cast(float, s_r2));

    float bx[NT], by[NT], bz[NT], b1[NT], b2[NT], thr[NT];
    bool full[NT];
    float v1[NT], v2[NT], v3[NT], v4[NT];
    int i1[NT], i2[NT], i3[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int i = (slice * 4 + w) * (NT * 32) + t * 32 + col;
        i = i < P ? i : P - 1;
        bx[t] = basis[3 * i];
        by[t] = basis[3 * i + 1];
        bz[t] = basis[3 * i + 2];
        b1[t] = -2.f * (h ? by[t] : bx[t]);
        b2[t] = h ? 1.f : -2.f * bz[t];
        const float nb = __builtin_sqrtf(__builtin_fmaf(bz[t], bz[t], __builtin_fmaf(by[t], by[t], bx[t] * bx[t])));
        const float S = (R + nb) * (R + nb);
        thr[t] = S * 0x1p-19f;  // 2 eps
        full[t] = cloud_bad ||
                  !(__builtin_fabsf(bx[t]) < 1e3f && __builtin_fabsf(by[t]) < 1e3f && __builtin_fabsf(bz[t]) < 1e3f) ||
                  !(S > 1e-20f);
        v1[t] = v2[t] = v3[t] = v4[t] = __builtin_inff();
        i1[t] = i2[t] = i3[t] = -1;
    }
    if (!cloud_bad) {
        const float* pf = reinterpret_cast<const float*>(pts) + col * 4 + h;
        float a1 = pf[0], a2 = pf[2];
        for (int i = 0; i < ntile; ++i) {
            const float c1 = a1, c2 = a2;
            const int nx = i + 1 < ntile ? i + 1 : i;
            a1 = pf[nx * 128];
            a2 = pf[nx * 128 + 2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, b1[t], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c2, b2[t], acc, 0, 0, 0);
                auto min3 = [](float a, float b_, float c_) { return __builtin_fminf(__builtin_fminf(a, b_), c_); };
                const float m0 = min3(acc[0], acc[1], acc[2]), m1 = min3(acc[3], acc[4], acc[5]);
                const float m2 = min3(acc[6], acc[7], acc[8]), m3 = min3(acc[9], acc[10], acc[11]);
                const float m4 = min3(acc[12], acc[13], acc[14]);
                const float m = min3(min3(m0, m1, m2), min3(m3, m4, acc[15]), __builtin_inff());
                const bool l1 = m < v1[t], l2 = m < v2[t], l3 = m < v3[t];
                int n3 = l3 ? i : i3[t], n2 = l2 ? i : i2[t];  // plain selects (nested ones become branches)
                n3 = l2 ? i2[t] : n3;
                n2 = l1 ? i1[t] : n2;
                i3[t] = n3;
                i2[t] = n2;
                i1[t] = l1 ? i : i1[t];
                v4[t] = __builtin_amdgcn_fmed3f(v3[t], v4[t], m);
                v3[t] = __builtin_amdgcn_fmed3f(v2[t], v3[t], m);
                v2[t] = __builtin_amdgcn_fmed3f(v1[t], v2[t], m);
                v1[t] = __builtin_fminf(v1[t], m);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float best = __builtin_inff();
        int bid = 0;
        const float X = bx[t], Y = by[t], Z = bz[t];
        auto block = [&](int i) {  // this lane's 16 rows of point tile i, the contract's arithmetic
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                const int j = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const Pt4 p = pts[j];
                const float d2 = dist2(p.x, p.y, p.z, X, Y, Z);
                const bool take = (j < N) & ((d2 < best) | ((d2 == best) & (j < bid)));
                best = take ? d2 : best;
                bid = take ? j : bid;
            }
        };
        const float lim = v1[t] + thr[t];
#if defined(BPS_ABLATE) && BPS_ABLATE == 1   // no exact stage at all
        best = v1[t];
        bid = i1[t];
#elif defined(BPS_ABLATE) && BPS_ABLATE == 2  // first block only
        block(i1[t] < 0 ? 0 : i1[t]);
#elif defined(BPS_ABLATE) && BPS_ABLATE == 3  // no full fallback
        block(i1[t] < 0 ? 0 : i1[t]);
        if (i2[t] >= 0 && !(v2[t] > lim)) block(i2[t]);
        if (i3[t] >= 0 && !(v3[t] > lim)) block(i3[t]);
#else
        if (full[t] || !(v4[t] > lim)) {
            for (int i = 0; i < ntile; ++i) block(i);
        } else {
            block(i1[t]);
            if (i2[t] >= 0 && !(v2[t] > lim)) block(i2[t]);
            if (i3[t] >= 0 && !(v3[t] > lim)) block(i3[t]);
        }
#endif
        const float od = __shfl_xor(best, 32);
        const int oi = __shfl_xor(bid, 32);
        const bool take = (od < best) | ((od == best) & (oi < bid));
        best = take ? od : best;
        bid = take ? oi : bid;
        const int i = (slice * 4 + w) * (NT * 32) + t * 32 + col;
        if (h == 0 && i < P) {
            dists[(size_t)b * P + i] = __builtin_sqrtf(best);
            if (ids) ids[(size_t)b * P + i] = bid;
        }
    }
}

__global__ void bps_preset_kernel(float* __restrict__ dists, int32_t* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dists[i] = __builtin_inff();
        if (ids) ids[i] = 0x7fffffff;
    }
}
__global__ void bps_finish_kernel(float* __restrict__ dists, int32_t* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dists[i] = __builtin_sqrtf(dists[i]);
        if (ids && ids[i] == 0x7fffffff) ids[i] = 0;
    }
}

int bps_encode_launch(const float* clouds, int B, int N, const float* basis, int P, float* dists, int32_t* ids,
                      hipStream_t st) {
    if (!clouds || !basis || !dists || B <= 0 || N <= 0 || P <= 0) return DGG_ERR_INVALID;
    const int gx = (P + 1023) / 1024;
    // few clouds: cut them into ranges until ~768 workgroups exist, ranges of at least 256 points (multiples of 16)
    int nchunk = 1, chunk_len = N;
    if ((long long)gx * B < 192 && N >= 512) {
        int want = 768 / (gx * B);
        const int most = N / 256;
        want = want < most ? want : most;
        if (want > 1) {
            chunk_len = ((N + want - 1) / want + 15) / 16 * 16;
            nchunk = (N + chunk_len - 1) / chunk_len;
        }
    }
    static const bool valu_only = [] {  // DGG_BPS=valu: the vector-ALU scan for every shape (A/B timing, tests)
        const char* e = getenv("DGG_BPS");
        return e && !strcmp(e, "valu");
    }();
    if (nchunk <= 1 && N >= 64 && N <= 4096 && !valu_only) {
        const int g5 = (P + 511) / 512;
        const size_t lds = (size_t)((N + 31) / 32) * 32 * sizeof(Pt4);
        hipLaunchKernelGGL((bps_mfma_kernel<4>), dim3((unsigned)g5 * B), dim3(256), lds, st, clouds, N, basis, P, dists, ids,
                           B, g5);
    } else if (nchunk <= 1) {
        hipLaunchKernelGGL((bps_batched_kernel<4, 16, 0>), dim3((unsigned)gx * B), dim3(256), 0, st, clouds, N, basis, P,
                           dists, ids, B, gx, 1, N);
    } else {
        const int n = B * P;
        const dim3 grid((unsigned)gx * B * nchunk);
        hipLaunchKernelGGL(bps_preset_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dists, ids, n);
        hipLaunchKernelGGL((bps_batched_kernel<4, 16, 1>), grid, dim3(256), 0, st, clouds, N, basis, P, dists, ids, B, gx,
                           nchunk, chunk_len);
        if (ids)
            hipLaunchKernelGGL((bps_batched_kernel<4, 16, 2>), grid, dim3(256), 0, st, clouds, N, basis, P, dists, ids, B,
                               gx, nchunk, chunk_len);
        hipLaunchKernelGGL(bps_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dists, ids, n);
    }
    return check_launch("bps_encode");
}

}  // namespace dgg
