// Basis-Point-Set encode (A1): replaces bps_torch.encode -> chamfer_distance NN kernel as reached
// from DexGanGrasp/data/bps_encoder.py:33 and dexgangrasp_real_robot.py:139.
//   id_i = argmin_j |b_i - x_j|^2   (strict '<': the lowest j wins a tie),   dist_i = sqrt(min)
// Arithmetic contract (identical to oracle/bps_oracle.c, so ids are bit-exact against it):
//   dx = x - bx ...;  d2 = fmaf(dz, dz, fmaf(dy, dy, dx*dx));  dist = sqrtf(d2) correctly rounded.
//
// Arithmetic intensity ~1e3 flop/B (4096*N pair evaluations against 12*N + 32 KiB bytes per cloud): never HBM bound.
// Two kernels, one answer (ids and distance bits are identical on every route):
//  - bps_mfma_kernel (below, rounds 3-4): clouds of >= 64 points.  A two-plane fp16 coarse pass on the matrix cores SELECTS
//    candidate blocks of 16 points, the contract's arithmetic on the candidates DECIDES.  One workgroup per (cloud, 1024 basis
//    points) for many clouds; for few clouds of many points — the reference's own call: ONE cloud of 15k-73k points — one
//    workgroup per (chunk of the cloud, 1024 basis points), chunks meeting by a 64-bit minimum in a caller-owned workspace
//    (dgg_bps_encode_ws, bps_merge_kernel);
//  - bps_batched_kernel (round 2), the vector-ALU scan: a lane owns basis points in registers, the cloud point index is
//    wave-uniform, so x_j, y_j, z_j arrive through the SCALAR cache and are broadcast as SGPR operands; 256-thread
//    workgroups, 4 basis points per lane (two packed pairs).  Serves clouds of fewer than 64 points, large clouds when no
//    workspace is given (RANGES of the cloud merged through the output arrays with atomicMin in two passes) and
//    dgg_set_bps_algo(DGG_BPS_SCAN): the second opinion the regression tests compare every output with.
#include "dgg_internal.h"
#include <cstdlib>
#include <cstring>

namespace dgg {

__device__ __forceinline__ float dist2(float x, float y, float z, float bx, float by, float bz) {
    const float dx = x - bx, dy = y - by, dz = z - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Three consecutive floats (a basis point, a cloud point) of the SCAN: the compiler makes one global_load_dwordx3 of it.  The
// matrix-core kernel has no such load (round 4: basis points through LDS, ragged point loads as one asm statement — see
// "Operands that were not there yet" in DESIGN.md 3.4 and tools/vmcnt_audit.py).
__device__ __forceinline__ void load3(const float* p, float& x, float& y, float& z) {
    x = p[0];
    y = p[1];
    z = p[2];
}

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
        float tx, ty, tz;
        load3(basis + 3 * i, tx, ty, tz);
        bx[q >> 1][q & 1] = tx;
        by[q >> 1][q & 1] = ty;
        bz[q >> 1][q & 1] = tz;
        best[q] = __builtin_inff();
        blk[q] = -1;
    }
    if (MODE == 2) {
        // `best` is the TARGET, the global minimum pass 1 left in dists (+inf: nothing was smaller, leave ids).  These dword
        // loads are issued only once the 12-byte basis loads above have landed: no counted wait over two classes of loads
        // (tools/vmcnt_audit.py)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            int i = slice * (256 * BPT) + q * 256 + threadIdx.x;
            i = i < P ? i : P - 1;
            best[q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)dists + (size_t)b * P + i, __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT));
        }
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
                float px, py, pz;
                load3(c + 3 * (j0 + u), px, py, pz);
                const float d2 = dist2(px, py, pz, bx[q >> 1][q & 1], by[q >> 1][q & 1], bz[q >> 1][q & 1]);
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
// Many clouds: a coarse pass on the matrix cores SELECTS, the contract's arithmetic DECIDES (round 3).
// The scan above spends ~3.7 vector instructions per (basis point, cloud point) pair.  Here one pair costs 1/16 of a
// vector min3 and 1/1024 of ONE matrix instruction:
//     v_ij = x_j.(-2 b_i) + |x_j|^2 + c_i  =  |x_j - b_i|^2 + bias_i            (c_i ~ |b_i|^2 + bias_i, any constant does)
// as a K = 14 dot product of fp16 values.  Cloud and basis are scaled by ONE power of two per cloud so that the cloud's
// largest |x| sits in [32, 64) — exact, and far from both ends of the fp16 range (a basis point may be ~60 cloud radii
// away before its lane has to take the all-points path); every scaled fp32 operand is split into two fp16 planes,
// x = h + m + r with |r| <= 2^-23 |x| (round to nearest twice, the residual step exact), and three of the four plane
// products are kept per coordinate; |x|^2 goes in three planes, c as 256 x two planes of c / 256.  One
// v_mfma_f32_32x32x16_f16 evaluates 32 points x 32 basis points (32 cycles of the matrix pipe; the instruction keeps fp16
// denormal inputs, tools/exp/f16_mfma_probe.hip).  Operand rows (k-slots in pairs, one 32-bit word each):
//     point, lanes 0-31: (xh,xh) (xm,yh) (yh,ym) (zh,zh)      lanes 32-63: (zm,nh) (nm,nl) (256,256) (0,0)   n = |x|^2
//     basis            : (qxh,qxm) (qxh,qyh) (qym,qyh) (qzh,qzm)            (qzh,1) (1,1) (ch,cm) (0,0)     q = -2b
// — 32 B of LDS and one ds_read_b128 per point, tile and lane.  (History: the fp32 matrix instruction runs at the vector
// rate and does NOT overlap vector work, 1.12x only; three bf16 planes need 23 slots = two matrix instructions per tile,
// 0.53 ms against this version's 0.47: the matrix instructions share the vector issue port, 8 cycles each.)
// v orders the cloud points of one basis point like the squared distance does, but not bit for bit like the contract's
// d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx)),  so:
//   - a lane sees 16 of a tile's 32 points for its basis point (the accumulator rows of its half-wave) = one BLOCK.  The
//     block minimum (8 v_min3_f32) gets the tile index written over its 7 lowest mantissa bits: one register is value
//     and index, and the four smallest keys are kept by three v_med3_f32 and a v_min_u32 — no compare / select.
//     (v > 0 thanks to the bias, so the keys order like the values; the overwritten bits cost up to 206 ulp between two
//     keys, budgeted as 2^-15 v in lim.)
//   - |v_ij - (c_i - |b_i|^2) - d2_ij| <= eps_i = 35u S_i  (all in the scaled units; the keys' index bits are budgeted apart),  S_i = (R + |b_i|)^2,
//     R = the cloud's largest |x_j|, u = 2^-24: dropped plane products 12u S, 14 accumulations of terms whose magnitudes
//     sum to <= S: 14u S (worst case, whatever the order inside the instruction), |x|^2 3u S, the contract's own d2
//     within 5u of the true squared distance (tests/test_bps_coarse_bound.py restates the operands in numpy: 0.1 eps
//     observed).  Hence every point whose d2 can be the minimum or tie with it lies in a block whose key is
//     <= k1 + 2 eps + 2^-15 k1 =: lim (k1 the smallest key).
//   - the (at most three) blocks with keys <= lim are re-evaluated with the contract's instructions on the cloud's own
//     fp32 coordinates (global memory, 3 x 16-byte loads per 4 consecutive points), the answer is the lexicographic
//     minimum of (d2, j): the sequential scan's lowest-index rule.  A fourth key <= lim, a cloud or basis point with a
//     coordinate not below 1e3 in magnitude (NaN, inf, the 3e19 cases of the tests) or a degenerate scale send the lane
//     through ALL its points with the exact formula: always right, never fast.
// Ids and distances are bit-identical to the scan's (tests/test_gpu_parity.py, tools/exp/bps_ab.py on 4.2 M outputs,
// tools/exp/bps_fuzz.py).  512-thread workgroups (8 waves x 128 basis points) share one staged cloud; 64 KB at 2048
// points: two per CU, so one workgroup's staging / exact stage runs beside the other's matrix work.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float bf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bu32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 bh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bh16x2 __attribute__((ext_vector_type(2)));
struct Planes2 { _Float16 h, m; };  // x = h + m + r,  |r| <= 2^-23 |x|  (round to nearest twice; the residual step is exact)
__device__ __forceinline__ Planes2 split2(float x) {
    Planes2 p;
    p.h = (_Float16)x;
    p.m = (_Float16)(x - (float)p.h);
    return p;
}
__device__ __forceinline__ unsigned pk16(_Float16 even, _Float16 odd) {  // k-slots 2n, 2n+1 of one operand word
    return __builtin_bit_cast(unsigned, bh16x2{even, odd});
}

constexpr int kBpsMfmaMaxN = 2560;  // points per workgroup: 32 B of LDS per point, tile index in 7 bits
constexpr int kBpsChunkMax = 1536;  // MERGE: points per chunk (44 B of LDS per point + 12 KB of basis -> two workgroups per CU; <= NW * 256)
constexpr int kBpsQueue = 512;      // work-list entries per wave (128 winners + further candidates; the rest in place)
#ifndef BPS_NW
#define BPS_NW 8
#endif

// Four consecutive cloud points as three 16-byte vectors {x0 y0 z0 x1} {y1 z1 x2 y2} {z2 x3 y3 z3}.  AL: the cloud (chunk) is
// 16-byte aligned and the four points lie inside it -> three global_load_dwordx4; otherwise point by point with the index
// clamped to the last point (rows past the cloud repeat its LAST point: a copy inside a block changes no block minimum, a
// block of copies only is a candidate at worst, and the exact stage skips j >= n).  That path is ONE asm statement: twelve
// naturally aligned dword loads and the `s_waitcnt vmcnt(0)` that covers them — the compiler sees neither 12-byte loads
// at 4-byte alignment nor loads of a second class among its counted waits, and cannot read a result before the wait
// (an asm wait behind compiler-visible loads does not bind it: it copies loaded registers under its own counted waits
// in front of such an asm; DESIGN.md 3.4, tools/vmcnt_audit.py).
template <bool AL>
__device__ __forceinline__ void load_quad(const float* __restrict__ c, int j0, int n, bf32x4& p0, bf32x4& p1, bf32x4& p2) {
    if (AL && j0 + 4 <= n) {
        const bf32x4* pq = reinterpret_cast<const bf32x4*>(c + 3 * j0);
        p0 = pq[0], p1 = pq[1], p2 = pq[2];
    } else {
        const float* a0 = c + 3 * min(j0, n - 1);
        const float* a1 = c + 3 * min(j0 + 1, n - 1);
        const float* a2 = c + 3 * min(j0 + 2, n - 1);
        const float* a3 = c + 3 * min(j0 + 3, n - 1);
        float v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11;
        asm volatile(
            "global_load_dword %0, %12, off\n\tglobal_load_dword %1, %12, off offset:4\n\tglobal_load_dword %2, %12, off offset:8\n\t"
            "global_load_dword %3, %13, off\n\tglobal_load_dword %4, %13, off offset:4\n\tglobal_load_dword %5, %13, off offset:8\n\t"
            "global_load_dword %6, %14, off\n\tglobal_load_dword %7, %14, off offset:4\n\tglobal_load_dword %8, %14, off offset:8\n\t"
            "global_load_dword %9, %15, off\n\tglobal_load_dword %10, %15, off offset:4\n\tglobal_load_dword %11, %15, off offset:8\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7), "=&v"(v8), "=&v"(v9),
              "=&v"(v10), "=&v"(v11)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
            : "memory");
        p0 = bf32x4{v0, v1, v2, v3}, p1 = bf32x4{v4, v5, v6, v7}, p2 = bf32x4{v8, v9, v10, v11};
    }
}

// MERGE = false: one workgroup = one whole cloud x NW * 128 basis points, results straight into dists / ids (many clouds).
// MERGE = true (round 4: the reference's own call shape, ONE cloud of 15k-73k points, dexgangrasp_real_robot.py:133-141): the
// cloud is cut into `nchunk` chunks of CH points (a multiple of 32), a workgroup = one chunk x NW * 128 basis points with a
// power-of-two scale of the CHUNK's own; it leaves the chunk's answer per basis point — (d2 bits << 32 | global index j),
// ~0 if nothing was below +inf — in `scratch[cloud][chunk][P]`, and bps_merge_kernel takes the 64-bit minimum over the
// chunks: the lexicographic minimum of (d2, j), the sequential scan's lowest-index rule over the whole cloud.
template <int NT, int NW, bool AL, bool MERGE>  // 32-wide basis tiles per wave, waves per workgroup
__global__ __launch_bounds__(NW * 64, 4) void bps_mfma_kernel(const float* __restrict__ clouds, int Ncloud, int CH, int nchunk,
                                                           const float* __restrict__ basis, int P,
                                                           float* __restrict__ dists, int32_t* __restrict__ ids,
                                                           unsigned long long* __restrict__ scratch, int B, int gx) {
    extern __shared__ bu32x4 planes[];  // [ntile * 32][2]: the operand words of lanes 0-31 | lanes 32-63
    __shared__ int s_r2, s_bad;
    __shared__ int s_qn[NW];  // entries on each wave's work list
    const int L = blockIdx.x, per = 8 * gx;
    int b, slice, chunk = 0;
    if (MERGE) {  // (cloud, chunk, slice): slices fastest
        slice = L % gx;
        const int rest = L / gx;
        chunk = rest % nchunk;
        b = rest / nchunk;
    } else if (L < (B / 8) * per) {  // the gx slices of a cloud on one XCD (see bps_batched_kernel)
        b = (L / per) * 8 + (L & 7);
        slice = (L % per) >> 3;
    } else {
        b = L / gx;
        slice = L % gx;
    }
    const int jbase = MERGE ? chunk * CH : 0;
    const int N = MERGE ? min(CH, Ncloud - jbase) : Ncloud;  // points of this workgroup's (chunk of the) cloud
    const float* __restrict__ c = clouds + ((size_t)b * Ncloud + jbase) * 3;  // AL: 16-byte aligned (launcher)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, col = lane & 31, h = lane >> 5;
    const int ntile = (N + 31) >> 5;
    if (tid == 0) {
        s_r2 = 0;
        s_bad = 0;
    }
    // LDS behind the operand planes AND behind the exact stage's slots / lists (which overlay the planes): the workgroup's
    // NW * 128 basis points (`sb`), then — MERGE — the chunk's fp32 coordinates (`sc`).
    constexpr int kExactBytes = NW * (NT * 32 + kBpsQueue / 2) * 8, kBasisFloats = NW * NT * 32 * 3;
    const int boff = max(((CH + 31) >> 5) * 1024, kExactBytes);
    float* const sb = reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + boff);
    bf32x4* const sc = reinterpret_cast<bf32x4*>(sb + kBasisFloats);
    // MERGE: a chunk has at most NW * 256 points, i.e. at most ONE quad of four points per thread: it is requested together
    // with the basis points (one memory round trip for both), stays in registers from pass 1 to pass 2, and goes to LDS
    // (`sc`, 12 B per point in the cloud's own layout) — the exact stage gathers its candidate blocks from there: gathered
    // from global memory, 13 loads x 64 lanes each in a cache line of its own kept the texture path busy for ~6 us per
    // pass.
    bf32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;
    const bool mine = MERGE && tid < ntile * 8;
    {
        // The basis points reach the lanes through LDS: 16-byte loads of the slice (the same class as the cloud's loads) or,
        // for a ragged last slice / a basis that is not 16-byte aligned, aligned dwords with the index clamped to P - 1 —
        // never a 12-byte load at a 4-byte aligned address among other loads (the round-3 wrong-operand failure,
        // DESIGN.md 3.4), and coalesced instead of one 12-byte piece per lane.
        const int first = slice * (NW * NT * 32);
        if (((uintptr_t)basis & 15) == 0 && first + NW * NT * 32 <= P) {
            if (mine) load_quad<AL>(c, 4 * tid, N, q0, q1, q2);
            const bf32x4* src = reinterpret_cast<const bf32x4*>(basis + 3 * (size_t)first);
            for (int e = tid; e < kBasisFloats / 4; e += NW * 64) reinterpret_cast<bf32x4*>(sb)[e] = src[e];
        } else {
            for (int e = tid; e < kBasisFloats; e += NW * 64) sb[e] = basis[3 * (size_t)min(first + e / 3, P - 1) + e % 3];
            if (mine) {  // (behind the dword loads: consumed by their LDS stores before these are waited for)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                load_quad<AL>(c, 4 * tid, N, q0, q1, q2);
            }
        }
    }
    __syncthreads();
    auto load_basis = [&](float (&X)[NT], float (&Y)[NT], float (&Z)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* q = sb + 3 * (w * (NT * 32) + t * 32 + col);
            X[t] = q[0], Y[t] = q[1], Z[t] = q[2];
        }
    };
    auto sane = [](float x, float y, float z) {
        return __builtin_fabsf(x) < 1e3f && __builtin_fabsf(y) < 1e3f && __builtin_fabsf(z) < 1e3f;
    };
    float bxs[NT], bys[NT], bzs[NT];
    load_basis(bxs, bys, bzs);
    auto quad = [&](int j0, bf32x4& p0, bf32x4& p1, bf32x4& p2) {  // the exact stage's four points j0 .. j0 + 3
        if constexpr (MERGE) {
            const bf32x4* pq = sc + 3 * (j0 >> 2);
            p0 = pq[0], p1 = pq[1], p2 = pq[2];
        } else {
            load_quad<AL>(c, j0, N, p0, p1, p2);
        }
    };
    {
        // pass 1: the largest |x|^2 of the cloud -> ONE power-of-two scale
        float r2 = 0.f;
        bool bad = false;
        for (int g = tid; g < ntile * 8; g += NW * 64) {
            const int j0 = 4 * g;
            if (j0 < N) {
                bf32x4 p0 = q0, p1 = q1, p2 = q2;
                if (!MERGE) load_quad<AL>(c, j0, N, p0, p1, p2);
                const float xs[4] = {p0[0], p0[3], p1[2], p2[1]}, ys[4] = {p0[1], p1[0], p1[3], p2[2]};
                const float zs[4] = {p0[2], p1[1], p2[0], p2[3]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    bad |= !sane(xs[k], ys[k], zs[k]);
                    r2 = __builtin_fmaxf(r2, __builtin_fmaf(zs[k], zs[k], __builtin_fmaf(ys[k], ys[k], xs[k] * xs[k])));
                }
            }
        }
        if (bad) atomicOr(&s_bad, 1);
        else atomicMax(&s_r2, __builtin_bit_cast(int, r2));  // non-negative floats order like their bits
    }
    __syncthreads();
    const bool cloud_bad = s_bad != 0;
    const float R = __builtin_sqrtf(__builtin_bit_cast(float, s_r2));
    // scale = 2^e with R * scale in [32, 64): the fp16 planes of the scaled cloud stay far from both ends of the fp16 range
    // (|x'| < 64, |x'|^2 < 4096; a basis point may be up to 4000 / 64 cloud radii away: -2 b' < 8000, |b'|^2 / 256 < 62500),
    // and a power of two changes no mantissa bit.  Below 2^-14 fp16 values are spaced 2^-24: an absolute error of at most
    // 2^-25 per plane, 2^-25 (2 sum|b'_c| + sum|x'_c| + 1) on v — under 0.2 % of eps, which is >= 35u 32 (R' + |b'|).
    float scale = 1.f;
    {
        const float m = R;
        const int k = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u) - 126;  // m = f * 2^k, f in [0.5, 1)
        if (!cloud_bad && m > 1e-30f) scale = __builtin_bit_cast(float, (unsigned)(127 + 6 - k) << 23);
    }
    constexpr float kBpsFar = 4000.f;  // |b| * scale beyond this: the lane takes the all-points path
    {
        // pass 2: the operand rows (rows past the cloud: copies of its last point, load_quad)
        const unsigned C256 = pk16((_Float16)256.f, (_Float16)256.f);  // the constant's slots: c enters as 256 * (c / 256)
        for (int g = tid; g < ntile * 8; g += NW * 64) {
            const int j0 = 4 * g;
            bf32x4 p0 = q0, p1 = q1, p2 = q2;
            if (!MERGE) load_quad<AL>(c, j0, N, p0, p1, p2);
            if (MERGE) sc[3 * g] = p0, sc[3 * g + 1] = p1, sc[3 * g + 2] = p2;
            const float xs[4] = {p0[0], p0[3], p1[2], p2[1]}, ys[4] = {p0[1], p1[0], p1[3], p2[2]};
            const float zs[4] = {p0[2], p1[1], p2[0], p2[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float x = xs[k] * scale, y = ys[k] * scale, z = zs[k] * scale;
                const float n2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
                const Planes2 X = split2(x), Y = split2(y), Z = split2(z);
                const _Float16 nh = (_Float16)n2;
                const float r1 = n2 - (float)nh;
                const _Float16 nm = (_Float16)r1, nl = (_Float16)(r1 - (float)nm);
                planes[2 * (j0 + k)] = bu32x4{pk16(X.h, X.h), pk16(X.m, Y.h), pk16(Y.h, Y.m), pk16(Z.h, Z.h)};
                planes[2 * (j0 + k) + 1] = bu32x4{pk16(Z.m, nh), pk16(nm, nl), C256, 0u};
            }
        }
    }
    // per basis point: S = (R + |b|)^2 in the cloud's units, and whether the coarse pass can be trusted for it at all
    auto basis_point = [&](float X, float Y, float Z, float& bb, float& S) {
        bb = __builtin_fmaf(Z, Z, __builtin_fmaf(Y, Y, X * X));
        const float nb = __builtin_sqrtf(bb);
        S = (R + nb) * (R + nb);
        return cloud_bad || !sane(X, Y, Z) || !(S > 1e-20f) || !(nb * scale < kBpsFar);
    };
    bu32x4 B1[NT];
    float k1[NT], k2[NT], k3[NT], k4[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bx = bxs[t], by = bys[t], bz = bzs[t];
        float bb, S;
        basis_point(bx, by, bz, bb, S);
        const Planes2 X = split2(-2.f * bx * scale), Y = split2(-2.f * by * scale), Z = split2(-2.f * bz * scale);
        const float cc = (bb + S * 0x1p-14f) * (scale * scale) * 0x1p-8f;  // (|b'|^2 + bias') / 256; bias' > 2^-22 c' + eps'
        const _Float16 ch = (_Float16)cc, cm = (_Float16)(cc - (float)ch), one = (_Float16)1.f;
        B1[t] = h ? bu32x4{pk16(Z.h, one), pk16(one, one), pk16(ch, cm), 0u}
                  : bu32x4{pk16(X.h, X.m), pk16(X.h, Y.h), pk16(Y.m, Y.h), pk16(Z.h, Z.m)};
        k1[t] = k2[t] = k3[t] = k4[t] = __builtin_inff();
    }
    __syncthreads();
    if (!cloud_bad) {
        const bu32x4* pl = planes + col * 2 + h;
        bu32x4 a = pl[0];
        for (int i = 0; i < ntile; ++i) {
            const bu32x4 cur = a;
            a = pl[(i + 1 < ntile ? i + 1 : i) * 64];
            const unsigned iv = (unsigned)i, keep = ~127u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bh16x8, cur), __builtin_bit_cast(bh16x8, B1[t]),
                                                             acc, 0, 0, 0);
                // (all values are positive and finite: they order like their bit patterns, and v_min_u32 comes without
                //  the quieting v_max the compiler puts in front of a two-operand v_min_f32)
                auto min3 = [](float a_, float b_, float c_) { return __builtin_fminf(__builtin_fminf(a_, b_), c_); };
                auto min2 = [](float a_, float b_) {
                    const unsigned ua = __builtin_bit_cast(unsigned, a_), ub = __builtin_bit_cast(unsigned, b_);
                    return __builtin_bit_cast(float, __builtin_elementwise_min(ua, ub));
                };
                const float m0 = min3(acc[0], acc[1], acc[2]), m1 = min3(acc[3], acc[4], acc[5]);
                const float m2 = min3(acc[6], acc[7], acc[8]), m3 = min3(acc[9], acc[10], acc[11]);
                const float m4 = min3(acc[12], acc[13], acc[14]);
                const float m = min2(min3(m0, m1, m2), min3(m3, m4, acc[15]));
                float kf;  // (m & ~127) | tile index, one instruction
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(kf) : "v"(m), "v"(keep), "v"(iv));
                k4[t] = __builtin_amdgcn_fmed3f(k3[t], k4[t], kf);
                k3[t] = __builtin_amdgcn_fmed3f(k2[t], k3[t], kf);
                k2[t] = __builtin_amdgcn_fmed3f(k1[t], k2[t], kf);
                k1[t] = min2(k1[t], kf);
            }
        }
    }
    // ---- exact stage.  Every evaluated block offers (d2 bits, j) as one 64-bit key to its basis point's LDS slot:
    // ds_min_u64 IS the lexicographic minimum, for the two half-waves and for blocks in any order.  The candidate blocks
    // (keys <= lim, lim from the smaller k1 of the two half-waves) go onto a per-wave work list and are evaluated 64 at
    // a time by all lanes: ~128 winners + a few % further candidates = 3 passes of 16 rows, where "every lane its own
    // blocks" took 4 passes for the first candidates alone and 2 mostly idle ones per basis tile for the others.
    // The operand planes are dead once every wave has left the main loop: slots and lists live there.
    __syncthreads();
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(planes) + w * (NT * 32 + kBpsQueue / 2);
    unsigned* q = reinterpret_cast<unsigned*>(slot + NT * 32);
    slot[lane] = ~0ull;
    slot[lane + 64] = ~0ull;
    if (lane == 0) s_qn[w] = 0;
    __builtin_amdgcn_wave_barrier();
    auto block = [&](int i, int hh, float X, float Y, float Z, float& bb, int& bj) {
        // 16 rows of point tile i as half-wave hh sees them, the contract's arithmetic; rows ascend: strict '<'
        bf32x4 pt[4][3];
        if (!MERGE && AL && i * 32 + 32 <= N) {  // the whole tile inside the cloud: twelve 16-byte loads, one test
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf32x4* pq = reinterpret_cast<const bf32x4*>(c + 3 * (i * 32 + 8 * g + 4 * hh));
                pt[g][0] = pq[0], pt[g][1] = pq[1], pt[g][2] = pq[2];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) quad(i * 32 + 8 * g + 4 * hh, pt[g][0], pt[g][1], pt[g][2]);
        }
        // (No j < N test below.  Rows past the cloud are copies of its LAST point, bit for bit, at larger indices: inside a
        //  block the strict '<' never prefers a later copy; a block of copies only has a key >= the key of the block that
        //  holds the original — same value, smaller-or-equal minimum, smaller-or-equal tile — so whenever it is a candidate
        //  the original's block is one too, in its own lane, and the slots' 64-bit minimum of (d2, j) keeps the original.)
        // Everything this evaluation loaded — the 12 point vectors and the caller's basis point — is waited for HERE,
        // with vmcnt(0), not by the compiler's counted waits.  With those (correct IF loads return in order: the basis
        // point's 12-byte load third among thirteen 16-byte point loads) lanes 32-63 computed, in one run of three and only
        // with another workgroup busy on the CU, a handful of the 4.2 M distances from operands that were not the ones
        // loaded for them; changing only that load or only this wait cures it (profiles/r03_experiments.txt 7: the
        // elimination, and a probe that does NOT reproduce it in isolation; ROCm 7.2, gfx950).
        // (Round 4 rebuilt the failing load form behind a macro to show that the regression test — tests/test_gpu_parity.py::
        // test_bps_matrix_path_every_output_under_load_repeated — catches it: that build did not fail, profiles/
        // r04_experiments.txt 4; the variant is gone from the source since round 5.)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(pt[0][0]), "+v"(pt[0][1]), "+v"(pt[0][2]), "+v"(pt[1][0]), "+v"(pt[1][1]), "+v"(pt[1][2]),
                       "+v"(pt[2][0]), "+v"(pt[2][1]), "+v"(pt[2][2]), "+v"(pt[3][0]), "+v"(pt[3][1]), "+v"(pt[3][2]),
                       "+v"(X), "+v"(Y), "+v"(Z));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = i * 32 + 8 * g + 4 * hh;
            const bf32x4 p0 = pt[g][0], p1 = pt[g][1], p2 = pt[g][2];
            // two points per packed instruction (v_pk_add / v_pk_mul / v_pk_fma_f32: each half bit-identical to dist2())
            const f32x2 bX = {X, X}, bY = {Y, Y}, bZ = {Z, Z};
            const f32x2 ax = f32x2{p0[0], p0[3]} - bX, ay = f32x2{p0[1], p1[0]} - bY, az = f32x2{p0[2], p1[1]} - bZ;
            const f32x2 cx = f32x2{p1[2], p2[1]} - bX, cy = f32x2{p1[3], p2[2]} - bY, cz = f32x2{p2[0], p2[3]} - bZ;
            const f32x2 da = __builtin_elementwise_fma(az, az, __builtin_elementwise_fma(ay, ay, ax * ax));
            const f32x2 dc = __builtin_elementwise_fma(cz, cz, __builtin_elementwise_fma(cy, cy, cx * cx));
            const float d0 = da[0], d1 = da[1], d2 = dc[0], d3 = dc[1];
            bool lt = d0 < bb;
            bb = lt ? d0 : bb, bj = lt ? j0 : bj;
            lt = d1 < bb;
            bb = lt ? d1 : bb, bj = lt ? j0 + 1 : bj;
            lt = d2 < bb;
            bb = lt ? d2 : bb, bj = lt ? j0 + 2 : bj;
            lt = d3 < bb;
            bb = lt ? d3 : bb, bj = lt ? j0 + 3 : bj;
        }
    };
    auto offer = [&](int bl, float bb, int bj) {
        if (bb < __builtin_inff())
            atomicMin(&slot[bl], ((unsigned long long)__builtin_bit_cast(unsigned, bb) << 32) | (unsigned)bj);
    };
    auto tile_of = [](float k) { return (int)(__builtin_bit_cast(unsigned, k) & 127u); };
    float Xs[NT], Ys[NT], Zs[NT];  // (again from memory: nothing of this lives through the main loop)
    load_basis(Xs, Ys, Zs);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float X = Xs[t], Y = Ys[t], Z = Zs[t];
        float bq, S;
        const bool full = basis_point(X, Y, Z, bq, S);
        const int bl = t * 32 + col;
        const float o1 = __shfl_xor(k1[t], 32);
        const float g1 = __builtin_bit_cast(float, __builtin_elementwise_min(__builtin_bit_cast(unsigned, k1[t]),
                                                                             __builtin_bit_cast(unsigned, o1)));
        // 2 eps: 2 * 35u S (in the keys' units: scaled) rounded up (0x1.41: + the fp16 spacing below 2^-14), + the keys' own
        // bits: a key is (m & ~127) | tile, so the smallest key may sit 127 ulp BELOW its block's m while the key of the
        // block that holds the true minimum sits up to ntile - 1 <= 79 ulp ABOVE its m: 206 ulp, and an ulp is up to
        // 2^-23 of the value -> 2^-15 g1 (256 ulp just above a power of two, 512 just below the next; 2^-16 was 128: short
        // when v ~ S sits just above a power of two — tests/test_bps_coarse_bound.py::test_tile_index_bits_in_the_keys)
        const float lim = g1 + (S * (scale * scale) * 0x1.41p-18f + g1 * 0x1p-15f);
        if (full || !(k4[t] > lim)) {
            float bb = __builtin_inff();
            int bj = 0;
            for (int i = 0; i < ntile; ++i) block(i, h, X, Y, Z, bb, bj);
            offer(bl, bb, bj);
        } else {
            // keys ascend: the candidates are a prefix of (k1, k2, k3)
            const int cnt = (!(k1[t] > lim) ? 1 : 0) + (!(k2[t] > lim) ? 1 : 0) + (!(k3[t] > lim) ? 1 : 0);
            if (cnt > 0) {
                const int pos = atomicAdd(&s_qn[w], cnt);
                const float ks[3] = {k1[t], k2[t], k3[t]};
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (e < cnt) {
                        if (pos + e < kBpsQueue) {
                            q[pos + e] = ((unsigned)bl << 8) | ((unsigned)h << 7) | (unsigned)tile_of(ks[e]);
                        } else {  // list full: in place
                            float b2 = __builtin_inff();
                            int j2 = 0;
                            block(tile_of(ks[e]), h, X, Y, Z, b2, j2);
                            offer(bl, b2, j2);
                        }
                    }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int n = min(s_qn[w], kBpsQueue);
        const int first = (slice * NW + w) * (NT * 32);
        int base = 0;
        for (; n - base > 16; base += 64) {
            if (base + lane < n) {
                const unsigned it = q[base + lane];
                const int bl = it >> 8;
                const float* bq = sb + 3 * (w * (NT * 32) + bl);
                float X = bq[0], Y = bq[1], Z = bq[2], bb = __builtin_inff();
                int bj = 0;
                block(it & 127u, (it >> 7) & 1u, X, Y, Z, bb, bj);
                offer(bl, bb, bj);
            }
        }
        // the last <= 16 entries (the usual case: 128 winners fill two passes, a handful of further candidates remain):
        // four lanes per entry, four rows each — a quarter of a pass
        if (base + (lane >> 2) < n) {
            const unsigned it = q[base + (lane >> 2)];
            const int bl = it >> 8;
            const float* bq = sb + 3 * (w * (NT * 32) + bl);
            float X = bq[0], Y = bq[1], Z = bq[2], bb = __builtin_inff();
            int bj = 0;
            const int j0 = (int)(it & 127u) * 32 + 8 * (lane & 3) + 4 * (int)((it >> 7) & 1u);
            bf32x4 p0, p1, p2;
            quad(j0, p0, p1, p2);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(X), "+v"(Y), "+v"(Z));
            const float d0 = dist2(p0[0], p0[1], p0[2], X, Y, Z), d1 = dist2(p0[3], p1[0], p1[1], X, Y, Z);
            const float d2 = dist2(p1[2], p1[3], p2[0], X, Y, Z), d3 = dist2(p2[1], p2[2], p2[3], X, Y, Z);
            bool lt = d0 < bb;
            bb = lt ? d0 : bb, bj = lt ? j0 : bj;
            lt = d1 < bb;
            bb = lt ? d1 : bb, bj = lt ? j0 + 1 : bj;
            lt = d2 < bb;
            bb = lt ? d2 : bb, bj = lt ? j0 + 2 : bj;
            lt = d3 < bb;
            bb = lt ? d3 : bb, bj = lt ? j0 + 3 : bj;
            offer(bl, bb, bj);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int bl = lane + 64 * e, i = first + bl;
            const unsigned long long k = slot[bl];
            if (i < P) {
                const bool none = k == ~0ull;  // nothing was below +inf: the scan's answer
                if (MERGE) {
                    scratch[((size_t)b * nchunk + chunk) * P + i] = none ? ~0ull : k + (unsigned long long)jbase;
                } else {
                    dists[(size_t)b * P + i] = none ? __builtin_inff() : __builtin_sqrtf(__builtin_bit_cast(float, (unsigned)(k >> 32)));
                    if (ids) ids[(size_t)b * P + i] = none ? 0 : (int)(unsigned)k;
                }
            }
        }
    }
}

// The chunks of a cloud meet here: the 64-bit minimum of (d2 bits << 32 | j) over the chunks is the lexicographic minimum.
// 32 basis points x 8 groups of chunks per workgroup (256-byte rows of the scratch per load instruction).
__global__ __launch_bounds__(256) void bps_merge_kernel(const unsigned long long* __restrict__ scratch, int nchunk, int P,
                                                        float* __restrict__ dists, int32_t* __restrict__ ids) {
    __shared__ unsigned long long s[8][32];
    const int il = threadIdx.x & 31, cg = threadIdx.x >> 5, tiles = (P + 31) >> 5;
    const int b = blockIdx.x / tiles, i = (blockIdx.x % tiles) * 32 + il;
    unsigned long long best = ~0ull;
    if (i < P) {
        const unsigned long long* __restrict__ q = scratch + (size_t)b * nchunk * P + i;
#pragma unroll 8
        for (int ch = cg; ch < nchunk; ch += 8) {
            const unsigned long long v = q[(size_t)ch * P];
            best = v < best ? v : best;
        }
    }
    s[cg][il] = best;
    __syncthreads();
    if (cg == 0 && i < P) {
#pragma unroll
        for (int g = 1; g < 8; ++g) best = s[g][il] < best ? s[g][il] : best;
        const bool none = best == ~0ull;
        dists[(size_t)b * P + i] = none ? __builtin_inff() : __builtin_sqrtf(__builtin_bit_cast(float, (unsigned)(best >> 32)));
        if (ids) ids[(size_t)b * P + i] = none ? 0 : (int)(unsigned)best;
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

// Which arithmetic: 0 (default) = by shape, 1 = the vector-ALU scan for every shape (A/B timing; the regression tests
// compare the matrix-core path with it output by output).  Process-wide; DGG_BPS=valu selects 1 at start-up.
static int g_bps_algo = -1;
int bps_algo() {
    if (g_bps_algo < 0) {
        const char* e = getenv("DGG_BPS");
        g_bps_algo = (e && !strcmp(e, "valu")) ? 1 : 0;
    }
    return g_bps_algo;
}
void bps_set_algo(int a) { g_bps_algo = a == 1 ? 1 : 0; }

namespace {
constexpr int kBpsNW = BPS_NW;
struct BpsPlan {
    int kind;    // 0: plain scan, 1: scan over ranges (no workspace), 2: matrix cores, whole clouds, 3: matrix cores, chunks + merge
    int g;       // basis slices per cloud (matrix paths)
    int ch, nchunk;
    size_t ws;   // bytes of workspace kind 3 needs
};
// ws_avail: bytes the caller can offer (SIZE_MAX: planning for dgg_bps_workspace_bytes)
BpsPlan bps_plan(int B, int N, int P, size_t ws_avail) {
    BpsPlan p = {0, (P + kBpsNW * 128 - 1) / (kBpsNW * 128), N, 1, 0};
    const int gx = (P + 1023) / 1024;
    const bool few = (long long)gx * B < 192 && N >= 512;
    if (bps_algo() != 1 && N >= 64) {
        // chunks: ~512 workgroups (two per CU), 256..1536 points each (planes + coordinates + basis: 78 KB of LDS), multiples of 32
        if (few || N > kBpsMfmaMaxN) {
            long long ch = ((long long)N * p.g * B + 511) / 512;
            ch = (ch + 31) / 32 * 32;
            ch = ch < 256 ? 256 : ch > kBpsChunkMax ? kBpsChunkMax : ch;
            const int nchunk = (int)((N + ch - 1) / ch);
            const size_t ws = (size_t)B * nchunk * P * sizeof(unsigned long long);
            if (nchunk > 1 && ws <= ws_avail && ws <= ((size_t)256 << 20)) {
                p.kind = 3, p.ch = (int)ch, p.nchunk = nchunk, p.ws = ws;
                return p;
            }
        }
        if (N <= kBpsMfmaMaxN) {
            p.kind = 2;
            return p;
        }
    }
    if (few) {  // ranges of at least 256 points (multiples of 16) until ~768 workgroups exist
        int want = 768 / (gx * B);
        const int most = N / 256;
        want = want < most ? want : most;
        if (want > 1) {
            p.ch = ((N + want - 1) / want + 15) / 16 * 16;
            p.nchunk = (N + p.ch - 1) / p.ch;
            p.kind = 1;
        }
    }
    return p;
}

template <bool AL, bool MERGE>
void bps_mfma_launch(const BpsPlan& pl, const float* clouds, int B, int N, const float* basis, int P, float* dists,
                     int32_t* ids, unsigned long long* scratch, hipStream_t st) {
    const int rows = MERGE ? pl.ch : N;
    size_t lds = (size_t)((rows + 31) / 32) * 32 * 32;                       // operand planes, then reused by the exact stage:
    const size_t lds_exact = (size_t)kBpsNW * (4 * 32 + kBpsQueue / 2) * 8;   // per wave 128 slots + the work list
    lds = lds > lds_exact ? lds : lds_exact;
    lds += (size_t)kBpsNW * 128 * 12;                                        // + the workgroup's basis points (`sb`)
    if (MERGE) lds += (size_t)((rows + 31) / 32) * 32 * 12;                  // + the chunk's fp32 coordinates (`sc`)
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&bps_mfma_kernel<4, kBpsNW, AL, MERGE>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
    (void)big_lds;
    hipLaunchKernelGGL((bps_mfma_kernel<4, kBpsNW, AL, MERGE>), dim3((unsigned)pl.g * B * (MERGE ? pl.nchunk : 1)),
                       dim3(kBpsNW * 64), lds, st, clouds, N, pl.ch, pl.nchunk, basis, P, dists, ids, scratch, B, pl.g);
}
}  // namespace

long long bps_workspace_bytes(int B, int N, int P) {
    if (B <= 0 || N <= 0 || P <= 0) return DGG_ERR_INVALID;
    return (long long)bps_plan(B, N, P, SIZE_MAX).ws;
}

int bps_encode_launch(const float* clouds, int B, int N, const float* basis, int P, float* dists, int32_t* ids, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    if (!clouds || !basis || !dists || B <= 0 || N <= 0 || P <= 0) return DGG_ERR_INVALID;
    if (ws && ((uintptr_t)ws & 7)) return DGG_ERR_INVALID;
    const BpsPlan pl = bps_plan(B, N, P, ws ? ws_bytes : 0);
    const int gx = (P + 1023) / 1024;
    // 16-byte loads of four points: every cloud (and chunk: ch % 32 == 0) must start on a 16-byte boundary
    const bool al = ((uintptr_t)clouds & 15) == 0 && (B == 1 || N % 4 == 0);
    if (pl.kind == 3) {
        unsigned long long* scratch = static_cast<unsigned long long*>(ws);
        if (al) bps_mfma_launch<true, true>(pl, clouds, B, N, basis, P, dists, ids, scratch, st);
        else bps_mfma_launch<false, true>(pl, clouds, B, N, basis, P, dists, ids, scratch, st);
        hipLaunchKernelGGL(bps_merge_kernel, dim3((unsigned)((P + 31) / 32) * B), dim3(256), 0, st, scratch, pl.nchunk, P,
                           dists, ids);
    } else if (pl.kind == 2) {
        if (al) bps_mfma_launch<true, false>(pl, clouds, B, N, basis, P, dists, ids, nullptr, st);
        else bps_mfma_launch<false, false>(pl, clouds, B, N, basis, P, dists, ids, nullptr, st);
    } else if (pl.kind == 0) {
        hipLaunchKernelGGL((bps_batched_kernel<4, 16, 0>), dim3((unsigned)gx * B), dim3(256), 0, st, clouds, N, basis, P,
                           dists, ids, B, gx, 1, N);
    } else {
        const int n = B * P;
        const dim3 grid((unsigned)gx * B * pl.nchunk);
        hipLaunchKernelGGL(bps_preset_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dists, ids, n);
        hipLaunchKernelGGL((bps_batched_kernel<4, 16, 1>), grid, dim3(256), 0, st, clouds, N, basis, P, dists, ids, B, gx,
                           pl.nchunk, pl.ch);
        if (ids)
            hipLaunchKernelGGL((bps_batched_kernel<4, 16, 2>), grid, dim3(256), 0, st, clouds, N, basis, P, dists, ids, B,
                               gx, pl.nchunk, pl.ch);
        hipLaunchKernelGGL(bps_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dists, ids, n);
    }
    return check_launch("bps_encode");
}

}  // namespace dgg
