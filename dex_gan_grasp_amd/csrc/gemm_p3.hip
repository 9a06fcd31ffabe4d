// bf16x3 GEMM on PRE-SPLIT operands ("planes"): C[M,N] = sum_k A[m][k] B[n][k] where each fp32 operand arrives as its
// three bf16 planes (hi, mid, lo; gemm_x3.hip explains the split and the six products kept), k-contiguous rows,
// K padded with zeros to a multiple of 32 by the producer.
//
// Why: gemm_x3_kernel splits fp32 fragments in registers beside the MFMAs — 352 VALU per wave and k-tile, each element
// split by the two waves that share it — and the SIMD's issue port, not the matrix pipe, bounds it (2760 cycles per
// k-tile against 1536 of MFMA).  The split is a property of the OPERAND, not of the product: the kernels that produce
// the activations (and one pass over the weights per phase) emit the planes once, 6 bytes per element instead of 4,
// and this kernel has no vector ALU work at all in its k loop:
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4, 48 pieces of 1 KiB per k-tile and workgroup, 12 per wave),
//   LDS -> VGPR by ds_read_b128 (one per plane and 32-row block: the 8 consecutive-k bf16 a lane feeds to the MFMA),
//   48 v_mfma_f32_32x32x16_bf16 per wave and k-tile.
// (The loop lives in gemm_p3_block.h: it serves the planes problem inside a grouped gemm_x3_kernel weight-gradient launch;
// the stand-alone kernel of round 2 is gone — gemm_p4.hip, the same product with specialised waves, runs the forward.
// This file keeps the PRODUCERS: planes_kernel, rowdot_kernel.)
// Tile 128 x 128 x 32, 256 threads = 4 waves (2 x 2 of 64 x 64), ring of 3 stages x 48 KB.  A staged operand plane is
// [128 rows][64 B]; its four 16-byte chunks are XOR-swizzled by (row >> 2) & 3 on the SOURCE side of the DMA (the LDS
// image of a DMA is lane-linear), which makes the fragment reads bank-conflict-free: 16 consecutive rows x one logical
// chunk cover all 64 banks once.
// Ring: tile T is issued right after the mid-tile barrier of tile T-3 (every wave has fetched the last fragments of
// tile T-3 by then: its stage is free) and must have landed at the mid-tile barrier of tile T-1, whose second half
// prefetches the first fragments of tile T — two k-tiles of latency budget, one barrier per k-tile.
// The wgrad runs on the same kernel: dW[n][k] = sum_b dY[b][n] X[b][k] is "NT" over the TRANSPOSED planes dY^T [n][b],
// X^T [k][b], which the producers also emit (planes_kernel below) — with the BatchNorm-aware epilogue of GemmProb.
#include <stdint.h>
#include <type_traits>
#include "dgg_internal.h"
#include "gemm_epilogue.h"
#include "gemm_p3_block.h"

namespace dgg {

// ------------------------------------------------------------------------------------------------
// Producer: fp32 matrix [R][C] (pitch ld) -> bf16 planes.
//   dst  (optional): [3][R][ldp]   row-major   (k = column):  value = x * scale[c] + shift[c] if given, columns
//                    C .. ldp-1 zero (ldp a multiple of 64)
//   dstT (optional): [3][C][ldpT]  transposed  (k = row), rows R .. ldpT-1 zero (ldpT a multiple of 64); scale/shift
//                    are NOT applied here (the wgrad consumes the un-affined x_hat, see GemmProb::ep_gamma)
// grid: (ldp or round64(C)) / 64 x (ldpT or round64(R)) / 64 tiles of 64 x 64, z = job (a launch carries up to
// kPlanesJobs matrices: the weight pieces of all ResBlocks of a net go in one launch); 256 threads.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned p3_pack(float a, float b) {
    pf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pbf16x2));
}
__device__ __forceinline__ void p3_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = p3_pack(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = p3_pack(r0, r1);
    const float q0 = r0 - __builtin_bit_cast(float, mid << 16), q1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = p3_pack(q0, q1);
}

__global__ __launch_bounds__(256) void planes_kernel(const PlanesJobs jobs) {
    __shared__ unsigned short tr[3][64][66];  // [plane][column][row], pitch 66: transposing stores 2 banks apart
    const PlanesJob J = jobs.j[blockIdx.z];
    if ((int)blockIdx.x >= J.gx || (int)blockIdx.y >= J.gy) return;  // workgroup-uniform
    const float* __restrict__ src = J.src;
    const int ld = J.ld, R = J.R, C = J.C;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int cq = (tid & 15) * 4, rl = tid >> 4;  // 4 consecutive columns, rows rl + 16 i
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + cq + e;
        sc[e] = (J.scale && c < C) ? J.scale[c] : 1.f;
        sh[e] = (J.shift && c < C) ? J.shift[c] : 0.f;
    }
    float x[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rl + 16 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + cq + e;
            x[i][e] = (r < R && c < C) ? src[(size_t)r * ld + c] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rl + 16 * i;
        if (J.dst && r < R && c0 + cq < J.ldp) {
            unsigned hi[2], mid[2], lo[2];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const bool in0 = c0 + cq + 2 * e2 < C, in1 = c0 + cq + 2 * e2 + 1 < C;
                const float a0 = in0 ? __builtin_fmaf(x[i][2 * e2], sc[2 * e2], sh[2 * e2]) : 0.f;
                const float a1 = in1 ? __builtin_fmaf(x[i][2 * e2 + 1], sc[2 * e2 + 1], sh[2 * e2 + 1]) : 0.f;
                p3_split2(a0, a1, hi[e2], mid[e2], lo[e2]);
            }
            __bf16* o = J.dst + (size_t)r * J.ldp + c0 + cq;
            *reinterpret_cast<pu32x2*>(o) = (pu32x2){hi[0], hi[1]};
            *reinterpret_cast<pu32x2*>(o + J.dplane) = (pu32x2){mid[0], mid[1]};
            *reinterpret_cast<pu32x2*>(o + 2 * J.dplane) = (pu32x2){lo[0], lo[1]};
        }
        if (J.dstT) {
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                unsigned hi, mid, lo;
                p3_split2(x[i][2 * e2], x[i][2 * e2 + 1], hi, mid, lo);
                const int cc = cq + 2 * e2, rr = rl + 16 * i;
                tr[0][cc][rr] = (unsigned short)hi;  tr[0][cc + 1][rr] = (unsigned short)(hi >> 16);
                tr[1][cc][rr] = (unsigned short)mid; tr[1][cc + 1][rr] = (unsigned short)(mid >> 16);
                tr[2][cc][rr] = (unsigned short)lo;  tr[2][cc + 1][rr] = (unsigned short)(lo >> 16);
            }
        }
    }
    if (!J.dstT) return;
    __syncthreads();
    // thread -> column tid / 4, 16 consecutive rows (tid % 4) * 16: 32 bytes per plane
    const int cc = tid >> 2, rq = (tid & 3) * 16;
    if (c0 + cc < C && r0 + rq < J.ldpT) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            unsigned w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (unsigned)tr[p][cc][rq + 2 * e] | ((unsigned)tr[p][cc][rq + 2 * e + 1] << 16);
            __bf16* o = J.dstT + (size_t)p * J.tplane + (size_t)(c0 + cc) * J.ldpT + r0 + rq;
            *reinterpret_cast<pu32x4*>(o) = (pu32x4){w[0], w[1], w[2], w[3]};
            *reinterpret_cast<pu32x4*>(o + 8) = (pu32x4){w[4], w[5], w[6], w[7]};
        }
    }
}

int planes_add(PlanesJobs& jobs, const float* src, int ld, int R, int C, const float* scale, const float* shift, void* dst,
               int ldp, long long dplane, void* dstT, int ldpT, long long tplane) {
    if (jobs.n >= kPlanesJobs || !src || R < 1 || C < 1 || (!dst && !dstT)) return DGG_ERR_INVALID;
    if (dst && ((ldp & 63) || ldp < C || (dplane & 7) || ((uintptr_t)dst & 15))) return DGG_ERR_INVALID;
    if (dstT && ((ldpT & 63) || ldpT < R || (tplane & 7) || ((uintptr_t)dstT & 15))) return DGG_ERR_INVALID;
    PlanesJob& J = jobs.j[jobs.n++];
    J.src = src; J.ld = ld; J.R = R; J.C = C; J.scale = scale; J.shift = shift;
    J.dst = (__bf16*)dst; J.ldp = ldp; J.dplane = dplane; J.dstT = (__bf16*)dstT; J.ldpT = ldpT; J.tplane = tplane;
    J.gx = dst ? ldp / 64 : (C + 63) / 64;
    J.gy = dstT ? ldpT / 64 : (R + 63) / 64;
    return DGG_OK;
}

int planes_run(const PlanesJobs& jobs, hipStream_t st) {
    if (jobs.n < 1) return DGG_OK;
    int gx = 0, gy = 0;
    for (int i = 0; i < jobs.n; ++i) {
        if (jobs.j[i].gx > gx) gx = jobs.j[i].gx;
        if (jobs.j[i].gy > gy) gy = jobs.j[i].gy;
    }
    hipLaunchKernelGGL(planes_kernel, dim3(gx, gy, jobs.n), dim3(256), 0, st, jobs);
    return check_launch("planes");
}

// out[n] = bias[n] + sum_c W[n][c] v[c]: one wave per row, lane l takes the columns l, l + 64, ... in order, then a fixed
// butterfly: deterministic.  grid (ceil(maxR / 4), jobs), 256 threads.
__global__ __launch_bounds__(256) void rowdot_kernel(const RowDotJobs jobs) {
    const RowDotJob J = jobs.j[blockIdx.y];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= J.R) return;  // wave-uniform
    const float* __restrict__ w = J.w + (size_t)row * J.ld;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int c = lane;
    for (; c + 192 < J.C; c += 256) {   // four independent chains: 4 loads in flight per lane
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(w[c + 64 * u], J.v[c + 64 * u], acc[u]);
    }
    for (; c < J.C; c += 64) acc[0] = __builtin_fmaf(w[c], J.v[c], acc[0]);
    float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) J.out[row] = (J.bias ? J.bias[row] : 0.f) + t;
}

int rowdot_add(RowDotJobs& jobs, const float* w, int ld, int R, int C, const float* v, const float* bias, float* out) {
    if (jobs.n >= 3 || !w || !v || !out || R < 1 || C < 1) return DGG_ERR_INVALID;
    RowDotJob& J = jobs.j[jobs.n++];
    J.w = w; J.v = v; J.bias = bias; J.out = out; J.ld = ld; J.R = R; J.C = C;
    return DGG_OK;
}

int rowdot_run(const RowDotJobs& jobs, hipStream_t st) {
    if (jobs.n < 1) return DGG_OK;
    int maxr = 0;
    for (int i = 0; i < jobs.n; ++i)
        if (jobs.j[i].R > maxr) maxr = jobs.j[i].R;
    hipLaunchKernelGGL(rowdot_kernel, dim3((maxr + 3) / 4, jobs.n), dim3(256), 0, st, jobs);
    return check_launch("rowdot");
}

int planes_launch(const float* src, int ld, int R, int C, const float* scale, const float* shift, void* dst, int ldp,
                  long long dplane, void* dstT, int ldpT, long long tplane, hipStream_t st) {
    PlanesJobs jobs;
    jobs.n = 0;
    const int rc = planes_add(jobs, src, ld, R, C, scale, shift, dst, ldp, dplane, dstT, ldpT, tplane);
    return rc != DGG_OK ? rc : planes_run(jobs, st);
}

}  // namespace dgg
