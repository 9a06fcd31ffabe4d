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

namespace dgg {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char p_lds_char;
typedef __attribute__((address_space(3))) float p_lds_f32;
typedef __attribute__((address_space(3))) pu32x4 p_lds_u32x4;

constexpr int PK = 32;                  // k-tile
constexpr int PT = 128;                 // tile rows / columns
constexpr int PS = 3;                   // ring stages
constexpr int PPL = PT * PK * 2;        // bytes of one plane of one operand tile (8 KB)
constexpr int POP = 3 * PPL;            // one operand tile: 3 planes (24 KB)
constexpr int PSTAGE = 2 * POP;         // A + B (48 KB)
constexpr int PRG = PT / 16;            // 1-KiB DMA pieces (16 rows x 64 B) per plane (8)
constexpr int PNP = 2 * 3 * PRG;        // pieces per k-tile (48)
constexpr int PPW = PNP / 4;            // per wave (12)

#ifndef DGG_P3_ABLATE
#define DGG_P3_ABLATE 0  // diagnostics only: 1 no DMA in the loop, 2 no MFMA, 3 no LDS reads in the loop
#endif
#define P3_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int N>
__device__ __forceinline__ void p3_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void p3_barrier() {  // raw: __syncthreads() would also drain the DMA ring (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct P3Frags {
    pbf16x8 f[4][3];  // [A block 0, A block 1, B block 0, B block 1][plane]
};

template <bool WGRAD>
__global__ __launch_bounds__(256) void gemm_p3_kernel(const GemmBatch batch) {
#ifndef DGG_P3_LDS_PAD
#define DGG_P3_LDS_PAD 0
#endif
    __shared__ __attribute__((aligned(16))) char smem[PS * PSTAGE + DGG_P3_LDS_PAD];

    // ---- work assignment: as gemm_f32_kernel (problem select, XCD-contiguous remap, split-K range)
    GemmProb P = batch.p[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < batch.nprob && (int)blockIdx.x >= batch.p[q].blk_begin) P = batch.p[q];
    int local = (int)blockIdx.x - P.blk_begin;
    {
        const int nb = P.tiles_m * P.tiles_n * (P.splits > 0 ? P.splits : 1);
        const int q = nb >> 3, r = nb & 7;
        const int xcd = local & 7, pos = local >> 3;
        local = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    const int ntile = P.tiles_m * P.tiles_n;
    const int split = local / ntile;
    const int tile = local - split * ntile;
    const int tn_i = tile / P.tiles_m;
    const int tm_i = tile - tn_i * P.tiles_m;
    const int m0 = tm_i * PT, n0 = tn_i * PT;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nkt = (kend - kbeg) / PK;  // K and kchunk are multiples of 32 (checked by the launcher)

    // the compiler does not model the DMA's LDS writes: launder the LDS offset (see gemm_x3.hip)
    unsigned lds0 = (unsigned)(size_t)(p_lds_char*)smem;
    asm volatile("" : "+s"(lds0));

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;

    // ---- staging: piece j = wave + 4 q -> (operand j / 24, plane (j % 24) / 8, row group j % 8); lane -> row 16 rg +
    // lane / 4, LDS slot lane % 4, which receives the row's logical chunk slot ^ swizzle(row)
    unsigned voff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int j = wave + 4 * q;
        const int opnd = j / (3 * PRG), rg = j % PRG;
        const int row = 16 * rg + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        int gr = (opnd ? n0 : m0) + row;
        const int nrows = opnd ? P.N : P.M;
        gr = gr < nrows ? gr : nrows - 1;  // such rows only feed accumulator rows / columns that are never stored
        voff[q] = ((unsigned)gr * (unsigned)(opnd ? P.ldb : P.lda) + 8u * (unsigned)lc) * 2u;
    }
    auto issue_piece = [&](int t, int q, unsigned stage) {  // stage: byte offset of tile t's ring stage
        const int j = wave + 4 * q;
        const int opnd = j / (3 * PRG), plane = (j % (3 * PRG)) / PRG, rg = j % PRG;
        const char* base = reinterpret_cast<const char*>(opnd ? P.B : P.A) +
                           ((size_t)plane * (size_t)(opnd ? P.b_plane : P.a_plane) + (size_t)(kbeg + t * PK)) * 2u;  // uniform
        const unsigned dst = lds0 + stage + (unsigned)opnd * POP + (unsigned)plane * PPL + (unsigned)rg * 1024u;
        if (DGG_P3_ABLATE == 1 && t >= PS) return;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + voff[q]), (p_lds_f32*)(size_t)dst, 16, 0, 0);
    };

    // ---- compute: 2 x 2 waves over the tile, 64 x 64 each
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int h = lane >> 5;
    unsigned fa[2], fb[2];  // byte offsets inside an operand plane of this lane's fragment for k-step s (block 0)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ra = wm0 + (lane & 31), rb = wn0 + (lane & 31);
        fa[s] = (unsigned)ra * 64u + (unsigned)(((2 * s + h) ^ ((ra >> 2) & 3)) * 16);
        fb[s] = (unsigned)rb * 64u + (unsigned)(((2 * s + h) ^ ((rb >> 2) & 3)) * 16);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // read unit u (0..11) of k-step (t, s): fragment u / 3, plane u % 3
    auto read_unit = [&](unsigned stage, int s, int u, P3Frags& dst) {
        const int f = u / 3, p = u % 3;
        if (DGG_P3_ABLATE == 3 && (stage != 0 || s != 0)) return;
        const unsigned st = lds0 + stage;
        const unsigned a = st + (f < 2 ? fa[s] : POP + fb[s]) + (unsigned)p * PPL + (unsigned)(f & 1) * (32u * 64u);
        dst.f[f][p] = __builtin_bit_cast(pbf16x8, *(p_lds_u32x4*)(size_t)a);
    };
    // MFMA g (0..23) of a k-step: block pair (g / 12, (g / 6) % 2), product g % 6 — small terms first:
    // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    auto mfma_one = [&](int g, const P3Frags& F) {
        const int ta = g / 12, tb = (g / 6) & 1, pr = g % 6;
        const int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
        const int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
        if (DGG_P3_ABLATE == 2) return;
        P3_MFMA(F.f[ta][ia], F.f[2 + tb][ib], acc[ta][tb]);
    };

    P3Frags fr[2];
    using No = std::false_type;
    using Yes = std::true_type;
    // One k-tile = two basic blocks with a pinned issue order (left alone, the compiler clusters the MFMAs and the
    // matrix pipe idles while the reads drain).  Entering: fr[0] = fragments of (t, 0).
    //   first half : 24 MFMAs on fr[0], between them the 12 reads of (t, 1) -> fr[1]
    //   NEXT       : wait for this wave's pieces of tile t+1 (W12: tile t+2's stay in flight), barrier
    //   second half: 24 MFMAs on fr[1], between them the 12 reads of (t+1, 0) -> fr[0] and (ISSUE) the 12 DMA pieces
    //                of tile t+3 into the stage tile t has just vacated
    unsigned so = 0;  // ring stage (byte offset) of tile t; tile t+1 lives in the next one
    // DMA issue is spread evenly over the k loop — one piece every 4 MFMAs: the vector-memory path takes ~140 cycles per
    // 1-KiB piece and CU, and a wave that issues into a full queue stalls with its MFMAs behind it (in-order issue).
    // Tile T goes into the stage tile T-3 vacates at its mid-tile barrier: pieces 0-5 in the second half of tile T-3,
    // pieces 6-11 in the first half of tile T-2; it must have landed at the mid-tile barrier of tile T-1.
    //   first half : 24 MFMAs on fr[0]; the 12 reads of (t, 1) -> fr[1]; (ISSUE_A) pieces 6-11 of tile t+2
    //   NEXT       : wait for this wave's pieces of tile t+1 (W12: the 12 pieces of tile t+2 stay in flight), barrier
    //   second half: 24 MFMAs on fr[1]; the 12 reads of (t+1, 0) -> fr[0]; (ISSUE_B) pieces 0-5 of tile t+3
    auto ktile = [&](int t, auto issue_a_tag, auto issue_b_tag, auto next_tag, auto w12_tag) {
        constexpr bool ISSUE_A = decltype(issue_a_tag)::value, ISSUE_B = decltype(issue_b_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value, W12 = decltype(w12_tag)::value;
        const unsigned sn = so + PSTAGE == PS * PSTAGE ? 0u : so + PSTAGE;
        const unsigned sp = so == 0 ? (PS - 1) * PSTAGE : so - PSTAGE;  // stage of tile t-1 = of tile t+2
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            mfma_one(g, fr[0]);
            if ((g & 1) == 0) read_unit(so, 1, g >> 1, fr[1]);
            if constexpr (ISSUE_A) {
                if ((g & 3) == 1) issue_piece(t + 2, 6 + (g >> 2), sp);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NEXT) {
            if constexpr (W12) p3_wait_vm<PPW>();
            else p3_wait_vm<0>();
            p3_barrier();
        }
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            mfma_one(g, fr[1]);
            if constexpr (NEXT) {
                if ((g & 1) == 0) read_unit(sn, 0, g >> 1, fr[0]);
            }
            if constexpr (ISSUE_B) {
                if ((g & 3) == 3) issue_piece(t + 3, g >> 2, so);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        so = sn;
    };

    // prologue: tiles 0 .. 2 in flight, tile 0 landed, fragments of (0, 0)
    for (int t = 0; t < PS; ++t)
        if (t < nkt) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) issue_piece(t, q, (unsigned)t * PSTAGE);
        }
    if (nkt >= 3) p3_wait_vm<2 * PPW>();
    else if (nkt == 2) p3_wait_vm<PPW>();
    else p3_wait_vm<0>();
    p3_barrier();
    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 12; ++u) read_unit(0u, 0, u, fr[0]);
    }
    // tile t issues the second half of tile t+2 (if t >= 1: tiles 0-2 were staged whole by the prologue) and the first
    // half of tile t+3
    int t = 0;
    if (nkt > 3) { ktile(t, No{}, Yes{}, Yes{}, Yes{}); ++t; }          // t = 0: tile 3's first half
    for (; t + 3 < nkt; ++t) ktile(t, Yes{}, Yes{}, Yes{}, Yes{});
    if (t + 2 < nkt) { if (t >= 1) ktile(t, Yes{}, No{}, Yes{}, Yes{}); else ktile(t, No{}, No{}, Yes{}, Yes{}); ++t; }
    if (t + 1 < nkt) { ktile(t, No{}, No{}, Yes{}, No{}); ++t; }
    if (t < nkt) ktile(t, No{}, No{}, No{}, No{});

    p3_barrier();  // the ring is idle (every tile was waited for and read): the epilogue reuses its LDS
    gemm_epilogue<PT, PT, 2, 2, WGRAD>(P, acc, reinterpret_cast<float*>(smem), split, tm_i, m0, n0, (int)threadIdx.x);
}

// Operands: planes (a_plane / b_plane != 0), 16-byte aligned bases, pitches and plane distances multiples of 8 elements,
// K (and a split's kchunk) a multiple of 32 with the padding zero-filled.  128 x 128 tiles.
int gemm_p3_launch(GemmBatch& batch, bool wgrad, hipStream_t st) {
    if (batch.nprob < 1 || batch.nprob > 4) return DGG_ERR_INVALID;
    int blocks = 0;
    for (int i = 0; i < batch.nprob; ++i) {
        GemmProb& p = batch.p[i];
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.B || !p.C || !p.a_plane || !p.b_plane) return DGG_ERR_INVALID;
        if ((p.K & 31) || (p.lda & 7) || (p.ldb & 7) || (p.a_plane & 7) || (p.b_plane & 7) || p.lda < p.K || p.ldb < p.K ||
            (((uintptr_t)p.A | (uintptr_t)p.B) & 15) || (p.splits > 0 && (p.kchunk & 31)))
            return DGG_ERR_INVALID;
        p.tiles_m = (p.M + PT - 1) / PT;
        p.tiles_n = (p.N + PT - 1) / PT;
        p.blk_begin = blocks;
        blocks += p.tiles_m * p.tiles_n * (p.splits > 0 ? p.splits : 1);
    }
    batch.total_blocks = blocks;
    if (wgrad) hipLaunchKernelGGL((gemm_p3_kernel<true>), dim3(blocks), dim3(256), 0, st, batch);
    else hipLaunchKernelGGL((gemm_p3_kernel<false>), dim3(blocks), dim3(256), 0, st, batch);
    return check_launch("gemm_p3");
}

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
