// The k loop of the bf16x3 GEMM on PRE-SPLIT operands (gemm_p3.hip explains it), as a device function over one work item
// so that it can also serve a planes problem inside a grouped gemm_x3_kernel launch (gemm_x3.hip).
#pragma once
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

// One (problem, split, tile) work item `bid` of a grouped launch on the calling 256-thread workgroup; smem: PS * PSTAGE
// bytes, 16-byte aligned.  Every thread returns from it with the epilogue's stores issued.
template <bool WGRAD>
__device__ __forceinline__ void gemm_p3_block(const GemmBatch& batch, const int bid, char* smem) {
    // ---- work assignment: as gemm_f32_kernel (problem select, XCD-contiguous remap, split-K range)
    GemmProb P = batch.p[0];
#pragma unroll
    for (int q = 1; q < kMaxProb; ++q)
        if (q < batch.nprob && bid >= batch.p[q].blk_begin) P = batch.p[q];
    int local = bid - P.blk_begin;
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

}  // namespace dgg
