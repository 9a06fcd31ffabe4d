// bf16x3 GEMM on pre-split operands with SPECIALISED waves: C[M,N] = sum_k A[m][k] B[n][k], each fp32 operand given as
// its three bf16 planes (hi, mid, lo; gemm_x3.hip explains the split and the six products kept), rows k-contiguous, K
// zero-padded to a multiple of 32 by the producer.  Same operands, tile, LDS image, arithmetic, product order and
// epilogue as gemm_p3_kernel (gemm_p3.hip) — bit-identical results — but the two instruction streams that kernel
// interleaves in every wave are given to different waves:
//   waves 0-3 ("consumers", one per SIMD, 2 x 2 of 64 x 64): per k-tile 24 ds_read_b128 + 48 v_mfma_f32_32x32x16_bf16,
//             nothing else — no vector-memory instruction ever sits in front of an MFMA;
//   waves 4-7 ("loaders", the second wave of each SIMD): per k-tile 12 global_load_lds_dwordx4 each (the 48 one-KiB
//             pieces of a 128 x 32 x (3 + 3) planes stage), a counted s_waitcnt vmcnt and the barrier.
// Why (round-2 measurements, profiles/r02_experiments.txt 1): in gemm_p3_kernel the MFMA + LDS-read stream alone runs
// 24 us of a 768 x 4120 x 1024 product and the DMA stream alone 16 us, but together 34 us — a wave that issues a DMA
// piece into a busy vector-memory queue stalls for 60-185 cycles with its MFMAs behind it (in-order issue), so the two
// streams add instead of overlapping.  A loader wave that stalls holds nobody up: the matrix pipe of its SIMD belongs
// to the consumer, and a wave parked at s_barrier takes no issue slots.
// Ring: 3 stages of 48 KB.  One barrier per k-tile, between its two k-steps ("mid-tile barrier" M_t), joined by all
// eight waves:
//   consumers   ... MFMAs on (t,0), reads of (t,1) | M_t | MFMAs on (t,1), reads of (t+1,0) ...
//   loaders     ... issue tile t+2 (into the stage tile t-1 vacated at M_{t-1}), wait until tile t+1 has landed | M_t | ...
// so that a tile has one and a half k-tiles of latency budget and the consumers never wait for memory themselves.
// Loader waves leave the kernel after the last M_t (s_barrier counts surviving waves only); the consumers run the shared
// epilogue (gemm_epilogue.h).
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "dgg_internal.h"
#include "gemm_epilogue.h"

namespace dgg {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned qu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char q_lds_char;
typedef __attribute__((address_space(3))) float q_lds_f32;
typedef __attribute__((address_space(3))) qu32x4 q_lds_u32x4;

constexpr int QK = 32;                  // k-tile
constexpr int QT = 128;                 // tile rows / columns
constexpr int QS = 3;                   // ring stages
constexpr int QPL = QT * QK * 2;        // bytes of one plane of one operand tile (8 KB)
constexpr int QOP = 3 * QPL;            // one operand tile: 3 planes (24 KB)
constexpr int QSTAGE = 2 * QOP;         // A + B (48 KB)
constexpr int QRG = QT / 16;            // 1-KiB DMA pieces (16 rows x 64 B) per plane (8)
constexpr int QNP = 2 * 3 * QRG;        // pieces per k-tile (48)
constexpr int QPW = QNP / 4;            // per loader wave (12)

#ifndef DGG_P4_ABLATE
#define DGG_P4_ABLATE 0  // diagnostics only: 1 no DMA in the loop, 2 no MFMA, 3 no LDS reads in the loop
#endif
#define P4_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int N>
__device__ __forceinline__ void p4_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void p4_barrier() {  // raw: __syncthreads() would also drain the loaders' DMA ring (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct P4Frags {
    qbf16x8 f[4][3];  // [A block 0, A block 1, B block 0, B block 1][plane]
};

template <bool WGRAD>
__global__ __launch_bounds__(512) void gemm_p4_kernel(const GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) char smem[QS * QSTAGE];

    // ---- work assignment: as gemm_p3_kernel (problem select, XCD-contiguous remap, split-K range)
    GemmProb P = batch.p[0];
#pragma unroll
    for (int q = 1; q < kMaxProb; ++q)
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
    const int m0 = tm_i * QT, n0 = tn_i * QT;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nkt = (kend - kbeg) / QK;  // K and kchunk are multiples of 32 (checked by the launcher)

    // the compiler does not model the DMA's LDS writes: launder the LDS offset (see gemm_x3.hip)
    unsigned lds0 = (unsigned)(size_t)(q_lds_char*)smem;
    asm volatile("" : "+s"(lds0));

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;

    if (wave >= 4) {
        // =========================================================================== loader waves
        // piece j = lw + 4 q -> (operand j / 24, plane (j % 24) / 8, row group j % 8); lane -> row 16 rg + lane / 4, LDS
        // slot lane % 4, which receives the row's logical 16-byte chunk slot ^ swizzle(row) (the DMA's LDS image is
        // lane-linear: the swizzle is applied on the source side)
        const int lw = wave - 4;
        unsigned voff[QPW];
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            const int j = lw + 4 * q;
            const int opnd = j / (3 * QRG), rg = j % QRG;
            const int row = 16 * rg + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            int gr = (opnd ? n0 : m0) + row;
            const int nrows = opnd ? P.N : P.M;
            gr = gr < nrows ? gr : nrows - 1;  // such rows only feed accumulator rows / columns that are never stored
            voff[q] = ((unsigned)gr * (unsigned)(opnd ? P.ldb : P.lda) + 8u * (unsigned)lc) * 2u;
        }
        auto issue_tile = [&](int t, unsigned stage) {  // stage: byte offset of tile t's ring stage
#pragma unroll
            for (int q = 0; q < QPW; ++q) {
                const int j = lw + 4 * q;
                const int opnd = j / (3 * QRG), plane = (j % (3 * QRG)) / QRG, rg = j % QRG;
                const char* base = reinterpret_cast<const char*>(opnd ? P.B : P.A) +
                                   ((size_t)plane * (size_t)(opnd ? P.b_plane : P.a_plane) + (size_t)(kbeg + t * QK)) * 2u;
                const unsigned dst = lds0 + stage + (unsigned)opnd * QOP + (unsigned)plane * QPL + (unsigned)rg * 1024u;
                if (DGG_P4_ABLATE == 1 && t >= QS) continue;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + voff[q]), (q_lds_f32*)(size_t)dst, 16,
                                                 0, 0);
            }
        };
        // prologue: tiles 0 .. 2 in flight, tile 0 landed at the first barrier
        for (int t = 0; t < QS; ++t)
            if (t < nkt) issue_tile(t, (unsigned)t * QSTAGE);
        if (nkt >= 3) p4_wait_vm<2 * QPW>();
        else if (nkt == 2) p4_wait_vm<QPW>();
        else p4_wait_vm<0>();
        p4_barrier();
        // before M_t (t = 0 .. nkt-2): tile t+2 goes into the stage tile t-1 vacated at M_{t-1}; tile t+1 must have landed
        unsigned st2 = 2 * QSTAGE;  // stage of tile t+2
        for (int t = 0; t + 1 < nkt; ++t) {
            if (t >= 1 && t + 2 < nkt) issue_tile(t + 2, st2);
            if (t + 2 < nkt) p4_wait_vm<QPW>();
            else p4_wait_vm<0>();
            p4_barrier();
            st2 = st2 + QSTAGE == QS * QSTAGE ? 0u : st2 + QSTAGE;
        }
        return;  // s_barrier waits for the surviving waves only: the consumers' last barrier and epilogue need no loader
    }

    // =============================================================================== consumer waves: 2 x 2, 64 x 64 each
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
    auto read_unit = [&](unsigned stage, int s, int u, P4Frags& dst) {
        const int f = u / 3, p = u % 3;
        if (DGG_P4_ABLATE == 3 && (stage != 0 || s != 0)) return;
        const unsigned st = lds0 + stage;
        const unsigned a = st + (f < 2 ? fa[s] : QOP + fb[s]) + (unsigned)p * QPL + (unsigned)(f & 1) * (32u * 64u);
        dst.f[f][p] = __builtin_bit_cast(qbf16x8, *(q_lds_u32x4*)(size_t)a);
    };
    // MFMA g (0..23) of a k-step: block pair (g / 12, (g / 6) % 2), product g % 6 — small terms first:
    // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi  (the order of gemm_p3_kernel: same bits)
    auto mfma_one = [&](int g, const P4Frags& F) {
        const int ta = g / 12, tb = (g / 6) & 1, pr = g % 6;
        const int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
        const int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
        if (DGG_P4_ABLATE == 2) return;
        P4_MFMA(F.f[ta][ia], F.f[2 + tb][ib], acc[ta][tb]);
    };

    P4Frags fr[2];
    using No = std::false_type;
    using Yes = std::true_type;
    // One k-tile = two basic blocks with a pinned issue order (left alone, the compiler clusters the MFMAs and the matrix
    // pipe idles while the reads drain).  Entering: fr[0] = fragments of (t, 0).
    unsigned so = 0;  // ring stage (byte offset) of tile t; tile t+1 lives in the next one
    auto ktile = [&](auto next_tag) {
        constexpr bool NEXT = decltype(next_tag)::value;
        const unsigned sn = so + QSTAGE == QS * QSTAGE ? 0u : so + QSTAGE;
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            mfma_one(g, fr[0]);
            if ((g & 1) == 0) read_unit(so, 1, g >> 1, fr[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NEXT) p4_barrier();  // M_t: tile t+1 has landed (the loaders saw to it), tile t's stage is free
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            mfma_one(g, fr[1]);
            if constexpr (NEXT) {
                if ((g & 1) == 0) read_unit(sn, 0, g >> 1, fr[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        so = sn;
    };

    p4_barrier();  // tile 0 has landed
    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 12; ++u) read_unit(0u, 0, u, fr[0]);
    }
    int t = 0;
    for (; t + 1 < nkt; ++t) ktile(Yes{});
    if (t < nkt) ktile(No{});

    p4_barrier();  // every consumer has read its last fragments (the loaders are gone): the epilogue reuses the ring's LDS
    gemm_epilogue<QT, QT, 2, 2, WGRAD, 1>(P, acc, reinterpret_cast<float*>(smem), split, tm_i, m0, n0, (int)threadIdx.x);
}

// Operands: planes (a_plane / b_plane != 0), 16-byte aligned bases, pitches and plane distances multiples of 8 elements,
// K (and a split's kchunk) a multiple of 32 with the padding zero-filled.  128 x 128 tiles, 512 threads.
int gemm_p4_launch(GemmBatch& batch, bool wgrad, hipStream_t st) {
    if (batch.nprob < 1 || batch.nprob > kMaxProb) return DGG_ERR_INVALID;
    int blocks = 0;
    for (int i = 0; i < batch.nprob; ++i) {
        GemmProb& p = batch.p[i];
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.B || !p.C || !p.a_plane || !p.b_plane) return DGG_ERR_INVALID;
        if ((p.K & 31) || (p.lda & 7) || (p.ldb & 7) || (p.a_plane & 7) || (p.b_plane & 7) || p.lda < p.K || p.ldb < p.K ||
            (((uintptr_t)p.A | (uintptr_t)p.B) & 15) || (p.splits > 0 && (p.kchunk & 31)))
            return DGG_ERR_INVALID;
        if ((long long)p.M * p.ldc * 4 >= (1ll << 31) || (p.ep_gamma && (long long)p.M * p.ep_ldw * 4 >= (1ll << 31))) {
            set_error_text("GEMM output of %d rows x pitch %d floats: the epilogues address C through 32-bit byte offsets, "
                           "rows * pitch * 4 must stay below 2 GiB (chunk the batch: at most %lld rows at this pitch)",
                           p.M, p.ldc, ((1ll << 31) - 1) / (4ll * (p.ldc > 0 ? p.ldc : 1)));
            return DGG_ERR_INVALID;
        }   // 32-bit byte offsets in the epilogue (gemm_epilogue.h)
        p.tiles_m = (p.M + QT - 1) / QT;
        p.tiles_n = (p.N + QT - 1) / QT;
        p.blk_begin = blocks;
        blocks += p.tiles_m * p.tiles_n * (p.splits > 0 ? p.splits : 1);
    }
    batch.total_blocks = blocks;
    if (wgrad) hipLaunchKernelGGL((gemm_p4_kernel<true>), dim3(blocks), dim3(512), 0, st, batch);
    else hipLaunchKernelGGL((gemm_p4_kernel<false>), dim3(blocks), dim3(512), 0, st, batch);
    return check_launch("gemm_p4");
}

int gemm_planes_launch(GemmBatch& batch, bool wgrad, hipStream_t st) { return gemm_p4_launch(batch, wgrad, st); }

}  // namespace dgg
