// bf16x3 GEMM on fp32 operands with SPECIALISED waves (round 3; since round 5 the kernel of launches of >= 1024 workgroups
// only — the inference sweep — beside gemm_x5.hip, which explains the arithmetic): problem descriptors, split-K slabs and
// epilogues as every GEMM kernel of the library, 128 x 128 x 32 tiles, and the fp32 -> (hi, mid, lo) split outside the
// MFMA waves.  ("gemm_x3_kernel" below is the first in-register-split kernel, rounds 2-4, 2 x 2 waves of 64 x 64 that
// each split their own fragments; "gemm_p3 / gemm_p4" were the pre-split planes kernels.  Both are deleted; the
// measurements against them stay in the comments because they are why this kernel looks the way it does.)
//   waves 4-7 ("loaders", the second wave of each SIMD): global -> VGPR (coalesced: 16 rows x 128 B of a k-contiguous
//             operand, 8 k-rows x 256 B of a k-strided one), [input-BatchNorm affine, k-tail mask], split ONCE per workgroup
//             (gemm_x3_kernel: every element split by the two waves that share it), three ds_write_b128 per lane and
//             task into the bf16 plane tiles of an LDS stage;
//   waves 0-3 ("consumers", 2 x 2 of 64 x 64): 24 ds_read_b128 + 48 v_mfma_f32_32x32x16_bf16 per k-tile, nothing else —
//             (LDS image: plane tile = [128 rows][64 B], 16-byte chunks
//             XOR-swizzled by (row >> 2) & 3: fragment reads and loader stores are both bank-conflict-free).
// Why this works where gemm_x3_kernel is issue-bound (2760 cycles per k-tile against 1536 of MFMA): a SIMD's vector ALU
// work of ANOTHER wave does fit into the shadow of an MFMA stream — tools/exp/split_probe.hip: with 176 split
// instructions per k-tile in the partner wave (one fragment-equivalent per 12 MFMAs) the MFMA wave keeps 1537 cycles per
// k-tile and the split wave needs 1505; with the 352 of the in-register scheme it would need 2679.  A loader that waits for
// memory or for the LDS holds nobody up.  No operand planes in HBM, no producers: fp32 operands as they are (any 4-byte
// alignment), k-strided operands transposed for free by which lane loads what.
// Ring: 2 LDS stages of 48 KB.  One barrier per k-tile between its two k-steps (M_t), joined by all eight waves:
//   consumers   ... MFMAs on (t,0), reads of (t,1) | M_t | MFMAs on (t,1), reads of (t+1,0) ...
//   loaders     ... loads of tile t+2 issued, tile t+1 (loaded one interval earlier) split into the stage tile t-1 left | M_t
// Loader waves leave after the last M_t (s_barrier counts surviving waves); the consumers run the shared epilogue.
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "dgg_internal.h"
#include "gemm_epilogue.h"

namespace dgg {

typedef __bf16 ybf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ybf16x2 __attribute__((ext_vector_type(2)));
typedef float yf32x2 __attribute__((ext_vector_type(2)));
typedef float yf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned yu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char y_lds_char;
typedef __attribute__((address_space(3))) yu32x4 y_lds_u32x4;

constexpr int YK = 32;                  // k-tile
constexpr int YT = 128;                 // tile rows / columns
constexpr int YS = 2;                   // LDS stages
constexpr int YPL = YT * YK * 2;        // bytes of one plane of one operand tile (8 KB)
constexpr int YOP = 3 * YPL;            // one operand tile: 3 planes (24 KB)
constexpr int YSTAGE = 2 * YOP;         // A + B (48 KB)

#ifndef DGG_X4_ABLATE
#define DGG_X4_ABLATE 0  // diagnostics only: 1 loaders skip the split (planes = raw bits), 2 no MFMA, 3 no global loads in the loop, 4 no LDS stores in the loop
#endif
#define X4_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ void x4_barrier() {  // LDS traffic of this wave retired, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ unsigned x4_pack(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE): low half = a
    yf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ybf16x2));
}
// 8 consecutive-k fp32 values of one operand row -> 16 bytes of each plane (the exact 3-way split, stage by stage so
// that the four dependent chains overlap)
__device__ __forceinline__ void x4_split(const float (&x)[8], yu32x4& hi, yu32x4& mid, yu32x4& lo) {
    unsigned h[4], m[4], l[4];
    float r[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) h[p] = x4_pack(x[2 * p], x[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        r[2 * p] = x[2 * p] - __builtin_bit_cast(float, h[p] << 16);
        r[2 * p + 1] = x[2 * p + 1] - __builtin_bit_cast(float, h[p] & 0xffff0000u);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) m[p] = x4_pack(r[2 * p], r[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        r[2 * p] -= __builtin_bit_cast(float, m[p] << 16);
        r[2 * p + 1] -= __builtin_bit_cast(float, m[p] & 0xffff0000u);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) l[p] = x4_pack(r[2 * p], r[2 * p + 1]);
    hi = (yu32x4){h[0], h[1], h[2], h[3]};
    mid = (yu32x4){m[0], m[1], m[2], m[3]};
    lo = (yu32x4){l[0], l[1], l[2], l[3]};
}

struct X4Frags {
    ybf16x8 f[4][3];  // [A block 0, A block 1, B block 0, B block 1][plane]
};

// ------------------------------------------------------------------------------------------------
// Loader.  Loader wave lw stages rows 64 (lw & 1) .. + 63 of operand lw >> 1 (A, B): per k-tile 32 fp32 values per lane =
// four UNITS of 8 consecutive k of one row, fetched by 8 global_load_dwordx4:
//   KC  (stored [row][k]): unit u -> row 64 hf + 16 u + lane / 4, k-chunk lane % 4; two 16-byte loads per unit (a wave
//        instruction reads 16 rows x 128 contiguous bytes);
//   !KC (stored [k][row]): lane -> row quad lane % 16 (rows 64 hf + 4 quad .. + 3), k-chunk lane / 16; load i fetches the
//        quad at k = 8 chunk + i (a wave instruction reads 4 k-rows x 256 contiguous bytes) — after 8 loads the lane holds
//        4 rows x 8 k: the transposition costs nothing.  Row indices are clamped into the matrix (a
//        quad may extend up to 12 bytes past the last row index: such bytes lie inside the operand's row pitch or the
//        next row and feed accumulator columns that are never stored).
// The loads of tile j+1 are issued before tile j is split and stay in flight meanwhile; the compiler places the counted
// s_waitcnt vmcnt itself (it tracks the loaded registers — hand-written asm loads were tried: the register copies the
// compiler may insert between an asm load and an asm wait read registers still in flight).  The one partial k-tile at the
// end of K goes through clamped loads AFTER the loop, so that the loop's bookkeeping stays exact.
// ------------------------------------------------------------------------------------------------
typedef float yf32x4a __attribute__((ext_vector_type(4)));
typedef float yuf4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte load from a dword-aligned address

template <bool KC, bool AFFM>
__device__ __forceinline__ void x4_loader(const float* __restrict__ G, const int ld, const int row0, const int nrows,
                                          const int hf, const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                          const bool mask_tail, const int kbeg, const int kend, const int K, const int nkt,
                                          const unsigned lds_op, const int lane) {
    // ---- per-lane geometry
    int chunk;
    unsigned voff[4];   // KC: byte offset of unit u's row (+ chunk);  !KC: voff[0] = byte offset of the lane's quad
    unsigned lds[4];    // byte offset of unit u inside a plane tile
    if constexpr (KC) {
        chunk = lane & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 64 * hf + 16 * u + (lane >> 2);
            int gr = row0 + row;
            gr = gr < nrows ? gr : nrows - 1;  // such rows only feed accumulator rows / columns that are never stored
            voff[u] = ((unsigned)gr * (unsigned)ld + 8u * (unsigned)chunk) * 4u;
            lds[u] = (unsigned)row * 64u + (unsigned)((chunk ^ ((row >> 2) & 3)) * 16);
        }
    } else {
        chunk = lane >> 4;
        const int quad = lane & 15;
        int gc = row0 + 64 * hf + 4 * quad;
        const int last = (nrows - 1) & ~3;
        gc = gc < last ? gc : last;
        voff[0] = (unsigned)gc * 4u;
        voff[1] = voff[2] = voff[3] = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 64 * hf + 4 * quad + u;
            lds[u] = (unsigned)row * 64u + (unsigned)((chunk ^ ((row >> 2) & 3)) * 16);
        }
    }
    const bool aff = AFFM && a_scale != nullptr;   // a grouped launch mixes problems with and without the affine
    const int nf = min(nkt, (K - kbeg) / YK);       // full k-tiles of this workgroup's range (a split may end inside K)

    yf32x4a raw[2][8];   // [parity][load]: KC load 2u + e = k 4e .. 4e+3 of unit u;  !KC load i = the quad at k = 8 chunk + i
    yf32x4a qv[2][4];    // [parity][scale lo, scale hi, shift lo, shift hi] of the lane's k-chunk
    auto issue = [&](int t, auto par_tag) {    // full tile t
        constexpr int PAR = decltype(par_tag)::value;
        const int k0 = kbeg + t * YK;
        if constexpr (KC) {
            const char* base = reinterpret_cast<const char*>(G + k0);          // uniform
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const yuf4* p = reinterpret_cast<const yuf4*>(base + voff[u]);
                raw[PAR][2 * u] = p[0];
                raw[PAR][2 * u + 1] = p[1];
            }
        } else {
            const char* base = reinterpret_cast<const char*>(G + (size_t)k0 * (size_t)ld);   // uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // (the k-row differs per lane group: its offset goes into the vector part)
                const unsigned vo = voff[0] + (unsigned)(8 * chunk + i) * (unsigned)ld * 4u;
                raw[PAR][i] = *reinterpret_cast<const yuf4*>(base + vo);
            }
        }
        if constexpr (AFFM) {
            if (aff) {
                const yuf4* ps = reinterpret_cast<const yuf4*>(a_scale + k0 + 8 * chunk);
                const yuf4* ph = reinterpret_cast<const yuf4*>(a_shift + k0 + 8 * chunk);
                qv[PAR][0] = ps[0];
                qv[PAR][1] = ps[1];
                qv[PAR][2] = ph[0];
                qv[PAR][3] = ph[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // issued here, not sunk into the split below
    };
    // the 8 k of unit u out of the tile's registers
    auto unit = [&](const yf32x4a (&r)[8], int u, float (&x)[8]) {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = r[2 * u][i]; x[4 + i] = r[2 * u + 1][i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = r[i][u];
        }
    };
    // [affine, k-tail mask (A operand: k >= kend contributes nothing; B's clamped duplicates are finite)], split, store
    auto finish = [&](int t, const yf32x4a (&r)[8], const yf32x4a (&q)[4]) {
        const int k0 = kbeg + t * YK;
        const unsigned st = lds_op + (unsigned)(t & 1) * YSTAGE;
        const bool tail = mask_tail && k0 + YK > kend;   // uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x[8];
            unit(r, u, x);
            if constexpr (AFFM) {
                if (aff) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        x[i] = __builtin_fmaf(x[i], q[0][i], q[2][i]);
                        x[4 + i] = __builtin_fmaf(x[4 + i], q[1][i], q[3][i]);
                    }
                }
            }
            if (tail) {
                const int kv = kend - (k0 + 8 * chunk);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = i < kv ? x[i] : 0.f;
            }
            yu32x4 hi, mid, lo;
            if constexpr (DGG_X4_ABLATE == 1) {
                hi = mid = lo = __builtin_bit_cast(yu32x4, (yf32x4){x[0], x[1], x[2], x[3]});
            } else {
                x4_split(x, hi, mid, lo);
            }
            const unsigned a = st + lds[u];
            if (DGG_X4_ABLATE == 4 && t >= 2) { asm volatile("" ::"v"(hi), "v"(mid), "v"(lo)); continue; }
            *(y_lds_u32x4*)(size_t)a = hi;
            *(y_lds_u32x4*)(size_t)(a + YPL) = mid;
            *(y_lds_u32x4*)(size_t)(a + 2 * YPL) = lo;
        }
    };
    // the partial k-tile at the end of K: plain loads, every k clamped into the operand, no run-ahead
    auto tail_tile = [&](int t) {
        const int k0 = kbeg + t * YK;
        yf32x4a r[8], q[4];
        if constexpr (KC) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* row = reinterpret_cast<const float*>(reinterpret_cast<const char*>(G) + (voff[u] - 32u * (unsigned)chunk));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int k = k0 + 8 * chunk + i;
                    k = k < K ? k : K - 1;
                    r[2 * u + (i >> 2)][i & 3] = row[k];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int k = k0 + 8 * chunk + i;
                k = k < K ? k : K - 1;
                const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(G + (size_t)k * (size_t)ld) + voff[0]);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i][e] = p[e];
            }
        }
        if constexpr (AFFM) {
            if (aff) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int k = k0 + 8 * chunk + i;
                    k = k < K ? k : K - 1;
                    q[i >> 2][i & 3] = a_scale[k];
                    q[2 + (i >> 2)][i & 3] = a_shift[k];
                }
            }
        }
        finish(t, r, q);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // One barrier behind every staged tile: the first lets the consumers start on tile 0, the one behind tile t+1 is M_t.
    // Full tiles: tile j+1's loads go out before tile j is split (they fly meanwhile and during the wait at the barrier).
    auto staged = [&](int j, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        finish(j, raw[PAR], qv[PAR]);
        x4_barrier();
    };
    if (nf > 0) issue(0, P0{});
    int j = 0;
    for (; j + 1 < nf; j += 2) {
        if (!(DGG_X4_ABLATE == 3 && j >= 1)) issue(j + 1, P1{});
        staged(j, P0{});
        if (j + 2 < nf && DGG_X4_ABLATE != 3) issue(j + 2, P0{});
        staged(j + 1, P1{});
    }
    if (j < nf) staged(j, P0{});
    if (nkt > nf) {   // the partial k-tile at the end of K (always the last one)
        tail_tile(nf);
        x4_barrier();
    }
}

// One (problem, split, tile) work item.  `first`: the workgroup's first item; `last`: its last one (the loader waves
// return from the kernel behind it).
template <bool A_KC, bool B_KC, bool AFF>
__device__ __forceinline__ void gemm_x4_block(const GemmBatch& batch, const int bid, char* smem, const bool first, const bool last) {
    // ---- work assignment: as in gemm_x5_kernel (problem select, XCD-contiguous remap, split-K range)
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
    const int m0 = tm_i * YT, n0 = tn_i * YT;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nkt = (kend - kbeg + YK - 1) / YK;

    const unsigned lds0 = (unsigned)(size_t)(y_lds_char*)smem;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;

    if (wave >= 4) {
        // =========================================================================== loader waves
        const int lw = wave - 4;
        // (a further item of the same workgroup: the consumers' last barrier of the previous item — every fragment of
        // it has been read — is joined before its LDS stages are written again)
        if (!first) x4_barrier();
        if (lw < 2)
            x4_loader<A_KC, AFF>(P.A, P.lda, m0, P.M, lw & 1, P.a_scale, P.a_shift, true, kbeg, kend, P.K, nkt, lds0, lane);
        else
            x4_loader<B_KC, false>(P.B, P.ldb, n0, P.N, lw & 1, nullptr, nullptr, false, kbeg, kend, P.K, nkt, lds0 + YOP, lane);
        return;  // (last item: s_barrier waits for the surviving waves only — the consumers' last barrier and epilogue)
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
    auto read_unit = [&](unsigned stage, int s, int u, X4Frags& dst) {
        const int f = u / 3, p = u % 3;
        const unsigned a = lds0 + stage + (f < 2 ? fa[s] : YOP + fb[s]) + (unsigned)p * YPL + (unsigned)(f & 1) * (32u * 64u);
        dst.f[f][p] = __builtin_bit_cast(ybf16x8, *(y_lds_u32x4*)(size_t)a);
    };
    // MFMA g (0..23) of a k-step: block pair (g / 12, (g / 6) % 2), product g % 6 — small terms first:
    // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    auto mfma_one = [&](int g, const X4Frags& F) {
        const int ta_ = g / 12, tb_ = (g / 6) & 1, pr = g % 6;
        const int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
        const int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
        if (DGG_X4_ABLATE == 2) return;
        X4_MFMA(F.f[ta_][ia], F.f[2 + tb_][ib], acc[ta_][tb_]);
    };

    X4Frags fr[2];
    using No = std::false_type;
    using Yes = std::true_type;
    unsigned so = 0;  // LDS stage (byte offset) of tile t; tile t+1 lives in the other one
    auto ktile = [&](auto next_tag) {
        constexpr bool NEXT = decltype(next_tag)::value;
        const unsigned sn = so ^ (unsigned)YSTAGE;
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            mfma_one(g, fr[0]);
            if ((g & 1) == 0) read_unit(so, 1, g >> 1, fr[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NEXT) x4_barrier();  // M_t: tile t+1 is staged (the loaders saw to it), tile t's stage is free
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

    x4_barrier();  // tile 0 is staged
#pragma unroll
    for (int u = 0; u < 12; ++u) read_unit(0u, 0, u, fr[0]);
    int t = 0;
    for (; t + 1 < nkt; ++t) ktile(Yes{});
    ktile(No{});

    // every consumer has read its last fragments: the epilogue reuses the LDS (weight gradients) / the loaders, which join
    // this barrier unless they have left, may stage the next item
    x4_barrier();
    (void)last;
    gemm_epilogue<YT, YT, 2, 2, !A_KC && !B_KC, 1>(P, acc, reinterpret_cast<float*>(smem), split, tm_i, m0, n0, (int)threadIdx.x);
}

// One workgroup per work item — or, launched with fewer workgroups than items ("side work", forward products only: their
// epilogue neither touches the LDS nor holds a barrier), a stride loop over them (see gemm_x5_kernel).
template <bool A_KC, bool B_KC, bool AFF>
__global__ __launch_bounds__(512) void gemm_x4_kernel(const GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) char smem[YS * YSTAGE];
    bool first = true;
    for (int bid = (int)blockIdx.x; bid < batch.total_blocks; bid += (int)gridDim.x) {
        gemm_x4_block<A_KC, B_KC, AFF>(batch, bid, smem, first, bid + (int)gridDim.x >= batch.total_blocks);
        first = false;
    }
}

// 128 x 128 tiles.  Operand bases and pitches need 4-byte alignment only; nothing is read outside [0, K) of an operand
// row (a partial last k-tile is fetched element by element with clamped indices).
void gemm_x4_dispatch(int mode, const GemmBatch& batch, hipStream_t st) {
    int nwg = batch.total_blocks;
    // side work (dgg_set_side_work): the forward products walk two work items per workgroup — same items, same sums,
    // half the CUs for twice as long (gemm_x5.hip: launch_x5)
    static const int kdiv = [] { const char* e = getenv("DGG_SIDE_DIV"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
    if (gemm_side_work() && mode == GEMM_NT && nwg > 64) { const int dv = kdiv + gemm_side_work() - 1; nwg = ((nwg + dv - 1) / dv + 7) / 8 * 8; }
    const dim3 grid(nwg), blk(512);
    if (mode == GEMM_NT) {
        bool any = false;
        for (int i = 0; i < batch.nprob; ++i) any = any || batch.p[i].a_scale != nullptr;
        if (any) hipLaunchKernelGGL((gemm_x4_kernel<true, true, true>), grid, blk, 0, st, batch);
        else hipLaunchKernelGGL((gemm_x4_kernel<true, true, false>), grid, blk, 0, st, batch);
    } else if (mode == GEMM_NN) {
        hipLaunchKernelGGL((gemm_x4_kernel<true, false, false>), grid, blk, 0, st, batch);
    } else {
        hipLaunchKernelGGL((gemm_x4_kernel<false, false, false>), grid, blk, 0, st, batch);
    }
}

bool gemm_x4_enabled(long total_blocks) {
    // 1.09-1.16x gemm_x3_kernel as a kernel (forward 61.6 -> 52.9 us, weight gradient 61.7 -> 56.8 us,
    // profiles/r03_experiments.txt), but no faster as cgan_step / eva_step (1.49 / 0.717 vs 1.485 / 0.712 ms) and SLOWER
    // inside joint_step (1.84 vs 1.74 ms): its 8-wave workgroups hold 384 of a SIMD's 512 registers, so the chain's short
    // kernels of the other streams find fewer CUs to co-reside on than beside gemm_x3_kernel's 4 waves.  Where ONE stream
    // runs long launches it does pay: the config-5 inference sweep (1250 objects x 200 grasps, products of 250 000 rows)
    // 15.15 -> 13.95 ms.  Hence: launches of >= 1024 workgroups (four rounds of the chip; the train step's largest has
    // 240) take it, everything else gemm_x5_kernel (round 5: sweep 13.1 ms here against 13.9 there).  DGG_X4=1 / 0 forces it
    // on / off for every launch.
    static const int mode = [] {
        const char* e = getenv("DGG_X4");
        return !e ? -1 : (e[0] == '1' ? 1 : 0);
    }();
    return mode < 0 ? total_blocks >= 1024 : mode == 1;
}

}  // namespace dgg
