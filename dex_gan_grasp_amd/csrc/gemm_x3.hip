// fp32 GEMMs of the fc / ResBlock stacks on the bf16 matrix pipe ("bf16x3"): on gfx950 the f32 MFMA
// (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s while v_mfma_f32_32x32x16_bf16 runs 16x faster, so every fp32
// operand element x is split EXACTLY into three bf16 pieces
//     hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)         (round-to-nearest-even each)
// (|x - hi - mid - lo| <= 2^-27 |x|: the three 8-bit significands cover fp32's 24 bits) and a product a*b is
// accumulated as the six leading partial products
//     a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid
// — each exact in the MFMA's fp32 accumulator.  The dropped terms (mid*lo, lo*mid, lo*lo) are below 2^-25 |a b|,
// i.e. under the rounding error of the fp32 accumulation itself, so the result is an fp32-accurate GEMM at
// 6/16 of the f32-MFMA cost.  Same problem descriptors, split-K slabs and epilogues as gemm_f32.hip; 128 x 128
// workgroup tiles only (other tile shapes stay on gemm_f32_kernel).
//
// Structure (round 2).  What bounded the round-1 kernel was the vector-memory path of its producer waves: global ->
// VGPR -> split -> 3 bf16 planes -> LDS moved 48 KB of LDS stores and 32 KB of register loads per k-tile through waves
// that issue in order (4200 cycles per k-tile against 1536 cycles of MFMA).  Now one role per wave, 256 threads =
// 4 waves (2 x 2 of 64 x 64), one workgroup per CU:
//   * the fp32 operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no LDS store
//     instructions), 32 pieces of 1 KiB per k-tile (8 per wave), a ring of 4 stages (32.25 KB each) with a counted
//     vmcnt so that 2 k-tiles stay in flight across the one raw s_barrier per k-tile.  A k-contiguous operand tile is
//     [row][32 k] with 128-byte rows fetched as whole 128-byte lines; its 16-byte chunks are XOR-swizzled by
//     (row >> 1) & 7 on the SOURCE side (the DMA's LDS image is lane-linear) so that the ds_read_b128 fragment
//     fetches are bank-conflict-free.  A k-strided operand tile is [32 k][128] as it lies in memory.
//   * every wave reads the fp32 fragments of its 64 x 64 block from LDS, applies the folded input-BatchNorm affine,
//     splits them in registers (44 VALU per fragment) and feeds 24 MFMAs per 16-deep k-step.  Software-pipelined over
//     k-steps: the split of step s+1 is interleaved with the MFMAs of step s, across the barrier, in a pinned issue
//     order (sched_barrier) — left alone the compiler clusters the MFMAs and the matrix pipe idles during the VALU tail.
// Every element is split by the two waves that share its row / column block — 352 VALU per wave and k-tile beside
// 48 MFMAs (1536 matrix-pipe cycles).  Measured 2760 cycles per k-tile: the SIMD's issue port is the bound (see the
// comment at the kernel and DESIGN.md 3.1); gemm_p3.hip is the variant without the split.
#include <stdint.h>
#include <type_traits>
#include "dgg_internal.h"
#include "gemm_epilogue.h"
#include "gemm_p3_block.h"

namespace dgg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

#ifndef DGG_X3_ABLATE
#define DGG_X3_ABLATE 0  // diagnostics only: 1 no DMA, 2 no split (planes = raw bits), 3 no MFMA
#endif

#ifndef DGG_X3_STAMP
#define DGG_X3_STAMP 0  // diagnostics only: per-workgroup s_memtime / s_memrealtime stamps (never in the shipped build)
#endif
#if DGG_X3_STAMP
__device__ unsigned long long g_x3_stamp[2048][8];
#endif

constexpr int XK = 32;                      // k-tile
constexpr int XT = 128;                     // tile rows / columns
constexpr int XS = 4;                       // ring stages
constexpr int XOP = XT * XK * 4;            // bytes of one operand tile (16 KB)
constexpr int XAFF = 2 * XOP;               // offset of the affine vectors [scale 32 | shift 32] inside a stage
constexpr int XSTAGE = 2 * XOP + 1024;      // A tile, B tile, affine (one 256-byte copy per wave, copy 0 is read)
constexpr int XNP = XOP / 1024;             // DMA pieces per operand tile (16)
constexpr int XNW = 4;                      // waves per workgroup
constexpr int XNPW = 2 * XNP / XNW;         // pieces per wave and k-tile (8)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE): low half = a
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// 8 consecutive-k fp32 values of one operand row -> the (hi, mid, lo) bf16 fragments of a 32x32x16 MFMA.
// Stage by stage so that the four dependent chains overlap.
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    unsigned h[4], m[4], l[4];
    float r[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) h[p] = pack_bf16(x[2 * p], x[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        r[2 * p] = x[2 * p] - __builtin_bit_cast(float, h[p] << 16);
        r[2 * p + 1] = x[2 * p + 1] - __builtin_bit_cast(float, h[p] & 0xffff0000u);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) m[p] = pack_bf16(r[2 * p], r[2 * p + 1]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        r[2 * p] -= __builtin_bit_cast(float, m[p] << 16);
        r[2 * p + 1] -= __builtin_bit_cast(float, m[p] & 0xffff0000u);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) l[p] = pack_bf16(r[2 * p], r[2 * p + 1]);
    hi = __builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]});
    mid = __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]});
    lo = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
}

struct Frag {
    bf16x8 p[3];  // hi, mid, lo
};

// A grouped launch mixes problems with and without the A-operand affine (the [x_hat | previous block] segments of a
// ResBlock input): the affine variant serves both — problems without one stage these identity vectors.
__device__ float g_aff_identity[2][XK] = {
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, {0}};

#define X3_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int N>
__device__ __forceinline__ void wait_vm() {  // all but the N youngest vector-memory operations of this wave are done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wg_barrier() {
    // raw barrier: __syncthreads() would add vmcnt(0) and drain the DMA ring.  LDS reads of the consumers are
    // retired first (lgkmcnt), the "memory" clobber keeps the compiler from moving LDS accesses across.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// Loader: one 1-KiB DMA piece.  KC (operand stored [row][k]): piece j = tile rows 8j .. 8j+7, lane -> (row, 16-byte
// chunk); the chunk a lane fetches is its LDS slot XOR the row's swizzle.  !KC (stored [k][col]): piece j = tile k-rows
// 2j, 2j+1, lane -> (k-row, column quad).  Row / column indices are clamped into the matrix (such rows only feed
// accumulator rows / columns that are never stored); k beyond the operand's extent is clamped to its last chunk
// (the consumer zeroes the A operand's k >= kend, so the duplicate never contributes).
// ------------------------------------------------------------------------------------------------
template <bool KC>
struct PieceAddr {
    unsigned voff;  // loop-invariant per-lane byte offset from the tile's wave-uniform base pointer
};

template <bool KC>
__device__ __forceinline__ void piece_coords(int j, int lane, int ld, int row0, int nrows, unsigned& a, unsigned& b) {
    // KC: a = row * ld (elements), b = 4 * logical chunk;  !KC: a = tile k-row, b = clamped column
    if constexpr (KC) {
        const int row = 8 * j + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        a = (unsigned)gr * (unsigned)ld;
        b = 4u * (unsigned)lc;
    } else {
        const int krow = 2 * j + (lane >> 5);
        int gc = row0 + 4 * (lane & 31);
        const int last = (nrows - 1) & ~3;
        gc = gc < last ? gc : last;
        a = (unsigned)krow;
        b = (unsigned)gc;
    }
}

template <bool KC>
__device__ __forceinline__ PieceAddr<KC> piece_setup(int j, int lane, int ld, int row0, int nrows) {
    PieceAddr<KC> r;
    unsigned a, b;
    piece_coords<KC>(j, lane, ld, row0, nrows, a, b);
    r.voff = KC ? (a + b) * 4u : (a * (unsigned)ld + b) * 4u;
    return r;
}

// FAST: the k-tile lies inside the operand (k0 + 32 <= K): address = uniform base (advanced per k-tile with scalar
// adds) + the invariant lane offset, no further vector ALU.  !FAST (the one partial k-tile at the end of K): the
// coordinates are recomputed and k is clamped per lane.
template <bool KC, bool FAST>
__device__ __forceinline__ void piece_issue(const float* __restrict__ G, int ld, const PieceAddr<KC>& pa, int j, int lane,
                                            int row0, int nrows, int k0, int K, unsigned lds_dst) {
    const float* src;
    if constexpr (FAST) {
        const char* base = reinterpret_cast<const char*>(G + (KC ? (size_t)k0 : (size_t)k0 * (size_t)ld));  // uniform
        src = reinterpret_cast<const float*>(base + pa.voff);
    } else {
        unsigned a, b;
        piece_coords<KC>(j, lane, ld, row0, nrows, a, b);
        if constexpr (KC) {
            int k = k0 + (int)b;
            const int klast = (K - 1) & ~3;
            k = k < klast ? k : klast;
            src = G + (size_t)a + (size_t)k;
        } else {
            int k = k0 + (int)a;
            k = k < K ? k : K - 1;
            src = G + (size_t)k * (size_t)ld + (size_t)b;
        }
    }
    if constexpr (DGG_X3_ABLATE != 1) __builtin_amdgcn_global_load_lds(src, (lds_f32*)(size_t)lds_dst, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Consumer: raw fp32 fragment (8 consecutive k of one row / column) of k-step s from a staged operand tile.
//   KC : two ds_read_b128 at the swizzled chunks;  !KC: eight ds_read_b32, lanes = consecutive columns.
// `lane_off` carries the per-lane part, `blk` the 32-row block of the wave's piece.
// ------------------------------------------------------------------------------------------------
template <bool KC>
struct FragAddr {
    unsigned ch[2][2];  // KC: byte offsets (row + physical chunk) of (k-step, 16-byte half);  !KC: ch[s][0] = offset of k 16 s
};

template <bool KC>
__device__ __forceinline__ FragAddr<KC> frag_setup(int lane, int w0) {
    FragAddr<KC> f;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (KC) {
            const unsigned base = (unsigned)(w0 + r) * 128u;
            const int sw = (r >> 1) & 7;  // (row >> 1) & 7 with row = w0 + 32 blk + r, w0 a multiple of 64
#pragma unroll
            for (int e = 0; e < 2; ++e) f.ch[s][e] = base + (unsigned)(((4 * s + 2 * h + e) ^ sw) * 16);
        } else {
            f.ch[s][0] = (unsigned)(16 * s + 8 * h) * (XT * 4u) + (unsigned)(w0 + r) * 4u;
            f.ch[s][1] = 0;
        }
    }
    return f;
}

// AFF: the A operand carries a per-k affine (GemmProb::a_scale / a_shift; NT only), a compile-time variant so that
// the k loop stays free of branches: MFMAs and split VALU are interleaved only inside one basic block.
//
// 256 threads = 4 waves, one per SIMD, 2 x 2 over the 128 x 128 block (64 x 64 each); every wave stages a quarter of
// the DMA pieces, splits the fragments it needs and issues 48 MFMAs per k-tile.
// Measured (in-kernel s_memtime stamps, M=1024 N=768 K=4120 split-K 4, 2.08 GHz under load): 2760 cycles per k-tile —
// MFMA + DMA + LDS reads alone 1883, split + DMA + reads alone 1052: the instruction streams ADD.  A SIMD issues one
// vector instruction at a time at ~4-5 cycles per plain VALU op whatever the number of waves (a two-waves-per-SIMD
// variant with half the VALU per wave ran at 2900), so 380 VALU + 48 MFMA (8 issue cycles each) + 8 DMA pieces (~60
// each) per k-tile fill the issue port.  Next lever: operands arriving pre-split (producers emit the bf16 planes).
template <bool A_KC, bool B_KC, bool AFF, bool X1 = false>
__device__ __forceinline__ void gemm_x3_block(const GemmBatch& batch, const int bid, char* smem);

// One workgroup per (problem, split, tile) work item — or, launched with fewer workgroups than items ("side work",
// launch_x3), a stride loop over them: a GEMM that runs BESIDE a latency-critical chain on another stream then holds
// half the CUs for twice as long instead of all of them, and the chain's kernels always find free CUs.  The grid
// is a multiple of 8 so that an item's XCD (bid & 7) does not depend on which round of the loop takes it.
// PL (weight-gradient launches only): the launch carries a problem on pre-split planes (GemmProb::a_plane != 0, "NT over
// transposed planes": the BPS block's weight gradient) beside its fp32 riders; its work items run gemm_p3_block.
// X1 (DGG_GEMM_BF16, BASELINE config 3 "bf16 MFMA fc stacks"; never the default): plain bf16 — only the hi plane, ONE
// product per fragment pair, fp32 accumulate, fp32 master operands converted while they are staged.
template <bool A_KC, bool B_KC, bool AFF, bool PL = false, bool X1 = false>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const GemmBatch batch) {
    constexpr int kSmem = (PL && PS * PSTAGE > XS * XSTAGE) ? PS * PSTAGE : XS * XSTAGE;
    __shared__ __attribute__((aligned(16))) char smem[kSmem];
    for (int bid = (int)blockIdx.x; bid < batch.total_blocks; bid += (int)gridDim.x) {
        bool planes = false;
        if constexpr (PL) {
#pragma unroll
            for (int q = 0; q < kMaxProb; ++q)   // workgroup-uniform
                if (q < batch.nprob && bid >= batch.p[q].blk_begin) planes = batch.p[q].a_plane != 0;
        }
        if (planes) gemm_p3_block<true>(batch, bid, smem);
        else gemm_x3_block<A_KC, B_KC, AFF, X1>(batch, bid, smem);
        if (bid + (int)gridDim.x < batch.total_blocks) __syncthreads();  // the epilogue's LDS scratch is done with
    }
}

template <bool A_KC, bool B_KC, bool AFF, bool X1>
__device__ __forceinline__ void gemm_x3_block(const GemmBatch& batch, const int bid, char* smem) {

    // ---- work assignment: identical to gemm_f32_kernel (problem select, XCD-contiguous remap, split-K range)
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
    const int m0 = tm_i * XT, n0 = tn_i * XT;
    const int M = P.M, N = P.N;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nkt = (kend - kbeg + XK - 1) / XK;

    // The compiler does not model the DMA's lane-dependent LDS writes: the LDS offset is laundered so that reads of
    // the staged bytes cannot be reasoned about (a __shared__ array nobody stores to folds its loads to undef).
    unsigned lds0 = (unsigned)(size_t)(lds_char*)smem;
    asm volatile("" : "+s"(lds0));

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
#if DGG_X3_STAMP
    unsigned long long stamp[8];
    stamp[0] = __builtin_amdgcn_s_memtime();
    stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- staging: this wave's DMA pieces w, w + 4, ... of the A tile and of the B tile (+ a copy of the affine vectors)
    PieceAddr<A_KC> pa[XNP / XNW];
    PieceAddr<B_KC> pb[XNP / XNW];
#pragma unroll
    for (int q = 0; q < XNP / XNW; ++q) {
        pa[q] = piece_setup<A_KC>(wave + XNW * q, lane, P.lda, m0, M);
        pb[q] = piece_setup<B_KC>(wave + XNW * q, lane, P.ldb, n0, N);
    }
    constexpr int PER = XNPW + (AFF ? 1 : 0);  // DMA instructions per wave and k-tile
    auto issue_aff = [&](int t) {  // lanes 0-31: scale[k0 + lane], lanes 32-63: shift[k0 + lane - 32]; every wave
        if constexpr (AFF) {       // stages a copy so that all waves count the same number of DMA operations per tile
            const int k0 = kbeg + t * XK;
            const unsigned st = lds0 + (unsigned)(t & (XS - 1)) * XSTAGE;
            int k = k0 + (lane & 31);
            k = k < P.K ? k : P.K - 1;
            const float* src = P.a_scale ? (lane < 32 ? P.a_scale : P.a_shift) + k
                                         : &g_aff_identity[lane >> 5][lane & 31];  // a problem without affine
            if constexpr (DGG_X3_ABLATE != 1)
                __builtin_amdgcn_global_load_lds(src, (lds_f32*)(size_t)(st + XAFF + (unsigned)wave * 256u), 4, 0, 0);
        }
    };
    auto issue_piece = [&](int t, int q, auto fast_tag) {  // piece q (0 .. PER-1) of this wave for k-tile t
        constexpr bool FAST = decltype(fast_tag)::value;
        const int k0 = kbeg + t * XK;
        const unsigned st = lds0 + (unsigned)(t & (XS - 1)) * XSTAGE;
        if (q < XNP / XNW)
            piece_issue<A_KC, FAST>(P.A, P.lda, pa[q], wave + XNW * q, lane, m0, M, k0, P.K,
                                    st + (unsigned)(wave + XNW * q) * 1024u);
        else if (q < 2 * (XNP / XNW))
            piece_issue<B_KC, FAST>(P.B, P.ldb, pb[q - XNP / XNW], wave + XNW * (q - XNP / XNW), lane, n0, N, k0, P.K,
                                    st + XOP + (unsigned)(wave + XNW * (q - XNP / XNW)) * 1024u);
        else issue_aff(t);
    };
    auto issue = [&](int t) {  // prologue form
        if (kbeg + (t + 1) * XK <= P.K) {
#pragma unroll
            for (int q = 0; q < PER; ++q) issue_piece(t, q, std::true_type{});
        } else {
#pragma unroll
            for (int q = 0; q < PER; ++q) issue_piece(t, q, std::false_type{});
        }
    };

    // ---- compute: 2 x 2 waves over the tile, 64 x 64 each
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int h = lane >> 5;
    const FragAddr<A_KC> fa = frag_setup<A_KC>(lane, wm0);
    const FragAddr<B_KC> fb = frag_setup<B_KC>(lane, wn0);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Software pipeline over 16-deep k-steps i = 2 t + s (t: k-tile, s: half), three k-steps in flight:
    //   planes P(i)   — consumed by the 24 MFMAs of half i;
    //   raw(i+1)      — fp32 fragments in registers, split into planes P(i+1) between those MFMAs;
    //   raw(i+2)      — LDS reads issued during half i (their latency never shows).
    // Fragments 0, 1 = A blocks, 2, 3 = B blocks.
    float raw[2][4][8];   // [parity of the k-step][fragment][8 k]
    float aq[2][2][8];    // [parity][scale / shift][8 k]
    Frag pl[2][4];        // [parity][fragment] planes

    // one LDS read unit of k-step (t, s): unit u = fragment u / 2, elements 4 (u % 2) .. + 3
    auto read_unit = [&](int t, int s, int u, float (&dst)[4][8]) {
        const unsigned st = lds0 + (unsigned)(t & (XS - 1)) * XSTAGE;
        const int f = u >> 1, e = u & 1;
        if (f < 2) {
            if constexpr (A_KC) {
                const f32x4 v = *(lds_f32x4*)(size_t)(st + fa.ch[s][e] + (unsigned)f * (32u * 128u));
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[f][4 * e + j] = v[j];
            } else {
                const unsigned p = st + fa.ch[s][0] + (unsigned)f * 128u;
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[f][4 * e + j] = *(lds_f32*)(size_t)(p + (unsigned)(4 * e + j) * (XT * 4u));
            }
        } else {
            const int g = f - 2;
            if constexpr (B_KC) {
                const f32x4 v = *(lds_f32x4*)(size_t)(st + XOP + fb.ch[s][e] + (unsigned)g * (32u * 128u));
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[f][4 * e + j] = v[j];
            } else {
                const unsigned p = st + XOP + fb.ch[s][0] + (unsigned)g * 128u;
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[f][4 * e + j] = *(lds_f32*)(size_t)(p + (unsigned)(4 * e + j) * (XT * 4u));
            }
        }
    };
    auto read_aff = [&](int t, int s, int e, float (&dst)[2][8]) {  // e: 0 scale, 1 shift
        const unsigned p = lds0 + (unsigned)(t & (XS - 1)) * XSTAGE + XAFF + (unsigned)e * 128u + (unsigned)(16 * s + 8 * h) * 4u;
        const f32x4 v0 = *(lds_f32x4*)(size_t)p, v1 = *(lds_f32x4*)(size_t)(p + 16u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dst[e][j] = v0[j];
            dst[e][4 + j] = v1[j];
        }
    };
    // Split of a fragment in six slices of ~8 VALU, one per MFMA of the half (44 VALU per fragment; + 8 for the affine,
    // + 8 for the k-tail mask on A fragments).  x: raw values, out: the fragment's planes.
    unsigned sh_[4], sm_[4];
    float sr_[8], sx_[8];
    auto slice = [&](int f, int sl, float (&x)[8], const float (&q)[2][8], Frag& out, int kvalid, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if constexpr (DGG_X3_ABLATE == 2) {
            if (sl == 5) out.p[0] = out.p[1] = out.p[2] = __builtin_bit_cast(bf16x8, (f32x4){x[0], x[1], x[2], x[3]});
            return;
        }
        if constexpr (X1) {   // plain bf16: affine / tail mask, one conversion, nothing else
            if (sl != 0) return;
            if (f < 2) {
                if constexpr (AFF) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], q[0][j], q[1][j]);
                }
                if constexpr (TAIL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = j < kvalid ? x[j] : 0.f;
                }
            }
            unsigned hh[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) hh[p2] = pack_bf16(x[2 * p2], x[2 * p2 + 1]);
            out.p[0] = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
            return;
        }
        if (sl == 0) {
            if (f < 2) {
                if constexpr (AFF) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], q[0][j], q[1][j]);
                }
                if constexpr (TAIL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = j < kvalid ? x[j] : 0.f;
                }
            }
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) sh_[p2] = pack_bf16(x[2 * p2], x[2 * p2 + 1]);
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                sr_[2 * p2] = __builtin_bit_cast(float, sh_[p2] << 16);
                sr_[2 * p2 + 1] = __builtin_bit_cast(float, sh_[p2] & 0xffff0000u);
            }
        } else if (sl == 1) {
#pragma unroll
            for (int p2 = 2; p2 < 4; ++p2) {
                sr_[2 * p2] = __builtin_bit_cast(float, sh_[p2] << 16);
                sr_[2 * p2 + 1] = __builtin_bit_cast(float, sh_[p2] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sr_[j] = x[j] - sr_[j];
        } else if (sl == 2) {
#pragma unroll
            for (int j = 4; j < 8; ++j) sr_[j] = x[j] - sr_[j];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) sm_[p2] = pack_bf16(sr_[2 * p2], sr_[2 * p2 + 1]);
        } else if (sl == 3) {
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                sx_[2 * p2] = __builtin_bit_cast(float, sm_[p2] << 16);
                sx_[2 * p2 + 1] = __builtin_bit_cast(float, sm_[p2] & 0xffff0000u);
            }
        } else if (sl == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sr_[j] -= sx_[j];
        } else {
            unsigned l[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) l[p2] = pack_bf16(sr_[2 * p2], sr_[2 * p2 + 1]);
            out.p[0] = __builtin_bit_cast(bf16x8, (u32x4){sh_[0], sh_[1], sh_[2], sh_[3]});
            out.p[1] = __builtin_bit_cast(bf16x8, (u32x4){sm_[0], sm_[1], sm_[2], sm_[3]});
            out.p[2] = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
        }
    };
    // MFMA g (0..23) of a half: block pair (g / 12, (g / 6) % 2), product g % 6
    auto mfma_one = [&](int g, const Frag (&P4)[4]) {
        if constexpr (DGG_X3_ABLATE == 3) return;
        const int ta = g / 12, tb = (g / 6) & 1, pr = g % 6;
        if constexpr (X1) {
            if (pr != 5) return;   // hi * hi only
        }
        const Frag& a = P4[ta];
        const Frag& b = P4[2 + tb];
        // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi  (small terms first)
        const int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
        const int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
        X3_MFMA(a.p[ia], b.p[ib], acc[ta][tb]);
    };

    using No = std::false_type;
    using Yes = std::true_type;
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    using D2 = std::integral_constant<int, 2>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // Ring protocol (4 stages, one barrier per k-tile).  Barrier C(t), at the head of tile t's first half, = "tile t
    // fully read (its fragments were fetched during tile t-1), tile t+1 landed": every wave first waits (counted
    // vmcnt) for its own pieces of tile t+1 — tiles t+2 and t+3 stay in flight across the barrier.  Right after it the
    // stage of tile t is free: tile t+4 goes there.
    // One half = ONE basic block with a pinned issue order (sched_barrier): 24 x {1 MFMA, one split slice (~8 VALU),
    // [1 LDS read unit], [1 DMA piece]}.  Left to itself the compiler clusters the MFMAs at the head of the block and
    // the matrix pipe idles during the VALU tail.
    auto half = [&](int t, int s, auto par_tag, auto rd_tag, auto sp_tag, auto tail_tag, auto dma_tag) {
        constexpr int PAR = decltype(par_tag)::value;       // parity of k-step i = 2 t + s
        constexpr bool RD = decltype(rd_tag)::value;        // k-step i + 2 exists: fetch its fragments
        constexpr bool SP = decltype(sp_tag)::value;        // k-step i + 1 exists: split it
        constexpr int DMA = decltype(dma_tag)::value;       // (s == 0 only) stage tile t + 4: 1 fast, 2 clamped (partial tile)
        const int t1 = s == 0 ? t : t + 1, s1 = s ^ 1;      // k-step i + 1
        const int t2 = t + 1, s2 = s;                       // k-step i + 2
        const int kvalid = kend - (kbeg + t1 * XK) - 16 * s1 - 8 * h;
        constexpr int NU = 8;                                // read units per k-step
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            mfma_one(g, pl[PAR]);
            if constexpr (SP) slice(g / 6, g % 6, raw[PAR ^ 1][g / 6], aq[PAR ^ 1], pl[PAR ^ 1][g / 6], kvalid, tail_tag);
            if constexpr (RD) {
                if (g < NU) read_unit(t2, s2, g, raw[PAR]);
                if constexpr (AFF) {
                    if (g == NU || g == NU + 1) read_aff(t2, s2, g - NU, aq[PAR]);
                }
            }
            if constexpr (DMA != 0) {
                if (g >= 10 && g < 10 + PER) issue_piece(t + XS, g - 10, std::integral_constant<bool, DMA == 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool ktail = (kend - kbeg) % XK != 0;  // the last tile is partial (workgroup-uniform)
    for (int t = 0; t < XS; ++t)
        if (t < nkt) issue(t);
    // tiles 0 and 1 landed (the prologue reads k-steps 0, 1 of tile 0 and the first half reads into tile 1)
    if (nkt >= XS) wait_vm<(XS - 2) * PER>();
    else wait_vm<0>();
    wg_barrier();
    // prologue: planes of k-step 0, raw fragments of k-step 1
#pragma unroll
    for (int u = 0; u < 8; ++u) read_unit(0, 0, u, raw[1]);
    if constexpr (AFF) {
        read_aff(0, 0, 0, aq[1]);
        read_aff(0, 0, 1, aq[1]);
    }
    {
        const int kvalid = kend - kbeg - 8 * h;
#pragma unroll
        for (int g = 0; g < 24; ++g) {
            if (nkt == 1 && ktail) slice(g / 6, g % 6, raw[1][g / 6], aq[1], pl[0][g / 6], kvalid, Yes{});
            else slice(g / 6, g % 6, raw[1][g / 6], aq[1], pl[0][g / 6], kvalid, No{});
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) read_unit(0, 1, u, raw[1]);
    if constexpr (AFF) {
        read_aff(0, 1, 0, aq[1]);
        read_aff(0, 1, 1, aq[1]);
    }
    wg_barrier();  // every wave's prologue reads of tile 0 are retired before any wave restages its buffer
#if DGG_X3_STAMP
    stamp[2] = __builtin_amdgcn_s_memtime();
#endif
    // k-step i = 2 t + s has parity s.  Steady state while tile t + 4 exists and lies fully inside K (=> every k-step
    // used below exists and no tile touched is the last one); then at most one iteration that stages the partial tile.
    const int nfast = min(nkt, (P.K - kbeg) / XK);  // tiles [0, nfast) need no k clamp
    int t = 0;
    for (; t + XS < nfast; ++t) {
        if (t > 0) {
            wait_vm<2 * PER>();
            wg_barrier();
        }
        half(t, 0, P0{}, Yes{}, Yes{}, No{}, D1{});
        half(t, 1, P1{}, Yes{}, Yes{}, No{}, D0{});
    }
    if (t + XS < nkt) {
        if (t > 0) {
            wait_vm<2 * PER>();
            wg_barrier();
        }
        half(t, 0, P0{}, Yes{}, Yes{}, No{}, D2{});
        half(t, 1, P1{}, Yes{}, Yes{}, No{}, D0{});
        ++t;
    }
    // drain: no more staging; the k-tail mask applies to the splits of the last tile's two k-steps
    for (; t < nkt; ++t) {
        if (t > 0) {
            wait_vm<0>();
            wg_barrier();
        }
        const bool last = t + 1 == nkt, before_last = t + 2 == nkt;
        // half (t, 0): splits k-step (t, 1), fetches (t+1, 0)
        if (last) {
            if (ktail) half(t, 0, P0{}, No{}, Yes{}, Yes{}, D0{});
            else half(t, 0, P0{}, No{}, Yes{}, No{}, D0{});
        } else {
            half(t, 0, P0{}, Yes{}, Yes{}, No{}, D0{});
        }
        // half (t, 1): splits k-step (t+1, 0), fetches (t+1, 1)
        if (last) half(t, 1, P1{}, No{}, No{}, No{}, D0{});
        else if (before_last && ktail) half(t, 1, P1{}, Yes{}, Yes{}, Yes{}, D0{});
        else half(t, 1, P1{}, Yes{}, Yes{}, No{}, D0{});
    }
#if DGG_X3_STAMP
    stamp[3] = __builtin_amdgcn_s_memtime();
#endif
    wg_barrier();  // the ring is idle (every tile was waited for and read): the epilogue reuses its LDS
    gemm_epilogue<XT, XT, 2, 2, !A_KC && !B_KC>(P, acc, reinterpret_cast<float*>(smem), split, tm_i, m0, n0, (int)threadIdx.x);
#if DGG_X3_STAMP
    stamp[4] = __builtin_amdgcn_s_memtime();
    stamp[5] = __builtin_amdgcn_s_memrealtime();
    stamp[6] = nkt;
    stamp[7] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    if (threadIdx.x == 0 && bid < 2048)
        for (int i = 0; i < 8; ++i) g_x3_stamp[bid][i] = stamp[i];
#endif
}

template <bool A_KC, bool B_KC, bool AFF, bool PL = false, bool X1 = false>
static void launch_x3(const GemmBatch& b, hipStream_t st) {
    int grid = b.total_blocks;
    // side work (dgg_set_side_work): the forward products walk two work items per workgroup — same items, same sums,
    // half the CUs for twice as long.  (Halving the weight gradient's grid as well costs 30-70 us per step: measured.)
    // diagnostics: DGG_SIDE_DIV items per workgroup (measured us per step: 1 -> 1814 / 1846, 2 -> 1757 / 1748, 3 -> 1863 /
    // 1850, 4 -> 1923 / 1893, 6 -> 1928 / 1924)
    static const int kdiv = [] { const char* e = getenv("DGG_SIDE_DIV"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
    if (gemm_side_work() && A_KC && B_KC && grid > 64) grid = ((grid + kdiv - 1) / kdiv + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, AFF, PL, X1>), dim3(grid), dim3(256), 0, st, b);
}

// 128 x 128 tiles only.  Operand bases and pitches need 4-byte alignment only (the DMA fetches 16-byte chunks from any
// dword address; 16-byte aligned operands — every activation, the weights of D and E — are ~10 % faster).  A k-tail
// chunk may extend up to 12 bytes past an operand row's last element: operands live inside larger buffers (flat
// parameter buffers, workspaces), the bytes are fetched and never used.
void gemm_x3_dispatch(int mode, const GemmBatch& batch, hipStream_t st) {
    if (gemm_algo() == 3) {   // DGG_GEMM_BF16: the same kernel, hi plane only
        if (mode == GEMM_NT) {
            bool any = false;
            for (int i = 0; i < batch.nprob; ++i) any = any || batch.p[i].a_scale != nullptr;
            if (any) launch_x3<true, true, true, false, true>(batch, st);
            else launch_x3<true, true, false, false, true>(batch, st);
        } else if (mode == GEMM_NN) {
            launch_x3<true, false, false, false, true>(batch, st);
        } else {
            launch_x3<false, false, false, false, true>(batch, st);
        }
        return;
    }
    if (mode == GEMM_NT) {
        bool any = false;
        for (int i = 0; i < batch.nprob; ++i) any = any || batch.p[i].a_scale != nullptr;
        if (any) launch_x3<true, true, true>(batch, st);
        else launch_x3<true, true, false>(batch, st);
    } else if (mode == GEMM_NN) {
        launch_x3<true, false, false>(batch, st);
    } else {
        bool planes = false;
        for (int i = 0; i < batch.nprob; ++i) planes = planes || batch.p[i].a_plane != 0;
        if (planes) launch_x3<false, false, false, true>(batch, st);
        else launch_x3<false, false, false>(batch, st);
    }
}

#if DGG_X3_STAMP
extern "C" int dgg_x3_debug_stamps(unsigned long long* host_out, int nblocks) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_x3_stamp), (size_t)nblocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

int gemm_x3_occupancy(int mode) {
    int n = -1;
    const void* f = (const void*)gemm_x3_kernel<true, true, false>;
    if (mode == GEMM_NN) f = (const void*)gemm_x3_kernel<true, false, false>;
    if (mode == GEMM_TN) f = (const void*)gemm_x3_kernel<false, false, false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, 0) != hipSuccess) return -1;
    return n;
}

}  // namespace dgg
