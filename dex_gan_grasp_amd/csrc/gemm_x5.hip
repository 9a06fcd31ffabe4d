// bf16x3 GEMM, round 5: ONE wave owns the whole 128 x 128 accumulator block.
//
// Arithmetic ("bf16x3"): on gfx950 the f32 MFMA (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s while
// v_mfma_f32_32x32x16_bf16 runs 16x faster, so every fp32 operand element x is split EXACTLY into three bf16 pieces
//     hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)         (round-to-nearest-even each)
// (|x - hi - mid - lo| <= 2^-27 |x|: the three 8-bit significands cover fp32's 24 bits) and a product a*b is accumulated
// as the six leading partial products  a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid  — each
// exact in the MFMA's fp32 accumulator.  The dropped terms are below 2^-25 |a b|, under the rounding error of the fp32
// accumulation itself: an fp32-accurate GEMM at 6/16 of the f32-MFMA cost.  Problem descriptors, work assignment and
// epilogues are those of gemm_f32.hip.
// Who computes what.  The kernel of rounds 2-4 (gemm_x3_kernel, deleted) gave each of its four waves a 64 x 64 quarter of the output tile: per
// 16-deep k-step a wave splits 4 fragments for 24 MFMAs — 7.3 vector instructions per MFMA, where an MFMA's 32 cycles
// hide about five (MI355X_MICROARCH.md, cycle constants) — and every element is split by two waves; measured 57
// cycles per MFMA.  Here the four waves of a workgroup split the K range instead: wave w takes the k-steps w, w + 4,
// w + 8, ... of the tile and accumulates the FULL 128 x 128 block (16 MFMA tiles, 256 accumulator registers) over them:
// 8 fragments per 96 MFMAs = 3.7 vector instructions per MFMA, every element split once, every staged byte read from
// LDS once.  The four partial blocks meet in LDS at the end (fixed order w0 + w1 + w2 + w3: deterministic), each wave
// keeps one 64 x 64 quadrant and runs the shared epilogue on it (gemm_epilogue.h, unchanged).
// tools/exp/x5_probe.hip is the timing probe of the two instruction mixes (42 against 66 cycles per MFMA).
//
// No wave shares staged data with another, so nothing in the k loop needs a workgroup barrier: each wave has a
// PRIVATE two-stage LDS ring (2 x 16 KB + the affine vectors), fills it by LDS-DMA (global_load_lds_dwordx4, 17 pieces
// per k-step) and waits for it with its own vmcnt.
//
// Software pipeline of one wave (local k-steps i = 0, 1, ...; stage = i & 1).  The planes of k-step i feed 96 MFMAs; the
// 8 fragment splits issued between them produce what the NEXT MFMAs need.  Two planes sets would not fit beside 256
// accumulators, so consecutive k-steps alternate the loop order of the 4 x 4 MFMA tiles:
//   even steps ("AO"): A block outermost — the B planes persist, the A planes rotate: A[a] is dead after its 24 MFMAs;
//   odd steps  ("BO"): B block outermost — the A planes persist, the B planes rotate.
// An AO step splits, in windows of 12 MFMAs: its own A[1], A[2], A[3] (each ready before its group of MFMAs starts),
// then the next step's four A fragments (persistent there) and B[0]; a BO step mirrors that.  At most 10 fragments'
// planes (120 registers) are live.  Raw fp32 fragments travel LDS -> registers one window ahead (two 8-register
// buffers); the stage a step reads for itself is free after its third window, and the DMA of k-step i + 2 goes there
// during MFMAs 24 .. 88 of step i.
//
// K tail: a partial last k-step is moved BACK to end exactly at the split's last k (k0 = kend - 16) and the A elements
// the previous step already covered are zeroed while they are staged — every DMA address lies inside the operand's k
// range, no clamped variants of the loop.
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "dgg_internal.h"
#include "gemm_epilogue.h"

namespace dgg {


namespace x5 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));   // a 16-byte global access at any dword address (row pitches are the checkpoint's: 4101, 4613, ...)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr int YT = 128;                 // tile rows / columns
constexpr int YK = 16;                  // k-step
constexpr int YOP = YT * YK * 4;        // bytes of one operand's k-step (8 KB)
constexpr int YAFF = 2 * YOP;           // affine vectors inside a stage: [scale 16 | shift 16 | unused]
constexpr int YSTAGE = 2 * YOP + 256;
constexpr int YWAVE = 2 * YSTAGE;       // a wave's private ring
constexpr int YSMEM = 4 * YWAVE;        // 133120 B; the reduction at the end reuses it (128 KB + 2 KB of epilogue scratch)
constexpr int YNP = YOP / 1024;         // DMA pieces per operand and k-step (8)
constexpr int YMINK = 64;               // shorter problems run on gemm_f32_kernel

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, 0>) ... : unrolled by construction
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE): low half = a
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

struct Frag {
    bf16x8 p[3];  // hi, mid, lo
};

__device__ float g_aff_identity[2][YK] = {{1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, {0}};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Which fragment a window of 12 MFMAs splits.  op: 0 = A, 1 = B; par: parity of the k-step the planes belong to (= the
// stage its raw values lie in); blk: 32-row block.
struct FD {
    int op, par, blk;
};
// TY: 0 = AO step (parity 0), 1 = BO step (parity 1).  Windows 0..7, and 8 = window 0 of the step that follows.
__host__ __device__ constexpr FD frag_of(int TY, int n) {
    const int own = TY;        // rotating operand of this step: A in AO, B in BO
    const int cur = TY, nxt = TY ^ 1;
    if (n < 3) return FD{own, cur, n + 1};
    if (n < 7) return FD{own, nxt, n - 3};    // persistent operand of the next step
    if (n == 7) return FD{own ^ 1, nxt, 0};   // first block of the next step's rotating operand
    return FD{own ^ 1, nxt, 1};               // n == 8: the next step's window 0
}

__host__ __device__ constexpr int fcode(FD f) { return f.op * 16 + f.par * 8 + f.blk; }   // C++17: no class-type template arguments
__host__ __device__ constexpr FD fdec(int c) { return FD{c >> 4, (c >> 3) & 1, c & 7}; }
template <int C>
using FC = std::integral_constant<int, C>;

// Epilogue of a tile that lies inside the matrix, with 16-byte stores and loads.  The MFMA layout hands a lane ONE column
// and 4 consecutive rows per register quad — row-major C then takes 64 one-dword store instructions per wave, and four
// waves storing at once are bound by store ISSUE (3.4-4.3k cycles; MI355X_MICROARCH.md: a store tail of narrow stores).  The
// wave turns its 64 x 64 quadrant through its own (now idle) 16 KB of LDS: 64 ds_write_b32 in the MFMA layout (lanes = 32
// consecutive columns: conflict-free), 16 ds_read_b128 row-major (lane = row (lane >> 4) + 4 i, columns 4 (lane & 15) ..:
// with a 256-byte row pitch the 16 lanes of a read group cover all 64 banks).  Then 16 instructions of 4 rows x 256
// contiguous bytes each.  Same values as gemm_epilogue() for the plain / slab / accumulate forms; the BatchNorm-aware
// weight-gradient form (GemmProb::ep_gamma) adds up its per-column partials in another — fixed — order.
template <bool WGRAD>
__device__ __forceinline__ void x5_epilogue_rows(const GemmProb& P, const f32x16 (&res)[2][2], unsigned lds_wave, float* red,
                                                 int split, int tm_i, int m0, int n0, int wave, int lane, int tid) {
    const int row0 = m0 + 64 * (wave >> 1) + (lane >> 4);          // this lane's rows: row0 + 4 i
    const int col0 = n0 + 64 * (wave & 1) + 4 * (lane & 15);        // ... and its 4 columns
    // BatchNorm-aware weight gradient: the per-column affine and the column sums of dY per output row are requested first and
    // land while the quadrant turns through LDS; they are waited for all at once (one-dword and 16-byte loads in flight
    // together: never a counted vmcnt over such a mix, tools/vmcnt_audit.py).  The wave's 64 rows lie on one side of
    // ep_split (a multiple of 64: the caller checks).
    bool bn_form = false;
    f32x4 ga = {0.f, 0.f, 0.f, 0.f}, be = {0.f, 0.f, 0.f, 0.f};
    float csn[16];
    if constexpr (WGRAD) {
        if (P.ep_gamma) {
            bn_form = true;
            ga = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const char*>(P.ep_gamma) + (unsigned)col0 * 4u);
            be = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const char*>(P.ep_beta) + (unsigned)col0 * 4u);
            float cst[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) csn[i] = cst[i] = 0.f;
            const int rbase = m0 + 64 * (wave >> 1);
            bool two = false;
            if (rbase + 64 <= P.ep_split) {
                if (P.ep_cs0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) csn[i] = P.ep_cs0[row0 + 4 * i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) csn[i] = P.ep_cs1[row0 + 4 * i - P.ep_split];
                if (P.ep_cs1b) {
                    two = true;
#pragma unroll
                    for (int i = 0; i < 16; ++i) cst[i] = P.ep_cs1b[row0 + 4 * i - P.ep_split];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (two) {
#pragma unroll
                for (int i = 0; i < 16; ++i) csn[i] += cst[i];
            }
        }
    }
    {
        const unsigned wb = lds_wave + (unsigned)((4 * (lane >> 5)) * 64 + (lane & 31)) * 4u;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *(lds_f32*)(size_t)(wb + (unsigned)(((32 * ta + (r & 3) + 8 * (r >> 2)) * 64 + 32 * tb) * 4)) = res[ta][tb][r];
    }
    f32x4 v[16];
    {
        const unsigned rb_ = lds_wave + (unsigned)((lane >> 4) * 64 + 4 * (lane & 15)) * 4u;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *(lds_f32x4*)(size_t)(rb_ + (unsigned)(4 * i * 64 * 4));
    }
    float* Cb = P.C + (P.splits > 0 ? (size_t)split * (size_t)P.slab_stride : (size_t)0);
    const bool accum = (P.splits == 0) && P.accumulate;
    char* const Cbytes = reinterpret_cast<char*>(Cb);
    auto c_at = [&](int i) { return reinterpret_cast<f32x4_u*>(Cbytes + ((unsigned)(row0 + 4 * i) * (unsigned)P.ldc + (unsigned)col0) * 4u); };
    if constexpr (WGRAD) {
        if (bn_form) {   // dW[n][k] = gamma[k] T + beta[k] cs[n]; per-column partials of sum_n W T and sum_n W cs (GemmProb)
            const char* const Wbytes = reinterpret_cast<const char*>(P.ep_w);
            auto w_at = [&](int i) {
                return reinterpret_cast<const f32x4_u*>(Wbytes + ((unsigned)(row0 + 4 * i) * (unsigned)P.ep_ldw + (unsigned)col0) * 4u);
            };
            f32x4 pg = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h8 = 0; h8 < 16; h8 += 8) {   // 8 rows per batch of loads
                f32x4 w[8], old[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = *w_at(h8 + i);
                if (accum) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) old[i] = *c_at(h8 + i);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no load of the epilogue outlives it (see gemm_epilogue.h)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 t = v[h8 + i];
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pg[j] = __builtin_fmaf(w[i][j], t[j], pg[j]);
                        pb[j] = __builtin_fmaf(w[i][j], csn[h8 + i], pb[j]);
                        o[j] = __builtin_fmaf(ga[j], t[j], be[j] * csn[h8 + i]);
                        if (accum) o[j] += old[i][j];
                    }
                    *c_at(h8 + i) = o;
                }
            }
            // the wave's 64 rows: lanes l, l + 16, l + 32, l + 48 hold the same columns
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pg[j] += lane_xor(pg[j], lane, 16);
                pg[j] += lane_xor(pg[j], lane, 32);
                pb[j] += lane_xor(pb[j], lane, 16);
                pb[j] += lane_xor(pb[j], lane, 32);
            }
            if (lane < 16) {
                const int c = 64 * (wave & 1) + 4 * lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    red[((wave >> 1) * 2 + 0) * YT + c + j] = pg[j];
                    red[((wave >> 1) * 2 + 1) * YT + c + j] = pb[j];
                }
            }
            __syncthreads();
            {
                const int q = tid >> 7, cidx = tid & 127;   // 256 threads: 2 quantities x 128 columns; the two row halves in order
                const float t = red[q * YT + cidx] + red[(2 + q) * YT + cidx];
                P.ep_part[((size_t)tm_i * 2 + q) * (P.ep_part_ld ? P.ep_part_ld : P.N) + n0 + cidx] = t;
            }
            return;
        }
    }
    if (accum) {
        f32x4 old[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) old[i] = *c_at(i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) old[i] = v[i] + old[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) *c_at(i) = old[i];
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *c_at(i) = v[i];
}

// X1 (DGG_GEMM_BF16, BASELINE config 3 "bf16 MFMA fc stacks"; never the default): plain bf16 — only the hi plane, ONE
// product per fragment pair, fp32 accumulate, fp32 master operands rounded while they are staged.
template <bool A_KC, bool B_KC, bool AFF, bool X1>
__device__ __forceinline__ void gemm_x5_block(const GemmBatch& batch, const int bid, char* smem) {
    // ---- work assignment (problem select, XCD-contiguous remap: each XCD gets a contiguous run of (split, tile) items; split-K range)
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kMaxProb; ++q)
        if (q < batch.nprob && bid >= batch.p[q].blk_begin) pi = q;
    const GemmProb& P = batch.p[pi];   // left in the kernel-argument segment: the epilogue's fields are fetched when it runs
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
    const int M = P.M, N = P.N;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nks = (kend - kbeg + YK - 1) / YK;   // k-steps of this work item (kend - kbeg >= 16: gemm_x5_usable)

    unsigned lds0 = (unsigned)(size_t)(lds_char*)smem;
    // The compiler does not model the DMA's lane-dependent LDS writes: the LDS offset is laundered so that reads of the staged
    // bytes cannot be reasoned about (a __shared__ array nobody stores to folds its loads to undef).
    asm volatile("" : "+s"(lds0));

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, r32 = lane & 31;
    const unsigned wl = lds0 + (unsigned)wave * YWAVE;
    const int nw = nks > wave ? (nks - wave + 3) >> 2 : 0;    // this wave's k-steps: wave, wave + 4, ...
    auto k0_of = [&](int i) {   // first k of local step i (the partial last step of the work item ends at kend)
        const int k = kbeg + YK * (wave + 4 * i);
        return k + YK <= kend ? k : kend - YK;
    };

    // ---- DMA addressing.  KC operand ([row][k]): piece j = rows 16 j .. 16 j + 15 of the k-step, 64 B each; lane -> (row,
    // 16-byte chunk); the LDS image is lane-linear, so the swizzle — physical chunk = logical chunk ^ ((row >> 2) & 3), which
    // makes the ds_read_b128 fragment fetches bank-conflict-free — is applied on the source side.  !KC ([k][col]): piece
    // j = k-rows 2 j, 2 j + 1, lane -> (k-row, column quad).  Rows / columns are clamped into the matrix.
    unsigned va[A_KC ? YNP : 1], vb[B_KC ? YNP : 1];
    if constexpr (A_KC) {
#pragma unroll
        for (int j = 0; j < YNP; ++j) {
            int gr = m0 + 16 * j + (lane >> 2);
            gr = gr < M ? gr : M - 1;
            va[j] = ((unsigned)gr * (unsigned)P.lda + 4u * (unsigned)((lane & 3) ^ ((lane >> 4) & 3))) * 4u;
        }
    } else {
        int gc = m0 + 4 * r32;
        const int last = (M - 1) & ~3;
        gc = gc < last ? gc : last;
        va[0] = ((unsigned)h * (unsigned)P.lda + (unsigned)gc) * 4u;
    }
    if constexpr (B_KC) {
#pragma unroll
        for (int j = 0; j < YNP; ++j) {
            int gr = n0 + 16 * j + (lane >> 2);
            gr = gr < N ? gr : N - 1;
            vb[j] = ((unsigned)gr * (unsigned)P.ldb + 4u * (unsigned)((lane & 3) ^ ((lane >> 4) & 3))) * 4u;
        }
    } else {
        int gc = n0 + 4 * r32;
        const int last = (N - 1) & ~3;
        gc = gc < last ? gc : last;
        vb[0] = ((unsigned)h * (unsigned)P.ldb + (unsigned)gc) * 4u;
    }
    // the affine's source per lane (a problem without one stages the identity, which does not move with k)
    const float* aff_src = nullptr;
    int aff_step = 0;
    if constexpr (AFF) {
        const int l = lane & 31;
        const bool has = P.a_scale != nullptr;
        aff_src = has ? (l < 16 ? P.a_scale : P.a_shift) + (l & 15) : &g_aff_identity[l >> 4][l & 15];
        aff_step = has ? 1 : 0;
    }
    // piece t (0..16: A pieces, B pieces, affine) of the k-step starting at k0, into stage st.  rowA / rowB: first byte of
    // k-row k0 of a k-strided operand (dma_rows), advanced here by two rows per piece.
    const size_t strideA2 = (size_t)P.lda * 8u, strideB2 = (size_t)P.ldb * 8u;
    auto dma_piece = [&](auto tc, int k0, int st, const char* rowA, const char* rowB) {
        constexpr int t = decltype(tc)::value;
        const unsigned dst = wl + (unsigned)st * YSTAGE;
        if constexpr (t < YNP) {
            const char* base;
            unsigned v;
            if constexpr (A_KC) { base = reinterpret_cast<const char*>(P.A + (size_t)k0); v = va[t]; }
            else { base = rowA + (size_t)t * strideA2; v = va[0]; }
            asm volatile("" : "+v"(v));   // keeps the offset 32 bits wide: `uniform base + VGPR offset` is one address form of the
                                          // load; hoisted out of the loop the compiler widens it to a 64-bit pair and adds per piece
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + v), (lds_f32*)(size_t)(dst + (unsigned)t * 1024u), 16, 0, 0);
        } else if constexpr (t < 2 * YNP) {
            constexpr int u = t - YNP;
            const char* base;
            unsigned v;
            if constexpr (B_KC) { base = reinterpret_cast<const char*>(P.B + (size_t)k0); v = vb[u]; }
            else { base = rowB + (size_t)u * strideB2; v = vb[0]; }
            asm volatile("" : "+v"(v));
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + v), (lds_f32*)(size_t)(dst + YOP + (unsigned)u * 1024u), 16, 0, 0);
        } else if constexpr (AFF) {   // lanes 0-15: scale[k0 + lane], 16-31: shift[k0 + lane - 16]; lanes 32-63 repeat
            __builtin_amdgcn_global_load_lds(aff_src + (size_t)(k0 * aff_step), (lds_f32*)(size_t)(dst + YAFF), 4, 0, 0);
        }
    };
    auto row_a = [&](int k0) { return reinterpret_cast<const char*>(P.A) + (A_KC ? (size_t)0 : (size_t)k0 * (size_t)P.lda * 4u); };
    auto row_b = [&](int k0) { return reinterpret_cast<const char*>(P.B) + (B_KC ? (size_t)0 : (size_t)k0 * (size_t)P.ldb * 4u); };
    constexpr int NDMA = 2 * YNP + (AFF ? 1 : 0);

    // ---- fragment reads: 8 consecutive k of one row of a 32-row block, lane = (row, k half).
    // Wave W owns quadrant (qa, qb) = (W >> 1, W & 1) of the output tile at the end.  So that its OWN quadrant always sits in
    // the accumulators acc[0..1][0..1] — register names are compile-time, the wave number is not — it reads A block
    // blk ^ 2 qa and B block blk ^ 2 qb where the schedule says blk: acc[a][b] holds output tile (a ^ 2 qa, b ^ 2 qb).
    // The remap lives in the per-lane base addresses ("lo" serves the schedule's blocks 0, 1, "hi" the blocks 2, 3), the
    // stage / operand / block offsets stay instruction immediates.
    const int qa = wave >> 1, qb = wave & 1;
    constexpr unsigned A_BLK = A_KC ? 2048u : 128u, B_BLK = B_KC ? 2048u : 128u;   // bytes from one 32-row block to the next
    unsigned ra[2][2], rb[2][2];   // [lo / hi][16-byte half (k-contiguous operands only)]
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned lane_a = A_KC ? (unsigned)r32 * 64u + (unsigned)(((2 * h + e) ^ ((r32 >> 2) & 3)) * 16) : (unsigned)h * (8u * 512u) + (unsigned)r32 * 4u;
        const unsigned lane_b = B_KC ? (unsigned)r32 * 64u + (unsigned)(((2 * h + e) ^ ((r32 >> 2) & 3)) * 16) : (unsigned)h * (8u * 512u) + (unsigned)r32 * 4u;
        ra[0][e] = wl + lane_a + 2u * A_BLK * (unsigned)qa;
        ra[1][e] = wl + lane_a + 2u * A_BLK * (unsigned)(1 - qa);
        rb[0][e] = wl + YOP + lane_b + 2u * B_BLK * (unsigned)qb;
        rb[1][e] = wl + YOP + lane_b + 2u * B_BLK * (unsigned)(1 - qb);
    }
    auto read_frag = [&](auto fc, float (&dst)[8]) {
        constexpr FD f = fdec(decltype(fc)::value);
        constexpr bool KC = f.op == 0 ? A_KC : B_KC;
        constexpr int hi = f.blk >> 1;
        constexpr unsigned imm = (unsigned)f.par * YSTAGE + (unsigned)(f.blk & 1) * (KC ? 2048u : 128u);
        if constexpr (KC) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 v = *(lds_f32x4*)(size_t)((f.op == 0 ? ra[hi][e] : rb[hi][e]) + imm);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[4 * e + j] = v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = *(lds_f32*)(size_t)((f.op == 0 ? ra[hi][0] : rb[hi][0]) + imm + (unsigned)j * 512u);
        }
    };
    float aq[2][2][8];   // [parity of the k-step][scale / shift][this lane's 8 k]
    auto read_aff = [&](int par) {
        if constexpr (AFF) {
            const unsigned p = wl + (unsigned)par * YSTAGE + YAFF + (unsigned)h * 32u;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 v0 = *(lds_f32x4*)(size_t)(p + (unsigned)e * 64u), v1 = *(lds_f32x4*)(size_t)(p + (unsigned)e * 64u + 16u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    aq[par][e][j] = v0[j];
                    aq[par][e][4 + j] = v1[j];
                }
            }
        }
    };

    f32x16 acc[4][4];
    auto zero_acc = [&]() {
        float z = 0.f;
        asm volatile("" : "+v"(z));   // (a literal zero vector is hoisted to the kernel's entry and spilled across the k loop)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = z;
    };
    Frag PA[2][4], PB[2][4];   // [parity of the k-step][32-row block]
    float rawb[2][8];          // raw fragments, one window ahead of the split
    unsigned sh_[4], sm_[4];
    float sr_[8], se_[4];

    // A fragment as staged -> what the split sees: the folded BatchNorm affine, and (MASK: only where a wave's LAST k-step
    // is being prepared) zeros for the k a moved-back last step shares with its predecessor
    auto prep = [&](auto fc, float (&x)[8], int mlo, auto mask_tag) {
        constexpr FD f = fdec(decltype(fc)::value);
        if constexpr (f.op == 0) {
            if constexpr (AFF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], aq[f.par][0][j], aq[f.par][1][j]);
            }
            if constexpr (decltype(mask_tag)::value) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = j < mlo ? 0.f : x[j];
            }
        }
    };
    // the split in 11 slices of 4 vector instructions, one per MFMA of the window
    auto slice = [&](auto qc, const float (&x)[8], Frag& out) {
        constexpr int q = decltype(qc)::value;
        if constexpr (X1) {   // plain bf16: one conversion, nothing else
            if constexpr (q == 0) {
                unsigned hh[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) hh[p] = pack_bf16(x[2 * p], x[2 * p + 1]);
                out.p[0] = __builtin_bit_cast(bf16x8, (u32x4){hh[0], hh[1], hh[2], hh[3]});
            }
        } else if constexpr (q == 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) sh_[p] = pack_bf16(x[2 * p], x[2 * p + 1]);
        } else if constexpr (q == 1 || q == 3) {
            constexpr int o = q == 1 ? 0 : 2;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                se_[2 * p] = __builtin_bit_cast(float, sh_[o + p] << 16);
                se_[2 * p + 1] = __builtin_bit_cast(float, sh_[o + p] & 0xffff0000u);
            }
        } else if constexpr (q == 2 || q == 4) {
            constexpr int o = q == 2 ? 0 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) sr_[o + j] = x[o + j] - se_[j];
        } else if constexpr (q == 5) {
#pragma unroll
            for (int p = 0; p < 4; ++p) sm_[p] = pack_bf16(sr_[2 * p], sr_[2 * p + 1]);
        } else if constexpr (q == 6 || q == 8) {
            constexpr int o = q == 6 ? 0 : 2;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                se_[2 * p] = __builtin_bit_cast(float, sm_[o + p] << 16);
                se_[2 * p + 1] = __builtin_bit_cast(float, sm_[o + p] & 0xffff0000u);
            }
        } else if constexpr (q == 7 || q == 9) {
            constexpr int o = q == 7 ? 0 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) sr_[o + j] -= se_[j];
        } else if constexpr (q == 10) {
            unsigned l[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) l[p] = pack_bf16(sr_[2 * p], sr_[2 * p + 1]);
            out.p[0] = __builtin_bit_cast(bf16x8, (u32x4){sh_[0], sh_[1], sh_[2], sh_[3]});
            out.p[1] = __builtin_bit_cast(bf16x8, (u32x4){sm_[0], sm_[1], sm_[2], sm_[3]});
            out.p[2] = __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]});
        }
    };
    auto planes_of = [&](auto fc) -> Frag& {
        constexpr FD f = fdec(decltype(fc)::value);
        if constexpr (f.op == 0) return PA[f.par][f.blk];
        else return PB[f.par][f.blk];
    };

    // One k-step: 96 MFMAs in 8 windows of 12; slot g = {MFMA g, slice g % 12 of window g / 12's fragment, [the LDS reads
    // of the next window's fragment], [its prep], [one DMA piece of step i + 2]}, issue order pinned.
    // DMA: stage k-step i + 2 (k0_dma).  MASK: the NEXT step is this wave's last: its A fragments get the cover mask.
    // NEXT: a next step exists (else no fragment of it is read or split).
    auto step = [&](auto ty_tag, auto dma_tag, auto mask_tag, auto next_tag, int k0_dma, int mlo) {
        constexpr int TY = decltype(ty_tag)::value;
        constexpr bool DMA = decltype(dma_tag)::value, MASK = decltype(mask_tag)::value, NEXT = decltype(next_tag)::value;
        constexpr int cur = TY;
        const char* const rowA = row_a(k0_dma);
        const char* const rowB = row_b(k0_dma);
        static_for<96>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int n = g / 12, q = g % 12;
            {   // the MFMA: small terms first (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi)
                constexpr int grp = g / 24, inner = (g / 6) % 4, pr = g % 6;
                constexpr int a = TY == 0 ? grp : inner, b = TY == 0 ? inner : grp;
                constexpr int ia = pr == 0 ? 2 : pr == 2 || pr == 3 ? 1 : 0;
                constexpr int ib = pr == 1 ? 2 : pr == 2 || pr == 4 ? 1 : 0;
                if constexpr (!X1 || pr == 5)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[cur][a].p[ia], PB[cur][b].p[ib], acc[a][b], 0, 0, 0);
            }
            constexpr FD f = frag_of(TY, n);
            constexpr bool have = NEXT || f.par == cur;
            if constexpr (have && q <= 10) slice(std::integral_constant<int, q>{}, rawb[n & 1], planes_of(FC<fcode(f)>{}));
            constexpr FD f1 = frag_of(TY, n + 1);
            constexpr bool have1 = NEXT || (f1.par == cur && n + 1 < 8);
            if constexpr (g == 24 && NEXT) {
                wait_vm<0>();   // the next step's stage has landed (issued during the previous step); nothing else is in flight
                read_aff(cur ^ 1);
            }
            if constexpr (have1 && q == 0) read_frag(FC<fcode(f1)>{}, rawb[(n + 1) & 1]);
            if constexpr (have1 && q == 11)
                prep(FC<fcode(f1)>{}, rawb[(n + 1) & 1], mlo, std::integral_constant<bool, MASK && f1.par != cur>{});
            if constexpr (DMA && g >= 25 && (g - 25) % 4 == 0 && (g - 25) / 4 < NDMA)
                dma_piece(std::integral_constant<int, (g - 25) / 4>{}, k0_dma, cur, rowA, rowB);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    if (nw > 0) {
        // A wave's last step is always a BO step: with an odd number of steps the first one is a BO step (stage 1), run from a
        // body of its own in front of the pair loop.  Step j has type (j + odd) & 1; stage = type.
        using AO = std::integral_constant<int, 0>;
        using BO = std::integral_constant<int, 1>;
        using Yes = std::true_type;
        using No = std::false_type;
        const bool odd = nw & 1;
        const int cover_last = kbeg + YK * (wave + 4 * (nw - 1)) - k0_of(nw - 1);   // > 0 only for a moved-back last step
        const int mlo_last = cover_last - 8 * h;
        const int mlo0 = nw == 1 ? mlo_last : -64;
        // ---- prologue: stage k-steps 0 and 1, split what step 0 needs at its first MFMA and prepare its window 0
        if (odd) {
            const int k0 = k0_of(0), k1 = k0_of(nw > 1 ? 1 : 0);   // (a single step is staged twice: the count in flight is fixed)
            static_for<NDMA>([&](auto tc) { dma_piece(tc, k0, 1, row_a(k0), row_b(k0)); });
            __builtin_amdgcn_sched_barrier(0);
            zero_acc();   // while the first stage lands
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<0>();   // (no counted wait anywhere in this kernel: its pieces come in two widths)
            // the second stage is requested only now: it lands behind the splits below and is waited for 24 MFMAs into step 0
            static_for<NDMA>([&](auto tc) { dma_piece(tc, k1, 0, row_a(k1), row_b(k1)); });
            __builtin_amdgcn_sched_barrier(0);
            read_aff(1);
            static_for<4>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                constexpr FD f{0, 1, a};
                read_frag(FC<fcode(f)>{}, rawb[0]);
                prep(FC<fcode(f)>{}, rawb[0], mlo0, Yes{});
                static_for<11>([&](auto qc) { slice(qc, rawb[0], PA[1][a]); });
            });
            {
                constexpr FD f{1, 1, 0};
                read_frag(FC<fcode(f)>{}, rawb[0]);
                static_for<11>([&](auto qc) { slice(qc, rawb[0], PB[1][0]); });
            }
            {
                constexpr FD f{1, 1, 1};
                read_frag(FC<fcode(f)>{}, rawb[0]);
            }
            // the odd step in front: BO, step 0
            if (nw > 1) step(BO{}, Yes{}, No{}, Yes{}, k0_of(nw > 2 ? 2 : 1), 0);
            else step(BO{}, No{}, No{}, No{}, 0, 0);
        } else {
            const int k0 = k0_of(0), k1 = k0_of(1);
            static_for<NDMA>([&](auto tc) { dma_piece(tc, k0, 0, row_a(k0), row_b(k0)); });
            __builtin_amdgcn_sched_barrier(0);
            zero_acc();   // while the first stage lands
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<0>();
            static_for<NDMA>([&](auto tc) { dma_piece(tc, k1, 1, row_a(k1), row_b(k1)); });
            __builtin_amdgcn_sched_barrier(0);
            read_aff(0);
            static_for<4>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                constexpr FD f{1, 0, b};
                read_frag(FC<fcode(f)>{}, rawb[0]);
                static_for<11>([&](auto qc) { slice(qc, rawb[0], PB[0][b]); });
            });
            {
                constexpr FD f{0, 0, 0};
                read_frag(FC<fcode(f)>{}, rawb[0]);
                prep(FC<fcode(f)>{}, rawb[0], 0, No{});
                static_for<11>([&](auto qc) { slice(qc, rawb[0], PA[0][0]); });
            }
            {
                constexpr FD f{0, 0, 1};
                read_frag(FC<fcode(f)>{}, rawb[0]);
                prep(FC<fcode(f)>{}, rawb[0], 0, No{});
            }
        }
        // ---- pairs (AO, BO): all but the last one stage two steps ahead; the last pair masks the last step's A fragments
        int i = odd ? 1 : 0;
        for (; i + 2 < nw; i += 2) {
            step(AO{}, Yes{}, No{}, Yes{}, k0_of(i + 2), 0);
            step(BO{}, Yes{}, No{}, Yes{}, k0_of(i + 3), 0);
        }
        if (i < nw) {
            step(AO{}, No{}, Yes{}, Yes{}, 0, mlo_last);
            step(BO{}, No{}, No{}, No{}, 0, 0);
        }
        wait_vm<0>();   // nothing of this wave is in flight: the ring's LDS is reused below
    } else {
        zero_acc();
    }

    // ---- the four partial blocks meet in LDS.  Wave W keeps its quadrant — acc[0..1][0..1], see read_frag — and needs the other
    // three waves' partial sums of it; its accumulators acc[0..1][2..3], acc[2..3][0..1] and acc[2..3][2..3] are the
    // quadrants of the waves W ^ 1, W ^ 2 and W ^ 3.  Round 1: the quadrants of W ^ 1 and W ^ 2 (8 tiles, 32 KB per wave:
    // 128 KB), round 2: the quadrant of W ^ 3 (64 KB) — 192 KB through the LDS store path instead of the 256 KB of
    // "everybody parks everything", and all four waves add in both rounds.  Order of the sum: ((own + W^1) + W^2) + W^3 —
    // fixed per element: deterministic.
    f32x16 res[2][2];
    __syncthreads();   // every wave's ring is idle
    auto slot = [&](int owner, int s, int t, int q4) {   // [owner][source slot][tile of the quadrant][register quad], lane-linear
        return lds0 + (unsigned)((((owner * 2 + s) * 4 + t) * 4 + q4) * 1024) + (unsigned)lane * 16u;
    };
    auto send = [&](auto ac, auto bc, int owner, int s) {   // accumulators acc[A0 + 0..1][B0 + 0..1] to `owner`'s slot s
        constexpr int A0 = decltype(ac)::value, B0 = decltype(bc)::value;
        static_for<4>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const f32x16& v = acc[A0 + (t >> 1)][B0 + (t & 1)];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *(lds_f32x4*)(size_t)slot(owner, s, t, q4) = (f32x4){v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]};
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    send(I0{}, I2{}, wave ^ 1, 0);   // the owner W ^ 1 finds its partner (W ^ 1) ^ 1 = W in slot 0 ...
    send(I2{}, I0{}, wave ^ 2, 1);   // ... and the owner W ^ 2 its partner (W ^ 2) ^ 2 = W in slot 1
    __syncthreads();
    static_for<4>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        f32x4 v[2][4];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) v[s2][q4] = *(lds_f32x4*)(size_t)slot(wave, s2, t, q4);
        const f32x16& own = acc[t >> 1][t & 1];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) res[t >> 1][t & 1][4 * q4 + j] = (own[4 * q4 + j] + v[0][q4][j]) + v[1][q4][j];
    });
    __syncthreads();
    send(I2{}, I2{}, wave ^ 3, 0);
    __syncthreads();
    static_for<4>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        f32x4 v[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) v[q4] = *(lds_f32x4*)(size_t)slot(wave, 0, t, q4);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int j = 0; j < 4; ++j) res[t >> 1][t & 1][4 * q4 + j] += v[q4][j];
    });
    __syncthreads();   // (the epilogue's scratch lies behind the slots; the next work item's ring does not)
    // a tile inside the matrix (and, for the BatchNorm-aware form, quadrants on one side of the fc1 | fc3 row split) stores
    // row-major with 16-byte accesses; the others take the shared element-wise epilogue (workgroup-uniform choice)
    // (lane-derived values of the epilogue are computed from here on: formed at the kernel's entry they are spilled across the k
    //  loop, and a kernel with scratch pays for its private segment at every launch)
    int lane_e = lane, tid_e = (int)threadIdx.x;
    asm volatile("" : "+v"(lane_e), "+v"(tid_e));
    if (!(batch.diag & 1) && m0 + YT <= M && n0 + YT <= N && (!P.ep_gamma || (P.ep_split & 63) == 0))
        x5_epilogue_rows<!A_KC && !B_KC>(P, res, lds0 + (unsigned)wave * 32768u, reinterpret_cast<float*>(smem + 128 * 1024), split, tm_i,
                                         m0, n0, wave, lane_e, tid_e);
    else
        gemm_epilogue<YT, YT, 2, 2, !A_KC && !B_KC>(P, res, reinterpret_cast<float*>(smem + 128 * 1024), split, tm_i, m0, n0, tid_e);
}

template <bool A_KC, bool B_KC, bool AFF, bool X1>
__global__ __launch_bounds__(256) void gemm_x5_kernel(const GemmBatch batch) {
    __shared__ __attribute__((aligned(16))) char smem[YSMEM];
    for (int bid = (int)blockIdx.x; bid < batch.total_blocks; bid += (int)gridDim.x) {
        gemm_x5_block<A_KC, B_KC, AFF, X1>(batch, bid, smem);
        if (bid + (int)gridDim.x < batch.total_blocks) __syncthreads();  // the epilogue's LDS scratch is done with
    }
}

template <bool A_KC, bool B_KC, bool AFF, bool X1 = false>
static void launch_x5(const GemmBatch& b, hipStream_t st) {
    int grid = b.total_blocks;
    // side work (dgg_set_side_work): as launch_x3 — the forward products walk two work items per workgroup
    static const int kdiv = [] { const char* e = getenv("DGG_SIDE_DIV"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
    // (forward products only: halving the weight-gradient grids of side work as well costs 8-10 us per step — re-measured in
    //  round 6 on the deferred, grouped launches: 1.349 -> 1.356 ms)
    if (gemm_side_work() && A_KC && B_KC && grid > 64) {
        const int dv = kdiv + gemm_side_work() - 1;   // dgg_set_side_work(n): n = 1 the default division, n = 2 one more, ...
        grid = ((grid + dv - 1) / dv + 7) / 8 * 8;
    }
    static const int diag = [] { const char* e = getenv("DGG_X5_WIDE"); return (e && e[0] == '0') ? 1 : 0; }();
    GemmBatch bb = b;
    bb.diag = diag;
    hipLaunchKernelGGL((gemm_x5_kernel<A_KC, B_KC, AFF, X1>), dim3(grid), dim3(256), 0, st, bb);
}

}  // namespace x5


// A problem needs at least YMINK of K inside each split (a k-step moved back to the split's end must stay inside the split).
bool gemm_x5_enabled() {
    static const int on = [] { const char* e = getenv("DGG_X5"); return e ? atoi(e) : 1; }();
    return on != 0;
}
bool gemm_x5_prob_ok(const GemmProb& p) {
    if (p.K < x5::YMINK) return false;
    if (p.splits > 0) {
        if (p.kchunk < x5::YMINK) return false;
        const int lastk = p.K - (p.splits - 1) * p.kchunk;   // the last split's share
        if (lastk < x5::YK) return false;
    }
    return true;
}
bool gemm_x5_usable(const GemmBatch& batch) {
    if (!gemm_x5_enabled()) return false;
    for (int i = 0; i < batch.nprob; ++i)
        if (!gemm_x5_prob_ok(batch.p[i])) return false;
    return true;
}

void gemm_x5_dispatch(int mode, const GemmBatch& batch, hipStream_t st) {
    const bool x1 = gemm_algo() == 3;   // DGG_GEMM_BF16: the same kernel, hi plane only
    if (mode == GEMM_NT) {
        bool any = false;
        for (int i = 0; i < batch.nprob; ++i) any = any || batch.p[i].a_scale != nullptr;
        if (x1) {
            if (any) x5::launch_x5<true, true, true, true>(batch, st);
            else x5::launch_x5<true, true, false, true>(batch, st);
        } else if (any) x5::launch_x5<true, true, true>(batch, st);
        else x5::launch_x5<true, true, false>(batch, st);
    } else if (mode == GEMM_NN) {
        if (x1) x5::launch_x5<true, false, false, true>(batch, st);
        else x5::launch_x5<true, false, false>(batch, st);
    } else {
        if (x1) x5::launch_x5<false, false, false, true>(batch, st);
        else x5::launch_x5<false, false, false>(batch, st);
    }
}

}  // namespace dgg
