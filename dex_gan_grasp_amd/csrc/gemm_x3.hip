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
// One 1024-thread workgroup (16 waves, four per SIMD) per 128 x 128 x K-range tile, in two roles:
//   * waves 0-7, PRODUCERS: each stages 1/8 of the next k-tile: global fp32 -> registers (ring of XNS tiles in
//     flight) -> [input-BN affine] -> split3 -> three bf16 planes per operand in LDS, [row][k] with an 80-byte row
//     pitch (32 bf16 + 16 B pad: the ds_read_b128 operand fetches of a 16-lane group hit 16 distinct 4-bank sets);
//   * waves 8-15, CONSUMERS: each owns a 32 x 64 piece of the tile (4 x 2): 18 fragment reads + 24 MFMAs per k-tile.
//   LDS is double-buffered, one barrier per k-tile hands a buffer over.
// Why roles, and why two producers per SIMD: waves issue in order, and the per-CU L2 -> L1 rate (~70 GB/s measured)
// makes the loads of a 32 KB k-tile cost about as much as its 1536 MFMA cycles — a wave that issues them sits in the
// address unit's back-pressure for that long and can do nothing else.  Measured on MI355X (1024 x 768 x 36904):
// one instruction stream doing both (any interleaving, also two 256-thread workgroups per CU) ran at
// time(memory) + time(MFMA); 4 producers + 4 consumers left the consumers idle 70% of the time behind producers that
// serialise VALU, LDS-store and load back-pressure.  With two producers per SIMD one splits while the other waits.
// Global loads are 16 bytes per lane whenever base and pitch of the operand allow (all activations; the weights of
// D and E): the address unit handles 4 lanes per clock whatever the width, so dword loads cap a CU at 16 B/clk.
#include <stdint.h>
#include <type_traits>
#include "dgg_internal.h"
#include "gemm_epilogue.h"

namespace dgg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef DGG_X3_ABLATE
#define DGG_X3_ABLATE 0  // diagnostics only: 1 no global loads, 2 no split/park, 3 no compute, 4 MFMA without LDS reads, 5 LDS reads without MFMA
#endif

constexpr int XK = 32;    // k-tile
constexpr int XP = 80;    // LDS row pitch in bytes
constexpr int XT = 128;   // tile rows / columns
constexpr int XE = 8;     // fp32 elements staged per thread, operand and k-tile (128 * 32 / 512)
constexpr int XNS = 3;    // register ring depth (k-tiles in flight)
constexpr int XPL = XT * XP;  // bytes of one plane

__device__ __forceinline__ unsigned pack_bf16(f32x2 v) {  // v_cvt_pk_bf16_f32 (RNE): low half = v.x
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 unpack_bf16(unsigned p) {
    f32x2 r;
    r.x = __builtin_bit_cast(float, p << 16);
    r.y = __builtin_bit_cast(float, p & 0xffff0000u);
    return r;
}
// 4 adjacent-k pairs -> packed (hi, mid, lo); stage by stage so that the four dependent chains overlap.
__device__ __forceinline__ void split3x4(const f32x2 (&v)[4], unsigned (&h)[4], unsigned (&m)[4], unsigned (&l)[4]) {
    f32x2 r[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) h[p] = pack_bf16(v[p]);
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = v[p] - unpack_bf16(h[p]);
#pragma unroll
    for (int p = 0; p < 4; ++p) m[p] = pack_bf16(r[p]);
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = r[p] - unpack_bf16(m[p]);
#pragma unroll
    for (int p = 0; p < 4; ++p) l[p] = pack_bf16(r[p]);
}

// ------------------------------------------------------------------------------------------------
// Staging of one 128 x 32 operand tile by 512 threads, 8 elements each.  KC: operand stored [row][k] (k contiguous);
// !KC: stored [k][row].  V: 16-byte loads (base, pitch 16-byte aligned).  FAST: the k-tile lies fully inside
// [kbeg, kend) (no k predication); else k tails are clamped and zeroed.  Row indices are always clamped.
//   KC  V : lane = (k quad tid&7, row tid>>3 (+64));           r[4e + j] = (row + 64e, k 4*k4 + j)
//   KC !V : lane = (k tid&31,     row tid>>5 (+16e, e < 8));   r[e]
//  !KC  V : lane = (k half ih = tid&1, k quad kq = tid[1:2] + 4 tid[6], row quad n4 = tid[3:5] + 8 tid[7:8]);
//           r[4i + j] = (k 4*kq + 2*ih + i, row 4*n4 + j); a wave-load reads 8 k rows x 128 contiguous bytes
//  !KC !V : lane = (row tid&127,  k octet tid>>7);             r[e] = (k 8*kb + e)
// ------------------------------------------------------------------------------------------------
template <bool KC, bool V, bool FAST>
__device__ __forceinline__ void stage_load(const float* __restrict__ G, int ld, int row0, int nrows, int k0, int kend,
                                           float (&r)[XE]) {
    const int tid = threadIdx.x;
    // Row (column) indices are clamped into the matrix in both variants: the per-lane part of every address is
    // loop-invariant, so the last M / N tile costs nothing extra (rows beyond the edge only feed accumulator rows
    // that are never stored).  !FAST additionally predicates and zero-fills k >= kend, element by element.
    if constexpr (KC && V) {
        const int k4 = tid & 7, rl = tid >> 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int gr = row0 + rl + 64 * e;
            gr = gr < nrows ? gr : nrows - 1;
            const float* p = G + (size_t)gr * ld + (k0 + 4 * k4);
            if constexpr (FAST) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                for (int j = 0; j < 4; ++j) r[4 * e + j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool kok = k0 + 4 * k4 + j < kend;
                    const float v = kok ? p[j] : p[kend - 1 - (k0 + 4 * k4)];
                    r[4 * e + j] = kok ? v : 0.f;
                }
            }
        }
    } else if constexpr (KC) {
        const int k = tid & 31, rl = tid >> 5;
        const int gk = k0 + k;
        const bool kok = FAST || gk < kend;
        const int ck = kok ? gk : kend - 1;
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            int gr = row0 + rl + 16 * e;
            gr = gr < nrows ? gr : nrows - 1;
            const float v = G[(size_t)gr * ld + ck];
            r[e] = kok ? v : 0.f;
        }
    } else if constexpr (V) {
        const int ih = tid & 1, kq = ((tid >> 1) & 3) + 4 * ((tid >> 6) & 1), n4 = ((tid >> 3) & 7) + 8 * (tid >> 7);
        // the aligned column quad holding the last valid column lies inside the pitch (ld % 4 == 0, ld >= nrows)
        int gc = row0 + 4 * n4;
        const int last4 = (nrows - 1) & ~3;
        gc = gc < last4 ? gc : last4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gk = k0 + 4 * kq + 2 * ih + i;
            const bool kok = FAST || gk < kend;
            const f32x4 v = *reinterpret_cast<const f32x4*>(G + (size_t)(kok ? gk : kend - 1) * ld + gc);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[4 * i + j] = kok ? v[j] : 0.f;
        }
    } else {
        const int col = tid & 127, kb = (tid >> 7) * XE;
        int gc = row0 + col;
        gc = gc < nrows ? gc : nrows - 1;
#pragma unroll
        for (int e = 0; e < XE; ++e) {
            const int gk = k0 + kb + e;
            const bool kok = FAST || gk < kend;
            const float v = G[(size_t)(kok ? gk : kend - 1) * ld + gc];
            r[e] = kok ? v : 0.f;
        }
    }
}

// Per-k affine of the k-contiguous A operand (input BatchNorm folded into the forward GEMM).  The (scale, shift) of
// a thread's k columns travel through the register ring with the tile itself, so their loads sit under the same
// counted vmcnt waits as the data (loaded inside the k-tile body they would force vmcnt(0) — draining the whole
// prefetch ring — before every split).  `on` is wave-uniform; when off, the (always valid) pointers are still read
// and the values replaced by 1 / 0, which keeps the k-tile body free of branches.  k >= kend gets (0, 0): zero-
// filled k tails stay zero.
//   V: q[0..3] = scale, q[4..7] = shift of k = k0 + 4*(tid&7) + j;   !V: q[0], q[1] of k = k0 + (tid&31)
template <bool V, bool FAST>
__device__ __forceinline__ void affine_load(const float* __restrict__ scale, const float* __restrict__ shift, bool on,
                                            int k0, int kend, float (&q)[XE]) {
    if constexpr (V) {
        const int kb = k0 + 4 * (threadIdx.x & 7);
        if constexpr (FAST) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(scale + kb);
            const f32x4 b = *reinterpret_cast<const f32x4*>(shift + kb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[j] = on ? a[j] : 1.f;
                q[4 + j] = on ? b[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool kok = kb + j < kend;
                const int kk = kok ? kb + j : kend - 1;
                const float a = scale[kk], b = shift[kk];
                q[j] = on ? (kok ? a : 0.f) : 1.f;
                q[4 + j] = on ? (kok ? b : 0.f) : 0.f;
            }
        }
    } else {
        const int gk = k0 + (threadIdx.x & 31);
        const bool kok = FAST || gk < kend;
        const int kk = kok ? gk : kend - 1;
        const float a = scale[kk], b = shift[kk];
        q[0] = on ? (kok ? a : 0.f) : 1.f;
        q[1] = on ? (kok ? b : 0.f) : 0.f;
    }
}
template <bool V>
__device__ __forceinline__ void affine_apply(float (&r)[XE], const float (&q)[XE]) {
    if constexpr (V) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) r[4 * e + j] = __builtin_fmaf(r[4 * e + j], q[j], q[4 + j]);
    } else {
#pragma unroll
        for (int e = 0; e < XE; ++e) r[e] = __builtin_fmaf(r[e], q[0], q[1]);
    }
}

// registers -> three LDS planes (hi, mid, lo; XPL bytes apart) of one operand
template <bool KC, bool V>
__device__ __forceinline__ void stage_park(char* __restrict__ planes, const float (&r)[XE]) {
    const int tid = threadIdx.x;
    f32x2 v[4];
    unsigned h[4], m[4], l[4];
    if constexpr (KC && V) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p].x = r[2 * p];
            v[p].y = r[2 * p + 1];
        }
        split3x4(v, h, m, l);
        char* base = planes + (tid >> 3) * XP + (tid & 7) * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) {  // 8-byte stores; lanes 0-7 / 8-15 of a 16-lane group are two adjacent rows
            char* p = base + 64 * e * XP;
            const u32x2 hh = {h[2 * e], h[2 * e + 1]}, mm = {m[2 * e], m[2 * e + 1]}, ll = {l[2 * e], l[2 * e + 1]};
            *reinterpret_cast<u32x2*>(p) = hh;
            *reinterpret_cast<u32x2*>(p + XPL) = mm;
            *reinterpret_cast<u32x2*>(p + 2 * XPL) = ll;
        }
    } else if constexpr (KC) {
        // lanes k and k^1 swap half of their rows (one DPP quad_perm move per pair): each lane then owns
        // (k even, k odd) pairs — the packed form cvt_pk produces — for 4 of its 8 rows
        const bool odd = (tid & 1) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e0 = r[2 * i], e1 = r[2 * i + 1];
            const float send = odd ? e0 : e1;
            const float recv = __builtin_bit_cast(
                float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
            v[i].x = odd ? recv : e0;
            v[i].y = odd ? e1 : recv;
        }
        split3x4(v, h, m, l);
        char* base = planes + ((tid >> 5) + (odd ? 16 : 0)) * XP + ((tid & 31) >> 1) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // row = rl + 16 * (2 i + odd)
            char* p = base + 32 * i * XP;
            *reinterpret_cast<unsigned*>(p) = h[i];
            *reinterpret_cast<unsigned*>(p + XPL) = m[i];
            *reinterpret_cast<unsigned*>(p + 2 * XPL) = l[i];
        }
    } else if constexpr (V) {
        // lanes l, l^1 hold the two k halves (ih) of the same 4 rows: they swap two rows each (DPP quad_perm), after
        // which a lane owns 2 rows x 4 consecutive k = one 8-byte store per row and plane
        const bool hi = (tid & 1) != 0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float keep[2], recv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                keep[i] = hi ? r[4 * i + 2 + jj] : r[4 * i + jj];
                const float send = hi ? r[4 * i + jj] : r[4 * i + 2 + jj];
                recv[i] = __builtin_bit_cast(
                    float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
            }
            v[2 * jj].x = hi ? recv[0] : keep[0];      // k = 4 kq + 0, 1
            v[2 * jj].y = hi ? recv[1] : keep[1];
            v[2 * jj + 1].x = hi ? keep[0] : recv[0];  // k = 4 kq + 2, 3
            v[2 * jj + 1].y = hi ? keep[1] : recv[1];
        }
        split3x4(v, h, m, l);
        const int kq = ((tid >> 1) & 3) + 4 * ((tid >> 6) & 1), n4 = ((tid >> 3) & 7) + 8 * (tid >> 7);
        char* base = planes + (4 * n4 + (hi ? 2 : 0)) * XP + kq * 8;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {  // a 16-lane group (2 ih x 4 kq x 2 n4) covers all 32 banks once
            char* p = base + jj * XP;
            const u32x2 hh = {h[2 * jj], h[2 * jj + 1]}, mm = {m[2 * jj], m[2 * jj + 1]}, ll = {l[2 * jj], l[2 * jj + 1]};
            *reinterpret_cast<u32x2*>(p) = hh;
            *reinterpret_cast<u32x2*>(p + XPL) = mm;
            *reinterpret_cast<u32x2*>(p + 2 * XPL) = ll;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i].x = r[2 * i];
            v[i].y = r[2 * i + 1];
        }
        split3x4(v, h, m, l);
        char* p = planes + (tid & 127) * XP + (tid >> 7) * 16;  // 8 consecutive k = one 16-byte store per plane
        const u32x4 hh = {h[0], h[1], h[2], h[3]}, mm = {m[0], m[1], m[2], m[3]}, ll = {l[0], l[1], l[2], l[3]};
        *reinterpret_cast<u32x4*>(p) = hh;
        *reinterpret_cast<u32x4*>(p + XPL) = mm;
        *reinterpret_cast<u32x4*>(p + 2 * XPL) = ll;
    }
}

#define X3_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <bool A_KC, bool B_KC, bool VA, bool VB>
__global__ __launch_bounds__(1024, 1) void gemm_x3_kernel(const GemmBatch batch) {
    constexpr int BUF = 6 * XPL;  // A planes then B planes
    static_assert(BUF >= 8 * XT * (int)sizeof(float), "epilogue scratch");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    // ---- work assignment: identical to gemm_f32_kernel (problem select, XCD-contiguous remap, split-K range)
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
    const int m0 = tm_i * XT, n0 = tn_i * XT;
    const int M = P.M, N = P.N;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nkt = (kend - kbeg + XK - 1) / XK;

    if (threadIdx.x < 512) {
        // ================= producer waves 0-7 (two per SIMD): global -> register ring -> split3 -> LDS planes
        float ra[XNS][XE], rb[XNS][XE], rq[XNS][XE];  // k-tile t lives in set t % XNS (rq: affine scale / shift)
        const bool aff = A_KC && P.a_scale != nullptr;
        const float* const sc_p = aff ? P.a_scale : P.A;  // valid either way, see affine_load
        const float* const sh_p = aff ? P.a_shift : P.A;
        const int nfull = (kend - kbeg) / XK;  // k-tiles [0, nfull) are full: loads without k predication
        auto fetch = [&](int kt, float (&a)[XE], float (&b)[XE], float (&q)[XE]) {
            if constexpr (DGG_X3_ABLATE == 1) return;
            const int k0 = kbeg + kt * XK;
            if (kt < nfull) {
                stage_load<A_KC, VA, true>(P.A, P.lda, m0, M, k0, kend, a);
                stage_load<B_KC, VB, true>(P.B, P.ldb, n0, N, k0, kend, b);
                if constexpr (A_KC) affine_load<VA, true>(sc_p, sh_p, aff, k0, kend, q);
            } else if (kt < nkt) {
                stage_load<A_KC, VA, false>(P.A, P.lda, m0, M, k0, kend, a);
                stage_load<B_KC, VB, false>(P.B, P.ldb, n0, N, k0, kend, b);
                if constexpr (A_KC) affine_load<VA, false>(sc_p, sh_p, aff, k0, kend, q);
            }
        };
        auto park = [&](int kt, float (&a)[XE], float (&b)[XE], const float (&q)[XE]) {
            if constexpr (DGG_X3_ABLATE == 2) return;
            char* buf = smem + (kt & 1) * BUF;
            if constexpr (A_KC) affine_apply<VA>(a, q);
            stage_park<A_KC, VA>(buf, a);
            stage_park<B_KC, VB>(buf + 3 * XPL, b);
        };
#pragma unroll
        for (int u = 0; u < XNS; ++u) fetch(u, ra[u], rb[u], rq[u]);
        if (nkt > 0) park(0, ra[0], rb[0], rq[0]);
        // Identical producers would move in lock-step — all splitting, then all storing, then all waiting on the
        // address unit — and use those resources one after the other.  Half of them (one wave of each SIMD pair)
        // therefore run "load first": in k-tile kt they fetch tile kt+XNS into the set freed one k-tile earlier
        // and then park tile kt+1; the others park tile kt+1 and then fetch tile kt+1+XNS into the same set.
        const bool load_first = (threadIdx.x >> 8) != 0;  // waves 4-7
        if (!load_first) fetch(XNS, ra[0], rb[0], rq[0]);
        __syncthreads();  // tile 0 is in buffer 0
        for (int kt0 = 0; kt0 < nkt; kt0 += XNS) {
#pragma unroll
            for (int u = 0; u < XNS; ++u) {
                const int kt = kt0 + u;
                if (kt < nkt) {
                    // buffer (kt+1)&1 was last read for tile kt-1, i.e. before the barrier just passed
                    if (load_first) {
                        fetch(kt + XNS, ra[u], rb[u], rq[u]);
                        if (kt + 1 < nkt) park(kt + 1, ra[(u + 1) % XNS], rb[(u + 1) % XNS], rq[(u + 1) % XNS]);
                    } else if (kt + 1 < nkt) {
                        park(kt + 1, ra[(u + 1) % XNS], rb[(u + 1) % XNS], rq[(u + 1) % XNS]);
                        fetch(kt + 1 + XNS, ra[(u + 1) % XNS], rb[(u + 1) % XNS], rq[(u + 1) % XNS]);
                    }
                    __syncthreads();
                }
            }
        }
        if constexpr (!A_KC && !B_KC) {
            if (P.ep_gamma) __syncthreads();  // matches the barrier inside the BatchNorm-aware epilogue
        }
        return;
    }

    // ===================== consumer waves 8-15 (two per SIMD): LDS -> MFMA, 4 x 2 over the tile, 32 x 64 each
    const int tid = threadIdx.x - 512;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[1][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

    // operand fragment addresses: row li of the wave's piece, 16-byte k chunk lh (+2 for the second k16 step)
    const int offA = (wm0 + li) * XP + lh * 16;
    const int offB = 3 * XPL + (wn0 + li) * XP + lh * 16;
#ifdef DGG_X3_PRIO
    __builtin_amdgcn_s_setprio(DGG_X3_PRIO);
#endif
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (DGG_X3_ABLATE != 3) {
            const char* fA = smem + (kt & 1) * BUF + offA;
            const char* fB = smem + (kt & 1) * BUF + offB;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 a[3], b[2][3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if constexpr (DGG_X3_ABLATE == 4) a[p] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, 1u, 2u, (unsigned)p});
                    else a[p] = *reinterpret_cast<const bf16x8*>(fA + p * XPL + 32 * st);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        if constexpr (DGG_X3_ABLATE == 4) b[t][p] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)kt, 3u, (unsigned)t, (unsigned)p});
                        else b[t][p] = *reinterpret_cast<const bf16x8*>(fB + t * 32 * XP + p * XPL + 32 * st);
                    }
                if constexpr (DGG_X3_ABLATE == 5) {  // reads only: fold them into the accumulator cheaply
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const u32x4 x = __builtin_bit_cast(u32x4, a[p]) ^ __builtin_bit_cast(u32x4, b[t][p]);
                            acc[0][t][p] += __builtin_bit_cast(float, x[0] ^ x[1] ^ x[2] ^ x[3]);
                        }
                    continue;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    X3_MFMA(a[2], b[t][0], acc[0][t]);  // lo  * hi
                    X3_MFMA(a[0], b[t][2], acc[0][t]);  // hi  * lo
                    X3_MFMA(a[1], b[t][1], acc[0][t]);  // mid * mid
                    X3_MFMA(a[1], b[t][0], acc[0][t]);  // mid * hi
                    X3_MFMA(a[0], b[t][1], acc[0][t]);  // hi  * mid
                    X3_MFMA(a[0], b[t][0], acc[0][t]);  // hi  * hi
                }
            }
        }
        __syncthreads();
    }
    gemm_epilogue<XT, XT, 4, 2, !A_KC && !B_KC>(P, acc, reinterpret_cast<float*>(smem), split, tm_i, m0, n0, tid);
}

template <bool A_KC, bool B_KC>
static void launch_x3(bool va, bool vb, const GemmBatch& b, hipStream_t st) {
    dim3 grid(b.total_blocks), block(1024);
    if (va && vb) hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, true, true>), grid, block, 0, st, b);
    else if (va) hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, true, false>), grid, block, 0, st, b);
    else if (vb) hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, false, true>), grid, block, 0, st, b);
    else hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, false, false>), grid, block, 0, st, b);
}

// 16-byte staging needs base and pitch of the operand 16-byte aligned in every problem of the launch (k origins
// are multiples of 32 elements, tile origins multiples of 128 rows).
static bool aligned16(const float* p, int ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// 128 x 128 tiles only.
void gemm_x3_dispatch(int mode, const GemmBatch& batch, hipStream_t st) {
    bool va = true, vb = true;
    for (int i = 0; i < batch.nprob; ++i) {
        va = va && aligned16(batch.p[i].A, batch.p[i].lda);
        vb = vb && aligned16(batch.p[i].B, batch.p[i].ldb);
    }
    if (mode == GEMM_NT) launch_x3<true, true>(va, vb, batch, st);
    else if (mode == GEMM_NN) launch_x3<true, false>(va, vb, batch, st);
    else launch_x3<false, false>(va, vb, batch, st);
}

int gemm_x3_occupancy(int mode) {
    int n = -1;
    const void* f = (const void*)gemm_x3_kernel<true, true, true, true>;
    if (mode == GEMM_NN) f = (const void*)gemm_x3_kernel<true, false, true, true>;
    if (mode == GEMM_TN) f = (const void*)gemm_x3_kernel<false, false, true, true>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 1024, 0) != hipSuccess) return -1;
    return n;
}

}  // namespace dgg
