// Epilogue shared by the GEMM kernels of libdgg.so: both accumulate 32x32 MFMA tiles with the same D layout
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Plain store / read-modify-write into C (or a split-K slab), or — wgrad only — the BatchNorm-aware epilogue
// described at GemmProb::ep_gamma.
#pragma once
#include "dgg_internal.h"

namespace dgg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// The workgroup's WM x WN waves tile the BM x BN block (wave w: m-slot w / WN, n-slot w % WN), each holding
// (BM/WM/32) x (BN/WN/32) MFMA tiles.
// smem: at least 2 * WM * BN floats of workgroup-shared scratch, free once the k loop has passed its last barrier.
// tid: 0 .. 64*WM*WN-1 over those waves; every one of them reaches the barrier inside (and no other wave may).
template <int BM, int BN, int WM, int WN, bool WGRAD>
__device__ __forceinline__ void gemm_epilogue(const GemmProb& P, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                              int split, int tm_i, int m0, int n0, int tid) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const int M = P.M, N = P.N;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int li = lane & 31, lh = lane >> 5;
    float* Cb = P.C + (P.splits > 0 ? (size_t)split * (size_t)P.slab_stride : (size_t)0);
    const bool accum = (P.splits == 0) && P.accumulate;
    // Every global LOAD below is unconditional, on indices clamped into the matrix, and all loads of a phase are issued
    // before the first is used; only the STORES are predicated.  (Round 4: with `if (row < M && col < N) { load; fma; store }`
    // per element the compiler could not move a load across its branch — 64 dependent L2 round trips per lane, 27 us of an
    // 86 us weight-gradient launch; profiles/r04_experiments.txt 9.)  Values and their order of accumulation are unchanged.
    if constexpr (WGRAD) {
        if (P.ep_gamma) {  // BatchNorm-aware wgrad epilogue (see GemmProb); wave-uniform branch
            float* red = smem;  // [WM m-slots][2][BN], free after the last barrier of the k loop
            float csn[TM][16];
            {
                const int split = P.ep_split;
                float c0[TM][16], c1[TM][16], c1b[TM][16];
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        row = row < M ? row : M - 1;
                        const int r0 = row < split ? row : (split > 0 ? split - 1 : 0);
                        const int r1 = row >= split ? row - split : 0;
                        c0[ta][r] = P.ep_cs0 ? P.ep_cs0[r0] : 0.f;      // (pointer tests: wave-uniform)
                        c1[ta][r] = P.ep_cs1[r1];
                        c1b[ta][r] = P.ep_cs1b ? P.ep_cs1b[r1] : 0.f;
                    }
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float v1 = P.ep_cs1b ? c1[ta][r] + c1b[ta][r] : c1[ta][r];
                        csn[ta][r] = row < M ? (row < split ? c0[ta][r] : v1) : 0.f;
                    }
            }
#pragma unroll
            for (int tb = 0; tb < TN; ++tb) {
                const int col = n0 + wn0 + tb * 32 + li;
                const bool colok = col < N;
                const int colc = colok ? col : N - 1;
                const float ga = P.ep_gamma[colc], be = P.ep_beta[colc];
                float w[TM][16], old[TM][16];
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        row = row < M ? row : M - 1;
                        w[ta][r] = P.ep_w[(size_t)row * P.ep_ldw + colc];
                        old[ta][r] = accum ? Cb[(size_t)row * P.ldc + colc] : 0.f;
                    }
                float pg = 0.f, pb = 0.f;
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const bool ok = row < M && colok;
                        const float t = acc[ta][tb][r];
                        const float pg1 = __builtin_fmaf(w[ta][r], t, pg), pb1 = __builtin_fmaf(w[ta][r], csn[ta][r], pb);
                        pg = ok ? pg1 : pg;
                        pb = ok ? pb1 : pb;
                        float v = __builtin_fmaf(ga, t, be * csn[ta][r]);
                        if (accum) v += old[ta][r];
                        if (ok) Cb[(size_t)row * P.ldc + col] = v;
                    }
                pg += __shfl_xor(pg, 32);
                pb += __shfl_xor(pb, 32);
                if (lh == 0) {
                    red[((wave / WN) * 2 + 0) * BN + wn0 + tb * 32 + li] = pg;
                    red[((wave / WN) * 2 + 1) * BN + wn0 + tb * 32 + li] = pb;
                }
            }
            __syncthreads();
            for (int idx = tid; idx < 2 * BN; idx += 64 * WM * WN) {
                const int q = idx / BN, cidx = idx - q * BN;
                const int col = n0 + cidx;
                if (col < N) {
                    float t = red[q * BN + cidx];
#pragma unroll
                    for (int w = 1; w < WM; ++w) t += red[(w * 2 + q) * BN + cidx];  // fixed order: deterministic
                    P.ep_part[((size_t)tm_i * 2 + q) * (P.ep_part_ld ? P.ep_part_ld : N) + col] = t;
                }
            }
            return;
        }
    }
    if (accum) {   // wave-uniform: read-modify-write — all the old values first
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const int col = n0 + wn0 + tb * 32 + li;
            const int colc = col < N ? col : N - 1;
            float old[TM][16];
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    row = row < M ? row : M - 1;
                    old[ta][r] = Cb[(size_t)row * P.ldc + colc];
                }
            // (sums into registers of their own first: a store whose data register the next sum overwrites makes the
            //  compiler wait for that store — vmcnt(0) after every one of them)
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) old[ta][r] = acc[ta][tb][r] + old[ta][r];
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < M && col < N) Cb[(size_t)row * P.ldc + col] = old[ta][r];
                }
        }
        return;
    }
#pragma unroll
    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const int col = n0 + wn0 + tb * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) Cb[(size_t)row * P.ldc + col] = acc[ta][tb][r];
            }
        }
}

}  // namespace dgg
