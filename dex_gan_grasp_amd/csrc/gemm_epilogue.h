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
    if constexpr (WGRAD) {
        if (P.ep_gamma) {  // BatchNorm-aware wgrad epilogue (see GemmProb); wave-uniform branch
            float* red = smem;  // [WM m-slots][2][BN], free after the last barrier of the k loop
            float csn[TM][16];
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = 0.f;
                    if (row < M)
                        v = row < P.ep_split ? (P.ep_cs0 ? P.ep_cs0[row] : 0.f)
                                             : P.ep_cs1[row - P.ep_split] + (P.ep_cs1b ? P.ep_cs1b[row - P.ep_split] : 0.f);
                    csn[ta][r] = v;
                }
#pragma unroll
            for (int tb = 0; tb < TN; ++tb) {
                const int col = n0 + wn0 + tb * 32 + li;
                const bool colok = col < N;
                const float ga = colok ? P.ep_gamma[col] : 0.f, be = colok ? P.ep_beta[col] : 0.f;
                float pg = 0.f, pb = 0.f;
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (row < M && colok) {
                            const float t = acc[ta][tb][r];
                            const float w = P.ep_w[(size_t)row * P.ep_ldw + col];
                            pg = __builtin_fmaf(w, t, pg);
                            pb = __builtin_fmaf(w, csn[ta][r], pb);
                            float* p = Cb + (size_t)row * P.ldc + col;
                            float v = __builtin_fmaf(ga, t, be * csn[ta][r]);
                            if (accum) v += *p;
                            *p = v;
                        }
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
#pragma unroll
    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < TN; ++tb) {
            const int col = n0 + wn0 + tb * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N) {
                    float* p = Cb + (size_t)row * P.ldc + col;
                    float v = acc[ta][tb][r];
                    if (accum) v += *p;
                    *p = v;
                }
            }
        }
}

}  // namespace dgg
