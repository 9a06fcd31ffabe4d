// Epilogue shared by the GEMM kernels of libdgg.so: both accumulate 32x32 MFMA tiles with the same D layout
//   col = lane & 31,  row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Plain store / read-modify-write into C (or a split-K slab), or — wgrad only — the BatchNorm-aware epilogue
// described at GemmProb::ep_gamma.
#pragma once
#include "dgg_internal.h"

namespace dgg {

// v of lane (lane ^ mask): __shfl_xor with the lane index handed in.  __shfl_xor takes it from v_mbcnt, which the compiler hoists
// to the kernel's entry and — in a kernel whose k loop leaves no register free — spills to scratch across the loop; a
// kernel with a private segment pays for it at every launch (gemm_x5.hip launders the index it passes after its k loop).
__device__ __forceinline__ float lane_xor(float v, int lane, int mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ mask) << 2, __builtin_bit_cast(int, v)));
}


using f32x16 = __attribute__((ext_vector_type(16))) float;

// The workgroup's WM x WN waves tile the BM x BN block (wave w: m-slot w / WN, n-slot w % WN), each holding
// (BM/WM/32) x (BN/WN/32) MFMA tiles.
// smem: at least 2 * WM * BN floats of workgroup-shared scratch, free once the k loop has passed its last barrier.
// tid: 0 .. 64*WM*WN-1 over those waves; every one of them reaches the barrier inside (and no other wave may).
template <int BM, int BN, int WM, int WN, bool WGRAD, int TA_B = BM / WM / 32>
__device__ __forceinline__ void gemm_epilogue(const GemmProb& P, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                              int split, int tm_i, int m0, int n0, int tid) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const int M = P.M, N = P.N;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int li = lane & 31, lh = lane >> 5;
    float* Cb = P.C + (P.splits > 0 ? (size_t)split * (size_t)P.slab_stride : (size_t)0);
    const bool accum = (P.splits == 0) && P.accumulate;
    // Element (row, col) of C / of ep_w through a 32-bit byte offset from the (wave-uniform) base: the address is
    // `SGPR pair + VGPR`, one register per element in flight instead of two and no 64-bit multiply-add per element.
    // gemm_launch refuses a problem whose C or ep_w does not fit 2 GB.
    char* const Cbytes = reinterpret_cast<char*>(Cb);
    const char* const Wbytes = reinterpret_cast<const char*>(P.ep_w);
    auto c_at = [&](int row, int col) -> float* {
        return reinterpret_cast<float*>(Cbytes + ((unsigned)row * (unsigned)P.ldc + (unsigned)col) * 4u);
    };
    auto f_at = [](const float* base, int i) -> float {   // base[i], i >= 0, through a 32-bit byte offset
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (unsigned)i * 4u);
    };
    auto w_at = [&](int row, int col) -> const float* {
        return reinterpret_cast<const float*>(Wbytes + ((unsigned)row * (unsigned)P.ep_ldw + (unsigned)col) * 4u);
    };
    // Every global LOAD below is unconditional, on indices clamped into the matrix, and all loads of a phase are issued
    // before the first is used; only the STORES are predicated.  (Round 4: with `if (row < M && col < N) { load; fma; store }`
    // per element the compiler could not move a load across its branch — 64 dependent L2 round trips per lane, 27 us of an
    // 86 us weight-gradient launch; profiles/r04_experiments.txt 9.)  Values and their order of accumulation are unchanged.
    if constexpr (WGRAD) {
        if (P.ep_gamma) {  // BatchNorm-aware wgrad epilogue (see GemmProb); wave-uniform branch
            float* red = smem;  // [WM m-slots][2][BN], free after the last barrier of the k loop
            // column sums of dY per output row: rows below ep_split take ep_cs0 (NULL: 0), the others ep_cs1 (+ ep_cs1b).
            // A 32-row MFMA tile lies on one side of the split wherever the split is a multiple of 32 (it is the hidden
            // width, 256): ONE uniform branch per tile around its 16 loads, no per-element select of two loads
            float csn[TM][16];
            {
                const int split = P.ep_split;
#pragma unroll
                for (int ta = 0; ta < TM; ++ta) {
                    const int base = m0 + wm0 + ta * 32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) csn[ta][r] = 0.f;
                    if (base + 32 <= split) {
                        if (P.ep_cs0) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) csn[ta][r] = f_at(P.ep_cs0, min(base + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1));
                        }
                    } else if (base >= split) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) csn[ta][r] = f_at(P.ep_cs1, min(base + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1) - split);
                        if (P.ep_cs1b) {
                            float t[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) t[r] = f_at(P.ep_cs1b, min(base + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1) - split);
#pragma unroll
                            for (int r = 0; r < 16; ++r) csn[ta][r] += t[r];
                        }
                    } else {   // a tile that straddles the split (never with the networks' shapes): element by element
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = min(base + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
                            if (row < split) csn[ta][r] = P.ep_cs0 ? f_at(P.ep_cs0, row) : 0.f;
                            else csn[ta][r] = f_at(P.ep_cs1, row - split) + (P.ep_cs1b ? f_at(P.ep_cs1b, row - split) : 0.f);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (base + (r & 3) + 8 * (r >> 2) + 4 * lh >= M) csn[ta][r] = 0.f;
                }
            }
#pragma unroll
            for (int tb = 0; tb < TN; ++tb) {
                const int col = n0 + wn0 + tb * 32 + li;
                const bool colok = col < N;
                const int colc = colok ? col : N - 1;
                const float ga = f_at(P.ep_gamma, colc), be = f_at(P.ep_beta, colc);
                // TA_B tile rows per batch of loads: all TM of them (64 loads in flight) where the kernel has the registers,
                // one (32) in the 512-thread kernels, whose budget is 128 + 128 — same values, same order of the sums
                float pg = 0.f, pb = 0.f;
#pragma unroll
                for (int tab = 0; tab < TM; tab += TA_B) {
                    float w[TA_B][16], old[TA_B][16];
#pragma unroll
                    for (int tq = 0; tq < TA_B; ++tq)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int row = m0 + wm0 + (tab + tq) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            row = row < M ? row : M - 1;
                            w[tq][r] = *w_at(row, colc);
                            old[tq][r] = 0.f;
                        }
                    if (accum) {   // one uniform branch around all the loads, not a select per element
#pragma unroll
                        for (int tq = 0; tq < TA_B; ++tq)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                int row = m0 + wm0 + (tab + tq) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                row = row < M ? row : M - 1;
                                old[tq][r] = *c_at(row, colc);
                            }
                    }
                    // every load of this batch has landed before the first store: none of them is still in flight when the
                    // workgroup's next work item starts its LDS-DMA ring (whose waits are counted; tools/vmcnt_audit.py)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int tq = 0; tq < TA_B; ++tq)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ta = tab + tq;
                            const int row = m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const bool ok = row < M && colok;
                            const float t = acc[ta][tb][r];
                            const float pg1 = __builtin_fmaf(w[tq][r], t, pg), pb1 = __builtin_fmaf(w[tq][r], csn[ta][r], pb);
                            pg = ok ? pg1 : pg;
                            pb = ok ? pb1 : pb;
                            float v = __builtin_fmaf(ga, t, be * csn[ta][r]);
                            if (accum) v += old[tq][r];
                            if (ok) *c_at(row, col) = v;
                        }
                }
                pg += lane_xor(pg, tid & 63, 32);
                pb += lane_xor(pb, tid & 63, 32);
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
                    old[ta][r] = *c_at(row, colc);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // as above: no load of the epilogue outlives it
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
                    if (row < M && col < N) *c_at(row, col) = old[ta][r];
                }
        }
        return;
    }
    if (m0 + BM <= M && n0 + BN <= N) {   // a tile inside the matrix (workgroup-uniform): 64 stores, no exec masks, no branches
#pragma unroll
        for (int ta = 0; ta < TM; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN; ++tb) {
                const int col = n0 + wn0 + tb * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) *c_at(m0 + wm0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col) = acc[ta][tb][r];
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
                if (row < M && col < N) *c_at(row, col) = acc[ta][tb][r];
            }
        }
}

}  // namespace dgg
