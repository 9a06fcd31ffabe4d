// fp32 GEMM family for the fc / ResBlock stacks (networks.py:32,35,39,75-77,121,293 and their
// autograd) on the CDNA4 matrix cores: v_mfma_f32_32x32x2_f32 — f32 in, f32 accumulate, bit-for-bit
// a k-ordered fmaf chain, so results are exact-fp32 GEMMs (no TF32-like path exists on gfx950).
//
// One kernel template serves the three operand orientations of forward / dgrad / wgrad:
//   NT  C[M,N] = A[M,K]  * B[N,K]^T   forward:  Y  = X  W^T       (both operands k-contiguous)
//   NN  C[M,N] = A[M,K]  * B[K,N]     dgrad:    dX = dY W
//   TN  C[M,N] = A[K,M]^T* B[K,N]     wgrad:    dW = dY^T X
// Tiles: BM x BN x 32 per 256-thread workgroup (4 waves as 2x2, each wave (BM/2)x(BN/2) as
// 32x32 MFMA tiles).  LDS holds both operands k-major ([k][m], row pitch BM+1 so the transposing
// store of a k-contiguous operand is bank-conflict-free) and double-buffered; the next k-tile is
// fetched global->registers while the current one feeds the MFMAs, one barrier per k-tile.
// Shapes of this model are awkward (K = 4101..4632, N = 21/256/512/768): every global access is
// predicated, K/M/N tails are zero-filled in LDS, never padded in the checkpoint tensors.
// A launch carries up to 4 independent problems (grouped GEMM) so the two column segments of a
// concatenated ResBlock input ([X0 | X1], networks.py:95,154,330) share one grid, and split-K
// partial products go to slabs that the bias/BN-statistics reduce kernel sums.
#include <stdlib.h>
#include <type_traits>
#include <string.h>
#include "dgg_internal.h"
#include "gemm_epilogue.h"

namespace dgg {

constexpr int BK = 32;
#ifndef DGG_ABLATE
#define DGG_ABLATE 0
#endif

// Global -> register staging of one BR x 32 operand tile.
//  FAST: the tile lies fully inside the matrix and the k-tile inside [kbeg,kend): every address is a
//        wave-uniform base (SGPR pair, advanced with scalar adds) + one loop-invariant per-lane offset,
//        so a load is a single `global_load_dword v, v_off, s[base]` with no predicate — the whole
//        k-tile body stays one basic block and the scheduler can sink the loads between the MFMAs.
//  !FAST: tails (last M/N tile, last k-tile of a split): indices are clamped into the matrix and the k
//        tail is zeroed with a select — branch-free, never out of bounds.  Rows/columns beyond M/N only
//        feed accumulator rows/columns that the epilogue never stores, so they need no zeroing.
template <int BR, bool KC, bool FAST>
__device__ __forceinline__ void load_tile(const float* __restrict__ G, int ld, int row0, int nrows, int k0, int kend,
                                          float (&r)[BR * BK / 256]) {
    const int tid = threadIdx.x;
    constexpr int NE = BR * BK / 256;
    if constexpr (KC) {
        // global [row][k]: a half-wave reads 32 consecutive k of one row (128 B)
        const int k = tid & 31, rl = tid >> 5;
        if constexpr (FAST) {
            const unsigned lane_byte = ((unsigned)rl * (unsigned)ld + (unsigned)k) * 4u;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const char* pu = reinterpret_cast<const char*>(G + ((size_t)(row0 + 8 * e) * ld + k0));  // uniform
                r[e] = *reinterpret_cast<const float*>(pu + lane_byte);
            }
        } else {
            const int gk = k0 + k;
            const bool kok = gk < kend;
            const int ck = kok ? gk : kend - 1;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                int gr = row0 + rl + 8 * e;
                gr = gr < nrows ? gr : nrows - 1;
                const float v = G[(size_t)gr * ld + ck];
                r[e] = kok ? v : 0.f;
            }
        }
    } else {
        // global [k][col]: a wave reads 64 consecutive columns of one k row (256 B)
        constexpr int KSTEP = 256 / BR;
        const int col = tid % BR, kl = tid / BR;
        if constexpr (FAST) {
            const unsigned lane_byte = ((unsigned)kl * (unsigned)ld + (unsigned)col) * 4u;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const char* pu = reinterpret_cast<const char*>(G + ((size_t)(k0 + KSTEP * e) * ld + row0));  // uniform
                r[e] = *reinterpret_cast<const float*>(pu + lane_byte);
            }
        } else {
            int gc = row0 + col;
            gc = gc < nrows ? gc : nrows - 1;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int gk = k0 + kl + KSTEP * e;
                const bool kok = gk < kend;
                const float v = G[(size_t)(kok ? gk : kend - 1) * ld + gc];
                r[e] = kok ? v : 0.f;
            }
        }
    }
}

// Per-k affine of a k-contiguous operand (the input BatchNorm's gamma/beta folded into the forward GEMM),
// applied when the staged registers are parked in LDS — NOT right after the loads, which would put a
// vmcnt(0) in front of the MFMAs and expose the full memory latency on every k-tile.  Zero-filled k tails
// (k >= kend) must stay zero.
template <int BR>
__device__ __forceinline__ void affine_tile(float (&r)[BR * BK / 256], const float* __restrict__ scale,
                                            const float* __restrict__ shift, int k0, int kend) {
    const int gk = k0 + (threadIdx.x & 31);
    const bool kok = gk < kend;
    const float sc = kok ? scale[gk] : 0.f, sh = kok ? shift[gk] : 0.f;
#pragma unroll
    for (int e = 0; e < BR * BK / 256; ++e) r[e] = __builtin_fmaf(r[e], sc, sh);
}

template <int BR, bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float (&r)[BR * BK / 256]) {
    const int tid = threadIdx.x;
    constexpr int NE = BR * BK / 256;
    constexpr int LDS_PITCH = BR + 1;
    if constexpr (KC) {
        const int k = tid & 31;
#pragma unroll
        for (int e = 0; e < NE; ++e) S[k * LDS_PITCH + (tid >> 5) + 8 * e] = r[e];
    } else {
        constexpr int KSTEP = 256 / BR;
        const int col = tid % BR;
#pragma unroll
        for (int e = 0; e < NE; ++e) S[(tid / BR + KSTEP * e) * LDS_PITCH + col] = r[e];
    }
}

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmBatch batch) {
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave along m / n
    constexpr int BUF = BK * (LDA_S + LDB_S);
    __shared__ float smem[2 * BUF];

    // ---- which problem / tile / k-split does this workgroup own (all wave-uniform)
    // (copy by scalar selects, not by a dynamic index: every field stays in SGPRs)
    GemmProb P = batch.p[0];
#pragma unroll
    for (int q = 1; q < kMaxProb; ++q)
        if (q < batch.nprob && (int)blockIdx.x >= batch.p[q].blk_begin) P = batch.p[q];
    // XCD-aware work assignment: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with
    // its own 4 MiB L2), so ids congruent mod 8 share an L2.  Give each XCD a CONTIGUOUS run of work items
    // (split-major, then tile): the 48 tiles of one k-split re-read the same A/W k-chunk (~3.3 MB), which then
    // stays in that XCD's L2 instead of being fetched by all eight.  Bijective for any block count; affects
    // speed only, never results.
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
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int M = P.M, N = P.N;
    int kbeg = 0, kend = P.K;
    if (P.splits > 0) {
        kbeg = split * P.kchunk;
        kend = min(P.K, kbeg + P.kchunk);
    }
    const int nkt = (kend - kbeg + BK - 1) / BK;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float ra[BM * BK / 256], rb[BN * BK / 256];
    const bool mn_full = (m0 + BM <= M) && (n0 + BN <= N);  // wave-uniform
    const bool aff = A_KC && P.a_scale != nullptr;          // wave-uniform
    // k-tiles [0, nfull) can use the predicate-free loads; tile nfull (if any) is the zero-filled tail
    const int nfull = mn_full ? (kend - kbeg) / BK : 0;

    // One k-tile: prefetch tile kt+1 (PF: 1 = fast loads, 2 = tail loads, 0 = none), MFMAs on tile kt, park the
    // prefetched registers in the other LDS buffer, barrier.  The three prefetch kinds are separate
    // instantiations on purpose: merging fast and tail loads behind one branch makes the compiler copy the
    // loaded registers at the join, i.e. wait vmcnt(0) BEFORE the MFMAs and expose the memory latency.
    auto ktile = [&](int kt, auto pf_tag) {
        constexpr int PF = decltype(pf_tag)::value;
        const float* As = smem + (kt & 1) * BUF;
        const float* Bs = As + BK * LDA_S;
        const int k0n = kbeg + (kt + 1) * BK;
        if constexpr (PF == 1 && DGG_ABLATE == 0) {
            load_tile<BM, A_KC, true>(P.A, P.lda, m0, M, k0n, kend, ra);
            load_tile<BN, B_KC, true>(P.B, P.ldb, n0, N, k0n, kend, rb);
        } else if constexpr (PF == 2) {
            load_tile<BM, A_KC, false>(P.A, P.lda, m0, M, k0n, kend, ra);
            load_tile<BN, B_KC, false>(P.B, P.ldb, n0, N, k0n, kend, rb);
        }
        // LDS -> register operand fetch runs one k-pair ahead of the MFMAs that consume it, so the ~100-cycle
        // ds_read latency overlaps the 4 x 64 cycles the matrix pipe spends on the previous k-pair
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[0][t] = As[lh * LDA_S + wm0 + t * 32 + li];
#pragma unroll
        for (int t = 0; t < TN; ++t) b[0][t] = Bs[lh * LDB_S + wn0 + t * 32 + li];
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const int cur = kp & 1, nxt = cur ^ 1;
            if (kp + 1 < BK / 2) {
                const int k = 2 * (kp + 1) + lh;
#pragma unroll
                for (int t = 0; t < TM; ++t) a[nxt][t] = (DGG_ABLATE >= 4) ? a[cur][t] * 1.0001f : As[k * LDA_S + wm0 + t * 32 + li];
#pragma unroll
                for (int t = 0; t < TN; ++t) b[nxt][t] = (DGG_ABLATE >= 4) ? b[cur][t] * 1.0001f : Bs[k * LDB_S + wn0 + t * 32 + li];
            }
            // pin the order in the emitted code: next pair's LDS reads are issued BEFORE this pair's MFMAs
            // (left alone, the scheduler sinks the reads below the MFMAs into the same registers and waits
            // lgkmcnt(0) on every pair, exposing the LDS latency 16 times per k-tile)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ta], b[cur][tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PF != 0 && DGG_ABLATE < 2) {
            float* Sn = smem + ((kt + 1) & 1) * BUF;
            if constexpr (A_KC) { if (aff) affine_tile<BM>(ra, P.a_scale, P.a_shift, k0n, kend); }
            store_tile<BM, A_KC>(Sn, ra);
            store_tile<BN, B_KC>(Sn + BK * LDA_S, rb);
        }
        if constexpr (DGG_ABLATE < 3) __syncthreads();
    };
    using PF0 = std::integral_constant<int, 0>;
    using PF1 = std::integral_constant<int, 1>;
    using PF2 = std::integral_constant<int, 2>;

    if (nkt > 0) {  // prologue: tile 0
        if (nfull > 0) {
            load_tile<BM, A_KC, true>(P.A, P.lda, m0, M, kbeg, kend, ra);
            load_tile<BN, B_KC, true>(P.B, P.ldb, n0, N, kbeg, kend, rb);
        } else {
            load_tile<BM, A_KC, false>(P.A, P.lda, m0, M, kbeg, kend, ra);
            load_tile<BN, B_KC, false>(P.B, P.ldb, n0, N, kbeg, kend, rb);
        }
        if constexpr (A_KC) { if (aff) affine_tile<BM>(ra, P.a_scale, P.a_shift, kbeg, kend); }
        store_tile<BM, A_KC>(smem, ra);
        store_tile<BN, B_KC>(smem + BK * LDA_S, rb);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nfull; ++kt) ktile(kt, PF1{});   // next tile full
    for (; kt + 1 < nkt; ++kt) ktile(kt, PF2{});     // next tile is a tail (or the M/N edge)
    if (kt < nkt) ktile(kt, PF0{});                  // last tile: nothing to prefetch

    gemm_epilogue<BM, BN, 2, 2, !A_KC && !B_KC, 1>(P, acc, smem, split, tm_i, m0, n0, tid);
}

// The ResBlock's middle stage (GemmProb::bn_part set) has a kernel of its own: fc2_stage.hip.
int gemm_fc2_fused_launch(GemmBatch& batch, hipStream_t st) {
    if (batch.nprob != 1) return DGG_ERR_INVALID;
    return fc2_stage_launch(batch.p[0], st);
}

// ------------------------------------------------------------------------------------------------
void gemm_tile_shape(int mode, int M, int N, int& bm, int& bn, int K) {
    // Big tiles when they alone (or with split-K in NT mode) can fill 256 CUs; otherwise 64-wide.
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    bm = 128;
    bn = 128;
    if (mode == GEMM_NT) {
        const long ksteps = (K + BK - 1) / BK;
        const long maxs = ksteps / 4 > 0 ? ksteps / 4 : 1;  // split-K keeps >= 4 k-tiles per workgroup
        if (t128 * maxs < 200) { bm = 64; bn = 64; }
        return;
    }
    // the bf16x3 kernel (128 x 128 only, one workgroup per CU) is worth a partly filled chip: 198 tiles of the
    // 768 x 4101 weight gradient at 128 x 128 beat 780 tiles of 64 x 64 on the f32 pipe by 1.5x
    const long enough = gemm_algo() >= 1 ? 128 : 200;
    // A weight gradient with few tiles and a long K (the real pass' pose columns + the fc2 rider at B = 1024: 44 tiles of
    // 64 x 64, each a chain of 32 k-tiles: 45-79 us) is a latency chain on the f32 kernel; on gemm_x5_kernel the four waves of
    // a workgroup split K: ~37 us at K = 1024 however few the tiles.  Below K = 768 the f32 kernel's chain is the shorter one.
    // (profiles/r05_experiments.txt 23)
    if (mode == GEMM_TN && gemm_algo() >= 1 && gemm_x5_enabled() && K >= 768 && K < (1 << 30)) return;
    if (t128 < enough) {
        bm = (M >= 1024) ? 128 : 64;
        bn = 64;
        const long t = (long)((M + bm - 1) / bm) * ((N + 63) / 64);
        if (t < 200) bm = 64;
    }
}

int gemm_pick_splits(int mode, int M, int N, int K, int other_blocks) {
    int bm, bn;
    gemm_tile_shape(mode, M, N, bm, bn, K);
    const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const int ksteps = (K + BK - 1) / BK;
    static int target = -1;
    if (target < 0) {
        const char* e = getenv("DGG_SPLITK_TARGET");
        target = e ? atoi(e) : (gemm_algo() >= 1 ? 192 : 512);  // bf16x3: one workgroup per CU, a quarter of the chip left to the other streams
    }
    int s = (target - other_blocks) / (tiles > 0 ? tiles : 1);
    if (s < 1) s = 1;
    if (s > 16) s = 16;
    const int max_by_k = ksteps / 4 > 0 ? ksteps / 4 : 1;  // keep >= 4 k-tiles per split
    if (s > max_by_k) s = max_by_k;
    return s;
}

int gemm_plan(GemmProb& p, int mode, int splits) {
    int bm, bn;
    gemm_tile_shape(mode, p.M, p.N, bm, bn, p.K);
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = (p.N + bn - 1) / bn;
    p.splits = splits;
    if (splits > 0) {
        const int ksteps = (p.K + BK - 1) / BK;
        const int per = (ksteps + splits - 1) / splits;
        p.kchunk = per * BK;
        // drop empty trailing splits
        int used = (p.K + p.kchunk - 1) / p.kchunk;
        if (used < 1) used = 1;
        p.splits = used;
    } else {
        p.kchunk = p.K;
    }
    return p.tiles_m * p.tiles_n * (p.splits > 0 ? p.splits : 1);
}

template <bool A_KC, bool B_KC>
static void launch_shape(int bm, int bn, const GemmBatch& b, hipStream_t st) {
    dim3 grid(b.total_blocks), block(256);
    if (bm == 128 && bn == 128)
        hipLaunchKernelGGL((gemm_f32_kernel<128, 128, A_KC, B_KC>), grid, block, 0, st, b);
    else if (bm == 128 && bn == 64)
        hipLaunchKernelGGL((gemm_f32_kernel<128, 64, A_KC, B_KC>), grid, block, 0, st, b);
    else if (bm == 64 && bn == 128)
        hipLaunchKernelGGL((gemm_f32_kernel<64, 128, A_KC, B_KC>), grid, block, 0, st, b);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<64, 64, A_KC, B_KC>), grid, block, 0, st, b);
}

// Which arithmetic the large GEMMs (128 x 128 tiles) run: 1 (default) = bf16x3 on fp32 operands, split inside the GEMM
// (gemm_x5.hip: one wave per 128 x 128 accumulator block; launches of >= 1024 workgroups: gemm_x4.hip, loader / consumer
// waves); 0 = native f32 MFMA (this file); 3 = plain bf16 (the hi plane only: one product, fp32 accumulate — BASELINE
// config 3, never the default).  Process-wide; DGG_GEMM_ALGO=f32 / bf16 select 0 / 3 at start-up.  (2 was the pre-split
// "planes" path of rounds 3-4, removed in round 5.)
static int g_algo = -1;
int gemm_algo() {
    if (g_algo < 0) {
        const char* e = getenv("DGG_GEMM_ALGO");
        g_algo = (e && (e[0] == 'f' || e[0] == '0')) ? 0 : (e && (e[0] == 'b' || e[0] == '3')) ? 3 : 1;
    }
    return g_algo;
}
void gemm_set_algo(int a) { g_algo = a == 0 ? 0 : a == 3 ? 3 : 1; }

// "Side work" (dgg_set_side_work, per calling thread): the launches that follow run beside a latency-critical chain on
// another stream — their large forward products take half the chip for twice as long (the bf16x3 kernel walks two work
// items per workgroup) so that the chain's kernels always find free CUs.
static thread_local int g_side_work = 0;
int gemm_side_work() { return g_side_work; }
void gemm_set_side_work(int on) { g_side_work = on < 0 ? 0 : on; }   // 0: off; n >= 1: n + 1 work items per workgroup where the default divides by 2

// Plans the tiles of a grouped launch and decides which kernel(s) run it — without launching anything:
//   returns DGG_OK and fills bm / bn, every problem's tiles_m / tiles_n / blk_begin, batch.total_blocks and `route`:
//   GEMM_ROUTE_F32 (this file's f32-MFMA kernels, any tile shape), GEMM_ROUTE_X4 (gemm_x4_kernel: bf16x3, launches of >= 1024
//   workgroups), GEMM_ROUTE_X5 (gemm_x5_kernel: bf16x3 / plain bf16; `large` / `small` then hold the problems it takes and the
//   short-K ones left to the f32-MFMA kernel at the same 128 x 128 tiling).
static int gemm_route(int mode, GemmBatch& batch, int& bm, int& bn, int& route, GemmBatch& large, GemmBatch& small) {
    if (batch.nprob < 1 || batch.nprob > kMaxProb) return DGG_ERR_INVALID;
    if (mode != GEMM_NT && mode != GEMM_NN && mode != GEMM_TN) return DGG_ERR_INVALID;
    int big = 0;  // the problem with the most work decides the tile shape
    for (int i = 1; i < batch.nprob; ++i)
        if ((double)batch.p[i].M * batch.p[i].N * batch.p[i].K > (double)batch.p[big].M * batch.p[big].N * batch.p[big].K) big = i;
    gemm_tile_shape(mode, batch.p[big].M, batch.p[big].N, bm, bn, batch.p[big].K);
    int blocks = 0;
    for (int i = 0; i < batch.nprob; ++i) {
        GemmProb& p = batch.p[i];
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.B || !p.C) return DGG_ERR_INVALID;
        // the epilogue addresses C (one slab of it) and ep_w through 32-bit byte offsets (gemm_epilogue.h)
        if ((long long)p.M * p.ldc * 4 >= (1ll << 31) || (p.ep_gamma && (long long)p.M * p.ep_ldw * 4 >= (1ll << 31))) {
            set_error_text("GEMM output of %d rows x pitch %d floats: the epilogues address C through 32-bit byte offsets, "
                           "rows * pitch * 4 must stay below 2 GiB (chunk the batch: at most %lld rows at this pitch)",
                           p.M, p.ldc, ((1ll << 31) - 1) / (4ll * (p.ldc > 0 ? p.ldc : 1)));
            return DGG_ERR_INVALID;
        }
        // ... and the bf16x3 kernels form a k-contiguous operand's row base as a 32-bit byte offset (gemm_x5.hip `va` / `vb`,
        // gemm_x4.hip): rows * pitch * 4 of such an operand must stay below 4 GiB (the f32-MFMA kernel of this file indexes in
        // 64 bits, but a launch is refused whatever route it would take: the limit is a property of the entry, not of the route)
        {
            const bool a_kc = mode != GEMM_TN, b_kc = mode == GEMM_NT;
            const long long abytes = a_kc ? (long long)p.M * p.lda * 4 : 0, bbytes = b_kc ? (long long)p.N * p.ldb * 4 : 0;
            if (abytes >= (1ll << 32) || bbytes >= (1ll << 32)) {
                const bool a_bad = abytes >= (1ll << 32);
                const int pitch = a_bad ? p.lda : p.ldb;
                set_error_text("GEMM operand %s of %d rows x pitch %d floats: k-contiguous operands are addressed through 32-bit "
                               "byte offsets, rows * pitch * 4 must stay below 4 GiB (chunk the batch: at most %lld rows at this pitch)",
                               a_bad ? "A" : "B", a_bad ? p.M : p.N, pitch, ((1ll << 32) - 1) / (4ll * (pitch > 0 ? pitch : 1)));
                return DGG_ERR_INVALID;
            }
        }
        p.tiles_m = (p.M + bm - 1) / bm;
        p.tiles_n = (p.N + bn - 1) / bn;
        p.blk_begin = blocks;
        blocks += p.tiles_m * p.tiles_n * (p.splits > 0 ? p.splits : 1);
    }
    batch.total_blocks = blocks;
    route = GEMM_ROUTE_F32;
    large.nprob = small.nprob = 0;
    if (gemm_algo() >= 1 && bm == 128 && bn == 128) {
        // gemm_x4_kernel fetches a k-strided operand ([K][rows]: B of NN, A and B of TN) in quads of 4 rows: with
        // rows % 4 != 0 the last quad of EVERY k-row reads up to 12 bytes past the row — past the operand itself on its last
        // k-row when the pitch is the row count.  Such problems stay on gemm_x5_kernel, whose padding rule is
        // in include/dgg.h (dgg_gemm_f32); inside the library every pitch covers the quad.
        bool quads_ok = true;
        for (int i = 0; i < batch.nprob; ++i) {
            const GemmProb& p = batch.p[i];
            if (mode == GEMM_TN && p.lda < ((p.M + 3) & ~3)) quads_ok = false;
            if (mode != GEMM_NT && p.ldb < ((p.N + 3) & ~3)) quads_ok = false;
        }
        if (quads_ok && gemm_algo() != 3 && gemm_x4_enabled(batch.total_blocks)) {
            route = GEMM_ROUTE_X4;
            return DGG_OK;
        }
        if (gemm_x5_enabled()) {
            // gemm_x5_kernel takes every problem with at least 64 of K; a grouped launch that also carries short ones (the
            // 24 pose columns beside the shared BPS block of a discriminator pass) is cut in two: the short problems are a
            // handful of tiles on the f32-MFMA kernel of this file
            large.total_blocks = small.total_blocks = 0;
            large.diag = small.diag = 0;
            for (int i = 0; i < batch.nprob; ++i) {
                GemmBatch& dst = gemm_x5_prob_ok(batch.p[i]) ? large : small;
                GemmProb& q = dst.p[dst.nprob++];
                q = batch.p[i];
                q.blk_begin = dst.total_blocks;
                dst.total_blocks += q.tiles_m * q.tiles_n * (q.splits > 0 ? q.splits : 1);
            }
            if (large.nprob > 0) route = GEMM_ROUTE_X5;
        }
    }
    return DGG_OK;
}

// All problems of one grouped launch must use the same tile shape: the launcher derives it from the problem with the
// most work and re-plans the others with it.
#define DGG_TRY_RC(expr) do { const int rc__ = (expr); if (rc__ != DGG_OK) return rc__; } while (0)
int gemm_launch(int mode, GemmBatch& batch, hipStream_t st) {
    int bm, bn, route;
    GemmBatch large, small;
    DGG_TRY_RC(gemm_route(mode, batch, bm, bn, route, large, small));
    if (route == GEMM_ROUTE_X4) {
        gemm_x4_dispatch(mode, batch, st);
        return check_launch("gemm_x4");
    }
    if (route == GEMM_ROUTE_X5) {
        gemm_x5_dispatch(mode, large, st);
        DGG_TRY_RC(check_launch("gemm_x5"));
        if (small.nprob == 0) return DGG_OK;
    }
    // (the short problems of a cut launch are tiled 128 x 128 by gemm_route; the caller's batch keeps its own plan — callers
    //  read p[i].splits back after the launch)
    const GemmBatch& rest = route == GEMM_ROUTE_X5 ? small : batch;
    if (mode == GEMM_NT) launch_shape<true, true>(bm, bn, rest, st);
    else if (mode == GEMM_NN) launch_shape<true, false>(bm, bn, rest, st);
    else launch_shape<false, false>(bm, bn, rest, st);
    return check_launch("gemm_f32");
}

// Which kernel a single product would run on (dgg_gemm_route: diagnostics and CPU tests of the routing rules; nothing is
// launched, no pointer is dereferenced).  10 * kernel + (short problems left to the f32 kernel ? 1 : 0) is not needed for one
// problem: returns GEMM_ROUTE_*.
int gemm_route_single(int mode, int M, int N, int K, int lda, int ldb, int ldc, int splits) {
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    g.nprob = 1;
    GemmProb& p = g.p[0];
    static float dummy;
    p.A = p.B = &dummy;
    p.C = &dummy;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    gemm_plan(p, mode, splits);
    int bm, bn, route;
    GemmBatch large, small;
    const int rc = gemm_route(mode, g, bm, bn, route, large, small);
    return rc != DGG_OK ? rc : route;
}

// Residency query (advisory, see MI355X notes on the occupancy API): workgroups of the given GEMM variant that
// fit one CU at once.
int gemm_occupancy(int mode, int bm, int bn) {
    int n = -1;
    const void* f = nullptr;
    if (mode == GEMM_NT && bm == 128 && bn == 128) f = (const void*)gemm_f32_kernel<128, 128, true, true>;
    else if (mode == GEMM_NT) f = (const void*)gemm_f32_kernel<64, 64, true, true>;
    else if (mode == GEMM_NN && bm == 128 && bn == 128) f = (const void*)gemm_f32_kernel<128, 128, true, false>;
    else if (mode == GEMM_NN) f = (const void*)gemm_f32_kernel<64, 64, true, false>;
    else if (mode == GEMM_TN && bm == 128 && bn == 128) f = (const void*)gemm_f32_kernel<128, 128, false, false>;
    else f = (const void*)gemm_f32_kernel<64, 64, false, false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 256, 0) != hipSuccess) return -1;
    return n;
}

}  // namespace dgg
