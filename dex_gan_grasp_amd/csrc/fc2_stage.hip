// The ResBlock's middle stage in one launch:  h = lrelu(bn1(a1)),  a2 = h W2^T + b2,  + the row-block statistics of a2
// (networks.py:49-52; GemmProb::bn_part and friends, dgg_internal.h).  K = 256 (the hidden width), N = 512.
//
// Until round 5 this stage ran as an instantiation of gemm_f32_kernel with 64 x 64 tiles: 128 workgroups at B = 1024 (32 at
// B = 256), each walking 8 k-tiles of {stage through LDS, barrier, 16 dependent MFMAs} behind a statistics merge — 18 us for
// 0.27 GFLOP, 15 us at B = 256: a latency chain, not work (profiles/r04_experiments.txt 15).  Here a workgroup owns ONE
// 32 x 32 output tile and its four waves split K: wave w takes k in [64 w, 64 w + 64).  A lane loads its 32 consecutive k of
// one A row and one W2 row straight into registers (128 contiguous bytes each, no LDS staging, no k loop), the wave issues 32
// MFMAs (v_mfma_f32_32x32x2_f32: slot 0 = lanes 0-31, slot 1 = lanes 32-63, so MFMA s pairs k = 64 w + s with 64 w + 32 + s),
// the four partial tiles meet in LDS (fixed order w0 + w1 + w2 + w3: deterministic) and wave 0 finishes: + bias, store, and
// the exact two-pass (mean, M2) of the tile's 32 rows per column — a 32-row tile IS one row block.  512 workgroups at B = 1024
// (128 at B = 256): the chip is covered and every dependent chain is a quarter as long.  With more than 256 such tiles a
// workgroup takes TWO column tiles (NT = 2: 32 x 64, the A registers shared, wave t < 2 finishes tile t): half as many
// workgroups repeat the statistics merge and re-read A.
#include <stdint.h>
#include "dgg_internal.h"

namespace dgg {

namespace fc2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));   // 16-byte access at any dword address

constexpr int kK = 256;      // hidden width (cfg_ok: hidden == 256)
constexpr int kKW = kK / 4;  // k per wave
constexpr int kKL = kKW / 2; // k per lane (32)

template <int NT>   // column tiles of 32 per workgroup
__global__ __launch_bounds__(256) void fc2_stage_kernel(const GemmProb P) {
    __shared__ __attribute__((aligned(16))) float sc_s[kK], sh_s[kK];
    __shared__ float red[4][NT][16][64];   // the waves' partial tiles, lane-linear (a wave's own final tile is not parked)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.M, N = P.N;
    const int tiles_m = (M + 31) / 32;
    const int tn = (int)blockIdx.x / tiles_m, tm = (int)blockIdx.x - tn * tiles_m;   // neighbours share the W2 rows
    const int m0 = tm * 32, n0 = tn * 32 * NT;
    const int kb = kKW * wave + kKL * lh;   // this lane's first k

    // ---- operands and parameters are requested first; the statistics merge runs while they are in flight
    const int arow = min(m0 + li, M - 1);
    f32x4 a4[kKL / 4], b4[NT][kKL / 4];
    {
        const f32x4_u* ap = reinterpret_cast<const f32x4_u*>(P.A + (size_t)arow * P.lda + kb);
#pragma unroll
        for (int q = 0; q < kKL / 4; ++q) a4[q] = ap[q];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4_u* bp = reinterpret_cast<const f32x4_u*>(P.B + (size_t)min(n0 + 32 * t + li, N - 1) * P.ldb + kb);
#pragma unroll
            for (int q = 0; q < kKL / 4; ++q) b4[t][q] = bp[q];
        }
    }
    // (used by the epilogue of the wave that finishes tile `wave`: column n0 + 32 wave + li)
    const float bias = P.c_bias[min(n0 + 32 * (wave < NT ? wave : 0) + li, N - 1)];
    {
        // BatchNorm statistics of the K columns: thread = column; Chan merge of the row-block partials in row-block order — 32
        // row blocks per round with all 64 loads in flight, merged as four interleaved chains (block j goes to chain j % 4)
        // that are then merged in order.  Every workgroup: same order, same bits.  Eval: running statistics.
        const float gamma_k = P.bn_gamma[tid], beta_k = P.bn_beta[tid];
        float mean = 0.f, rstd;
        if (P.bn_train) {
            float cnt = 0.f, m2 = 0.f;
            int b0 = 0;
            do {   // (bn_nrb >= 1: the launcher checks — with a loop that may not run the compiler waits for the parameters with a
                   //  counted vmcnt among the operand loads on the path that never happens)
                float mb[32], qb[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const float* q = P.bn_part + (size_t)min(b0 + j, P.bn_nrb - 1) * 2 * kK;
                    mb[j] = q[tid];
                    qb[j] = q[kK + tid];
                }
                // (16-byte operand loads and one-dword loads are in flight together: waited for all at once — the partials were
                //  requested last, so nothing is given up — never with a counted vmcnt over the mix: tools/vmcnt_audit.py)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                float cc[4] = {0.f, 0.f, 0.f, 0.f}, cm[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if (b0 + j < P.bn_nrb) {
                        const int ch = j & 3;
                        const float nb = (float)min(kRowBlock, P.bn_rows - (b0 + j) * kRowBlock);
                        const float tot = cc[ch] + nb;
                        const float d = mb[j] - cm[ch];
                        cm[ch] += d * (nb / tot);
                        cq[ch] += qb[j] + d * d * (cc[ch] * nb / tot);
                        cc[ch] = tot;
                    }
                }
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    if (cc[ch] > 0.f) {
                        const float tot = cnt + cc[ch];
                        const float d = cm[ch] - mean;
                        mean += d * (cc[ch] / tot);
                        m2 += cq[ch] + d * d * (cnt * cc[ch] / tot);
                        cnt = tot;
                    }
                }
            } while ((b0 += 32) < P.bn_nrb);
            rstd = 1.0f / sqrtf(m2 / cnt + kBnEps);
            if (blockIdx.x == 0) {
                P.bn_mean_out[tid] = mean;
                P.bn_rstd_out[tid] = rstd;
                if (P.bn_run_mean) P.bn_run_mean[tid] = (1.f - kBnMom) * P.bn_run_mean[tid] + kBnMom * mean;
                if (P.bn_run_var) P.bn_run_var[tid] = (1.f - kBnMom) * P.bn_run_var[tid] + kBnMom * (m2 / (cnt - 1.f));
            }
        } else {
            mean = P.bn_run_mean[tid];
            const float var_k = P.bn_run_var[tid];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (as above)
            rstd = 1.0f / sqrtf(var_k + kBnEps);
            if (blockIdx.x == 0) {
                P.bn_mean_out[tid] = mean;
                P.bn_rstd_out[tid] = rstd;
            }
        }
        const float sc = rstd * gamma_k;
        sc_s[tid] = sc;
        sh_s[tid] = beta_k - mean * sc;
    }
    __syncthreads();

    // ---- h = lrelu(bn1(a1)) on this lane's 32 k of its row; the first column of tiles writes h out for the backward
    float a[kKL];
    {
        const f32x4* scp = reinterpret_cast<const f32x4*>(sc_s + kb);
        const f32x4* shp = reinterpret_cast<const f32x4*>(sh_s + kb);
        const bool out_h = tn == 0 && P.a_out && m0 + li < M;
        f32x4_u* hp = reinterpret_cast<f32x4_u*>(P.a_out + (size_t)arow * kK + kb);
#pragma unroll
        for (int q = 0; q < kKL / 4; ++q) {
            const f32x4 s4 = scp[q], t4 = shp[q];
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = __builtin_fmaf(a4[q][j], s4[j], t4[j]);
                x = x > 0.f ? x : kLrelu * x;
                v[j] = x;
                a[4 * q + j] = x;
            }
            if (out_h) hp[q] = v;
        }
    }
    // ---- 32 MFMAs per tile: the wave's share of K
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < kKL; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b4[t][s >> 2][s & 3], acc[t], 0, 0, 0);
    // ---- the four partial tiles meet in LDS; wave t < NT adds tile t's in wave order and finishes it
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    if (wave >= NT) return;
    const int nvalid = min(kRowBlock, M - m0);   // rows of this tile inside the batch
    const int col = n0 + 32 * wave + li;
    const bool cok = col < N;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        v[r] = (((red[0][wave][r][lane] + red[1][wave][r][lane]) + red[2][wave][r][lane]) + red[3][wave][r][lane]) + bias;
        if (rr < nvalid) {
            s += v[r];
            if (cok) P.C[(size_t)(m0 + rr) * P.ldc + col] = v[r];
        }
    }
    // the exact two-pass (mean, M2) of the row block per column: lanes l and l + 32 hold its two 16-row halves
    s += __shfl_xor(s, 32);
    const float mean = s / (float)nvalid;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float d = v[r] - mean;
        if (rr < nvalid) q += d * d;
    }
    q += __shfl_xor(q, 32);
    if (lh == 0 && cok) {
        float* o = P.c_part + (size_t)tm * 2 * N;
        o[col] = mean;
        o[N + col] = q;
    }
}

// The backward mirror: dh = da2 W2 (K = 512), dq = dh * lrelu'(h), and the row-block partial sums of bn1's backward
// (sum dq, sum dq * x_hat1 per 32 rows and column) in ONE launch — until round 5 a split-K GEMM on gemm_f32_kernel (four
// slabs) plus act_bwd_kernel, which summed the slabs: two dependent launches and 4 MB of slabs per ResBlock backward.
// One 32 x 32 output tile per workgroup, EIGHT waves split K = 512 (64 k each: a lane holds 32 k of one da2 row and of
// one W2 column), 32 dependent MFMAs per wave, the partial tiles meet in LDS in wave order; wave 0 — which requested its
// h and a1 values at the start — finishes the tile.
struct Fc2Bwd {
    const float* da2; int ldda;     // (B, 512)
    const float* w2; int ldw;       // (512, 256): W2[k][n]
    const float* h; int ldh;        // (B, 256) the forward's activated hidden values
    const float* a1; int lda1;      // (B, .) bn1's input
    const float* mean; const float* rstd;   // (256) bn1's batch (train) or running (eval) statistics
    int B;
    float* dq; int lddq;            // (B, 256)
    float* part;                    // [row block][2][256]
};
constexpr int kKB = 512, kNB = 256;   // K = n_neurons, N = hidden

__global__ __launch_bounds__(512) void fc2_bwd_stage_kernel(const Fc2Bwd P) {
    __shared__ float red[8][16][64];   // 32 KB: the waves' partial tiles, lane-linear
    __shared__ float cs[8][2][64];     // per finishing wave: its rows' share of the two column sums
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int M = P.B;
    const int tiles_m = (M + 31) / 32;
    const int tn = (int)blockIdx.x / tiles_m, tm = (int)blockIdx.x - tn * tiles_m;
    const int m0 = tm * 32, n0 = tn * 32;
    const int kb = 64 * wave + 32 * lh;   // this lane's first k
    const int arow = min(m0 + li, M - 1), col = n0 + li;
    // ---- every load of the kernel is requested here; one wait (two widths in flight: never a counted vmcnt over them)
    f32x4 a4[8];
    float b[32];
    {
        const f32x4_u* ap = reinterpret_cast<const f32x4_u*>(P.da2 + (size_t)arow * P.ldda + kb);
#pragma unroll
        for (int q = 0; q < 8; ++q) a4[q] = ap[q];
        const float* bp = P.w2 + (size_t)kb * P.ldw + col;
#pragma unroll
        for (int s = 0; s < 32; ++s) b[s] = bp[(size_t)s * P.ldw];
    }
    // Wave w finishes the accumulator registers 2 w and 2 w + 1 of the tile: rows (r & 3) + 8 (r >> 2) + 4 lh, column li — its
    // h and a1 values are requested with the operands
    float hv[2], av[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * wave + e;
        const int row = min(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
        hv[e] = P.h[(size_t)row * P.ldh + col];
        av[e] = P.a1[(size_t)row * P.lda1 + col];
    }
    const float mu = P.mean[col], rs = P.rstd[col];
    __builtin_amdgcn_sched_barrier(0);   // (nothing that uses a loaded value is scheduled in front of the wait)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- 32 MFMAs: the wave's share of K (slot 0 = lanes 0-31: k = 64 w + s, slot 1: k = 64 w + 32 + s)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s >> 2][s & 3], b[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const int nvalid = min(kRowBlock, M - m0);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * wave + e;
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float d = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) d += red[w][r][lane];   // wave (= k) order
        d *= hv[e] > 0.f ? 1.f : kLrelu;
        if (rr < nvalid) {
            P.dq[(size_t)(m0 + rr) * P.lddq + col] = d;
            const float xh = (av[e] - mu) * rs;
            s0 += d;
            s1 = __builtin_fmaf(d, xh, s1);
        }
    }
    cs[wave][0][lane] = s0;
    cs[wave][1][lane] = s1;
    __syncthreads();
    if (tid < 64) {   // column (tid & 31), quantity (tid >> 5): the 8 waves' shares, both row halves, in a fixed order
        const int q = tid >> 5, c = tid & 31;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += cs[w][q][c] + cs[w][q][32 + c];
        P.part[(size_t)tm * 2 * kNB + (size_t)q * kNB + n0 + c] = t;
    }
}

// The generator's pose heads (networks.py:98-100): out = y Wh^T + bh, (B x 512)(512 x nh <= 32) — one 32-row tile per
// workgroup, eight waves split K = 512 (a lane: 32 k of one y row and of one head's weight row), partial tiles through LDS in
// wave order, wave w stores the accumulator registers 2 w, 2 w + 1.  Until round 5: a split-K GEMM on gemm_f32_kernel (64
// workgroups, four slabs) + reduce_bias_kernel.
struct HeadsFwd {
    const float* y; int ldy;    // (B, 512)
    const float* wh;            // (nh, 512)
    const float* bh;            // (nh)
    int nh, B;
    float* out; int ldo;        // (B, nh)
};
__global__ __launch_bounds__(512) void heads_stage_kernel(const HeadsFwd P) {
    __shared__ float red[8][16][64];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = (int)blockIdx.x * 32;
    const int kb = 64 * wave + 32 * lh;
    const int arow = min(m0 + li, P.B - 1), brow = min(li, P.nh - 1);
    f32x4 a4[8], b4[8];
    {
        const f32x4_u* ap = reinterpret_cast<const f32x4_u*>(P.y + (size_t)arow * P.ldy + kb);
        const f32x4_u* bp = reinterpret_cast<const f32x4_u*>(P.wh + (size_t)brow * kKB + kb);
#pragma unroll
        for (int q = 0; q < 8; ++q) a4[q] = ap[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) b4[q] = bp[q];
    }
    const float bias = P.bh[brow];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s >> 2][s & 3], b4[s >> 2][s & 3], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (li >= P.nh) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * wave + e;
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float d = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) d += red[w][r][lane];   // wave (= k) order
        if (row < P.B) P.out[(size_t)row * P.ldo + li] = d + bias;
    }
}

// A narrow NN product, C (M x N <= 32) (+)= A (M x K) B (K x N), K a multiple of 64 up to 1024: the WGAN-GP step's pose-column
// adjoint (1024 x 27 x 768; on gemm_f32_kernel: 16 workgroups walking 24 k-tiles, 27.6 us).  One 32-row tile per workgroup,
// eight waves split K, wave w stores the accumulator registers 2 w, 2 w + 1.
struct NarrowNN {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int M, N, K, accumulate;
};
__global__ __launch_bounds__(512) void narrow_nn_kernel(const NarrowNN P) {
    __shared__ float red[8][16][64];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = (int)blockIdx.x * 32;
    const int kl = P.K / 16;                // k per lane: K / 8 waves / 2 halves (a multiple of 4, <= 64)
    const int kb = 2 * kl * wave + kl * lh;
    const int arow = min(m0 + li, P.M - 1), col = min(li, P.N - 1);
    constexpr int KLMAX = 64;
    f32x4 a4[KLMAX / 4];
    float b[KLMAX];
    {
        const f32x4_u* ap = reinterpret_cast<const f32x4_u*>(P.A + (size_t)arow * P.lda + kb);
        const float* bp = P.B + (size_t)kb * P.ldb + col;
#pragma unroll
        for (int q = 0; q < KLMAX / 4; ++q) a4[q] = 4 * q < kl ? ap[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KLMAX; ++s) b[s] = s < kl ? bp[(size_t)s * P.ldb] : 0.f;
    }
    float old[2] = {0.f, 0.f};
    if (P.accumulate) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * wave + e;
            const int row = min(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, P.M - 1);
            old[e] = P.C[(size_t)row * P.ldc + col];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (two widths in flight: waited for at once, tools/vmcnt_audit.py)
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KLMAX; ++s)
        if (s < kl) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s >> 2][s & 3], b[s], acc, 0, 0, 0);   // (uniform)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (li >= P.N) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * wave + e;
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float d = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) d += red[w][r][lane];   // wave (= k) order
        if (row < P.M) P.C[(size_t)row * P.ldc + li] = d + old[e];
    }
}

}  // namespace fc2

// One problem with GemmProb::bn_part set (see there): K = 256, no split-K.
int fc2_stage_launch(const GemmProb& p, hipStream_t st) {
    if (p.K != fc2::kK || p.M <= 0 || p.N <= 0 || (p.bn_train && p.bn_nrb < 1) || !p.A || !p.B || !p.C || !p.bn_part || !p.bn_gamma || !p.bn_beta || !p.bn_mean_out ||
        !p.bn_rstd_out || !p.c_bias || !p.c_part || p.splits != 0 || p.accumulate || p.lda < p.K || p.ldb < p.K ||
        (!p.bn_train && (!p.bn_run_mean || !p.bn_run_var)))
        return DGG_ERR_INVALID;
    const int tm = (p.M + 31) / 32, tn = (p.N + 31) / 32;
    // (profiles/r05_experiments.txt 21: two column tiles per workgroup win at B = 1024, one at B <= 512)
    if (tm * tn > 256) hipLaunchKernelGGL(fc2::fc2_stage_kernel<2>, dim3(tm * ((tn + 1) / 2)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(fc2::fc2_stage_kernel<1>, dim3(tm * tn), dim3(256), 0, st, p);
    return check_launch("fc2_stage");
}

}  // namespace dgg

namespace dgg {
// dq = (da2 W2) * lrelu'(h) and the partial sums of bn1's backward per 32-row block: part [row block][2][256] (what
// act_bwd_kernel emitted).  K = 512, N = 256, one group of B rows.
int fc2_bwd_stage_launch(const float* da2, int ldda, const float* w2, int ldw, const float* h, int ldh, const float* a1,
                         int lda1, const float* mean, const float* rstd, int B, float* dq, int lddq, float* part,
                         hipStream_t st) {
    if (!da2 || !w2 || !h || !a1 || !mean || !rstd || !dq || !part || B < 1 || ldda < fc2::kKB || ldw < fc2::kNB) return DGG_ERR_INVALID;
    fc2::Fc2Bwd P{da2, ldda, w2, ldw, h, ldh, a1, lda1, mean, rstd, B, dq, lddq, part};
    const int tiles = ((B + 31) / 32) * (fc2::kNB / 32);
    hipLaunchKernelGGL(fc2::fc2_bwd_stage_kernel, dim3(tiles), dim3(512), 0, st, P);
    return check_launch("fc2_bwd_stage");
}
// C (M, N <= 32) (+)= A (M, K) B (K, N): K a multiple of 64, 64 <= K <= 1024
int narrow_nn_launch(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K, int accumulate,
                     hipStream_t st) {
    if (!A || !Bm || !C || M < 1 || N < 1 || N > 32 || K < 64 || K > 1024 || K % 64 || lda < K) return DGG_ERR_INVALID;
    fc2::NarrowNN P{A, lda, Bm, ldb, C, ldc, M, N, K, accumulate};
    hipLaunchKernelGGL(fc2::narrow_nn_kernel, dim3((M + 31) / 32), dim3(512), 0, st, P);
    return check_launch("narrow_nn");
}
// out (B, nh) = y (B, 512) wh (nh, 512)^T + bh, nh <= 32
int heads_stage_launch(const float* y, int ldy, int B, const float* wh, const float* bh, int nh, float* out, int ldo,
                       hipStream_t st) {
    if (!y || !wh || !bh || !out || B < 1 || nh < 1 || nh > 32 || ldy < fc2::kKB) return DGG_ERR_INVALID;
    fc2::HeadsFwd P{y, ldy, wh, bh, nh, B, out, ldo};
    hipLaunchKernelGGL(fc2::heads_stage_kernel, dim3((B + 31) / 32), dim3(512), 0, st, P);
    return check_launch("heads_stage");
}
}  // namespace dgg
