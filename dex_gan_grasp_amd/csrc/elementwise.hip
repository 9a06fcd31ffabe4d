// Everything around the GEMMs of the ResBlock stacks: BatchNorm1d statistics / apply / backward
// (networks.py:33,36,71,117,289), LeakyReLU(0.2) (networks.py:41), residual add, dropout
// (networks.py:294), the small heads (networks.py:75-77,121,293), losses
// (dexgangrasp.py:142-147,213-232,272-307; losses.py:61-76), rot_matrix_from_ortho6d
// (utils/utils.py:380-394) and Adam (dexgangrasp.py:97-112).
//
// These are HBM/L2-bound column kernels.  A 256-thread workgroup owns 64 feature columns x one block of
// kRowBlock (32) rows: lane = column (a wave reads 64 consecutive floats of a row: coalesced 256 B),
// the 4 waves take 8 rows each, keep their samples in registers and produce exact two-pass partial
// statistics (mean, M2) / partial sums that are merged across the 4 waves through LDS in a fixed order
// and across row blocks by tiny finalize kernels (Chan's formula) — deterministic, no atomics.
// Grids are (ceil(N/64), row blocks): 384 workgroups for a 1024 x 768 activation.
// A batch may consist of `groups` independent BatchNorm batches of rows_per_group rows
// (the discriminator's real/fake pair, dexgangrasp.py:788-789).
#include <string.h>
#include <type_traits>
#include "dgg_internal.h"

namespace dgg {

constexpr int RB = kRowBlock;
constexpr int SUB = 4;          // waves per workgroup, each takes RB/SUB rows of the row block
constexpr int RS = RB / SUB;    // rows per wave
constexpr int kNhMax = 32;      // heads of a network at most

// These kernels are latency-bound (a workgroup touches 32 rows x 64 columns): every wave loads ALL its RS rows before
// it uses any — a `for (r < n)` loop with a data-dependent trip count compiles to one load, one s_waitcnt vmcnt(0) per
// row, i.e. RS serial memory round trips.  Rows past the wave's last one re-read that last row (clamped index: always
// a valid address) and are masked out of stores and sums, which keep their row order (same bits as the plain loop).
__device__ __forceinline__ int row_clamp(int r, int n) { return r < n ? r : n - 1; }

// The (mean, rstd) of every row of a wave — from the fused finalize's LDS copy or from the finalized arrays — fetched
// BEFORE the per-row arithmetic and stores: a `fused ? lds : global` select inside the row loop is a global load behind
// the previous row's store, and the compiler waits for that store, once per row.
template <typename Res>
__device__ __forceinline__ void row_stats(bool fused, const Res& res, const float* __restrict__ mean, const float* __restrict__ rstd,
                                          int stride, int c, int lane, int r0, int n, int rpg, float (&mu)[RS], float (&rs)[RS]) {
    if (fused) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int g = (r0 + row_clamp(r, n > 0 ? n : 1)) / rpg;
            mu[r] = res[g][0][lane];
            rs[r] = res[g][1][lane];
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int g = (r0 + row_clamp(r, n > 0 ? n : 1)) / rpg;
            mu[r] = mean[g * stride + c];
            rs[r] = rstd[g * stride + c];
        }
    }
}

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : kLrelu * x; }
__device__ __forceinline__ float lrelu_grad(float y) { return y > 0.f ? 1.f : kLrelu; }

__device__ __forceinline__ const float* src_col(const ColSrcs& S, int c, int& ld) {
    // column c of the virtual concat -> pointer to element (row 0, c)
    int base = 0;
    const float* p = nullptr;
    ld = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < S.nsrc) {
            if (c >= base && c < base + S.s[i].n) {
                p = S.s[i].p + (c - base);
                ld = S.s[i].ld;
            }
            base += S.s[i].n;
        }
    }
    return p;
}

__device__ __forceinline__ void partial_stats(const float (&v)[RB], int n, float& mean, float& m2) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) s += (r < n) ? v[r] : 0.f;
    mean = s / (float)n;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const float d = v[r] - mean;
        q += (r < n) ? d * d : 0.f;
    }
    m2 = q;
}

// merge the per-wave (mean, M2) of RS-row sub-blocks into the row block's partial (thread sub == 0 writes)
__device__ __forceinline__ void merge_sub_stats(float mean, float m2, int nsub, int lane, int sub, float (*sh)[2][64],
                                                int n, float& out_mean, float& out_m2) {
    sh[sub][0][lane] = mean;
    sh[sub][1][lane] = m2;
    __syncthreads();
    float cnt = 0.f, mu = 0.f, q = 0.f;
    if (sub == 0) {
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            const int ns = min(RS, max(0, n - s * RS));
            if (ns > 0) {
                const float nb = (float)ns, tot = cnt + nb;
                const float d = sh[s][0][lane] - mu;
                mu += d * (nb / tot);
                q += sh[s][1][lane] + d * d * (cnt * nb / tot);
                cnt = tot;
            }
        }
    }
    (void)nsub;
    out_mean = mu;
    out_m2 = q;
}

__device__ __forceinline__ void sub_stats(const float (&v)[RS], int n, float& mean, float& m2) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RS; ++r) s += (r < n) ? v[r] : 0.f;
    mean = n > 0 ? s / (float)n : 0.f;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const float d = v[r] - mean;
        q += (r < n) ? d * d : 0.f;
    }
    m2 = q;
}

// grid: (ceil(ncols/64), groups*nrb)
__global__ __launch_bounds__(256) void ibn_stats_kernel(ColSrcs S, int rpg, int nrb, float* __restrict__ part) {
    __shared__ float sh[SUB][2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool cok = c < S.total;
    const int g = blockIdx.y / nrb, rb = blockIdx.y % nrb;
    const int r0 = rb * RB;
    const int n = min(RB, rpg - r0);
    const int ns = min(RS, max(0, n - sub * RS));
    int ld = 0;
    const float* p = src_col(S, cok ? c : S.total - 1, ld);
    float v[RS];
    // (unconditional loads on clamped indices, all in flight; the select comes afterwards)
#pragma unroll
    for (int r = 0; r < RS; ++r) v[r] = p[(size_t)(g * rpg + r0 + (ns > 0 ? sub * RS + row_clamp(r, ns) : 0)) * ld];
#pragma unroll
    for (int r = 0; r < RS; ++r) v[r] = (cok && r < ns) ? v[r] : 0.f;
    float mean, m2, om, oq;
    sub_stats(v, ns, mean, m2);
    merge_sub_stats(mean, m2, ns, lane, sub, sh, n, om, oq);
    if (sub == 0 && cok) {
        float* o = part + (size_t)blockIdx.y * 2 * S.total;
        o[c] = om;
        o[S.total + c] = oq;
    }
}

// BatchNorm statistics are finalized inside the kernels that consume them: every workgroup merges the row-block
// partials of ITS 64 columns — the 4 waves merge disjoint ranges of row blocks (Chan), wave 0 merges those in order;
// same order, same arithmetic, hence the same bits in every workgroup; groups in order, so the running statistics
// see the reference's call order —
// leaves mean / rstd of each group in `res`, and the workgroups with `writer` set also store them for the backward
// and advance the running statistics.  Must be reached by all 256 threads.  c: this lane's column, clamped into
// [0, ncols); cok: lane owns a real BN column.
struct BnFin {
    const float* part;   // NULL: statistics were finalized by an earlier launch (mean / rstd arrays are read)
    int nrb, train;
    float *run_mean, *run_var, *mean_out, *rstd_out;
    float* varu_out;     // optional: the unbiased batch variance (what the running variance integrates), group 0
};
__device__ __forceinline__ void bn_block_stats(const BnFin& f, int rpg, int groups, int ncols, int c, bool cok,
                                               bool writer, float (*sh)[3][64], float (*res)[2][64]) {
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    if (!f.train) {
        if (sub == 0) {
            const float m = f.run_mean[c];
            const float rs = 1.0f / sqrtf(f.run_var[c] + kBnEps);
            for (int g = 0; g < groups; ++g) {
                res[g][0][lane] = m;
                res[g][1][lane] = rs;
                if (writer && cok) {
                    f.mean_out[g * ncols + c] = m;
                    f.rstd_out[g * ncols + c] = rs;
                }
            }
        }
        __syncthreads();
        return;
    }
    float rm = f.run_mean ? f.run_mean[c] : 0.f, rv = f.run_var ? f.run_var[c] : 0.f;
    const int per = (f.nrb + SUB - 1) / SUB;
    const int rb_lo = min(f.nrb, sub * per), rb_hi = min(f.nrb, rb_lo + per);
    for (int g = 0; g < groups; ++g) {
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
        for (int rb8 = rb_lo; rb8 < rb_hi; rb8 += 8) {  // 16 loads in flight, merged in row-block order
            float mb[8], qb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* p = f.part + (size_t)(g * f.nrb + min(rb8 + j, rb_hi - 1)) * 2 * ncols;
                mb[j] = p[c];
                qb[j] = p[ncols + c];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (rb8 + j < rb_hi) {
                    const float nb = (float)min(RB, rpg - (rb8 + j) * RB);
                    const float tot = cnt + nb;
                    const float d = mb[j] - mean;
                    mean += d * (nb / tot);
                    m2 += qb[j] + d * d * (cnt * nb / tot);
                    cnt = tot;
                }
            }
        }
        __syncthreads();
        sh[sub][0][lane] = cnt;
        sh[sub][1][lane] = mean;
        sh[sub][2][lane] = m2;
        __syncthreads();
        if (sub == 0) {
            cnt = 0.f; mean = 0.f; m2 = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < SUB; ++s2) {
                const float nb = sh[s2][0][lane];
                if (nb > 0.f) {
                    const float tot = cnt + nb;
                    const float d = sh[s2][1][lane] - mean;
                    mean += d * (nb / tot);
                    m2 += sh[s2][2][lane] + d * d * (cnt * nb / tot);
                    cnt = tot;
                }
            }
            const float rstd = 1.0f / sqrtf(m2 / cnt + kBnEps);
            res[g][0][lane] = mean;
            res[g][1][lane] = rstd;
            if (writer && cok) {
                f.mean_out[g * ncols + c] = mean;
                f.rstd_out[g * ncols + c] = rstd;
            }
            const float var_u = m2 / (cnt - 1.f);
            if (writer && cok && g == 0 && f.varu_out) f.varu_out[c] = var_u;
            rm = (1.f - kBnMom) * rm + kBnMom * mean;
            rv = (1.f - kBnMom) * rv + kBnMom * var_u;
        }
    }
    if (sub == 0 && writer && cok) {
        if (f.run_mean) f.run_mean[c] = rm;
        if (f.run_var) f.run_var[c] = rv;
    }
    __syncthreads();
}

#ifndef DGG_IBN_BLOCKS
#define DGG_IBN_BLOCKS 4
#endif
constexpr int kIbnBlocks = DGG_IBN_BLOCKS;  // row blocks of 32 per ibn_apply workgroup

// grid: (ceil((nprefix+ncols)/64), ceil(B/(RB*kIbnBlocks))).  gamma == NULL: store the un-affined x_hat = (x-mean)*rstd
// (the forward GEMM then folds gamma/beta into its operand staging, see GemmProb::a_scale) and emit the
// per-column affine vectors aff_scale = [1.. | gamma], aff_shift = [0.. | beta] over prefix + BN columns.
__global__ __launch_bounds__(256) void ibn_apply_kernel(ColSrcs S, int B, int rpg, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ prefix, int nprefix,
                                                        float* __restrict__ X0, int ldx,
                                                        const float* __restrict__ aff_gamma,
                                                        const float* __restrict__ aff_beta,
                                                        float* __restrict__ aff_scale, float* __restrict__ aff_shift,
                                                        BnFin fin, float* __restrict__ X0b, int ldb, int nb) {
    // X0b (optional): the first nb BatchNorm columns are stored once more at X0b[row * ldb + c] (XShare: the real pass of a D
    // phase normalises the BPS block for the generator and the fake pass as well)
    __shared__ float sh[SUB][3][64];
    __shared__ float res[2][2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int oc = blockIdx.x * 64 + lane;
    const bool ook = oc < nprefix + S.total;
    const bool bnc = ook && oc >= nprefix;
    const int c = bnc ? oc - nprefix : 0;
    const int groups = B / rpg;
    const bool fused = fin.mean_out != nullptr;
    if (aff_scale && ook && blockIdx.y == 0 && sub == 0) {
        aff_scale[oc] = oc < nprefix ? 1.f : aff_gamma[oc - nprefix];
        aff_shift[oc] = oc < nprefix ? 0.f : aff_beta[oc - nprefix];
    }
    // A workgroup walks kIbnBlocks row blocks of 32 (the statistics merge above is paid once for all of them); the
    // loads of all of them are issued before the first use
    const int nrb_total = (B + RB - 1) / RB;
    int ld = 0;
    const float* p = bnc ? src_col(S, c, ld) : nullptr;
    const float ga = (bnc && gamma) ? gamma[c] : 1.f, be = (bnc && beta) ? beta[c] : 0.f;
    float xv[kIbnBlocks][RS];
    if (ook) {
#pragma unroll
        for (int k = 0; k < kIbnBlocks; ++k) {
            const int r0 = (blockIdx.y * kIbnBlocks + k) * RB + sub * RS;
            const int n = min(RS, B - r0);
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const size_t row = (size_t)(n > 0 ? r0 + row_clamp(r, n) : 0);
                const float* src = bnc ? p + row * ld : prefix + row * nprefix + oc;   // ONE load (a value select would merge two)
                xv[k][r] = *src;
            }
        }
    }
    // (the statistics merge — two barriers, a chain of dependent partial loads — runs while the 32 loads above are in flight)
    if (fused) bn_block_stats(fin, rpg, groups, S.total, c, bnc, blockIdx.y == 0, sh, res);
#pragma unroll
    for (int k = 0; k < kIbnBlocks; ++k) {
        const int rbk = blockIdx.y * kIbnBlocks + k;
        if (rbk >= nrb_total) break;
        const int r0 = rbk * RB + sub * RS;
        const int n = min(RS, B - r0);
        if (!ook || n <= 0) continue;
        if (!bnc) {  // prefix columns pass through
#pragma unroll
            for (int r = 0; r < RS; ++r)
                if (r < n) X0[(size_t)(r0 + r) * ldx + oc] = xv[k][r];
            continue;
        }
        float mu_[RS], rs_[RS];
        row_stats(fused, res, mean, rstd, S.total, c, lane, r0, n, rpg, mu_, rs_);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int row = r0 + row_clamp(r, n);
            const float mu = mu_[r], rs = rs_[r];
            const float sc = rs * ga;
            const float shf = be - mu * sc;
            xv[k][r] = fmaf(xv[k][r], sc, shf);
            if (r < n) X0[(size_t)row * ldx + oc] = xv[k][r];
        }
        if (X0b && c < nb) {
#pragma unroll
            for (int r = 0; r < RS; ++r)
                if (r < n) X0b[(size_t)(r0 + r) * ldb + c] = xv[k][r];
        }
    }
}

// grid: (ceil(N/64), groups*nrb)
__global__ __launch_bounds__(256) void reduce_bias_kernel(const float* __restrict__ slabs, int nslab,
                                                          long long slab_stride, int ld_slab,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int ld_out, int rpg, int nrb, int N, int stat_lo,
                                                          int stat_hi, int act_lo, int act_hi,
                                                          float* __restrict__ part, int accumulate,
                                                          const float* __restrict__ slabs2, int nslab2,
                                                          long long stride2, int ld2) {
    __shared__ float sh[SUB][2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool cok = c < N;
    const int g = blockIdx.y / nrb, rb = blockIdx.y % nrb;
    const int r0 = rb * RB;
    const int n = min(RB, rpg - r0);
    const int ns = min(RS, max(0, n - sub * RS));
    const size_t row0 = (size_t)(g * rpg + r0 + sub * RS);
    const float b = (bias && cok) ? bias[c] : 0.f;
    const bool act = c >= act_lo && c < act_hi;
    float v[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) v[r] = 0.f;
    if (cok) {
        // slab order = k order: fixed summation order.  All slabs of a round are in flight together: the kernel is
        // latency-bound, not bandwidth-bound, at these sizes.
        // slabs [0, nslab) come from `slabs`, [nslab, nslab + nslab2) from `slabs2` (same stride and pitch): a
        // partial product computed once and shared by two reduces (D real / fake pass) needs no copy
        const int ntot = nslab + nslab2;
        // (the second stack may have a geometry of its own: stride2 / ld2)
        auto slab = [&](int i, int& ld) -> const float* {
            ld = i < nslab ? ld_slab : ld2;
            return i < nslab ? slabs + (size_t)i * slab_stride + row0 * ld_slab + c
                             : slabs2 + (size_t)(i - nslab) * stride2 + row0 * ld2 + c;
        };
        // (the last group may hold 1-3 slabs: the same batch of loads, not one slab — one round trip per ROW — at a time;
        //  round 4: five or six slabs are the common case, [x_hat | previous block] pieces of a ResBlock input)
        // (round 5: up to eight slabs per round — the [x_hat | previous block] inputs arrive as five to seven slabs, and a
        //  second round of four was a second dependent memory round trip; four or fewer slabs keep the round of four)
        auto round = [&](auto nc, int s) {
            constexpr int NB = decltype(nc)::value;
            const int nu = min(NB, ntot - s);   // uniform
            float t[NB][RS];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                int ld;
                const float* p = slab(s + min(u, nu - 1), ld);
#pragma unroll
                for (int r = 0; r < RS; ++r) t[u][r] = (r < ns && u < nu) ? p[(size_t)r * ld] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u)
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    if (u < nu) v[r] += t[u][r];
        };
        if (ntot <= 4) {
            round(std::integral_constant<int, 4>{}, 0);
        } else {
            for (int s = 0; s < ntot; s += 8) round(std::integral_constant<int, 8>{}, s);
        }
        // (old values — accumulate — all at once, then the sums, then the stores: a load or a store per row behind the
        //  previous row's store makes the compiler wait for that store, 8 times in a row)
        float oldv[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) oldv[r] = 0.f;
        if (accumulate) {
#pragma unroll
            for (int r = 0; r < RS; ++r)
                if (r < ns) oldv[r] = out[(row0 + r) * ld_out + c];
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (r < ns) {
                v[r] += b;
                if (accumulate) v[r] += oldv[r];
            }
        }
        float ov[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) ov[r] = act ? lrelu(v[r]) : v[r];
#pragma unroll
        for (int r = 0; r < RS; ++r)
            if (r < ns) out[(row0 + r) * ld_out + c] = ov[r];
    }
    const bool any_stat = stat_hi > stat_lo;  // uniform
    if (any_stat) {
        float mean, m2, om, oq;
        sub_stats(v, ns, mean, m2);
        merge_sub_stats(mean, m2, ns, lane, sub, sh, n, om, oq);
        if (sub == 0 && c >= stat_lo && c < stat_hi) {
            const int nst = stat_hi - stat_lo;
            float* o = part + (size_t)blockIdx.y * 2 * nst;
            o[c - stat_lo] = om;
            o[nst + c - stat_lo] = oq;
        }
    }
}

// grid: (ceil(N/64), ceil(B/RB))
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ a, int lda, int B, int rpg, int N,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ h, int ldh, BnFin fin) {
    __shared__ float sh[SUB][3][64];
    __shared__ float res[2][2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64 + lane;
    const bool cok = c0 < N;
    const int c = cok ? c0 : N - 1;
    const bool fused = fin.mean_out != nullptr;
    const float ga = gamma[c], be = beta[c];   // (before the merge: see res_out_kernel)
    if (fused) bn_block_stats(fin, rpg, B / rpg, N, c, cok, blockIdx.y == 0, sh, res);
    if (!cok) return;
    const int r0 = blockIdx.y * RB + sub * RS;
    const int n = min(RS, B - r0);
    if (n <= 0) return;
    float av[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) av[r] = a[(size_t)(r0 + row_clamp(r, n)) * lda + c];
    float mu_[RS], rs_[RS];
    row_stats(fused, res, mean, rstd, N, c, lane, r0, n, rpg, mu_, rs_);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        if (r < n) {
            const int row = r0 + r;
            const float sc = rs_[r] * ga;
            const float shf = be - mu_[r] * sc;
            h[(size_t)row * ldh + c] = lrelu(fmaf(av[r], sc, shf));
        }
    }
}

// Philox4x32-10 (Salmon et al. 2011), one call -> 4 x 32 random bits for counter (idx, seed).
__device__ __forceinline__ uint4 philox4x32_10(uint64_t ctr, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float philox_keep(uint64_t seed, uint64_t elem) {
    const uint4 r = philox4x32_10(elem >> 2, seed);
    const uint32_t w = (elem & 3) == 0 ? r.x : (elem & 3) == 1 ? r.y : (elem & 3) == 2 ? r.z : r.w;
    // keep with probability 1-p: uniform in [0,1) >= p
    return ((float)(w >> 8) * (1.0f / 16777216.0f)) >= kDropP ? 1.f : 0.f;
}

// grid: (ceil(N/64), ceil(B/RB))
__global__ __launch_bounds__(256) void res_out_kernel(const float* __restrict__ s, int lds,
                                                      const float* __restrict__ a2, int lda2, int B, int rpg, int N,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int drop_mode, const float* __restrict__ keep_in, uint64_t seed,
                                                      uint64_t seed_off, uint8_t* __restrict__ keep_out,
                                                      float* __restrict__ y, int ldy, BnFin fin,
                                                      const float* __restrict__ a2_bias) {
    __shared__ float sh[SUB][3][64];
    __shared__ float res[2][2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64 + lane;
    const bool cok = c0 < N;
    const int c = cok ? c0 : N - 1;
    const bool fused = fin.mean_out != nullptr;
    const int r0 = blockIdx.y * RB + sub * RS;
    const int n = min(RS, B - r0);
    // (the two operand tiles are requested before the statistics merge — two barriers and a chain of dependent partial
    //  loads — so that one memory latency covers both; clamped indices, every lane)
    float sv[RS], av[RS], kv[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const size_t row = (size_t)(n > 0 ? r0 + row_clamp(r, n) : 0);
        sv[r] = s[row * lds + c];
        av[r] = a2[row * lda2 + c];
        kv[r] = 1.f;
    }
    // (... and so is everything else the rows need: behind the merge each of these was a dependent round trip of its own)
    const float ga = gamma[c], be = beta[c];
    const float ab = a2_bias ? a2_bias[c] : 0.f;  // a2 arrives without fc2's bias (inference path: plain GEMM output)
    // drop_mode 3: `seed` carries a device pointer to the seed (graph replay: the seed lives in device memory)
    const uint64_t sd = drop_mode == 3 ? *reinterpret_cast<const uint64_t*>(seed) : seed;
    if (drop_mode == 1) {   // (one uniform branch around the loop, not a load-or-constant per row)
#pragma unroll
        for (int r = 0; r < RS; ++r) kv[r] = keep_in[(size_t)(n > 0 ? r0 + row_clamp(r, n) : 0) * N + c];
    }
    if (fused) bn_block_stats(fin, rpg, B / rpg, N, c, cok, blockIdx.y == 0, sh, res);
    if (!cok || n <= 0) return;
    const float inv_keep = 1.0f / (1.0f - kDropP);
#pragma unroll
    for (int r = 0; r < RS; ++r) av[r] += ab;
    float mu_[RS], rs_[RS];
    row_stats(fused, res, mean, rstd, N, c, lane, r0, n, rpg, mu_, rs_);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        if (r < n) {
            const int row = r0 + r;
            const float sc = rs_[r] * ga;
            const float shf = be - mu_[r] * sc;
            float v = lrelu(sv[r] + fmaf(av[r], sc, shf));
            if (drop_mode) {
                const size_t e = (size_t)row * N + c;
                const float k = drop_mode == 1 ? kv[r] : philox_keep(sd, seed_off + e);
                keep_out[e] = (uint8_t)(k != 0.f);
                v = v * k * inv_keep;
            }
            y[(size_t)row * ldy + c] = v;
        }
    }
}

// one wave per row; K <= 512*? handled in chunks of 64 lanes
template <int NHMAX>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ y, int ldy, int B, int K,
                                                        const float* __restrict__ wh, const float* __restrict__ bh,
                                                        int nh, int sigmoid, float* __restrict__ out, int ldo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    float acc[NHMAX], bj[NHMAX];
#pragma unroll
    for (int j = 0; j < NHMAX; ++j) {
        acc[j] = 0.f;
        bj[j] = j < nh ? bh[j] : 0.f;   // (requested with the operands, not behind the reduction)
    }
    for (int k = lane; k < K; k += 64) {
        const float v = y[(size_t)row * ldy + k];
#pragma unroll
        for (int j = 0; j < NHMAX; ++j)
            if (j < nh) acc[j] = fmaf(v, wh[(size_t)j * K + k], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NHMAX; ++j) {
        if (j < nh) {
            float v = acc[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) {
                v += bj[j];
                if (sigmoid) v = 1.0f / (1.0f + expf(-v));
                out[(size_t)row * ldo + j] = v;
            }
        }
    }
}

// grid: (ceil(K/256), nrb): dy[m][k] = sum_j dout[m][j] wh[j][k]; partial dWh[j][k], dbh[j]
// ---------------------------------------------------------------- ResBlock backward stages
// grid: (ceil(N/256), groups*nrb)
// Head mode (hd_dout != NULL; single-score head of D / E, networks.py:121,293): the block output's gradient is the
// outer product dy[row][c] = d_score[row] * w_head[c], formed here instead of being materialised by a heads kernel;
// with nq == 5 two more partial sums give the head's parameter gradients, sum_row d_score*y (q 3) and sum_row
// d_score (q 4, the same in every column).
__global__ __launch_bounds__(256) void res_out_bwd_kernel(const float* __restrict__ dy, int lddy,
                                                          const float* __restrict__ y, int ldy,
                                                          const uint8_t* __restrict__ keep,
                                                          const float* __restrict__ s, int lds,
                                                          const float* __restrict__ a2, int lda2,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int rpg, int nrb, int N,
                                                          float* __restrict__ dpre, float* __restrict__ da3,
                                                          int ldda3, float* __restrict__ part,
                                                          const float* __restrict__ hd_dout,
                                                          const float* __restrict__ hd_w, int nq, int nslab,
                                                          long long slab_stride, SumOut SO, int hd_n) {
    // nslab > 1: `dy` is a stack of split-K slabs of the next block's dgrad GEMM, summed here in slab (= k) order
    // hd_n > 1 (round 6: the generator's pose heads, networks.py:98-100): dy[row][c] = sum_j d_out[row][j] * w[j][c] is formed
    // here — hd_n fused multiply-adds per element, the workgroup's 32 rows of d_out staged through LDS — instead of by a
    // K = 21 GEMM launch of its own in front of this one (rounds 1-5)
    __shared__ float sh[SUB][5][64];
    __shared__ float sh_d[RB][kNhMax];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64 + lane;
    const bool cok = c0 < N;
    const int c = cok ? c0 : N - 1;
    const int g = blockIdx.y / nrb, rb = blockIdx.y % nrb;
    const int r0 = rb * RB + sub * RS;
    const int n = min(RS, max(0, rpg - r0));
    const float mu = mean[g * N + c], rs = rstd[g * N + c];
    const float inv_keep = 1.0f / (1.0f - kDropP);
    const bool multi = hd_dout && hd_n > 1;
    const float wv = (hd_dout && !multi) ? hd_w[c] : 0.f;
    float wm[kNhMax];
    if (multi) {   // (workgroup-uniform: every thread reaches the barrier)
#pragma unroll
        for (int j = 0; j < kNhMax; ++j) wm[j] = j < hd_n ? hd_w[(size_t)j * N + c] : 0.f;
        const int rows = min(RB, max(0, rpg - rb * RB));
        for (int i = threadIdx.x; i < rows * hd_n; i += 256) {
            const int rr = i / hd_n, j = i - rr * hd_n;
            sh_d[rr][j] = hd_dout[(size_t)(g * rpg + rb * RB + rr) * hd_n + j];
        }
        __syncthreads();
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    if (n > 0) {
        float yv_[RS], dv_[RS], av_[RS], sv_[RS], ov_[RS];
        unsigned kp_[RS];   // (a register each: bytes packed into one register made every row's loads wait for the previous row's)
        // (one load per value, optional operands under ONE uniform branch around their loop: `p ? load_a : load_b` per
        //  element merges two registers at a join, and the compiler waits for both loads there — once per row)
        const float* dbase = multi ? y + c : hd_dout ? hd_dout : dy + c;   // (multi: a valid address; the value is replaced below)
        const size_t dstr = multi ? (size_t)ldy : hd_dout ? (size_t)1 : (size_t)lddy;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const size_t row = (size_t)(g * rpg + r0 + row_clamp(r, n));
            yv_[r] = y[row * ldy + c];
            dv_[r] = dbase[row * dstr];
            av_[r] = a2[row * lda2 + c];
            sv_[r] = s[row * lds + c];
            kp_[r] = 1u;
            ov_[r] = 0.f;
        }
        // every operand is requested before anything is waited for: the other pass' gradient, the further slabs of a split-K
        // dy (three per round: a slab per round was a dependent memory round trip each) and the keep bytes
        auto slab_loads = [&](int q0, float (&t)[3][RS]) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = min(q0 + u, nslab - 1);
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    t[u][r] = dy[(size_t)q * slab_stride + (size_t)(g * rpg + r0 + row_clamp(r, n)) * lddy + c];
            }
        };
        auto slab_adds = [&](int q0, const float (&t)[3][RS]) {   // slab (= k) order
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (q0 + u < nslab) {   // uniform
#pragma unroll
                    for (int r = 0; r < RS; ++r) dv_[r] += t[u][r];
                }
        };
        float t0[3][RS];
        if (nslab > 1) slab_loads(1, t0);
        if (SO.other) {
#pragma unroll
            for (int r = 0; r < RS; ++r) ov_[r] = SO.other[(size_t)(g * rpg + r0 + row_clamp(r, n)) * SO.ld + c];
        }
        if (keep) {
#pragma unroll
            for (int r = 0; r < RS; ++r) kp_[r] = (unsigned)keep[(size_t)(g * rpg + r0 + row_clamp(r, n)) * N + c];
            // byte loads among the dword loads above: everything in flight is waited for at once, no counted wait over the
            // two widths (tools/vmcnt_audit.py)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (nslab > 1) slab_adds(1, t0);
        for (int q0 = 4; q0 < nslab; q0 += 3) {
            float t[3][RS];
            slab_loads(q0, t);
            slab_adds(q0, t);
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (r < n) {
                const size_t row = (size_t)(g * rpg + r0 + r);
                const float yv = yv_[r];
                float d;
                if (multi) {
                    d = 0.f;
#pragma unroll
                    for (int j = 0; j < kNhMax; ++j)
                        if (j < hd_n) d = fmaf(sh_d[sub * RS + r][j], wm[j], d);
                } else if (hd_dout) {
                    const float ds = dv_[r];
                    d = ds * wv;
                    s3 = fmaf(ds, yv, s3);
                    s4 += ds;
                } else {
                    d = dv_[r];
                }
                if (keep) d = kp_[r] ? d * inv_keep : 0.f;
                d *= lrelu_grad(yv);
                if (cok) dpre[row * N + c] = d;
                const float xh = (av_[r] - mu) * rs;
                const float d3 = d * lrelu_grad(sv_[r]);
                if (cok) da3[row * ldda3 + c] = d3;
                if (SO.sum && cok) SO.sum[row * SO.ld + c] = d3 + ov_[r];
                s0 += d;
                s1 = fmaf(d, xh, s1);
                s2 += d3;
            }
        }
    }
    sh[sub][0][lane] = s0;
    sh[sub][1][lane] = s1;
    sh[sub][2][lane] = s2;
    sh[sub][3][lane] = s3;
    sh[sub][4][lane] = s4;
    __syncthreads();
    if (sub == 0 && cok) {
        float* o = part + (size_t)blockIdx.y * nq * N;
        for (int q = 0; q < nq; ++q)
            o[q * N + c] = ((sh[0][q][lane] + sh[1][q][lane]) + sh[2][q][lane]) + sh[3][q][lane];
    }
}

// grid: (ceil(N/256), groups*nrb)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, int lddz,
                                                           const float* __restrict__ a, int lda,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, int train, int rpg,
                                                           int nrb, int N, float* __restrict__ dx, int lddx,
                                                           float* __restrict__ part, ColFin fin, SumOut SO) {
    constexpr int kFinQ = 5;   // quantities a ColFin finalizes at most (fin.nq)
    __shared__ float sh[SUB][1][64];
    __shared__ float shq[SUB][kFinQ][64];
    __shared__ float tq[2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64 + lane;
    const bool cok = c0 < N;
    const int c = cok ? c0 : N - 1;
    const int g = blockIdx.y / nrb, rb = blockIdx.y % nrb;
    const int r0 = rb * RB + sub * RS;
    const int n = min(RS, max(0, rpg - r0));
    const float mu = mean[g * N + c], rs = rstd[g * N + c];
    const float k = gamma[c] * rs;
    const float inv_n = 1.0f / (float)rpg;
    float m0 = 0.f, m1 = 0.f;
    // the rows this thread transforms: requested before the finalize prologue below, one memory latency for both
    float dv_[RS], av_[RS], ov_[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const size_t row = (size_t)(g * rpg + (n > 0 ? r0 + row_clamp(r, n) : 0));
        dv_[r] = dz[row * lddz + c];
        av_[r] = a[row * lda + c];
        ov_[r] = 0.f;
    }
    if (SO.other) {
#pragma unroll
        for (int r = 0; r < RS; ++r) ov_[r] = SO.other[(size_t)(g * rpg + (n > 0 ? r0 + row_clamp(r, n) : 0)) * SO.ld + c];
    }
    if (fin.part) {  // one group: column totals from the row-block partials (see ColFin)
        const bool writer = blockIdx.y == 0;
        if (writer && blockIdx.x == 0) {
            for (int z = threadIdx.x; z < fin.nz0; z += 256) fin.zero0[z] = 0.f;
            for (int z = threadIdx.x; z < fin.nz1; z += 256) fin.zero1[z] = 0.f;
        }
        const int per = (nrb + SUB - 1) / SUB;
        const int rb_lo = min(nrb, sub * per), rb_hi = min(nrb, rb_lo + per);
        const int nqq = writer ? fin.nq : (train ? 2 : 0);  // workgroup-uniform
        // all quantities' partials in ONE round of loads and one pair of barriers (they were a round and a pair each); per
        // quantity the same sums in the same order.  The number of quantities is a compile-time constant of the body that runs
        // (a switch over it): with `if (q < nqq)` around each quantity's loads the compiler waited for one quantity's loads
        // before it requested the next — the "one round" was two to five dependent ones.
        auto fin_round = [&](auto nqc) {
            constexpr int NQ = decltype(nqc)::value;
            float sg[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) sg[q] = 0.f;
            for (int b8 = rb_lo; b8 < rb_hi; b8 += 8) {
                float t[NQ][8];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[q][j] = fin.part[((size_t)min(b8 + j, rb_hi - 1) * fin.nq + q) * N + c];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (b8 + j < rb_hi) sg[q] += t[q][j];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) shq[sub][q][lane] = sg[q];
        };
        switch (nqq) {   // workgroup-uniform
        case 1: fin_round(std::integral_constant<int, 1>{}); break;
        case 2: fin_round(std::integral_constant<int, 2>{}); break;
        case 3: fin_round(std::integral_constant<int, 3>{}); break;
        case 4: fin_round(std::integral_constant<int, 4>{}); break;
        case 5: fin_round(std::integral_constant<int, 5>{}); break;
        default: break;
        }
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int q = 0; q < kFinQ; ++q) {
                if (q < nqq) {
                    const float tot = ((shq[0][q][lane] + shq[1][q][lane]) + shq[2][q][lane]) + shq[3][q][lane];
                    if (q < 2) tq[q][lane] = tot;
                    if (writer && cok) {
                        float* d = q < 4 ? fin.d[q] : (c == 0 ? fin.d[4] : nullptr);  // q 4: one scalar
                        if (d) d[c] = fin.accumulate ? d[c] + tot : tot;
                        if (q == 2 && fin.copy2) fin.copy2[c] = tot;
                    }
                }
            }
        }
        __syncthreads();
        if (train) {
            m0 = tq[0][lane] * inv_n;
            m1 = tq[1][lane] * inv_n;
        }
    } else if (train) {
        m0 = sums[(size_t)(g * 2 + 0) * N + c] * inv_n;
        m1 = sums[(size_t)(g * 2 + 1) * N + c] * inv_n;
    }
    float sb = 0.f;
    if (n > 0) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (r < n) {
                const size_t row = (size_t)(g * rpg + r0 + r);
                const float xh = (av_[r] - mu) * rs;
                const float v = k * (dv_[r] - m0 - xh * m1);
                if (cok) dx[row * lddx + c] = v;
                if (SO.sum && cok) SO.sum[row * SO.ld + c] = v + ov_[r];
                sb += v;
            }
        }
    }
    sh[sub][0][lane] = sb;
    __syncthreads();
    if (sub == 0 && cok)
        part[(size_t)blockIdx.y * N + c] = ((sh[0][0][lane] + sh[1][0][lane]) + sh[2][0][lane]) + sh[3][0][lane];
}

// grid: ceil(nq*ncols/64): 64 (quantity,column) pairs per workgroup, the 4 waves sum disjoint row-block ranges.
// part[((g*nrb + rb)*nq + q)*ld + c_lo + c], c in [0, ncols)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int groups, int nrb,
                                                              int nq, int ncols, int ld, int c_lo,
                                                              float* __restrict__ sums_out, float* __restrict__ d0,
                                                              float* __restrict__ d1, float* __restrict__ d2,
                                                              int accumulate, float* __restrict__ copy0,
                                                              float* __restrict__ copy2, float* __restrict__ zero0,
                                                              int nz0, float* __restrict__ zero1, int nz1,
                                                              float* __restrict__ d3, float* __restrict__ d4,
                                                              int noacc_below, long long gstride) {
    // gstride: distance between the groups' partial blocks (0: they lie back to back, nrb * nq * ld apart)
    __shared__ float sh[SUB][64];
    if (blockIdx.x == 0) {  // ranges the caller wants cleared alongside (saves separate memset launches)
        for (int z = threadIdx.x; z < nz0; z += 256) zero0[z] = 0.f;
        for (int z = threadIdx.x; z < nz1; z += 256) zero1[z] = 0.f;
    }
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64 + lane;
    const bool ok = i0 < nq * ncols;
    const int i = ok ? i0 : nq * ncols - 1;
    const int q = i / ncols, c = i - q * ncols;
    const int per = (nrb + SUB - 1) / SUB;
    const int rb_lo = min(nrb, sub * per), rb_hi = min(nrb, rb_lo + per);
    float tot = 0.f;
    for (int g = 0; g < groups; ++g) {
        float sg = 0.f;
        for (int b8 = rb_lo; b8 < rb_hi; b8 += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                t[j] = part[(size_t)g * gstride + ((size_t)min(b8 + j, rb_hi - 1) * nq + q) * ld + c_lo + c];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (b8 + j < rb_hi) sg += t[j];
        }
        __syncthreads();
        sh[sub][lane] = sg;
        __syncthreads();
        if (sub == 0) {
            sg = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
            if (sums_out && ok) sums_out[((size_t)g * nq + q) * ncols + c] = sg;
            tot += sg;
        }
    }
    if (sub == 0 && ok) {
        float* d = q == 0 ? d0 : q == 1 ? d1 : q == 2 ? d2 : q == 3 ? d3 : (c == 0 ? d4 : nullptr);  // q 4: one scalar
        if (d) d[c] = (accumulate && c >= noacc_below) ? d[c] + tot : tot;
        if (q == 0 && copy0) copy0[c] = tot;
        if (q == 2 && copy2) copy2[c] = tot;
    }
}

// grid: (ceil(ncols/256), groups*nrb)
// nslab > 1: dX0 is a stack of split-K slabs (slab_stride apart) summed here in slab order; dsum (optional): the summed
// gradient, compact (pitch S.total), for ibn_bwd_apply
__global__ __launch_bounds__(256) void ibn_bwd_stats_kernel(ColSrcs S, const float* __restrict__ dX0, int ldx,
                                                            int col_off, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int rpg, int nrb,
                                                            float* __restrict__ part, int nslab,
                                                            long long slab_stride, float* __restrict__ dsum) {
    __shared__ float sh[SUB][2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64 + lane;
    const bool cok = c0 < S.total;
    const int c = cok ? c0 : S.total - 1;
    const int g = blockIdx.y / nrb, rb = blockIdx.y % nrb;
    const int r0 = rb * RB + sub * RS;
    const int n = min(RS, max(0, rpg - r0));
    int ld;
    const float* p = src_col(S, c, ld);
    const float mu = mean[g * S.total + c], rs = rstd[g * S.total + c];
    float s0 = 0.f, s1 = 0.f;
    if (n > 0) {
        float dv_[RS], xv_[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const size_t row = (size_t)(g * rpg + r0 + row_clamp(r, n));
            dv_[r] = dX0[row * ldx + col_off + c];
            xv_[r] = p[row * ld];
        }
        // (the pose columns' gradient arrives as ~12 split-K slabs, two ResBlocks x six K splits: seven more per round — with
        //  the loads above eight slabs in flight — instead of a dependent memory round trip per slab; slab order as before)
        for (int q0 = 1; q0 < nslab; q0 += 7) {
            float t[7][RS];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int q = min(q0 + u, nslab - 1);
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    t[u][r] = dX0[(size_t)q * slab_stride + (size_t)(g * rpg + r0 + row_clamp(r, n)) * ldx + col_off + c];
            }
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (q0 + u < nslab) {   // uniform
#pragma unroll
                    for (int r = 0; r < RS; ++r) dv_[r] += t[u][r];
                }
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (r < n) {
                const size_t row = (size_t)(g * rpg + r0 + r);
                if (dsum && cok) dsum[row * S.total + c] = dv_[r];
                const float xh = (xv_[r] - mu) * rs;
                s0 += dv_[r];
                s1 = fmaf(dv_[r], xh, s1);
            }
        }
    }
    sh[sub][0][lane] = s0;
    sh[sub][1][lane] = s1;
    __syncthreads();
    if (sub == 0 && cok) {
        float* o = part + (size_t)blockIdx.y * 2 * S.total;
        o[c] = ((sh[0][0][lane] + sh[1][0][lane]) + sh[2][0][lane]) + sh[3][0][lane];
        o[S.total + c] = ((sh[0][1][lane] + sh[1][1][lane]) + sh[2][1][lane]) + sh[3][1][lane];
    }
}

// grid: (ceil((c_hi-c_lo)/256), ceil(B/RB))
__global__ __launch_bounds__(256) void ibn_bwd_apply_kernel(ColSrcs S, const float* __restrict__ dX0, int ldx,
                                                            int col_off, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ sums, int train, int B, int rpg,
                                                            int c_lo, int c_hi, float* __restrict__ dst, int ldd,
                                                            const float* __restrict__ fin_part, int fin_nrb, int nslab,
                                                            long long slab_stride) {
    // fin_part (train, one group): the column sums come from ibn_bwd_stats' row-block partials [nrb][2][S.total],
    // summed by every workgroup for its own columns in colsum_finalize's order.  nslab > 1: dX0 = split-K slabs.
    __shared__ float shq[SUB][2][64];
    __shared__ float tq[2][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c0 = c_lo + blockIdx.x * 64 + lane;
    const bool cok = c0 < c_hi;
    const int c = cok ? c0 : c_hi - 1;
    const float inv_n = 1.0f / (float)rpg;
    // the rows this thread transforms and their statistics are requested first (clamped indices, every thread), the partials of
    // both column sums in one round behind them: one memory latency where there were four (partials of q = 0, of q = 1, rows,
    // statistics)
    const int r0 = blockIdx.y * RB + sub * RS;
    const int n = min(RS, B - r0);
    int ld;
    const float* p = src_col(S, c, ld);
    const float ga = gamma[c];
    float dv_[RS], xv_[RS], mu_[RS], rs_[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const size_t row = (size_t)(n > 0 ? r0 + row_clamp(r, n) : 0);
        const int g = (int)row / rpg;
        dv_[r] = dX0[row * ldx + col_off + c];
        xv_[r] = p[row * ld];
        mu_[r] = mean[g * S.total + c];
        rs_[r] = rstd[g * S.total + c];
    }
    if (fin_part && train) {
        const int per = (fin_nrb + SUB - 1) / SUB;
        const int rb_lo = min(fin_nrb, sub * per), rb_hi = min(fin_nrb, rb_lo + per);
        float sg[2] = {0.f, 0.f};
        for (int b8 = rb_lo; b8 < rb_hi; b8 += 8) {
            float t[2][8];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) t[q][j] = fin_part[((size_t)min(b8 + j, rb_hi - 1) * 2 + q) * S.total + c];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (b8 + j < rb_hi) sg[q] += t[q][j];
        }
        shq[sub][0][lane] = sg[0];
        shq[sub][1][lane] = sg[1];
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                tq[q][lane] = ((shq[0][q][lane] + shq[1][q][lane]) + shq[2][q][lane]) + shq[3][q][lane];
        }
        __syncthreads();
    }
    if (!cok || n <= 0) return;
    for (int q0 = 1; q0 < nslab; q0 += 7) {   // (eval mode: no statistics pass has summed the slabs; see ibn_bwd_stats_kernel)
        float t[7][RS];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int q = min(q0 + u, nslab - 1);
#pragma unroll
            for (int r = 0; r < RS; ++r)
                t[u][r] = dX0[(size_t)q * slab_stride + (size_t)(r0 + row_clamp(r, n)) * ldx + col_off + c];
        }
#pragma unroll
        for (int u = 0; u < 7; ++u)
            if (q0 + u < nslab) {   // uniform
#pragma unroll
                for (int r = 0; r < RS; ++r) dv_[r] += t[u][r];
            }
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        if (r < n) {
            const int row = r0 + r;
            const int g = row / rpg;
            const float mu = mu_[r], rs = rs_[r];
            float v;
            if (train) {
                const float m0 = (fin_part ? tq[0][lane] : sums[(size_t)(g * 2 + 0) * S.total + c]) * inv_n;
                const float m1 = (fin_part ? tq[1][lane] : sums[(size_t)(g * 2 + 1) * S.total + c]) * inv_n;
                const float xh = (xv_[r] - mu) * rs;
                v = ga * rs * (dv_[r] - m0 - xh * m1);
            } else {
                v = ga * rs * dv_[r];
            }
            dst[(size_t)row * ldd + (c - c_lo)] = v;
        }
    }
}

// ---------------------------------------------------------------- losses
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
    // deterministic: wave shuffle tree, then thread 0 adds the wave sums in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// single block; torch.nn.BCELoss(mean) with log clamped at -100, backward as torch
// (binary_cross_entropy_backward: (p - t) / max((1-p) p, 1e-12)) chained through the sigmoid.
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ p, const float* __restrict__ labels,
                                                   float target_const, int B, float weight, float loss_scale,
                                                   float* __restrict__ loss_out, float* __restrict__ d_logit,
                                                   float* __restrict__ combine, int real_slot,
                                                   unsigned* __restrict__ clear_ticket, float* __restrict__ loss_copy,
                                                   RunUpd ru) {
    __shared__ float sh[16];
    float acc = 0.f;
    if (clear_ticket && threadIdx.x == 0) clear_ticket[0] = 0u;   // see gen_loss_kernel (GenLossFin)
    // rider (round 6): the running statistics of an input BatchNorm advance once more with stored batch statistics — what
    // ibn_refresh_kernel did as a launch of its own at the head of the G phase's pass through the updated discriminator
    for (int c = threadIdx.x; c < ru.n; c += blockDim.x) {
        ru.run_mean[c] = (1.f - kBnMom) * ru.run_mean[c] + kBnMom * ru.mean[c];
        ru.run_var[c] = (1.f - kBnMom) * ru.run_var[c] + kBnMom * ru.var_u[c];
    }
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float x = p[i];
        const float t = labels ? labels[i] : target_const;
        const float l1 = fmaxf(logf(x), -100.f), l0 = fmaxf(logf(1.f - x), -100.f);
        acc += -(t * l1 + (1.f - t) * l0);
        if (d_logit) {
            const float gp = (x - t) / fmaxf((1.f - x) * x, 1e-12f) * (weight * loss_scale / (float)B);
            d_logit[i] = gp * (1.f - x) * x;
        }
    }
    const float tot = block_sum_1024(acc, sh);
    if (threadIdx.x == 0) {
        const float l = weight * (tot / (float)B);
        loss_out[0] = l;
        if (loss_copy) loss_copy[0] = l;   // (the evaluator reports its one loss under two names: a copy launch of its own until round 5)
        // combine (the D phase's fake-pass term; dexgangrasp.py:229-231): total = (real + fake) / 2 with the real pass' term
        // taken from `real_slot` (a prefetched real pass leaves it in a spare slot) — what a one-thread kernel of its own did
        // before round 4
        if (combine) {
            const float r = combine[real_slot];
            combine[DGG_L_BCE_REAL] = r;
            combine[DGG_L_TOTAL_DISC] = (r + l) / 2.f;
        }
    }
}

struct Rot {
    float x[3], y[3], z[3];
    float na, nc;      // |a| and |x cross b| before clamping
    float b[3];
};

// Rows of 6 / 9 floats at a 4-byte aligned address (pitches of 6, 9, 21 floats): the compiler merges their loads into
// dwordx4 + dwordx2 (+ dword) pieces at addresses that are not aligned to the piece and waits for them with counted
// vmcnt(N) among each other — the shape of load the round-3 wrong-operand failure of the BPS kernel went away without
// (tools/vmcnt_audit.py lists every such wait of the library).  These helpers are ONE asm statement each:
// naturally aligned dword loads and the s_waitcnt vmcnt(0) that covers them.
__device__ __forceinline__ void gld6(const float* p, float (&v)[6]) {
    asm volatile(
        "global_load_dword %0, %6, off\n\tglobal_load_dword %1, %6, off offset:4\n\tglobal_load_dword %2, %6, off offset:8\n\t"
        "global_load_dword %3, %6, off offset:12\n\tglobal_load_dword %4, %6, off offset:16\n\t"
        "global_load_dword %5, %6, off offset:20\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void gld9(const float* p, float (&v)[9]) {
    asm volatile(
        "global_load_dword %0, %9, off\n\tglobal_load_dword %1, %9, off offset:4\n\tglobal_load_dword %2, %9, off offset:8\n\t"
        "global_load_dword %3, %9, off offset:12\n\tglobal_load_dword %4, %9, off offset:16\n\t"
        "global_load_dword %5, %9, off offset:20\n\tglobal_load_dword %6, %9, off offset:24\n\t"
        "global_load_dword %7, %9, off offset:28\n\tglobal_load_dword %8, %9, off offset:32\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
        : "v"(p)
        : "memory");
}

__device__ __forceinline__ void ortho6d_of(const float (&o)[6], Rot& R) {   // the 6 values already in registers
    const float a0 = o[0], a1 = o[1], a2 = o[2];
    R.b[0] = o[3]; R.b[1] = o[4]; R.b[2] = o[5];
    R.na = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    const float da = fmaxf(R.na, 1e-8f);
    R.x[0] = a0 / da; R.x[1] = a1 / da; R.x[2] = a2 / da;
    const float c0 = R.x[1] * R.b[2] - R.x[2] * R.b[1];
    const float c1 = R.x[2] * R.b[0] - R.x[0] * R.b[2];
    const float c2 = R.x[0] * R.b[1] - R.x[1] * R.b[0];
    R.nc = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
    const float dc = fmaxf(R.nc, 1e-8f);
    R.z[0] = c0 / dc; R.z[1] = c1 / dc; R.z[2] = c2 / dc;
    R.y[0] = R.z[1] * R.x[2] - R.z[2] * R.x[1];
    R.y[1] = R.z[2] * R.x[0] - R.z[0] * R.x[2];
    R.y[2] = R.z[0] * R.x[1] - R.z[1] * R.x[0];
}
__device__ __forceinline__ void ortho6d(const float* op, Rot& R) {
    float o[6];
    gld6(op, o);
    ortho6d_of(o, R);
}

__device__ __forceinline__ void cross3(const float* u, const float* v, float* o) {
    o[0] = u[1] * v[2] - u[2] * v[1];
    o[1] = u[2] * v[0] - u[0] * v[2];
    o[2] = u[0] * v[1] - u[1] * v[0];
}

// gR row-major (3x3, R[i][0]=x_i, R[i][1]=y_i, R[i][2]=z_i) -> gradient w.r.t. the 6-vector
__device__ __forceinline__ void ortho6d_bwd(const Rot& R, const float* gR, float* go) {
    float gx[3], gy[3], gz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gx[i] = gR[3 * i]; gy[i] = gR[3 * i + 1]; gz[i] = gR[3 * i + 2]; }
    float t[3];
    // y = z x x
    cross3(R.x, gy, t);
    gz[0] += t[0]; gz[1] += t[1]; gz[2] += t[2];
    cross3(gy, R.z, t);
    gx[0] += t[0]; gx[1] += t[1]; gx[2] += t[2];
    // z = c / max(|c|, 1e-8)
    float gc[3];
    if (R.nc > 1e-8f) {
        const float dot = R.z[0] * gz[0] + R.z[1] * gz[1] + R.z[2] * gz[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) gc[i] = (gz[i] - R.z[i] * dot) / R.nc;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) gc[i] = gz[i] / 1e-8f;
    }
    // c = x x b
    cross3(R.b, gc, t);
    gx[0] += t[0]; gx[1] += t[1]; gx[2] += t[2];
    float gb[3];
    cross3(gc, R.x, gb);
    // x = a / max(|a|, 1e-8)
    if (R.na > 1e-8f) {
        const float dot = R.x[0] * gx[0] + R.x[1] * gx[1] + R.x[2] * gx[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) go[i] = (gx[i] - R.x[i] * dot) / R.na;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) go[i] = gx[i] / 1e-8f;
    }
    go[3] = gb[0]; go[4] = gb[1]; go[5] = gb[2];
}

__global__ __launch_bounds__(256) void ortho6d_fwd_kernel(const float* __restrict__ o6, int ld, int B,
                                                          float* __restrict__ rot9) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    Rot R;
    ortho6d(o6 + (size_t)i * ld, R);
    float* o = rot9 + (size_t)i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[3 * r] = R.x[r]; o[3 * r + 1] = R.y[r]; o[3 * r + 2] = R.z[r]; }
}

// The generator's output row [joint (nj) | rot6 | transl] -> what the discriminator takes: R = ortho6d(rot6) (9), and
// contiguous copies of transl (3) and joint (nj) — one launch instead of ortho6d + two column copies
__global__ __launch_bounds__(256) void fake_pose_kernel(const float* __restrict__ heads, int ld, int nj, int B,
                                                        float* __restrict__ rot9, float* __restrict__ transl,
                                                        float* __restrict__ joint) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const float* h = heads + (size_t)i * ld;
    Rot R;
    ortho6d(h + nj, R);
    float* o = rot9 + (size_t)i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) { o[3 * r] = R.x[r]; o[3 * r + 1] = R.y[r]; o[3 * r + 2] = R.z[r]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) transl[(size_t)i * 3 + k] = h[nj + 6 + k];
    for (int k = 0; k < nj; ++k) joint[(size_t)i * nj + k] = h[k];
}

__global__ __launch_bounds__(256) void ortho6d_bwd_kernel(const float* __restrict__ o6, int ld, int B,
                                                          const float* __restrict__ d_rot9, float* __restrict__ d_o6,
                                                          int ldd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    Rot R;
    ortho6d(o6 + (size_t)i * ld, R);
    float go[6], gR[9];
    gld9(d_rot9 + (size_t)i * 9, gR);
    ortho6d_bwd(R, gR, go);
#pragma unroll
    for (int k = 0; k < 6; ++k) d_o6[(size_t)i * ldd + k] = go[k];
}

// heads (B, nj+9) = [joint | rot6 | transl];  d_pose (B, 12+nj) = [d rot9 | d transl | d joint].
// One row per thread, rows strided by 21 / 24 floats: a single workgroup pushes ~70 uncoalesced wave-loads of 64
// cache lines each through one CU's address unit (44 us at B = 1024).  With `partial` the grid spreads the rows
// over CUs (64 rows = one wave per workgroup); every workgroup leaves its partial sums — the four loss terms and, with
// fin.gbh, the column sums of its 64 rows of d_heads (the heads' bias gradient, networks.py:98-100) — and takes a ticket;
// the workgroup that draws the last one adds the partials in workgroup order (deterministic) and writes the losses and the
// bias gradient: what gen_loss_final_kernel and narrow_colsum_kernel did as launches of their own until round 5 (two
// dependent launches less on the G phase's chain).  The ticket must read 0 when the kernel starts: the G phase's first
// single-workgroup kernel (bce_kernel / wass_kernel, clear_ticket) clears it; the last workgroup leaves it at 0 again.
// Without `partial`: one workgroup, final values written directly.
__device__ __forceinline__ float ld_agent(const float* p) {   // a load that other workgroups' (fenced) stores are visible to
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <int NT>   // threads per workgroup: 64 (the spread form) or 256 (one workgroup); a row's ~150 values stay in registers
__global__ __launch_bounds__(NT) void gen_loss_kernel(int nj, float bce_w, float c_tr, float c_rot, float c_cf,
                                                        int B, const float* __restrict__ heads,
                                                        const float* __restrict__ gt_rot9,
                                                        const float* __restrict__ gt_transl,
                                                        const float* __restrict__ gt_joint,
                                                        const float* __restrict__ fake_score,
                                                        const float* __restrict__ d_pose, float* __restrict__ losses,
                                                        float* __restrict__ d_heads, float* __restrict__ partial,
                                                        GenLossFin fin) {
    __shared__ float sh[16];
    __shared__ float dsh[64][kNhMax + 1];   // the workgroup's rows of d_heads (partial form with fin.gbh: 64 threads)
    __shared__ float tot_sh[4 + kNhMax];
    __shared__ int last_sh;
    const int nh = nj + 9;
    const bool colsum = partial && fin.gbh && d_heads;
    if (colsum) {
        for (int e = 0; e < nh; ++e) dsh[threadIdx.x][e] = 0.f;   // rows past B add nothing
    }
    float l_tr = 0.f, l_rot = 0.f, l_cf = 0.f, l_f = 0.f;
    constexpr int kNjMax = kNhMax - 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        // Everything the row needs is requested before anything is used (clamped indices where the joint count is short of
        // kNjMax) and waited for ONCE: as loops of load - use - store the row took ~30 dependent waits (9.9 us at B = 1024).
        // Rows of 21 / 24 / 27 floats at 4-byte aligned addresses: the compiler may merge neighbouring loads into 8- and 16-byte
        // pieces — all of them are waited for by the one vmcnt(0) below, never by a counted wait (tools/vmcnt_audit.py).
        const float* hrow = heads + (size_t)i * nh;
        // (optional operands: loaded from a valid stand-in address when absent and zeroed after the wait — a load under a branch
        //  merges registers at the join, where the compiler waits with a counted vmcnt)
        const float* prow = d_pose ? d_pose + (size_t)i * (12 + nj) : hrow;
        const int pmax = d_pose ? 11 + nj : nh - 1;
        const float* fsp = fake_score ? fake_score + i : hrow;
        float hj[kNjMax], gj[kNjMax], pj[kNjMax], o6[6], ht[3], gr[9], gt3[3], pr[9], pt[3];
#pragma unroll
        for (int e = 0; e < kNjMax; ++e) {
            const int ec = e < nj ? e : nj - 1;
            hj[e] = hrow[ec];
            gj[e] = gt_joint[(size_t)i * nj + ec];
            pj[e] = prow[min(12 + ec, pmax)];
        }
#pragma unroll
        for (int e = 0; e < 6; ++e) o6[e] = hrow[nj + e];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            ht[e] = hrow[nj + 6 + e];
            gt3[e] = gt_transl[(size_t)i * 3 + e];
            pt[e] = prow[min(9 + e, pmax)];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            gr[e] = gt_rot9[(size_t)i * 9 + e];
            pr[e] = prow[min(e, pmax)];
        }
        float fs = *fsp;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!d_pose) {
#pragma unroll
            for (int e = 0; e < kNjMax; ++e) pj[e] = 0.f;
#pragma unroll
            for (int e = 0; e < 3; ++e) pt[e] = 0.f;
#pragma unroll
            for (int e = 0; e < 9; ++e) pr[e] = 0.f;
        }
        Rot R;
        ortho6d_of(o6, R);
        float Rm[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) { Rm[3 * r] = R.x[r]; Rm[3 * r + 1] = R.y[r]; Rm[3 * r + 2] = R.z[r]; }
        float gR[9];
        const float k_rot = c_rot * 2.f / (9.f * (float)B);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float d = Rm[e] - gr[e];
            l_rot = fmaf(d, d, l_rot);
            gR[e] = k_rot * d + pr[e];
        }
        const float k_tr = c_tr * 2.f / (3.f * (float)B);
        float dtr[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float d = ht[e] - gt3[e];
            l_tr = fmaf(d, d, l_tr);
            dtr[e] = k_tr * d + pt[e];
        }
        const float k_cf = c_cf * 2.f / ((float)nj * (float)B);
        float dj[kNjMax];
#pragma unroll
        for (int e = 0; e < kNjMax; ++e) {
            const float d = hj[e] - gj[e];
            if (e < nj) l_cf = fmaf(d, d, l_cf);
            dj[e] = k_cf * d + pj[e];
        }
        if (d_heads) {
            float go[6];
            ortho6d_bwd(R, gR, go);
            float* drow = d_heads + (size_t)i * nh;
#pragma unroll
            for (int e = 0; e < kNjMax; ++e) {
                if (e < nj) {
                    drow[e] = dj[e];
                    if (colsum) dsh[threadIdx.x][e] = dj[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                drow[nj + e] = go[e];
                if (colsum) dsh[threadIdx.x][nj + e] = go[e];
            }
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                drow[nj + 6 + e] = dtr[e];
                if (colsum) dsh[threadIdx.x][nj + 6 + e] = dtr[e];
            }
        }
        if (fake_score) l_f += bce_w < 0.f ? -fs : -fmaxf(logf(fs), -100.f);  // bce_w < 0: Wasserstein
    }
    const float t_tr = block_sum_1024(l_tr, sh);
    const float t_rot = block_sum_1024(l_rot, sh);
    const float t_cf = block_sum_1024(l_cf, sh);
    const float t_f = block_sum_1024(l_f, sh);   // (its barriers also order the dsh stores above before the reads below)
    float ltot[4] = {t_tr, t_rot, t_cf, t_f};
    if (partial) {
        const int np = 4 + (colsum ? nh : 0);    // partial sums per workgroup
        float* o = partial + (size_t)np * blockIdx.x;
        if (threadIdx.x == 0) { o[0] = t_tr; o[1] = t_rot; o[2] = t_cf; o[3] = t_f; }
        if (colsum && (int)threadIdx.x < nh) {   // column threadIdx.x of the 64 rows, in row order
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) a += dsh[r][threadIdx.x];
            o[4 + threadIdx.x] = a;
        }
        if (!fin.ticket) return;   // (a caller that adds the partials itself)
        __threadfence();           // the partials are visible device-wide before this workgroup's ticket is
        __syncthreads();
        if (threadIdx.x == 0) last_sh = atomicAdd(fin.ticket, 1u) == gridDim.x - 1 ? 1 : 0;
        __syncthreads();
        if (!last_sh) return;
        __threadfence();
        if ((int)threadIdx.x < np) {
            float a = 0.f;
            for (int b0 = 0; b0 < (int)gridDim.x; b0 += 8) {   // 8 loads in flight, added in workgroup order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = ld_agent(partial + (size_t)np * min(b0 + u, (int)gridDim.x - 1) + threadIdx.x);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (b0 + u < (int)gridDim.x) a += t[u];
            }
            tot_sh[threadIdx.x] = a;
            if (threadIdx.x >= 4) fin.gbh[threadIdx.x - 4] = fin.accumulate ? fin.gbh[threadIdx.x - 4] + a : a;
        }
        __syncthreads();
        if (threadIdx.x != 0) return;
        fin.ticket[0] = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) ltot[q] = tot_sh[q];
    }
    if (threadIdx.x == 0) {
        const float lf = (bce_w < 0.f ? 1.f : bce_w) * (ltot[3] / (float)B);
        const float ltr = c_tr * (ltot[0] / (3.f * (float)B));
        const float lrot = c_rot * (ltot[1] / (9.f * (float)B));
        const float lcf = c_cf * (ltot[2] / ((float)nj * (float)B));
        losses[DGG_L_GEN_FAKE] = lf;
        losses[DGG_L_TRANSL] = ltr;
        losses[DGG_L_ROT] = lrot;
        losses[DGG_L_CONF] = lcf;
        losses[DGG_L_TOTAL_GEN] = lf + (ltr + lrot + lcf);
    }
}

// Graph replay: the optimiser's step number lives in device memory.  step[0] += 1; dyn = {lr / (1 - b1^t), sqrt(1 - b2^t)}
// formed in double, as adam_launch does on the host.  seed (optional): the dropout seed, advanced alongside.
__global__ void adam_prep_kernel(int64_t* __restrict__ step, float* __restrict__ dyn, double lr, double b1, double b2) {
    const int64_t t = step[0] + 1;
    step[0] = t;
    dyn[0] = (float)(lr / (1.0 - pow(b1, (double)t)));
    dyn[1] = (float)sqrt(1.0 - pow(b2, (double)t));
}
__global__ void seed_bump_kernel(uint64_t* __restrict__ seed) { seed[0] += 0x9E3779B97F4A7C15ull; }

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float step_size, float bc2_sqrt, float omb1, float b2, float omb2,
                                                   float eps, float wd, float gscale, const float* __restrict__ dyn) {
    if (dyn) {  // device-resident step-dependent scalars (dgg_adam_l2_graph)
        step_size = dyn[0];
        bc2_sqrt = dyn[1];
    }
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = fmaf(wd, pa[e], ga[e] * gscale);
            ma[e] = ma[e] + (gr - ma[e]) * omb1;
            va[e] = b2 * va[e] + omb2 * gr * gr;
            const float den = sqrtf(va[e]) / bc2_sqrt + eps;
            pa[e] = pa[e] - step_size * (ma[e] / den);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gr = fmaf(wd, p[t], g[t] * gscale);
        const float mn = m[t] + (gr - m[t]) * omb1;
        const float vn = b2 * v[t] + omb2 * gr * gr;
        m[t] = mn;
        v[t] = vn;
        p[t] = p[t] - step_size * (mn / (sqrtf(vn) / bc2_sqrt + eps));
    }
}

__global__ __launch_bounds__(256) void translate_axis_kernel(float* __restrict__ transl,
                                                             const float* __restrict__ rot9, int B, int axis,
                                                             float offset) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
#pragma unroll
    // (volatile reads: three single-dword loads, never a global_load_dwordx3 — see load3() in bps_encode.hip)
    for (int r = 0; r < 3; ++r) {
        const float t = static_cast<const volatile float*>(transl)[(size_t)i * 3 + r];
        transl[(size_t)i * 3 + r] = t - rot9[(size_t)i * 9 + 3 * r + axis] * offset;
    }
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int lds,
                                                        float* __restrict__ dst, int ldd, int B, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * n) return;
    const int r = (int)(i / n), c = (int)(i % n);
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
}

// ================================================================= host launchers
static inline int nrb_of(int rpg) { return (rpg + RB - 1) / RB; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int ibn_stats_launch(const ColSrcs& srcs, int rpg, int groups, float* part, hipStream_t st) {
    dim3 grid(cdiv(srcs.total, 64), groups * nrb_of(rpg));
    hipLaunchKernelGGL(ibn_stats_kernel, grid, dim3(256), 0, st, srcs, rpg, nrb_of(rpg), part);
    return check_launch("ibn_stats");
}

// host side of BnFin: NULL -> statistics already finalized (the kernel reads its mean / rstd arguments)
static int make_fin(const BnFused* fz, int rpg, BnFin& fin) {
    memset(&fin, 0, sizeof(fin));
    if (!fz) return DGG_OK;
    if (fz->train && rpg < 2) return DGG_ERR_INVALID;  // torch: "Expected more than 1 value per channel when training"
    fin.part = fz->part;
    fin.nrb = nrb_of(rpg);
    fin.train = fz->train;
    fin.run_mean = fz->run_mean;
    fin.run_var = fz->run_var;
    fin.mean_out = fz->mean_out;
    fin.rstd_out = fz->rstd_out;
    fin.varu_out = fz->varu_out;
    return DGG_OK;
}

// A second train-mode forward over the SAME inputs (the G phase's pass through the updated discriminator,
// dexgangrasp.py:808 after :789): x_hat, mean and rstd in the workspace are still right — only the running statistics
// advance once more (same batch statistics) and the affine vectors follow the updated gamma / beta.
__global__ __launch_bounds__(256) void ibn_refresh_kernel(const float* __restrict__ mean, const float* __restrict__ var_u,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ aff_scale, float* __restrict__ aff_shift,
                                                          int nprefix, int total) {
    const int oc = blockIdx.x * 256 + threadIdx.x;
    if (oc >= nprefix + total) return;
    if (oc < nprefix) {
        if (aff_scale) {
            aff_scale[oc] = 1.f;
            aff_shift[oc] = 0.f;
        }
        return;
    }
    const int c = oc - nprefix;
    run_mean[c] = (1.f - kBnMom) * run_mean[c] + kBnMom * mean[c];
    run_var[c] = (1.f - kBnMom) * run_var[c] + kBnMom * var_u[c];
    if (aff_scale) {   // (NULL: the GEMMs read gamma / beta themselves — a network without prefix columns)
        aff_scale[oc] = gamma[c];
        aff_shift[oc] = beta[c];
    }
}

int ibn_refresh_launch(const float* mean, const float* var_u, float* run_mean, float* run_var, const float* gamma,
                       const float* beta, float* aff_scale, float* aff_shift, int nprefix, int total, hipStream_t st) {
    hipLaunchKernelGGL(ibn_refresh_kernel, dim3(cdiv(nprefix + total, 256)), dim3(256), 0, st, mean, var_u, run_mean,
                       run_var, gamma, beta, aff_scale, aff_shift, nprefix, total);
    return check_launch("ibn_refresh");
}

int ibn_apply_launch(const ColSrcs& srcs, int B, int rpg, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, const float* prefix, int nprefix, float* X0, int ldx, const float* aff_gamma,
                     const float* aff_beta, float* aff_scale, float* aff_shift, hipStream_t st, const BnFused* fz,
                     float* X0b, int ldb, int nb) {
    BnFin fin;
    if (int rc = make_fin(fz, rpg, fin)) return rc;
    const int width = nprefix + srcs.total;
    dim3 grid(cdiv(width, 64), cdiv(B, RB * kIbnBlocks));
    hipLaunchKernelGGL(ibn_apply_kernel, grid, dim3(256), 0, st, srcs, B, rpg, mean, rstd, gamma, beta, prefix,
                       nprefix, X0, ldx, aff_gamma, aff_beta, aff_scale, aff_shift, fin, X0b, ldb, nb);
    return check_launch("ibn_apply");
}

// ---------------------------------------------------------------------------------------------------------------------
// XShare (round 6): the BPS block of a cGAN step — the same batch, hence the same column statistics and the same x_hat —
// is normalised ONCE, by the discriminator's real-batch pass (ibn_apply_kernel's second destination), for the generator
// (networks.py:88-95), the fake-batch pass and the G phase's pass through the updated discriminator (:148-154;
// dexgangrasp.py:772-789,808).  What is left of a consumer's input BatchNorm is this one launch:
//   workgroups 1..: for the shared columns, copy the producer's batch statistics into the consumer's own arrays (its
//     backward and a later DGG_REUSE_INPUT_NORM forward read them there), advance the consumer's running statistics with
//     them, write its affine vectors [1.. | gamma], [0.. | beta], and copy the generator's latent prefix into the slot in
//     front of every shared row;
//   workgroup 0 (the fake pass only): the generator's output row [joint | rot6 | transl] -> R = ortho6d(rot6), contiguous
//     copies of R / transl / joint (what fake_pose_kernel writes), and the train-mode BatchNorm of these npose columns over
//     the whole batch — two passes over rows the workgroup keeps re-deriving from `heads` (fixed order: deterministic) —
//     into the shared rows' pose columns, with statistics, running statistics and affine vectors.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPoseMax = 24;   // 12 + n_hand_joints of the shipped configuration (more: the caller keeps the separate launches)
template <int NJ>   // hand joints of the fake pose (compile-time: the flat copies below divide by it and by 12 + NJ —
                    // with run-time divisors those divisions, ~100 per thread, WERE the kernel: 28 us on one CU)
__global__ __launch_bounds__(1024) void ibn_shared_kernel(IbnShared P) {
    const int tid = threadIdx.x;
    if (blockIdx.x > 0 || P.npose == 0) {
        const int wg = P.npose > 0 ? (int)blockIdx.x - 1 : (int)blockIdx.x;
        const int nwg = P.npose > 0 ? (int)gridDim.x - 1 : (int)gridDim.x;
        for (int c = wg * 1024 + tid; c < P.n_alias; c += nwg * 1024) {
            const float m = P.src_mean[c], vu = P.src_varu[c];
            P.mean[c] = m;
            P.rstd[c] = P.src_rstd[c];
            P.varu[c] = vu;
            P.run_mean[c] = (1.f - kBnMom) * P.run_mean[c] + kBnMom * m;
            P.run_var[c] = (1.f - kBnMom) * P.run_var[c] + kBnMom * vu;
            if (P.aff_scale) {
                P.aff_scale[P.nprefix + c] = P.gamma[c];
                P.aff_shift[P.nprefix + c] = P.beta[c];
            }
        }
        for (int i = wg * 1024 + tid; i < (P.aff_scale ? P.nprefix : 0); i += nwg * 1024) {
            P.aff_scale[i] = 1.f;
            P.aff_shift[i] = 0.f;
        }
        if (P.nprefix > 0)
            for (int i = wg * 1024 + tid; i < P.B * P.nprefix; i += nwg * 1024) {
                const int m = i / P.nprefix, j = i - m * P.nprefix;
                P.X0[(size_t)m * P.ldx + j] = P.prefix[i];
            }
        return;
    }
    // ---- workgroup 0: the fake pose and its BatchNorm (npose = 12 + nj <= kPoseMax columns), B <= 1024: one row per thread.
    // A row is 21 floats at a pitch of 84 bytes: fetched row by row, every wave-load touches 64 cache lines and 16 waves
    // queue at one CU's address unit (the first version of this kernel: 44 us; gen_loss_kernel's history says the same).  So
    // the generator's rows come in as ONE flat, coalesced stream into LDS, every thread takes its row from there, and the
    // outputs leave the same way: through LDS, as flat coalesced stores.
    extern __shared__ float lds_dyn[];                 // B * max(ldh, kPoseMax) floats
    __shared__ float red[32][kPoseMax];
    __shared__ float tot[2][kPoseMax];
    constexpr int np = 12 + NJ, nj = NJ, ldh = NJ + 9;   // (the launcher checks P.npose / P.nj / P.ldh against these)
    const int B = P.B;
    {   // all loads of a thread in flight at once (a load-then-store loop is one memory round trip per iteration: 21 of them)
        const int nflt = B * ldh, nvec = nflt >> 2;          // <= 6144 float4 for B <= 1024, ldh <= 24
        const float4* h4 = reinterpret_cast<const float4*>(P.heads);
        float4 buf[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) buf[k] = h4[min(tid + k * 1024, nvec > 0 ? nvec - 1 : 0)];
        float tail = 0.f;
        const int ti = (nvec << 2) + tid;
        if (tid < 4) tail = P.heads[min(ti, nflt - 1)];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (tid + k * 1024 < nvec) reinterpret_cast<float4*>(lds_dyn)[tid + k * 1024] = buf[k];
        if (tid < 4 && ti < nflt) lds_dyn[ti] = tail;
    }
    __syncthreads();
    float v[kPoseMax];
#pragma unroll
    for (int c = 0; c < kPoseMax; ++c) v[c] = 0.f;
    const bool live = tid < B;
    if (live) {   // [rot9 | transl | joint], as the discriminator's concat has it
        const float* h = lds_dyn + tid * ldh;
        float o6[6];   // (ortho6d() itself fetches from GLOBAL memory: the values are taken from LDS here)
#pragma unroll
        for (int k = 0; k < 6; ++k) o6[k] = h[nj + k];
        Rot R;
        ortho6d_of(o6, R);
#pragma unroll
        for (int r = 0; r < 3; ++r) { v[3 * r] = R.x[r]; v[3 * r + 1] = R.y[r]; v[3 * r + 2] = R.z[r]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) v[9 + k] = h[nj + 6 + k];
#pragma unroll
        for (int k = 0; k < kPoseMax - 12; ++k) v[12 + k] = k < nj ? h[k] : 0.f;
    }
    __syncthreads();   // every row has been read: the LDS image turns into [row][np] pose values at a pitch of np + 1
    // floats (odd: a thread writing its row element by element, 24 floats apart from its neighbour's, would put 64 lanes on
    // 4 banks — a 16-way conflict on each of 48 writes and 16 waves was most of this kernel's first 23-44 us)
    constexpr int pp = np + 1;
    if (live) {
#pragma unroll
        for (int c = 0; c < np; ++c) lds_dyn[tid * pp + c] = v[c];
    }
    // column totals over all rows, from the LDS image: 32 threads per column take every 32nd row, one thread per column adds
    // their 32 partial sums in order — a fixed order, and no cross-lane traffic
    auto col_total = [&](int slot, bool centred) {
        __syncthreads();   // the image is complete
        if (tid < 32 * np) {
            const int c = tid % np, j = tid / np;
            const float mu = centred ? tot[0][c] : 0.f;
            float t = 0.f;
            for (int m = j; m < B; m += 32) {
                const float d = lds_dyn[m * pp + c] - mu;
                t += centred ? d * d : d;
            }
            red[j][c] = t;
        }
        __syncthreads();
        if (tid < np) {
            float t = 0.f;
            for (int j = 0; j < 32; ++j) t += red[j][tid];
            tot[slot][tid] = slot == 0 ? t / (float)B : t;   // slot 0: the column MEAN, slot 1: M2
        }
        __syncthreads();
    };
    col_total(0, false);
    // contiguous copies for the backward / the G phase: R (B,9), transl (B,3), joint (B,nj) — flat, coalesced
    for (int i = tid; i < B * 9; i += 1024) P.rot9[i] = lds_dyn[(i / 9) * pp + i % 9];
    for (int i = tid; i < B * 3; i += 1024) P.transl[i] = lds_dyn[(i / 3) * pp + 9 + i % 3];
    for (int i = tid; i < B * nj; i += 1024) P.joint[i] = lds_dyn[(i / nj) * pp + 12 + i % nj];
    const float fB = (float)B;
    col_total(1, true);   // (its barriers also order the copies above before the image is overwritten)
    if (tid < np) red[0][tid] = 1.0f / sqrtf(tot[1][tid] / fB + kBnEps);   // rstd per column, once
    __syncthreads();
    if (live) {
#pragma unroll
        for (int c = 0; c < np; ++c) {
            const float rstd = red[0][c];
            lds_dyn[tid * pp + c] = fmaf(v[c], rstd, 0.f - tot[0][c] * rstd);
        }
    }
    __syncthreads();
    {   // the pose columns of the shared rows: np contiguous floats per row
        float* const base = P.X0 + P.nprefix + P.n_alias;
        for (int i = tid; i < B * np; i += 1024) {
            const int m = i / np, c = i - m * np;
            base[(size_t)m * P.ldx + c] = lds_dyn[m * pp + c];
        }
    }
    if (tid < np) {
        const int c = P.n_alias + tid;
        const float m2 = tot[1][tid], mu = tot[0][tid];
        const float var_u = m2 / (fB - 1.f);
        P.mean[c] = mu;
        P.rstd[c] = red[0][tid];
        P.varu[c] = var_u;
        P.run_mean[c] = (1.f - kBnMom) * P.run_mean[c] + kBnMom * mu;
        P.run_var[c] = (1.f - kBnMom) * P.run_var[c] + kBnMom * var_u;
        if (P.aff_scale) {
            P.aff_scale[P.nprefix + c] = P.gamma[c];
            P.aff_shift[P.nprefix + c] = P.beta[c];
        }
    }
}

int ibn_shared_launch(const IbnShared& P, hipStream_t st) {
    if (P.npose > kPoseMax || P.npose < 0) return DGG_ERR_INVALID;
    // (one instantiation: 12 hand joints, the shipped configuration — carve_step's xs_ok asks for it)
    if (P.npose > 0 && (P.nj != 12 || P.npose != 24 || P.ldh != 21 || P.B < 2 || P.B > 1024)) return DGG_ERR_INVALID;
    const int work = P.n_alias > P.B * P.nprefix ? P.n_alias : P.B * P.nprefix;
    int nwg = cdiv(work, 1024);
    if (nwg < 1) nwg = 1;
    if (nwg > 16) nwg = 16;
    const size_t lds = P.npose > 0 ? (size_t)P.B * (P.npose + 1) * sizeof(float) : 0;   // <= 100 KB (pitch np + 1 >= ldh)
    static const bool big_lds = [] {   // (more than the 64 KB a launch gets without asking)
        return hipFuncSetAttribute((const void*)ibn_shared_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    }();
    (void)big_lds;
    hipLaunchKernelGGL(ibn_shared_kernel<12>, dim3(nwg + (P.npose > 0 ? 1 : 0)), dim3(1024), lds, st, P);
    return check_launch("ibn_shared");
}

int reduce_bias_launch(const float* slabs, int nslab, long long slab_stride, int ld_slab, const float* bias,
                       float* out, int ld_out, int B, int rpg, int groups, int N, int stat_lo, int stat_hi,
                       int act_lo, int act_hi, float* part, hipStream_t st, int accumulate, const float* slabs2,
                       int nslab2, long long stride2, int ld2) {
    (void)B;
    dim3 grid(cdiv(N, 64), groups * nrb_of(rpg));
    hipLaunchKernelGGL(reduce_bias_kernel, grid, dim3(256), 0, st, slabs, nslab, slab_stride, ld_slab, bias, out,
                       ld_out, rpg, nrb_of(rpg), N, stat_lo, stat_hi, act_lo, act_hi, part, accumulate, slabs2,
                       slabs2 ? nslab2 : 0, stride2 ? stride2 : slab_stride, ld2 ? ld2 : ld_slab);
    return check_launch("reduce_bias");
}

// Per-object factorisation of an eval-mode first layer: all n_per samples of an object share
// its BPS row, so x_hat_bps W13[:, bps]^T + b13 is ONE row per object (cobj).  A sample's pre-activation is
//   out[m][c] = cobj[m / n_per][c] + sum_s slabs[s][m][c] + sum_j small(m, j) * wsm[c * ldw + j]
// with the few per-sample input columns `small` (generator: Z; evaluator: eval-BatchNorm'd rot9 | transl), LeakyReLU on
// columns [act_lo, act_hi).  grid: (ceil(N/64), ceil(B/RB)).
struct SmallIn {
    ColSrcs srcs;                           // raw per-sample columns (virtual concat), srcs.total <= 16
    const float *gamma, *beta, *rm, *rv;    // eval BatchNorm of those columns (NULL gamma: identity)
};
// Optional: columns [0, hid) — fc1's pre-activation — leave as h = lrelu(bn1(.)) with bn1's running statistics, written
// to `h` (pitch ldh) instead of `out`: the fc2 product that follows is then a plain GEMM on the bf16x3 kernel.
struct HOut {
    float* h;
    int ldh, hid;
    const float *gamma, *beta, *rm, *rv;
};
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ cobj, int ldc, int n_per,
                                                          const float* __restrict__ slabs, int nslab,
                                                          long long slab_stride, int ld_slab, SmallIn sm,
                                                          const float* __restrict__ wsm, int ldw,
                                                          float* __restrict__ out, int ld_out, int B, int N, int act_lo,
                                                          int act_hi, HOut HO) {
    // the workgroup's 32 rows x <= 16 small columns, eval-BatchNorm'd, staged once through LDS (every column of the
    // output needs them: from global memory they were 128 broadcast loads per thread)
    __shared__ __attribute__((aligned(16))) float smx[RB][16];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int ns = sm.srcs.total;
    for (int idx = threadIdx.x; idx < RB * 16; idx += 256) {
        const int rl = idx >> 4, j = idx & 15;
        const int row = blockIdx.y * RB + rl;
        float x = 0.f;
        if (j < ns && row < B) {
            int ld;
            const float* p = src_col(sm.srcs, j, ld);
            float scj = 1.f, shj = 0.f;
            if (sm.gamma) {
                scj = sm.gamma[j] * (1.0f / sqrtf(sm.rv[j] + kBnEps));
                shj = sm.beta[j] - sm.rm[j] * scj;
            }
            x = fmaf(p[(size_t)row * ld], scj, shj);
        }
        smx[rl][j] = x;
    }
    // the 64 columns' weights of the small inputs, staged the same way: 16 consecutive floats per column lie 64 bytes
    // apart from the next column's — fetched per thread (lane = column) every one of the 16 loads touched 64 cache lines
    // (1024 line requests per wave against 32 for the rows themselves: 112 us per launch of the config-5 sweep, now see
    // profiles/r04_experiments.txt 21); staged, 16 lanes read one column's 64 bytes
    __shared__ float wl[64][17];
    for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
        const int cc = idx >> 4, j = idx & 15;
        const int col = blockIdx.x * 64 + cc;
        wl[cc][j] = (j < ns && col < N) ? wsm[(size_t)col * ldw + j] : 0.f;
    }
    __syncthreads();
    const int c = blockIdx.x * 64 + lane;
    if (c >= N) return;
    const bool to_h = HO.h && c < HO.hid;
    float h_mu = 0.f, h_rs = 0.f, h_ga = 0.f, h_be = 0.f;
    if (to_h) {
        h_mu = HO.rm[c];
        h_rs = 1.0f / sqrtf(HO.rv[c] + kBnEps);
        h_ga = HO.gamma[c];
        h_be = HO.beta[c];
    }
    float w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = wl[lane][j];
    const bool act = c >= act_lo && c < act_hi;
    const int r0 = blockIdx.y * RB + sub * RS;
    const int n = min(RS, B - r0);
    if (n <= 0) return;
    // all 8 rows of a wave in flight per load group (a row loop costs one memory round trip per row); per row the same
    // order of additions as before: object row, slabs in k order, then the small-column sum
    float v[RS], t[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int row = r0 + row_clamp(r, n);
        v[r] = cobj[(size_t)(row / n_per) * ldc + c];
        t[r] = 0.f;
    }
    for (int s2 = 0; s2 < nslab; ++s2) {
        float u[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r)
            u[r] = slabs[(size_t)s2 * slab_stride + (size_t)(r0 + row_clamp(r, n)) * ld_slab + c];
#pragma unroll
        for (int r = 0; r < RS; ++r) v[r] += u[r];
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j)          // columns >= ns hold 0 with weight 0: t + 0 * 0 leaves t as it is
            if (j < ns) t[r] = fmaf(smx[sub * RS + r][j], w[j], t[r]);
    }
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        if (r < n) {
            const float o = v[r] + t[r];
            if (to_h) HO.h[(size_t)(r0 + r) * HO.ldh + c] = lrelu(fmaf((o - h_mu) * h_rs, h_ga, h_be));
            else out[(size_t)(r0 + r) * ld_out + c] = act ? lrelu(o) : o;
        }
    }
}

int expand_rows_launch(const float* cobj, int ldc, int n_per, const float* slabs, int nslab, long long slab_stride,
                       int ld_slab, const ColSrcs& small, const float* gamma, const float* beta, const float* rm,
                       const float* rv, const float* wsm, int ldw, float* out, int ld_out, int B, int N, int act_lo,
                       int act_hi, hipStream_t st, float* h, int ldh, int hid, const float* g1, const float* b1,
                       const float* rm1, const float* rv1) {
    if (small.total > 16 || n_per < 1) return DGG_ERR_UNSUPPORTED;
    HOut ho = {h, ldh, hid, g1, b1, rm1, rv1};
    SmallIn sm;
    sm.srcs = small;
    sm.gamma = gamma; sm.beta = beta; sm.rm = rm; sm.rv = rv;
    dim3 grid(cdiv(N, 64), cdiv(B, RB));
    hipLaunchKernelGGL(expand_rows_kernel, grid, dim3(256), 0, st, cobj, ldc, n_per, slabs, nslab, slab_stride, ld_slab, sm,
                       wsm, ldw, out, ld_out, B, N, act_lo, act_hi, ho);
    return check_launch("expand_rows");
}

int bn_act_launch(const float* a, int lda, int B, int rpg, int N, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, float* h, int ldh, hipStream_t st, const BnFused* fz) {
    BnFin fin;
    if (int rc = make_fin(fz, rpg, fin)) return rc;
    dim3 grid(cdiv(N, 64), cdiv(B, RB));
    hipLaunchKernelGGL(bn_act_kernel, grid, dim3(256), 0, st, a, lda, B, rpg, N, mean, rstd, gamma, beta, h, ldh, fin);
    return check_launch("bn_act");
}

int res_out_launch(const float* s, int lds, const float* a2, int lda2, int B, int rpg, int N, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, int drop_mode, const float* keep_in,
                   uint64_t seed, uint64_t seed_off, uint8_t* keep_out, float* y, int ldy, hipStream_t st,
                   const BnFused* fz, const float* a2_bias) {
    BnFin fin;
    if (int rc = make_fin(fz, rpg, fin)) return rc;
    dim3 grid(cdiv(N, 64), cdiv(B, RB));
    hipLaunchKernelGGL(res_out_kernel, grid, dim3(256), 0, st, s, lds, a2, lda2, B, rpg, N, mean, rstd, gamma, beta,
                       drop_mode, keep_in, seed, seed_off, keep_out, y, ldy, fin, a2_bias);
    return check_launch("res_out");
}

// column sums of a narrow (B x n <= 32) matrix; one workgroup of 1024 threads = 32 row lanes x 32 columns, 8 loads in
// flight per thread (a row loop with one load per trip costs B / 8 serial memory round trips: 15 us at B = 1024), fixed
// summation order
__global__ __launch_bounds__(1024) void narrow_colsum_kernel(const float* __restrict__ x, int ld, int B, int n,
                                                             float* __restrict__ dst, int accumulate) {
    __shared__ float sh[32][33];
    const int j = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float a = 0.f;
    if (j < n) {
        for (int r0 = rl; r0 < B; r0 += 32 * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 32 * u;
                t[u] = x[(size_t)(r < B ? r : B - 1) * ld + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + 32 * u < B) a += t[u];
        }
    }
    sh[rl][j] = a;
    __syncthreads();
    if (rl == 0 && j < n) {
        float t = sh[0][j];
#pragma unroll
        for (int q = 1; q < 32; ++q) t += sh[q][j];
        dst[j] = accumulate ? dst[j] + t : t;
    }
}

static void one_problem(GemmBatch& g, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N,
                        int K, int splits, long long slab_stride) {
    memset(&g, 0, sizeof(g));
    g.nprob = 1;
    GemmProb& p = g.p[0];
    p.A = A; p.B = Bm; p.C = C;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.kchunk = K;
    if (splits > 0) {
        const int ksteps = (K + 31) / 32;
        const int per = (ksteps + splits - 1) / splits;
        p.kchunk = per * 32;
        p.splits = (K + p.kchunk - 1) / p.kchunk;
        p.slab_stride = slab_stride;
    }
}

// nh == 1 (discriminator / evaluator score): one wave per row.  nh > 1 (the generator's 21 pose outputs,
// networks.py:98-100): the heads are a (B x K)(K x nh) product — a stage kernel of its own at K = 512 (heads_stage_kernel), else an
// MFMA GEMM with split-K slabs + the bias reduce;
// the per-row kernel spent its time in nh x 6 cross-lane reduction steps (53 us at B = 1024 against 12).
// scratch: >= 4 * B * nh floats (nh > 1 only).
int heads_fwd_launch(const float* y, int ldy, int B, int K, const float* wh, const float* bh, int nh, int sigmoid,
                     float* out, int ldo, float* scratch, hipStream_t st) {
    if (nh > kNhMax) return DGG_ERR_UNSUPPORTED;
    if (nh == 1) {
        hipLaunchKernelGGL((heads_fwd_kernel<1>), dim3(cdiv(B, 4)), dim3(256), 0, st, y, ldy, B, K, wh, bh, nh, sigmoid,
                           out, ldo);
        return check_launch("heads_fwd");
    }
    if (sigmoid || !scratch) return DGG_ERR_UNSUPPORTED;
    if (K == 512) return heads_stage_launch(y, ldy, B, wh, bh, nh, out, ldo, st);   // one launch (fc2_stage.hip)
    GemmBatch g;
    one_problem(g, y, ldy, wh, K, scratch, nh, B, nh, K, 4, (long long)B * nh);
    int rc = gemm_launch(GEMM_NT, g, st);
    if (rc) return rc;
    return reduce_bias_launch(scratch, g.p[0].splits, (long long)B * nh, nh, bh, out, ldo, B, B, 1, nh, 0, 0, 0, 0,
                              nullptr, st);
}

// the heads' bias gradient alone: column sums of d_out (B, nh) (+)= gbh
int heads_bias_grad_launch(const float* dout, int ldd, int B, int nh, float* gbh, int accumulate, hipStream_t st) {
    if (nh > kNhMax) return DGG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(narrow_colsum_kernel, dim3(1), dim3(1024), 0, st, dout, ldd, B, nh, gbh, accumulate);
    return check_launch("narrow_colsum");
}

int res_out_bwd_launch(const float* dy, int lddy, const float* y, int ldy, const uint8_t* keep, const float* s,
                       int lds, const float* a2, int lda2, const float* mean, const float* rstd, int B, int rpg, int N,
                       float* dpre, float* da3, int ldda3, float* part, hipStream_t st, const float* hd_dout,
                       const float* hd_w, int nq, int nslab, long long slab_stride, const SumOut* sum, int hd_n) {
    if (!dy && !hd_dout) return DGG_ERR_INVALID;
    if (hd_dout && (hd_n < 1 || hd_n > kNhMax || (hd_n > 1 && nq != 3))) return DGG_ERR_INVALID;
    SumOut so = {nullptr, nullptr, 0};
    if (sum && sum->other) so = *sum;
    if (hd_dout) nslab = 0;
    const int groups = B / rpg;
    dim3 grid(cdiv(N, 64), groups * nrb_of(rpg));
    hipLaunchKernelGGL(res_out_bwd_kernel, grid, dim3(256), 0, st, dy, lddy, y, ldy, keep, s, lds, a2, lda2, mean,
                       rstd, rpg, nrb_of(rpg), N, dpre, da3, ldda3, part, hd_dout, hd_w, nq, nslab, slab_stride, so, hd_n);
    return check_launch("res_out_bwd");
}

int bn_bwd_apply_launch(const float* dz, int lddz, const float* a, int lda, const float* mean, const float* rstd,
                        const float* gamma, const float* sums, int train, int B, int rpg, int N, float* dx, int lddx,
                        float* part, hipStream_t st, const ColFin* fin, const SumOut* sum) {
    const int groups = B / rpg;
    SumOut so = {nullptr, nullptr, 0};
    if (sum && sum->other) so = *sum;
    dim3 grid(cdiv(N, 64), groups * nrb_of(rpg));
    ColFin f = {};
    if (fin) {
        if (groups != 1 || fin->part == part || fin->nq < 2 || fin->nq > 5) return DGG_ERR_INVALID;
        f = *fin;
        if (!f.zero0) f.nz0 = 0;
        if (!f.zero1) f.nz1 = 0;
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, st, dz, lddz, a, lda, mean, rstd, gamma, sums, train,
                       rpg, nrb_of(rpg), N, dx, lddx, part, f, so);
    return check_launch("bn_bwd_apply");
}

int colsum_finalize_launch(const float* part, int groups, int nrb, int nq, int ncols, float* sums_out, float* dst0,
                           float* dst1, float* dst2, int accumulate, hipStream_t st, int ld, int c_lo, float* copy0,
                           float* copy2, float* zero0, int nz0, float* zero1, int nz1, float* dst3, float* dst4,
                           int noacc_below, long long gstride) {
    if (ld <= 0) ld = ncols;
    if (gstride <= 0) gstride = (long long)nrb * nq * ld;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(nq * ncols, 64)), dim3(256), 0, st, part, groups, nrb, nq,
                       ncols, ld, c_lo, sums_out, dst0, dst1, dst2, accumulate, copy0, copy2, zero0, zero0 ? nz0 : 0,
                       zero1, zero1 ? nz1 : 0, dst3, dst4, noacc_below, gstride);
    return check_launch("colsum_finalize");
}

int ibn_bwd_stats_launch(const ColSrcs& srcs, const float* dX0, int ldx, int col_off, const float* mean,
                         const float* rstd, int rpg, int groups, float* part, hipStream_t st, int nslab,
                         long long slab_stride, float* dsum) {
    dim3 grid(cdiv(srcs.total, 64), groups * nrb_of(rpg));
    hipLaunchKernelGGL(ibn_bwd_stats_kernel, grid, dim3(256), 0, st, srcs, dX0, ldx, col_off, mean, rstd, rpg,
                       nrb_of(rpg), part, nslab, slab_stride, dsum);
    return check_launch("ibn_bwd_stats");
}

int ibn_bwd_apply_launch(const ColSrcs& srcs, const float* dX0, int ldx, int col_off, const float* mean,
                         const float* rstd, const float* gamma, const float* sums, int train, int B, int rpg, int c_lo,
                         int c_hi, float* dst, int ldd, hipStream_t st, const float* fin_part, int nslab,
                         long long slab_stride) {
    if (c_hi <= c_lo) return DGG_OK;
    if (fin_part && B != rpg) return DGG_ERR_INVALID;
    dim3 grid(cdiv(c_hi - c_lo, 64), cdiv(B, RB));
    hipLaunchKernelGGL(ibn_bwd_apply_kernel, grid, dim3(256), 0, st, srcs, dX0, ldx, col_off, mean, rstd, gamma, sums,
                       train, B, rpg, c_lo, c_hi, dst, ldd, fin_part, nrb_of(rpg), nslab, slab_stride);
    return check_launch("ibn_bwd_apply");
}

int bce_launch(const float* p, const float* labels, float target_const, int B, float weight, float loss_scale,
               float* loss_out, float* d_logit, hipStream_t st, float* combine, int real_slot, unsigned* clear_ticket,
               float* loss_copy, const RunUpd* ru) {
    RunUpd r = {nullptr, nullptr, nullptr, nullptr, 0};
    if (ru) r = *ru;
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, st, p, labels, target_const, B, weight, loss_scale,
                       loss_out, d_logit, combine, real_slot, clear_ticket, loss_copy, r);
    return check_launch("bce");
}

int ortho6d_fwd_launch(const float* o6, int ld, int B, float* rot9, hipStream_t st) {
    hipLaunchKernelGGL(ortho6d_fwd_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, o6, ld, B, rot9);
    return check_launch("ortho6d_fwd");
}

int fake_pose_launch(const float* heads, int ld, int nj, int B, float* rot9, float* transl, float* joint, hipStream_t st) {
    hipLaunchKernelGGL(fake_pose_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, heads, ld, nj, B, rot9, transl, joint);
    return check_launch("fake_pose");
}

int ortho6d_bwd_launch(const float* o6, int ld, int B, const float* d_rot9, float* d_o6, int ldd, hipStream_t st) {
    hipLaunchKernelGGL(ortho6d_bwd_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, o6, ld, B, d_rot9, d_o6, ldd);
    return check_launch("ortho6d_bwd");
}

int gen_loss_launch(const dgg_cfg* cfg, const dgg_hparams* hp, int B, const float* heads, const float* gt_rot9,
                    const float* gt_transl, const float* gt_joint, const float* fake_score, const float* d_pose,
                    float* losses, float* d_heads, hipStream_t st, float* scratch, int wass, const GenLossFin* fin) {
    // the kernels read a negative BCE weight as "Wasserstein term -mean(fake_score) with weight 1"
    const float bw = wass ? -1.f : (float)hp->bce_weight, ct = (float)hp->transl_coef, cr = (float)hp->rot_coef,
                cc = (float)hp->conf_coef;
    const int nblk = cdiv(B, 64), nh = cfg->n_hand_joints + 9;
    if (nh > kNhMax) return DGG_ERR_UNSUPPORTED;
    // the spread form needs the caller's scratch (gen_loss_scratch_floats) and a ticket that reads 0 (GenLossFin)
    if (!scratch || !fin || !fin->ticket || !gen_loss_spread(B)) {
        if (fin && fin->gbh) return DGG_ERR_INVALID;   // the bias gradient is formed by the spread form only
        hipLaunchKernelGGL(gen_loss_kernel<256>, dim3(1), dim3(256), 0, st, cfg->n_hand_joints, bw, ct, cr, cc, B, heads,
                           gt_rot9, gt_transl, gt_joint, fake_score, d_pose, losses, d_heads, (float*)nullptr, GenLossFin{});
        return check_launch("gen_loss");
    }
    hipLaunchKernelGGL(gen_loss_kernel<64>, dim3(nblk), dim3(64), 0, st, cfg->n_hand_joints, bw, ct, cr, cc, B, heads,
                       gt_rot9, gt_transl, gt_joint, fake_score, d_pose, losses, d_heads, scratch, *fin);
    return check_launch("gen_loss");
}
// the one place that decides between the two forms (the caller tells net_backward whether the heads' bias gradient exists)
bool gen_loss_spread(int B) { return cdiv(B, 64) >= 2; }
int gen_loss_scratch_floats(const dgg_cfg* cfg, int B) { return cdiv(B, 64) * (4 + cfg->n_hand_joints + 9); }

int adam_launch(float* params, const float* grads, float* m, float* v, int64_t n, int step, double lr, double b1,
                double b2, double eps, double wd, double grad_scale, hipStream_t st) {
    if (n <= 0 || step < 1) return DGG_ERR_INVALID;
    // everything that involves 1-beta or beta^t is formed in double (like torch) and rounded once
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    {   // side work (dgg_set_side_work: the evaluator's optimiser step beside the cGAN chain): one workgroup per CU instead of
        // eight — the launch is HBM-bound either way, and 2048 workgroups arriving at once keep the chain's short kernels
        // out of the CUs for its whole duration.  1.302 / 1.309 -> 1.294 / 1.294 ms per interleaved step; 128: 1.33, 512: as 256
        // (DGG_ADAM_SIDE_WGS: diagnostics, 0 = no limit)
        static const int side = [] { const char* e = getenv("DGG_ADAM_SIDE_WGS"); return e ? atoi(e) : 256; }();
        if (side > 0 && gemm_side_work() && blocks > side) blocks = side;
    }
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, m, v, n, step_size,
                       bc2_sqrt, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd,
                       (float)grad_scale, (const float*)nullptr);
    return check_launch("adam");
}

int adam_graph_launch(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step_dev, float* dyn,
                      double lr, double b1, double b2, double eps, double wd, double grad_scale, hipStream_t st) {
    if (n <= 0) return DGG_ERR_INVALID;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, st, step_dev, dyn, lr, b1, b2);
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, m, v, n, 0.f, 1.f,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, (float)grad_scale,
                       (const float*)dyn);
    return check_launch("adam_graph");
}

int seed_bump_launch(uint64_t* seed_dev, hipStream_t st) {
    hipLaunchKernelGGL(seed_bump_kernel, dim3(1), dim3(1), 0, st, seed_dev);
    return check_launch("seed_bump");
}

int translate_axis_launch(float* transl, const float* rot9, int B, int axis, float offset, hipStream_t st) {
    hipLaunchKernelGGL(translate_axis_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, transl, rot9, B, axis, offset);
    return check_launch("translate_axis");
}

int copy_cols_launch(const float* src, int lds, float* dst, int ldd, int B, int n, hipStream_t st) {
    const int64_t tot = (int64_t)B * n;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, lds, dst, ldd, B,
                       n);
    return check_launch("copy_cols");
}

}  // namespace dgg
