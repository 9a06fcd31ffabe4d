// Kernels of the WGAN-GP gradient penalty (DexGanGrasp/models/dexgangrasp.py:149-197, cfg["is_wgan"]).
//
// The penalty GP(theta) = mean over the three pose parts of mean_b (|dS/dx_part|_b - 1)^2, S = sum_b D(x_b), is a
// function of the discriminator's INPUT gradient g; its parameter gradient needs a second derivative of the
// network.  With v = dGP/dg held fixed,  dGP/dtheta = d/dtheta (v . g) = d/dtheta [ d/deps S(x + eps v) ]:
// (1) an ordinary forward + input-gradient backward gives g and v; (2) a TANGENT forward pass (JVP) propagates v through
// the network next to the stored primal activations and yields F = sum_b sigmoid'(l_b) ldot_b; (3) one reverse sweep of
// F through the primal + tangent graph gives every parameter gradient.  Linear layers need nothing new (tangents and
// both adjoint streams go through the GEMM kernels); this file holds what is specific: the tangent of a train-mode
// BatchNorm1d (batch statistics couple the samples), its second-order backward, the LeakyReLU masks (second derivative
// zero), the sigmoid head and the penalty itself.
//
// Layout: these are narrow launches (the discriminator has 2 ResBlocks, widths 256 / 512, batch <= a few thousand): a
// 256-thread workgroup owns 16 columns x ALL rows (16 row-lanes), makes its sweeps over the rows and reduces across the
// row-lanes through LDS in a fixed order — deterministic, no atomics, no partial buffers.  Every row sweep runs in
// batches of 8 rows per thread with all loads of a batch issued before the first use (a plain row loop is one memory
// round trip per row: 60 us per launch at B = 1024 instead of ~10); the order of every sum is that of the plain loop.
// WGAN-GP is off in the shipped configuration (config_dexgangrasp.yaml:55).
#include "dgg_internal.h"

namespace dgg {

constexpr int SC = 16;  // columns per workgroup
constexpr int SR = 16;  // row-lanes per workgroup
constexpr int SU = 8;   // rows per thread and load batch

// row index of batch slot u (clamped into the matrix: clamped slots are loaded, never used)
__device__ __forceinline__ size_t srow(int b0, int u, int B) {
    const int b = b0 + u * SR;
    return (size_t)(b < B ? b : B - 1);
}

__device__ __forceinline__ float lrelu_d(float post) { return post > 0.f ? 1.f : kLrelu; }

// sums over the NR row-lanes of a column, NQ quantities at once; every thread of the column gets the totals
template <int NQ, int NR = SR>
__device__ __forceinline__ void col_reduce(float (&q)[NQ], float (*sh)[NR][SC]) {
    const int cl = threadIdx.x & (SC - 1), rl = threadIdx.x / SC;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i) sh[i][rl][cl] = q[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) t += sh[i][r][cl];
        q[i] = t;
    }
}

// "Resident" variants (round 4) of the two BatchNorm kernels below, for B <= SRF * 16 = 1024 rows: 1024 threads = 16
// columns x 64 row-lanes, every thread keeps its RT = ceil(B / 64) rows of every operand in registers — ONE round of loads
// (up to 80 in flight per thread) instead of one per 8-row batch and sweep (24 dependent round trips in the second-order
// backward: 44 us at B = 1024, N = 512; 32 us in the tangent).  Same formulas; the row sums run over 64 lanes instead of 16.
constexpr int SRF = 64;
__device__ __forceinline__ unsigned frow(int rl, int u, int B, int nrl = SRF) {
    const int b = rl + u * nrl;
    return (unsigned)(b < B ? b : B - 1);
}
// Column totals for the resident variants (CW columns x 1024 / CW row-lanes): the 64 / CW row-lanes of a column inside a
// wave meet through xor-shuffles (both partners form the same sum), the 16 waves through LDS in wave order — 16 LDS reads per
// quantity and thread.
template <int NQ, int CW>
__device__ __forceinline__ void col_reduce_res(float (&q)[NQ], float (*sh)[16][SC], float (*tot)[SC]) {
    const int cl = threadIdx.x & (CW - 1), wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
#pragma unroll
        for (int d = CW; d < 64; d <<= 1) q[i] += __shfl_xor(q[i], d);
    }
    __syncthreads();
    if ((threadIdx.x & 63) < CW) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) sh[i][wv][cl] = q[i];
    }
    __syncthreads();
    // NQ * CW threads add the 16 waves' sums of one (quantity, column) each — in wave order —, everybody reads the totals
    // (each thread summing all NQ * 16 values itself kept ~100 registers busy with LDS reads in flight)
    if ((int)threadIdx.x < NQ * CW) {
        const int i = threadIdx.x / CW, c = threadIdx.x - i * CW;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sh[i][w][c];
        tot[i][c] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = tot[i][cl];
}

// base[i] through a 32-bit byte offset from the uniform base (SGPR pair + one VGPR per load in flight: B <= 1024 rows of at
// most a few thousand floats)
__device__ __forceinline__ float ldf(const float* __restrict__ base, unsigned i) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * 4u);
}

// x_i = alpha * real + (1 - alpha) * fake for the 12 + nj pose columns; alphas (B,3) = [rot | transl | joint]
__global__ void gp_interp_kernel(const float* __restrict__ alphas, const float* __restrict__ r_rot,
                                 const float* __restrict__ r_tr, const float* __restrict__ r_jt,
                                 const float* __restrict__ f_rot, const float* __restrict__ f_tr,
                                 const float* __restrict__ f_jt, int B, int nj, float* __restrict__ o_rot,
                                 float* __restrict__ o_tr, float* __restrict__ o_jt) {
    const int np = 12 + nj;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * np) return;
    const int b = i / np, j = i - b * np;
    if (j < 9) {
        const float a = alphas[b * 3 + 0];
        o_rot[b * 9 + j] = a * r_rot[b * 9 + j] + (1.f - a) * f_rot[b * 9 + j];
    } else if (j < 12) {
        const float a = alphas[b * 3 + 1];
        o_tr[b * 3 + j - 9] = a * r_tr[b * 3 + j - 9] + (1.f - a) * f_tr[b * 3 + j - 9];
    } else {
        const float a = alphas[b * 3 + 2];
        o_jt[b * nj + j - 12] = a * r_jt[b * nj + j - 12] + (1.f - a) * f_jt[b * nj + j - 12];
    }
}

// Wasserstein terms (dexgangrasp.py:199-211, :258): loss_out = mean(p); d_logit = sign * p (1 - p) / B
__global__ __launch_bounds__(256) void wass_kernel(const float* __restrict__ p, int B, float sign, float* __restrict__ loss_out,
                                                   float* __restrict__ d_logit, unsigned* __restrict__ clear_ticket) {
    __shared__ float sh[256];
    float s = 0.f;
    if (clear_ticket && threadIdx.x == 0) clear_ticket[0] = 0u;   // GenLossFin::ticket
    for (int b = threadIdx.x; b < B; b += 256) {
        const float v = p[b];
        s += v;
        if (d_logit) d_logit[b] = sign * v * (1.f - v) / (float)B;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_out) loss_out[0] = sh[0] / (float)B;
}

// d_logit = p (1 - p): upstream gradient 1 on every sigmoid output (grad_outputs = ones, dexgangrasp.py:160)
__global__ void sig_grad_kernel(const float* __restrict__ p, int B, float* __restrict__ d_logit) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) d_logit[b] = p[b] * (1.f - p[b]);
}

// GP value and v = gain * dGP/dg from the input gradient g (B, 12 + nj): parts rot (9), transl (3), joint (nj).
// One thread per row, all of the row's values requested at once (single dwords: rows of 12 + nj floats have no 16-byte
// alignment to offer); gp_part[workgroup] = the workgroup's sum of (|g_part| - 1)^2 — wgan_combine_kernel adds them up
// in workgroup order.  (Round 3: one workgroup walked all rows, four per thread, value by value: 47 us at B = 1024.)
constexpr int kGpMaxCols = 48;
__global__ __launch_bounds__(256) void gp_value_kernel(const float* __restrict__ g, int B, int nj, float gain,
                                                       float* __restrict__ gp_part, float* __restrict__ v) {
    __shared__ float sh[256];
    const int np = 12 + nj;
    const int b = blockIdx.x * 256 + threadIdx.x;
    const bool bok = b < B;
    const float* gr = g + (size_t)(bok ? b : B - 1) * np;
    float x[kGpMaxCols];
#pragma unroll
    for (int j = 0; j < kGpMaxCols; ++j) x[j] = gr[j < np ? j : np - 1];
    float pen = 0.f;
    float k[3];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const int lo = part == 0 ? 0 : part == 1 ? 9 : 12, hi = part == 0 ? 9 : part == 1 ? 12 : np;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kGpMaxCols; ++j)
            if (j >= lo && j < hi) s = fmaf(x[j], x[j], s);
        const float n = sqrtf(s);
        pen += (n - 1.f) * (n - 1.f);
        // d/dg of mean_parts mean_b (n - 1)^2 = 2 (n - 1) g / n / (3 B)
        k[part] = n > 0.f ? gain * 2.f * (n - 1.f) / (n * 3.f * (float)B) : 0.f;
    }
    if (bok) {
#pragma unroll
        for (int j = 0; j < kGpMaxCols; ++j)
            if (j < np) v[(size_t)b * np + j] = (j < 9 ? k[0] : j < 12 ? k[1] : k[2]) * x[j];
    }
    sh[threadIdx.x] = bok ? pen : 0.f;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) gp_part[blockIdx.x] = sh[0];
}

__global__ void wgan_combine_kernel(float* losses, float gain, int gp_slot, const float* __restrict__ gp_part, int npart,
                                    int B) {
    float s = 0.f;
    for (int i = 0; i < npart; ++i) s += gp_part[i];
    const float gp = s / (3.f * (float)B);
    losses[gp_slot] = gp;
    losses[DGG_L_TOTAL_DISC] = -losses[DGG_L_BCE_REAL] + losses[DGG_L_BCE_FAKE] + gain * gp;
}

// Tangent of a train-mode BatchNorm1d (+ optional addend, + optional LeakyReLU mask):
//   x_hat = (a - mean) rstd;  x_hat_dot = rstd (adot - mean(adot) - x_hat mean(x_hat adot))
//   out = lrelu'(mask_ref) * (gamma x_hat_dot + addend)          (mask_ref / addend may be NULL; gamma NULL = 1)
// xhd_out (optional): x_hat_dot itself.  grid: ceil(N / 16)
__global__ __launch_bounds__(256) void bn_jvp_kernel(const float* __restrict__ a, int lda, const float* __restrict__ adot,
                                                     int ldad, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ addend, int ldadd,
                                                     const float* __restrict__ mask_ref, int ldm, int B, int N,
                                                     float* __restrict__ out, int ldo, float* __restrict__ xhd_out,
                                                     int ldx, int a_is_xhat) {
    __shared__ float sh[2][SR][SC];
    const int cl = threadIdx.x & (SC - 1), rl = threadIdx.x / SC;
    const int c0 = blockIdx.x * SC + cl;
    const bool ok = c0 < N;
    const int c = ok ? c0 : N - 1;
    const float mu = a_is_xhat ? 0.f : mean[c], rs = rstd[c];
    float q[2] = {0.f, 0.f};
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        float av[SU], dv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            av[u] = a[srow(b0, u, B) * lda + c];
            dv[u] = adot[srow(b0, u, B) * ldad + c];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (b0 + u * SR < B) {
                const float xh = a_is_xhat ? av[u] : (av[u] - mu) * rs;
                q[0] += dv[u];
                q[1] = fmaf(xh, dv[u], q[1]);
            }
        }
    }
    col_reduce<2>(q, sh);
    const float mad = q[0] / (float)B, m = q[1] / (float)B;
    const float ga = gamma ? gamma[c] : 1.f;
    if (!ok) return;
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        float av[SU], dv[SU], ev[SU], mv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            av[u] = a[srow(b0, u, B) * lda + c];
            dv[u] = adot[srow(b0, u, B) * ldad + c];
            ev[u] = addend ? addend[srow(b0, u, B) * ldadd + c] : 0.f;
            mv[u] = mask_ref ? mask_ref[srow(b0, u, B) * ldm + c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int b = b0 + u * SR;
            if (b < B) {
                const float xh = a_is_xhat ? av[u] : (av[u] - mu) * rs;
                const float xhd = rs * (dv[u] - mad - xh * m);
                if (xhd_out) xhd_out[(size_t)b * ldx + c] = xhd;
                float t = ga * xhd;
                if (addend) t += ev[u];
                if (mask_ref) t *= lrelu_d(mv[u]);
                out[(size_t)b * ldo + c] = t;
            }
        }
    }
}

// x *= lrelu'(ref), in place (the fc3 branch: s3_dot = lrelu'(a3) a3_dot; ref stores lrelu(a3), same sign)
__global__ void mask_inplace_kernel(float* __restrict__ x, int ldx, const float* __restrict__ ref, int ldr, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, c = i - b * N;
    x[(size_t)b * ldx + c] *= lrelu_d(ref[(size_t)b * ldr + c]);
}

// Sigmoid head, second order.  l = y.w + b, s = sigmoid(l), ldot = ydot.w, F = sum_b s (1 - s) ldot:
//   c1 = dF/dldot = s (1 - s);  c2 = dF/dl = s (1 - s)(1 - 2 s) ldot.      One wave per row.
__global__ __launch_bounds__(256) void head2_kernel(const float* __restrict__ ydot, int ld, const float* __restrict__ w,
                                                    const float* __restrict__ p, int B, int K, float* __restrict__ c1,
                                                    float* __restrict__ c2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(ydot[(size_t)row * ld + k], w[k], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        const float s = p[row];
        const float d = s * (1.f - s);
        c1[row] = d;
        c2[row] = d * (1.f - 2.f * s) * acc;
    }
}

// head parameter gradients: gw[c] += sum_b c2_b y[b][c] + c1_b ydot[b][c];  gb += sum_b c2_b.   grid: ceil(K / 16)
__global__ __launch_bounds__(256) void head2_grads_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ ydot,
                                                          int ldyd, const float* __restrict__ c1,
                                                          const float* __restrict__ c2, int B, int K,
                                                          float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float sh[2][SR][SC];
    const int cl = threadIdx.x & (SC - 1), rl = threadIdx.x / SC;
    const int c0 = blockIdx.x * SC + cl;
    const bool ok = c0 < K;
    const int c = ok ? c0 : K - 1;
    float q[2] = {0.f, 0.f};
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        float yv[SU], dv[SU], k1[SU], k2[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            yv[u] = y[srow(b0, u, B) * ldy + c];
            dv[u] = ydot[srow(b0, u, B) * ldyd + c];
            k1[u] = c1[srow(b0, u, B)];
            k2[u] = c2[srow(b0, u, B)];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (b0 + u * SR < B) {
                q[0] = fmaf(k2[u], yv[u], q[0]);
                q[0] = fmaf(k1[u], dv[u], q[0]);
                q[1] += k2[u];
            }
        }
    }
    col_reduce<2>(q, sh);
    if (ok && rl == 0) {
        gw[c] += q[0];
        if (c0 == 0) gb[0] += q[1];
    }
}

// Output stage of a ResBlock, reverse: y = lrelu(p), p = s3 + z2, s3 = lrelu(a3).
//   pbar = lrelu'(y) ybar;  a3bar = lrelu'(s3) pbar   (and the same for the tangent adjoints).
// Head mode (c1 != NULL): ybar = c2_b w_c, ydotbar = c1_b w_c formed on the fly.
__global__ void out2_bwd_kernel(const float* __restrict__ ybar, const float* __restrict__ ydbar, int ldyb,
                                const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ w,
                                const float* __restrict__ y, int ldy, const float* __restrict__ s3, int lds, int B, int N,
                                float* __restrict__ pbar, float* __restrict__ pdbar, int ldp, float* __restrict__ a3bar,
                                float* __restrict__ a3dbar, int lda3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, c = i - b * N;
    const float my = lrelu_d(y[(size_t)b * ldy + c]), m3 = lrelu_d(s3[(size_t)b * lds + c]);
    const float yb = c1 ? c2[b] * w[c] : ybar[(size_t)b * ldyb + c];
    const float ydb = c1 ? c1[b] * w[c] : ydbar[(size_t)b * ldyb + c];
    const float pb = my * yb, pdb = my * ydb;
    pbar[(size_t)b * ldp + c] = pb;
    pdbar[(size_t)b * ldp + c] = pdb;
    a3bar[(size_t)b * lda3 + c] = m3 * pb;
    a3dbar[(size_t)b * lda3 + c] = m3 * pdb;
}

// Second-order backward of a train-mode BatchNorm1d.  Forward (primal) z = gamma x_hat + beta, tangent
// zdot = gamma x_hat_dot with x_hat_dot = rstd (adot_c - x_hat m), adot_c = adot - mean(adot), m = mean(x_hat adot).
// Given the adjoints zbar, zdbar of both outputs (already masked by lrelu'(mask_ref) when the BatchNorm feeds a
// LeakyReLU — folded in here), with w = gamma zdbar:
//   adotbar = rstd (w - mean(w) - x_hat mean(w x_hat))
//   xhbar   = -rstd (w m + adot_c mean(w x_hat)),   X = gamma zbar + xhbar
//   abar    = rstd (X - mean(X) - x_hat mean(X x_hat)) - rstd x_hat mean(w x_hat_dot)      [rstd's own dependence on a]
//   gamma_bar += sum zbar x_hat + sum zdbar x_hat_dot;   beta_bar += sum zbar
// grid: ceil(N / 16); three sweeps over the rows.
__global__ __launch_bounds__(256) void bn_second_bwd_kernel(
    const float* __restrict__ a, int lda, const float* __restrict__ adot, int ldad, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ zbar,
    const float* __restrict__ zdbar, int ldz, const float* __restrict__ mask_ref, int ldm, int B, int N,
    float* __restrict__ abar, float* __restrict__ adbar, int ldo, float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    __shared__ float sh[6][SR][SC];
    const int cl = threadIdx.x & (SC - 1), rl = threadIdx.x / SC;
    const int c0 = blockIdx.x * SC + cl;
    const bool ok = c0 < N;
    const int c = ok ? c0 : N - 1;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c];
    const float invB = 1.f / (float)B;
    // one batch of rows: x_hat, adot, mask, zbar, zdbar (masked) of the thread's SU rows
    struct Rows {
        float xh[SU], ad[SU], zb[SU], zdb[SU];
    };
    auto load = [&](int b0, Rows& R) {
        float av[SU], mv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            av[u] = a[srow(b0, u, B) * lda + c];
            R.ad[u] = adot[srow(b0, u, B) * ldad + c];
            mv[u] = mask_ref ? mask_ref[srow(b0, u, B) * ldm + c] : 1.f;
            R.zb[u] = zbar[srow(b0, u, B) * ldz + c];
            R.zdb[u] = zdbar[srow(b0, u, B) * ldz + c];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const float m_ = mask_ref ? lrelu_d(mv[u]) : 1.f;
            R.xh[u] = (av[u] - mu) * rs;
            R.zb[u] = m_ * R.zb[u];
            R.zdb[u] = m_ * R.zdb[u];
        }
    };
    // sweep 1
    float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // sum adot, sum xh adot, sum w, sum w xh, sum zbar, sum zbar xh
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        Rows R;
        load(b0, R);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (b0 + u * SR < B) {
                const float xh = R.xh[u], ad = R.ad[u], zb = R.zb[u];
                const float w = ga * R.zdb[u];
                q[0] += ad;
                q[1] = fmaf(xh, ad, q[1]);
                q[2] += w;
                q[3] = fmaf(w, xh, q[3]);
                q[4] += zb;
                q[5] = fmaf(zb, xh, q[5]);
            }
        }
    }
    col_reduce<6>(q, sh);
    const float mad = q[0] * invB, m = q[1] * invB, mw = q[2] * invB, mwx = q[3] * invB;
    const float s_zb = q[4], s_zbx = q[5];
    // sweep 2
    float r2[4] = {0.f, 0.f, 0.f, 0.f};  // sum w xhd, sum X, sum X xh, sum zdbar xhd
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        Rows R;
        load(b0, R);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (b0 + u * SR < B) {
                const float xh = R.xh[u];
                const float adc = R.ad[u] - mad;
                const float zb = R.zb[u], zdb = R.zdb[u];
                const float w = ga * zdb;
                const float xhd = rs * (adc - xh * m);
                const float X = ga * zb - rs * (w * m + adc * mwx);
                r2[0] = fmaf(w, xhd, r2[0]);
                r2[1] += X;
                r2[2] = fmaf(X, xh, r2[2]);
                r2[3] = fmaf(zdb, xhd, r2[3]);
            }
        }
    }
    col_reduce<4>(r2, reinterpret_cast<float(*)[SR][SC]>(sh));
    const float s_wxd = r2[0], mX = r2[1] * invB, mXx = r2[2] * invB;
    if (!ok) return;
    // sweep 3
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        Rows R;
        load(b0, R);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int b = b0 + u * SR;
            if (b < B) {
                const float xh = R.xh[u];
                const float adc = R.ad[u] - mad;
                const float zb = R.zb[u], w = ga * R.zdb[u];
                const float X = ga * zb - rs * (w * m + adc * mwx);
                abar[(size_t)b * ldo + c] = rs * (X - mX - xh * mXx) - rs * xh * (s_wxd * invB);
                adbar[(size_t)b * ldo + c] = rs * (w - mw - xh * mwx);
            }
        }
    }
    if (rl == 0) {
        g_gamma[c] += s_zbx + r2[3];
        g_beta[c] += s_zb;
    }
}

// ---- resident variants (see frow): B <= 64 * RT rows
template <int RT>
__global__ __launch_bounds__(1024) void bn_jvp_res_kernel(const float* __restrict__ a, int lda, const float* __restrict__ adot,
                                                          int ldad, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ addend, int ldadd,
                                                          const float* __restrict__ mask_ref, int ldm, int B, int N,
                                                          float* __restrict__ out, int ldo, float* __restrict__ xhd_out,
                                                          int ldx, int a_is_xhat) {
    __shared__ float sh[2][16][SC];
    __shared__ float tot[2][SC];
    const int cl = threadIdx.x & (SC - 1), rl = threadIdx.x / SC;
    const int c0 = blockIdx.x * SC + cl;
    const bool ok = c0 < N;
    const int c = ok ? c0 : N - 1;
    const float mu = a_is_xhat ? 0.f : ldf(mean, (unsigned)c), rs = ldf(rstd, (unsigned)c);
    const float ga = gamma ? ldf(gamma, (unsigned)c) : 1.f;
    float xh[RT], dv[RT], ev[RT], mv[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        xh[u] = ldf(a, frow(rl, u, B) * (unsigned)lda + (unsigned)c);
        dv[u] = ldf(adot, frow(rl, u, B) * (unsigned)ldad + (unsigned)c);
        ev[u] = 0.f;
        mv[u] = 1.f;
    }
    if (addend) {
#pragma unroll
        for (int u = 0; u < RT; ++u) ev[u] = ldf(addend, frow(rl, u, B) * (unsigned)ldadd + (unsigned)c);
    }
    if (mask_ref) {
#pragma unroll
        for (int u = 0; u < RT; ++u) mv[u] = lrelu_d(ldf(mask_ref, frow(rl, u, B) * (unsigned)ldm + (unsigned)c));
    }
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        if (!a_is_xhat) xh[u] = (xh[u] - mu) * rs;
        if (rl + u * SRF < B) {
            q[0] += dv[u];
            q[1] = fmaf(xh[u], dv[u], q[1]);
        }
    }
    col_reduce_res<2, SC>(q, sh, tot);
    const float mad = q[0] / (float)B, m = q[1] / (float)B;
    if (!ok) return;
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const int b = rl + u * SRF;
        if (b < B) {
            const float xhd = rs * (dv[u] - mad - xh[u] * m);
            if (xhd_out) xhd_out[(size_t)b * ldx + c] = xhd;
            out[(size_t)b * ldo + c] = (ga * xhd + ev[u]) * mv[u];
        }
    }
}

template <int RT, int CW>
__global__ __launch_bounds__(1024) void bn_second_bwd_res_kernel(
    const float* __restrict__ a, int lda, const float* __restrict__ adot, int ldad, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ zbar,
    const float* __restrict__ zdbar, int ldz, const float* __restrict__ mask_ref, int ldm, int B, int N,
    float* __restrict__ abar, float* __restrict__ adbar, int ldo, float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    __shared__ float sh[6][16][SC];
    __shared__ float tot[6][SC];
    constexpr int NRL = 1024 / CW;   // row-lanes
    const int cl = threadIdx.x & (CW - 1), rl = threadIdx.x / CW;
    const int c0 = blockIdx.x * CW + cl;
    const bool ok = c0 < N;
    const int c = ok ? c0 : N - 1;
    const float mu = ldf(mean, (unsigned)c), rs = ldf(rstd, (unsigned)c), ga = ldf(gamma, (unsigned)c);
    const float invB = 1.f / (float)B;
    float xh[RT], ad[RT], zb[RT], zdb[RT];
    {
        float mv[RT];
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            xh[u] = ldf(a, frow(rl, u, B, NRL) * (unsigned)lda + (unsigned)c);
            ad[u] = ldf(adot, frow(rl, u, B, NRL) * (unsigned)ldad + (unsigned)c);
            zb[u] = ldf(zbar, frow(rl, u, B, NRL) * (unsigned)ldz + (unsigned)c);
            zdb[u] = ldf(zdbar, frow(rl, u, B, NRL) * (unsigned)ldz + (unsigned)c);
            mv[u] = 1.f;
        }
        if (mask_ref) {
#pragma unroll
            for (int u = 0; u < RT; ++u) mv[u] = lrelu_d(ldf(mask_ref, frow(rl, u, B, NRL) * (unsigned)ldm + (unsigned)c));
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            xh[u] = (xh[u] - mu) * rs;
            zb[u] *= mv[u];
            zdb[u] *= mv[u];
        }
    }
    // sweep 1
    float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // sum adot, sum xh adot, sum w, sum w xh, sum zbar, sum zbar xh
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        if (rl + u * NRL < B) {
            const float w = ga * zdb[u];
            q[0] += ad[u];
            q[1] = fmaf(xh[u], ad[u], q[1]);
            q[2] += w;
            q[3] = fmaf(w, xh[u], q[3]);
            q[4] += zb[u];
            q[5] = fmaf(zb[u], xh[u], q[5]);
        }
    }
    col_reduce_res<6, CW>(q, sh, tot);
    const float mad = q[0] * invB, m = q[1] * invB, mw = q[2] * invB, mwx = q[3] * invB;
    const float s_zb = q[4], s_zbx = q[5];
    // sweep 2
    float r2[4] = {0.f, 0.f, 0.f, 0.f};  // sum w xhd, sum X, sum X xh, sum zdbar xhd
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        if (rl + u * NRL < B) {
            const float adc = ad[u] - mad;
            const float w = ga * zdb[u];
            const float xhd = rs * (adc - xh[u] * m);
            const float X = ga * zb[u] - rs * (w * m + adc * mwx);
            r2[0] = fmaf(w, xhd, r2[0]);
            r2[1] += X;
            r2[2] = fmaf(X, xh[u], r2[2]);
            r2[3] = fmaf(zdb[u], xhd, r2[3]);
        }
    }
    col_reduce_res<4, CW>(r2, reinterpret_cast<float(*)[16][SC]>(sh), tot);
    const float s_wxd = r2[0], mX = r2[1] * invB, mXx = r2[2] * invB;
    if (!ok) return;
    // sweep 3
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const int b = rl + u * NRL;
        if (b < B) {
            const float adc = ad[u] - mad;
            const float w = ga * zdb[u];
            const float X = ga * zb[u] - rs * (w * m + adc * mwx);
            abar[(size_t)b * ldo + c] = rs * (X - mX - xh[u] * mXx) - rs * xh[u] * (s_wxd * invB);
            adbar[(size_t)b * ldo + c] = rs * (w - mw - xh[u] * mwx);
        }
    }
    if (rl == 0) {
        g_gamma[c] += s_zbx + r2[3];
        g_beta[c] += s_zb;
    }
}

// dst[c] (+)= sum_b x[b][c] * (y ? y[b][c] : 1).   grid: ceil(N / 16)
__global__ __launch_bounds__(256) void colsum_prod_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y,
                                                          int ldy, int B, int N, float* __restrict__ dst, int accumulate) {
    __shared__ float sh[1][SR][SC];
    const int cl = threadIdx.x & (SC - 1), rl = threadIdx.x / SC;
    const int c0 = blockIdx.x * SC + cl;
    const bool ok = c0 < N;
    const int c = ok ? c0 : N - 1;
    float q[1] = {0.f};
    for (int b0 = rl; b0 < B; b0 += SR * SU) {
        float xv[SU], yv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            xv[u] = x[srow(b0, u, B) * ldx + c];
            yv[u] = y ? y[srow(b0, u, B) * ldy + c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u)
            if (b0 + u * SR < B) q[0] = fmaf(xv[u], yv[u], q[0]);
    }
    col_reduce<1>(q, sh);
    if (ok && rl == 0) dst[c] = (accumulate ? dst[c] : 0.f) + q[0];
}

static inline int cdiv2(int a, int b) { return (a + b - 1) / b; }

int gp_interp_launch(const float* alphas, const float* r_rot, const float* r_tr, const float* r_jt, const float* f_rot,
                     const float* f_tr, const float* f_jt, int B, int nj, float* o_rot, float* o_tr, float* o_jt,
                     hipStream_t st) {
    hipLaunchKernelGGL(gp_interp_kernel, dim3(cdiv2(B * (12 + nj), 256)), dim3(256), 0, st, alphas, r_rot, r_tr, r_jt,
                       f_rot, f_tr, f_jt, B, nj, o_rot, o_tr, o_jt);
    return check_launch("gp_interp");
}
int wass_launch(const float* p, int B, float sign, float* loss_out, float* d_logit, hipStream_t st, unsigned* clear_ticket) {
    hipLaunchKernelGGL(wass_kernel, dim3(1), dim3(256), 0, st, p, B, sign, loss_out, d_logit, clear_ticket);
    return check_launch("wass");
}
int sig_grad_launch(const float* p, int B, float* d_logit, hipStream_t st) {
    hipLaunchKernelGGL(sig_grad_kernel, dim3(cdiv2(B, 256)), dim3(256), 0, st, p, B, d_logit);
    return check_launch("sig_grad");
}
int gp_value_parts(int B) { return cdiv2(B, 256); }
int gp_value_launch(const float* g, int B, int nj, float gain, float* gp_part, float* v, hipStream_t st) {
    if (12 + nj > kGpMaxCols) return DGG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gp_value_kernel, dim3(gp_value_parts(B)), dim3(256), 0, st, g, B, nj, gain, gp_part, v);
    return check_launch("gp_value");
}
int wgan_combine_launch(float* losses, float gain, int gp_slot, const float* gp_part, int B, hipStream_t st) {
    hipLaunchKernelGGL(wgan_combine_kernel, dim3(1), dim3(1), 0, st, losses, gain, gp_slot, gp_part, gp_value_parts(B), B);
    return check_launch("wgan_combine");
}
int bn_jvp_launch(const float* a, int lda, const float* adot, int ldad, const float* mean, const float* rstd,
                  const float* gamma, const float* addend, int ldadd, const float* mask_ref, int ldm, int B, int N,
                  float* out, int ldo, float* xhd_out, int ldx, int a_is_xhat, hipStream_t st) {
    const dim3 grid(cdiv2(N, SC));
#define DGG_JVP_RES(RT)                                                                                                    \
    hipLaunchKernelGGL((bn_jvp_res_kernel<RT>), grid, dim3(1024), 0, st, a, lda, adot, ldad, mean, rstd, gamma, addend, ldadd, \
                       mask_ref, ldm, B, N, out, ldo, xhd_out, ldx, a_is_xhat)
    if (B <= 4 * SRF) DGG_JVP_RES(4);
    else if (B <= 8 * SRF) DGG_JVP_RES(8);
    else if (B <= 16 * SRF) DGG_JVP_RES(16);
    else
        hipLaunchKernelGGL(bn_jvp_kernel, grid, dim3(256), 0, st, a, lda, adot, ldad, mean, rstd, gamma, addend, ldadd,
                           mask_ref, ldm, B, N, out, ldo, xhd_out, ldx, a_is_xhat);
#undef DGG_JVP_RES
    return check_launch("bn_jvp");
}
int mask_inplace_launch(float* x, int ldx, const float* ref, int ldr, int B, int N, hipStream_t st) {
    hipLaunchKernelGGL(mask_inplace_kernel, dim3(cdiv2(B * N, 256)), dim3(256), 0, st, x, ldx, ref, ldr, B, N);
    return check_launch("mask_inplace");
}
int head2_launch(const float* ydot, int ld, const float* w, const float* p, int B, int K, float* c1, float* c2,
                 hipStream_t st) {
    hipLaunchKernelGGL(head2_kernel, dim3(cdiv2(B, 4)), dim3(256), 0, st, ydot, ld, w, p, B, K, c1, c2);
    return check_launch("head2");
}
int head2_grads_launch(const float* y, int ldy, const float* ydot, int ldyd, const float* c1, const float* c2, int B,
                       int K, float* gw, float* gb, hipStream_t st) {
    hipLaunchKernelGGL(head2_grads_kernel, dim3(cdiv2(K, SC)), dim3(256), 0, st, y, ldy, ydot, ldyd, c1, c2, B, K, gw, gb);
    return check_launch("head2_grads");
}
int out2_bwd_launch(const float* ybar, const float* ydbar, int ldyb, const float* c1, const float* c2, const float* w,
                    const float* y, int ldy, const float* s3, int lds, int B, int N, float* pbar, float* pdbar, int ldp,
                    float* a3bar, float* a3dbar, int lda3, hipStream_t st) {
    hipLaunchKernelGGL(out2_bwd_kernel, dim3(cdiv2(B * N, 256)), dim3(256), 0, st, ybar, ydbar, ldyb, c1, c2, w, y, ldy, s3,
                       lds, B, N, pbar, pdbar, ldp, a3bar, a3dbar, lda3);
    return check_launch("out2_bwd");
}
int bn_second_bwd_launch(const float* a, int lda, const float* adot, int ldad, const float* mean, const float* rstd,
                         const float* gamma, const float* zbar, const float* zdbar, int ldz, const float* mask_ref, int ldm,
                         int B, int N, float* abar, float* adbar, int ldo, float* g_gamma, float* g_beta, hipStream_t st) {
#define DGG_BN2_RES(RT, CW)                                                                                                \
    hipLaunchKernelGGL((bn_second_bwd_res_kernel<RT, CW>), dim3(cdiv2(N, CW)), dim3(1024), 0, st, a, lda, adot, ldad, mean,  \
                       rstd, gamma, zbar, zdbar, ldz, mask_ref, ldm, B, N, abar, adbar, ldo, g_gamma, g_beta)
    // (8 rows of five operands per thread at most: 16 would not fit the 128 registers of a 1024-thread workgroup)
    if (B <= 4 * 64) DGG_BN2_RES(4, 16);
    else if (B <= 8 * 64) DGG_BN2_RES(8, 16);
    else if (B <= 8 * 128) DGG_BN2_RES(8, 8);
    else
        hipLaunchKernelGGL(bn_second_bwd_kernel, dim3(cdiv2(N, SC)), dim3(256), 0, st, a, lda, adot, ldad, mean, rstd, gamma, zbar,
                           zdbar, ldz, mask_ref, ldm, B, N, abar, adbar, ldo, g_gamma, g_beta);
#undef DGG_BN2_RES
    return check_launch("bn_second_bwd");
}
int colsum_prod_launch(const float* x, int ldx, const float* y, int ldy, int B, int N, float* dst, int accumulate,
                       hipStream_t st) {
    hipLaunchKernelGGL(colsum_prod_kernel, dim3(cdiv2(N, SC)), dim3(256), 0, st, x, ldx, y, ldy, B, N, dst, accumulate);
    return check_launch("colsum_prod");
}

}  // namespace dgg
