// Internal declarations shared by the HIP translation units of libdgg.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dgg.h"

namespace dgg {

constexpr float kLrelu = 0.2f;   // networks.py:41
constexpr float kBnEps = 1e-5f;  // nn.BatchNorm1d default
constexpr float kBnMom = 0.1f;
constexpr float kDropP = 0.3f;   // networks.py:294
constexpr int kRowBlock = 32;    // rows per partial-statistics block

void set_error(const char* what, hipError_t e);
void set_error_text(const char* fmt, ...);   // dgg_last_error() text for an error without a HIP status
int check_launch(const char* what);

// ---------------------------------------------------------------- GEMM (gemm_f32.hip)
// C[M,N] (+)= sum_k A(m,k) * B(k,n) on v_mfma_f32_32x32x2_f32 (exact fp32).
//   a_kc: A is stored [M][K] (k contiguous) else [K][M];  b_kc: B stored [N][K] else [K][N].
struct GemmProb {
    const float* A;
    const float* B;
    float* C;          // direct output, or slab base when splits > 0 (slab s at C + s*slab_stride)
    int M, N, K;
    int lda, ldb, ldc;
    int splits;        // 0: write C directly (accumulate honoured); >=1: write `splits` slabs
    int kchunk;        // K range per split (multiple of BK)
    long long slab_stride;
    int accumulate;
    int tiles_m, tiles_n;
    int blk_begin;     // first block id of this problem in the grouped launch
    // NT only (may be NULL): per-k affine applied to the A operand while it is staged,
    //   A'(m,k) = A(m,k) * a_scale[k] + a_shift[k]   — the input BatchNorm's gamma/beta folded into the
    //   forward GEMM, so the stored operand is the un-affined normalised input x_hat.
    const float* a_scale;
    const float* a_shift;
    // TN only (ep_gamma != NULL): BatchNorm-aware weight-gradient epilogue.  The accumulator holds
    //   T[n][k] = sum_m dY[m][n] x_hat[m][k];  the kernel stores dW[n][k] = gamma[k] T + beta[k] cs[n]
    //   and emits per-tile column partials  sum_n W[n][k] T[n][k]  (-> d gamma[k])  and
    //   sum_n W[n][k] cs[n]  (-> d beta[k]), so the input gradient dX0 is never formed.
    const float* ep_gamma;
    const float* ep_beta;
    const float* ep_cs0;   // column sums of dY for output rows < ep_split (NULL = exactly zero)
    const float* ep_cs1;   // ... for rows >= ep_split (indexed row - ep_split)
    int ep_split;
    const float* ep_w;     // weights, indexed like C (row n, column k), pitch ep_ldw
    int ep_ldw;
    float* ep_part;        // [tiles_m][2][ep_part_ld]
    int ep_part_ld;        // pitch of ep_part rows (0 = N): a column sub-range of a wider matrix keeps the wide layout
    const float* ep_cs1b;  // optional: a second vector added to ep_cs1 (column sums of ANOTHER pass' dY summed into A)
    // gemm_f32 NT with 64 x 64 tiles only ("fc2 forward", bn_part != NULL): the A operand is a PRE-activation whose
    // train-mode BatchNorm + LeakyReLU is applied while it is staged — every workgroup first merges the row-block
    // partials bn_part [bn_nrb][2][K] of all K <= 256 columns (same order everywhere), the first one also stores
    // mean / rstd and advances the running statistics; the activated operand is written to a_out (pitch K) by the first
    // column of tiles.  Epilogue: C = acc + c_bias[n] and the (mean, M2) partials of every 32-row block of C to c_part
    // [row block][2][N] (what reduce_bias_kernel emits) — a whole ResBlock stage h = lrelu(bn1(a1)), a2 = h W2^T + b2
    // in one launch.
    const float* bn_part;
    int bn_nrb, bn_train, bn_rows;     // rows per batch (the partials' last block may be short)
    const float *bn_gamma, *bn_beta;
    float *bn_run_mean, *bn_run_var, *bn_mean_out, *bn_rstd_out;
    float* a_out;
    const float* c_bias;
    float* c_part;
};
constexpr int kMaxProb = 8;   // problems of one grouped launch (the weight gradients of all ResBlocks of a pass: <= 8)
struct GemmBatch {
    GemmProb p[kMaxProb];
    int nprob;
    int total_blocks;
    int diag;   // diagnostics (gemm_x5: bit 0 = element-wise epilogue everywhere, DGG_X5_WIDE=0); 0 in every product path
};
enum GemmMode { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum GemmRoute { GEMM_ROUTE_F32 = 0, GEMM_ROUTE_X4 = 4, GEMM_ROUTE_X5 = 5 };   // dgg.h: DGG_KERNEL_*
int gemm_route_single(int mode, int M, int N, int K, int lda, int ldb, int ldc, int splits);
// Fills tiling fields of one problem for the tile shape the launcher will use; returns #blocks.
int gemm_plan(GemmProb& p, int mode, int splits);
int gemm_launch(int mode, GemmBatch& batch, hipStream_t st);
// NT, one problem with GemmProb::bn_part set (see there): 64 x 64 tiles, no split-K, K <= 256
int gemm_fc2_fused_launch(GemmBatch& batch, hipStream_t st);
// fc2_stage.hip: the same stage with one 32 x 32 output tile per workgroup, its waves splitting K = 256
int fc2_stage_launch(const GemmProb& p, hipStream_t st);
// a narrow NN product in the same form: C (M, N <= 32) (+)= A (M, K) B (K, N), K a multiple of 64 up to 1024
int narrow_nn_launch(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K, int accumulate,
                     hipStream_t st);
// the generator's pose heads, out (B, nh <= 32) = y (B, 512) wh (nh, 512)^T + bh, in one launch of the same form
int heads_stage_launch(const float* y, int ldy, int B, const float* wh, const float* bh, int nh, float* out, int ldo,
                       hipStream_t st);
// ... and its backward mirror: dq = (da2 W2) * lrelu'(h) + bn1's backward partial sums per row block, one launch
int fc2_bwd_stage_launch(const float* da2, int ldda, const float* w2, int ldw, const float* h, int ldh, const float* a1,
                         int lda1, const float* mean, const float* rstd, int B, float* dq, int lddq, float* part,
                         hipStream_t st);
int gemm_pick_splits(int mode, int M, int N, int K, int other_blocks);
void gemm_tile_shape(int mode, int M, int N, int& bm, int& bn, int K = 1 << 30);
int gemm_occupancy(int mode, int bm, int bn);
int gemm_algo();
void gemm_set_algo(int a);
int gemm_side_work();            // per thread, see dgg_set_side_work
void gemm_set_side_work(int on);
// gemm_x4.hip: the bf16x3 products with the split done by loader waves (launches of >= 1024 workgroups; DGG_X4=1 / 0 forces)
void gemm_x4_dispatch(int mode, const GemmBatch& batch, hipStream_t st);
bool gemm_x4_enabled(long total_blocks);
// gemm_x5.hip: the same products with ONE wave per 128 x 128 accumulator block (the workgroup's waves split K); default
// for every launch it can take (DGG_X5=0 switches it off)
void gemm_x5_dispatch(int mode, const GemmBatch& batch, hipStream_t st);
bool gemm_x5_usable(const GemmBatch& batch);
bool gemm_x5_enabled();
bool gemm_x5_prob_ok(const GemmProb& p);
// dx / da3 of THIS pass plus the same tensor of another pass, written next to the plain output (the discriminator's
// weight gradient over the BPS columns takes dY_real + dY_fake in one product, see Share::bwd_mode)
struct SumOut {
    const float* other;   // NULL: none
    float* sum;
    int ld;               // pitch of both
};

// ---------------------------------------------------------------- BPS encode (bps_encode.hip)
int bps_encode_launch(const float* clouds, int B, int N, const float* basis, int P, float* dists, int32_t* ids, void* ws,
                      size_t ws_bytes, hipStream_t st);
long long bps_workspace_bytes(int B, int N, int P);
int bps_algo();
void bps_set_algo(int a);

// ---------------------------------------------------------------- element-wise / BN / losses (elementwise.hip)
struct ColSrc {           // one column block of a concatenated input
    const float* p;
    int ld;
    int n;
};
struct ColSrcs {
    ColSrc s[4];
    int nsrc;
    int total;            // sum of n
};

// Partial batch statistics (mean, M2) of the virtual concat of `srcs` per (group,row block,column).
int ibn_stats_launch(const ColSrcs& srcs, int rows_per_group, int groups, float* part, hipStream_t st);
// Finalize fused into the consumer (ibn_apply / bn_act / res_out): the kernel merges the partials of its own columns
// (train) or reads the running statistics (eval), uses them, and its first row of workgroups stores mean / rstd
// (mean_out / rstd_out: [groups][ncols]) and advances the running statistics.
struct BnFused {
    const float* part;
    int train;
    float *run_mean, *run_var, *mean_out, *rstd_out;
    float* varu_out = nullptr;  // optional (group 0): unbiased batch variance, for ibn_refresh_launch
};
// Advance the running statistics once more with stored batch statistics and rewrite the affine vectors (a second
// train-mode forward over the same inputs with updated gamma / beta).
int ibn_refresh_launch(const float* mean, const float* var_u, float* run_mean, float* run_var, const float* gamma,
                       const float* beta, float* aff_scale, float* aff_shift, int nprefix, int total, hipStream_t st);
// X0[m][col_off + c] = (x - mean)*rstd*gamma + beta over the concat; optionally copies `prefix`
// (B, nprefix) into columns [0, nprefix).
int ibn_apply_launch(const ColSrcs& srcs, int B, int rows_per_group, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, const float* prefix, int nprefix, float* X0, int ldx,
                     const float* aff_gamma, const float* aff_beta, float* aff_scale, float* aff_shift,
                     hipStream_t st, const BnFused* fused = nullptr,
                     float* X0b = nullptr, int ldb = 0, int nb = 0);   // the first nb BN columns stored once more (XShare)
// What is left of a consumer's input BatchNorm when the BPS block was normalised by another pass (elementwise.hip,
// ibn_shared_kernel): statistics / running statistics / affine vectors of the n_alias shared columns, the generator's
// latent prefix, and (npose > 0: the discriminator's fake pass) the fake pose from `heads` with its own BatchNorm.
struct IbnShared {
    int n_alias;
    const float *src_mean, *src_rstd, *src_varu;   // the producer's batch statistics of the shared columns
    float *mean, *rstd, *varu;                     // the consumer's own arrays [k_in]
    float *run_mean, *run_var;                     // ... its running statistics [k_in]
    const float *gamma, *beta;                     // ... its input BatchNorm's parameters [k_in]
    float *aff_scale, *aff_shift;                  // [nprefix + k_in]
    int nprefix;
    const float* prefix;                           // (B, nprefix) or NULL
    float* X0;                                     // the consumer's operand: row m = [prefix | shared x_hat | pose], pitch ldx
    int ldx, B;
    int npose, nj;
    const float* heads;                            // (B, ldh): [joint (nj) | rot6 | transl]
    int ldh;
    float *rot9, *transl, *joint;                  // contiguous copies (B,9) (B,3) (B,nj)
};
int ibn_shared_launch(const IbnShared& P, hipStream_t st);

// out[m][c] = sum_s slabs[s][m][c] (+ sum_s slabs2[s][m][c], same stride / pitch) + bias[c];  columns [stat_lo, stat_hi) get partial statistics,
// columns [act_lo, act_hi) get LeakyReLU applied before the store.
int reduce_bias_launch(const float* slabs, int nslab, long long slab_stride, int ld_slab, const float* bias,
                       float* out, int ld_out, int B, int rows_per_group, int groups, int N, int stat_lo,
                       int stat_hi, int act_lo, int act_hi, float* part, hipStream_t st, int accumulate = 0,
                       const float* slabs2 = nullptr, int nslab2 = 0,
                       long long stride2 = 0, int ld2 = 0);  // 0: the second stack has the first one's stride / pitch
// Per-object factorised first layer (eval mode): out[m][c] = cobj[m / n_per][c] + sum_s slabs[s][m][c]
//   + sum_j bn_eval(small)(m, j) * wsm[c * ldw + j]; LeakyReLU on [act_lo, act_hi).  small.total <= 16.
int expand_rows_launch(const float* cobj, int ldc, int n_per, const float* slabs, int nslab, long long slab_stride,
                       int ld_slab, const ColSrcs& small, const float* gamma, const float* beta, const float* rm,
                       const float* rv, const float* wsm, int ldw, float* out, int ld_out, int B, int N, int act_lo,
                       int act_hi, hipStream_t st,
                       float* h = nullptr, int ldh = 0, int hid = 0, const float* g1 = nullptr, const float* b1 = nullptr,
                       const float* rm1 = nullptr, const float* rv1 = nullptr);
// h[m][c] = lrelu((a[m][c]-mean)*rstd*gamma+beta)
int bn_act_launch(const float* a, int lda, int B, int rows_per_group, int N, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, float* h, int ldh, hipStream_t st,
                  const BnFused* fused = nullptr);
// y = lrelu(s + bn2(a2)) then dropout (keep mask given as floats, or drawn from Philox when
// drop_mode == 2); writes the keep mask (bytes) when dropping.  drop_mode: 0 none, 1 given, 2 rng.
int res_out_launch(const float* s, int lds, const float* a2, int lda2, int B, int rows_per_group, int N,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, int drop_mode,
                   const float* keep_in, uint64_t seed, uint64_t seed_off, uint8_t* keep_out, float* y, int ldy,
                   hipStream_t st, const BnFused* fused = nullptr, const float* a2_bias = nullptr);
// heads: out[m][j] = sum_k y[m][k] Wh[j][k] + bh[j]; sigmoid optionally.
int heads_fwd_launch(const float* y, int ldy, int B, int K, const float* wh, const float* bh, int nh, int sigmoid,
                     float* out, int ldo, float* scratch, hipStream_t st);
// ResBlock backward element-wise stages ----------------------------------------------------
// Stage 1 (head mode: dy = hd_dout (x) hd_w formed on the fly, nq == 5 adds the head's parameter-gradient sums):
//          dpre = dy * [keep/(1-p)] * lrelu'(y);  da3 = dpre * lrelu'(s) -> dY13[:, c3off + c];
//          partial sums: sum dpre, sum dpre*xhat2, sum da3.
int res_out_bwd_launch(const float* dy, int lddy, const float* y, int ldy, const uint8_t* keep, const float* s,
                       int lds, const float* a2, int lda2, const float* mean, const float* rstd, int B,
                       int rows_per_group, int N, float* dpre, float* da3, int ldda3, float* part,
                       hipStream_t st, const float* hd_dout = nullptr, const float* hd_w = nullptr, int nq = 3,
                       int nslab = 0, long long slab_stride = 0,  // nslab > 1: dy = split-K slabs, summed in order
                       const SumOut* sum = nullptr,
                       int hd_n = 1);   // heads of the head mode: 1 = d_score (x) w; > 1: dy = d_out (B, hd_n) w (hd_n, N), nq == 3
int heads_bias_grad_launch(const float* dout, int ldd, int B, int nh, float* gbh, int accumulate, hipStream_t st);
// Stage 2/4 (BN backward apply): dx = gamma*rstd*(dz - sum_dz/B - xhat*sum_dzx/B) (train) or
//          gamma*rstd*dz (eval);  partial sums of dx (-> bias gradient of the preceding Linear).
// `fin` (one group only): the column sums come straight from the producer's row-block partials fin->part
// [nrb][nq][N] — every workgroup sums quantities 0, 1 of its own 64 columns (4 waves over disjoint row-block ranges,
// then in order: same order and bits as colsum_finalize) and the first row of workgroups also stores what
// colsum_finalize would have: totals of quantity q into d[q] (NULL: skip; d[4] is one scalar), a plain copy of
// quantity 2 into copy2, and clears zero0 / zero1.  fin->part must not alias `part` (other workgroups still read it).
struct ColFin {
    const float* part;
    int nq;
    float* d[5];
    int accumulate;
    float* copy2;
    float *zero0, *zero1;
    int nz0, nz1;
};
int bn_bwd_apply_launch(const float* dz, int lddz, const float* a, int lda, const float* mean, const float* rstd,
                        const float* gamma, const float* sums /*[groups][2][N]*/, int train, int B,
                        int rows_per_group, int N, float* dx, int lddx, float* part, hipStream_t st,
                        const ColFin* fin = nullptr, const SumOut* sum = nullptr);
// Stage 3 (dq = dh * lrelu'(h); partial sums: sum dq, sum dq*xhat1) lives in fc2_bwd_stage_launch, behind the fc2 dgrad.
// Sum partial column sums over row blocks: part [groups][nrb][nq][ncols] ->
//   sums_out [groups][nq][ncols] (may be NULL) and, summed over groups, into up to 3 gradient
//   destinations (NULL to skip): dst[q][c] (+)= sum.
//   ld / c_lo: row pitch of `part` (0 = ncols) and first column; copy0 / copy2: extra plain copies of the
//   q == 0 / q == 2 totals; zero0 / zero1: ranges cleared alongside.
int colsum_finalize_launch(const float* part, int groups, int nrb, int nq, int ncols, float* sums_out, float* dst0,
                           float* dst1, float* dst2, int accumulate, hipStream_t st, int ld = 0, int c_lo = 0,
                           float* copy0 = nullptr, float* copy2 = nullptr, float* zero0 = nullptr, int nz0 = 0,
                           float* zero1 = nullptr, int nz1 = 0, float* dst3 = nullptr, float* dst4 = nullptr,
                           int noacc_below = 0,   // columns c < noacc_below are overwritten even when accumulating
                           long long gstride = 0);  // distance between the groups' partial blocks (0: back to back)
// Input BN backward: partial sums of dX0 and dX0*xhat over the concat sources.
int ibn_bwd_stats_launch(const ColSrcs& srcs, const float* dX0, int ldx, int col_off, const float* mean,
                         const float* rstd, int rows_per_group, int groups, float* part, hipStream_t st, int nslab = 0,
                         long long slab_stride = 0, float* dsum = nullptr);
// Input gradient for a column range [c_lo, c_hi) of the concat: dx written to dst (B, c_hi-c_lo).
int ibn_bwd_apply_launch(const ColSrcs& srcs, const float* dX0, int ldx, int col_off, const float* mean,
                         const float* rstd, const float* gamma, const float* sums, int train, int B,
                         int rows_per_group, int c_lo, int c_hi, float* dst, int ldd, hipStream_t st,
                         const float* fin_part = nullptr, int nslab = 0, long long slab_stride = 0);

// losses -------------------------------------------------------------------------------------
// a rider of bce_launch: run_mean / run_var (n columns) advance once more with the stored batch statistics mean / var_u
struct RunUpd {
    const float *mean, *var_u;
    float *run_mean, *run_var;
    int n;
};
int bce_launch(const float* p, const float* labels, float target_const, int B, float weight, float loss_scale,
               float* loss_out, float* d_logit, hipStream_t st, float* combine = nullptr, int real_slot = DGG_L_BCE_REAL,
               unsigned* clear_ticket = nullptr /* set to 0 by the launch (GenLossFin::ticket) */,
               float* loss_copy = nullptr /* the loss value once more */, const RunUpd* ru = nullptr);
int ortho6d_fwd_launch(const float* o6, int ld, int B, float* rot9, hipStream_t st);
int fake_pose_launch(const float* heads, int ld, int nj, int B, float* rot9, float* transl, float* joint, hipStream_t st);
int ortho6d_bwd_launch(const float* o6, int ld, int B, const float* d_rot9, float* d_o6, int ldd, hipStream_t st);
// The spread form of the generator's loss kernel (64 rows per workgroup) finishes inside its own launch: the workgroup that
// draws the last ticket adds the per-workgroup partials in order.  ticket: one word that reads 0 when the launch starts (an
// earlier launch on the same stream clears it: bce_launch / wass_launch `clear_ticket`) — left at 0 again.  gbh (optional):
// the column sums of d_heads = the heads' bias gradient, written (accumulate: added) there; net_backward is then told not
// to form it again (kHeadBiasDone).
struct GenLossFin {
    unsigned* ticket = nullptr;
    float* gbh = nullptr;
    int accumulate = 0;
};
constexpr unsigned kHeadBiasDone = 0x10000u;   // net_backward flag, internal: heads' bias gradient already in `grads`
constexpr unsigned kRefreshElsewhere = 0x20000u;   // net_forward flag, internal (with DGG_REUSE_INPUT_NORM): the caller advances
                                                   // the running statistics itself (a rider of its loss launch: RunUpd)
constexpr unsigned kPublicFlags = DGG_TRAIN | DGG_ACCUM_GRADS | DGG_SKIP_PARAM_GRADS | DGG_NEED_INPUT_GRADS | DGG_SEED_ON_DEVICE |
                                  DGG_REUSE_INPUT_NORM;   // what an extern "C" entry accepts: internal bits are masked off there
int gen_loss_scratch_floats(const dgg_cfg* cfg, int B);
bool gen_loss_spread(int B);   // true: gen_loss_launch (given scratch and a ticket) takes the spread form and writes fin->gbh
int gen_loss_launch(const dgg_cfg* cfg, const dgg_hparams* hp, int B, const float* heads, const float* gt_rot9,
                    const float* gt_transl, const float* gt_joint, const float* fake_score, const float* d_pose,
                    float* losses, float* d_heads, hipStream_t st, float* scratch = nullptr /* >= gen_loss_scratch_floats */,
                    int wass = 0 /* 1: gen_loss_fake = -mean(fake_score) (dexgangrasp.py:258) instead of BCE(., 1) */,
                    const GenLossFin* fin = nullptr);

// ---------------------------------------------------------------- WGAN-GP second-order pass (second_order.hip)
int gp_interp_launch(const float* alphas, const float* r_rot, const float* r_tr, const float* r_jt, const float* f_rot,
                     const float* f_tr, const float* f_jt, int B, int nj, float* o_rot, float* o_tr, float* o_jt,
                     hipStream_t st);
int wass_launch(const float* p, int B, float sign, float* loss_out, float* d_logit, hipStream_t st,
                unsigned* clear_ticket = nullptr);
int sig_grad_launch(const float* p, int B, float* d_logit, hipStream_t st);
int gp_value_parts(int B);   // floats of gp_part
int gp_value_launch(const float* g, int B, int nj, float gain, float* gp_part, float* v, hipStream_t st);
int wgan_combine_launch(float* losses, float gain, int gp_slot, const float* gp_part, int B, hipStream_t st);
int bn_jvp_launch(const float* a, int lda, const float* adot, int ldad, const float* mean, const float* rstd,
                  const float* gamma, const float* addend, int ldadd, const float* mask_ref, int ldm, int B, int N,
                  float* out, int ldo, float* xhd_out, int ldx, int a_is_xhat, hipStream_t st);
int mask_inplace_launch(float* x, int ldx, const float* ref, int ldr, int B, int N, hipStream_t st);
int head2_launch(const float* ydot, int ld, const float* w, const float* p, int B, int K, float* c1, float* c2,
                 hipStream_t st);
int head2_grads_launch(const float* y, int ldy, const float* ydot, int ldyd, const float* c1, const float* c2, int B,
                       int K, float* gw, float* gb, hipStream_t st);
int out2_bwd_launch(const float* ybar, const float* ydbar, int ldyb, const float* c1, const float* c2, const float* w,
                    const float* y, int ldy, const float* s3, int lds, int B, int N, float* pbar, float* pdbar, int ldp,
                    float* a3bar, float* a3dbar, int lda3, hipStream_t st);
int bn_second_bwd_launch(const float* a, int lda, const float* adot, int ldad, const float* mean, const float* rstd,
                         const float* gamma, const float* zbar, const float* zdbar, int ldz, const float* mask_ref, int ldm,
                         int B, int N, float* abar, float* adbar, int ldo, float* g_gamma, float* g_beta, hipStream_t st);
int colsum_prod_launch(const float* x, int ldx, const float* y, int ldy, int B, int N, float* dst, int accumulate,
                       hipStream_t st);
int adam_launch(float* params, const float* grads, float* m, float* v, int64_t n, int step, double lr, double b1,
                double b2, double eps, double wd, double grad_scale, hipStream_t st);
int adam_graph_launch(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step_dev, float* dyn,
                      double lr, double b1, double b2, double eps, double wd, double grad_scale, hipStream_t st);
int seed_bump_launch(uint64_t* seed_dev, hipStream_t st);
int translate_axis_launch(float* transl, const float* rot9, int B, int axis, float offset, hipStream_t st);
int copy_cols_launch(const float* src, int lds, float* dst, int ldd, int B, int n, hipStream_t st);

}  // namespace dgg
