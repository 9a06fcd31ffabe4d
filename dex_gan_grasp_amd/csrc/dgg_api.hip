// C-ABI of libdgg.so (include/dgg.h): flat-buffer layouts, workspace carving and the launch
// sequences of Generator / Discriminator / DexEvaluator forward+backward
// (DexGanGrasp/models/networks.py:43-56,88-103,132-159,315-338) and of the training phases
// (DexGanGrasp/models/dexgangrasp.py:739-823).  Host code only enqueues kernels on the caller's
// stream: no allocation, no synchronisation, no host<->device copies.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include "dgg_internal.h"

#include <stdarg.h>

namespace dgg {

static thread_local char g_err[512] = "";

void set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

void set_error_text(const char* fmt, ...) {   // a reason for a DGG_ERR_INVALID that the caller could not guess
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what, e);
        return DGG_ERR_LAUNCH;
    }
    return DGG_OK;
}

#define DGG_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != DGG_OK) return rc__; \
    } while (0)

static inline int64_t round64(int64_t x) { return (x + 63) / 64 * 64; }

// ------------------------------------------------------------------------------------------------
// Layout of the flat parameter / BN-statistics buffers.
//   fc1.weight and fc3.weight of a ResBlock are stored back to back so that they form ONE
//   (hidden+n_neurons) x Fin matrix W13 (they share the A operand: fc1 || fc3 fused along N);
//   likewise their biases, and the generator's three head matrices form one (nj+9) x 512 matrix.
// ------------------------------------------------------------------------------------------------
struct RbL {
    int fin;
    int64_t w13, b13, bn1w, bn1b, w2, b2, bn2w, bn2b;  // parameter offsets
    int64_t rm1, rv1, rm2, rv2;                         // running-stat offsets
};
struct NetL {
    int net;
    int hid, nn, n13;  // 256, 512, 768
    int k_in;          // width of the BatchNorm'd input concat
    int nprefix;       // columns copied in front of it (generator: Z)
    int k0, ldx0;      // k0 = nprefix + k_in; row pitch of X0
    int in_bps;
    int nrb;
    bool x1_first;     // evaluator: cat[X, X0]; generator/discriminator: cat[X0, X]
    RbL rb[3];
    int nh;
    int64_t ibnw, ibnb, wh, bh;
    int64_t irm, irv;
    int64_t nparam, nbn;
};

static inline int64_t align4(int64_t x) { return (x + 3) & ~(int64_t)3; }

static bool cfg_ok(const dgg_cfg* c) {
    return c && c->latent_d >= 0 && c->latent_d <= 64 && c->n_hand_joints >= 1 && c->n_hand_joints + 9 <= 32 &&
           c->in_bps >= 1 && c->n_neurons == 512 && c->hidden == 256;
}

static int build_layout(int net, const dgg_cfg* c, NetL& L) {
    if (!cfg_ok(c)) return DGG_ERR_INVALID;
    memset(&L, 0, sizeof(L));
    L.net = net;
    L.hid = c->hidden;
    L.nn = c->n_neurons;
    L.n13 = L.hid + L.nn;
    L.in_bps = c->in_bps;
    if (net == DGG_NET_GEN) {
        L.k_in = c->in_bps;
        L.nprefix = c->latent_d;
        L.nrb = 2;
        L.nh = c->n_hand_joints + 9;
        L.x1_first = false;
    } else if (net == DGG_NET_DISC) {
        L.k_in = c->in_bps + 12 + c->n_hand_joints;
        L.nprefix = 0;
        L.nrb = 2;
        L.nh = 1;
        L.x1_first = false;
    } else if (net == DGG_NET_EVA) {
        L.k_in = c->in_bps + 12;
        L.nprefix = 0;
        L.nrb = 3;
        L.nh = 1;
        L.x1_first = true;
    } else {
        return DGG_ERR_INVALID;
    }
    L.k0 = L.nprefix + L.k_in;
    L.ldx0 = (int)((L.k0 + 31) / 32 * 32);   // 128-byte rows: a k-tile row segment of the GEMMs is exactly one cache line
    int64_t po = 0, bo = 0;
    auto P = [&](int64_t n) { int64_t o = po; po = align4(po + n); return o; };
    auto S = [&](int64_t n) { int64_t o = bo; bo = align4(bo + n); return o; };
    L.ibnw = P(L.k_in);
    L.ibnb = P(L.k_in);
    L.irm = S(L.k_in);
    L.irv = S(L.k_in);
    for (int r = 0; r < L.nrb; ++r) {
        RbL& R = L.rb[r];
        R.fin = (r == 0) ? L.k0 : L.k0 + L.nn;
        R.w13 = P((int64_t)L.n13 * R.fin);
        R.b13 = P(L.n13);
        R.bn1w = P(L.hid);
        R.bn1b = P(L.hid);
        R.w2 = P((int64_t)L.nn * L.hid);
        R.b2 = P(L.nn);
        R.bn2w = P(L.nn);
        R.bn2b = P(L.nn);
        R.rm1 = S(L.hid);
        R.rv1 = S(L.hid);
        R.rm2 = S(L.nn);
        R.rv2 = S(L.nn);
    }
    // heads: weight matrix immediately followed by its bias vector (nh*512 is a multiple of 4)
    L.wh = P((int64_t)L.nh * L.nn);
    L.bh = L.wh + (int64_t)L.nh * L.nn;
    po = align4(L.bh + L.nh);
    L.nparam = po;
    L.nbn = bo;
    return DGG_OK;
}

// Stand-alone ResBlock (networks.py:26-56): one block, no input BatchNorm, no heads.  Same parameter order as a
// block inside a network: [W13 = fc1.weight ; fc3.weight | b13 | bn1.w | bn1.b | W2 | b2 | bn2.w | bn2.b].
static const int kNetResBlock = 3;
static int build_rb_layout(int fin, NetL& L) {
    if (fin < 1) return DGG_ERR_INVALID;
    memset(&L, 0, sizeof(L));
    L.net = kNetResBlock;
    L.hid = 256;
    L.nn = 512;
    L.n13 = 768;
    L.k_in = 0;
    L.nprefix = 0;
    L.k0 = fin;
    L.ldx0 = 4;  // no X0 / dX0 buffers: the block reads the caller's x directly
    L.nrb = 1;
    int64_t po = 0, bo = 0;
    auto P = [&](int64_t n) { int64_t o = po; po = align4(po + n); return o; };
    auto S = [&](int64_t n) { int64_t o = bo; bo = align4(bo + n); return o; };
    RbL& R = L.rb[0];
    R.fin = fin;
    R.w13 = P((int64_t)L.n13 * fin);
    R.b13 = P(L.n13);
    R.bn1w = P(L.hid);
    R.bn1b = P(L.hid);
    R.w2 = P((int64_t)L.nn * L.hid);
    R.b2 = P(L.nn);
    R.bn2w = P(L.nn);
    R.bn2b = P(L.nn);
    R.rm1 = S(L.hid);
    R.rv1 = S(L.hid);
    R.rm2 = S(L.nn);
    R.rv2 = S(L.nn);
    L.nparam = po;
    L.nbn = bo;
    return DGG_OK;
}

static int add_entry(dgg_entry* out, int cap, int& n, const char* name, int64_t off, int rows, int cols, int kind) {
    if (out && n < cap) {
        memset(&out[n], 0, sizeof(dgg_entry));
        snprintf(out[n].name, sizeof(out[n].name), "%s", name);
        out[n].offset = off;
        out[n].rows = rows;
        out[n].cols = cols;
        out[n].kind = kind;
    }
    ++n;
    return n;
}

static int fill_table(const NetL& L, const dgg_cfg* c, dgg_entry* out, int cap) {
    int n = 0;
    char nm[64];
    const char* ibn = L.net == DGG_NET_GEN ? "gen_bn1" : L.net == DGG_NET_DISC ? "disc_bn1" : "bn1";
    const char* rbp = L.net == DGG_NET_GEN ? "gen_rb" : L.net == DGG_NET_DISC ? "disc_rb" : "rb";
    auto E = [&](const char* a, const char* b, int64_t off, int rows, int cols, int kind) {
        if (a[0]) snprintf(nm, sizeof(nm), "%s.%s", a, b);
        else snprintf(nm, sizeof(nm), "%s", b);
        add_entry(out, cap, n, nm, off, rows, cols, kind);
    };
    const bool bare = L.net == kNetResBlock;  // stand-alone block: keys "fc1.weight", ... without a prefix
    if (!bare) {
        E(ibn, "weight", L.ibnw, L.k_in, 1, 0);
        E(ibn, "bias", L.ibnb, L.k_in, 1, 0);
        E(ibn, "running_mean", L.irm, L.k_in, 1, 1);
        E(ibn, "running_var", L.irv, L.k_in, 1, 1);
    }
    for (int r = 0; r < L.nrb; ++r) {
        const RbL& R = L.rb[r];
        char pre[32];
        if (bare) pre[0] = 0;
        else snprintf(pre, sizeof(pre), "%s%d", rbp, r + 1);
        E(pre, "fc1.weight", R.w13, L.hid, R.fin, 0);
        E(pre, "fc3.weight", R.w13 + (int64_t)L.hid * R.fin, L.nn, R.fin, 0);
        E(pre, "fc1.bias", R.b13, L.hid, 1, 0);
        E(pre, "fc3.bias", R.b13 + L.hid, L.nn, 1, 0);
        E(pre, "bn1.weight", R.bn1w, L.hid, 1, 0);
        E(pre, "bn1.bias", R.bn1b, L.hid, 1, 0);
        E(pre, "fc2.weight", R.w2, L.nn, L.hid, 0);
        E(pre, "fc2.bias", R.b2, L.nn, 1, 0);
        E(pre, "bn2.weight", R.bn2w, L.nn, 1, 0);
        E(pre, "bn2.bias", R.bn2b, L.nn, 1, 0);
        E(pre, "bn1.running_mean", R.rm1, L.hid, 1, 1);
        E(pre, "bn1.running_var", R.rv1, L.hid, 1, 1);
        E(pre, "bn2.running_mean", R.rm2, L.nn, 1, 1);
        E(pre, "bn2.running_var", R.rv2, L.nn, 1, 1);
    }
    if (bare) return n;
    if (L.net == DGG_NET_GEN) {
        const int nj = c->n_hand_joints;
        add_entry(out, cap, n, "gen_joint_conf.weight", L.wh, nj, L.nn, 0);
        add_entry(out, cap, n, "gen_rot.weight", L.wh + (int64_t)nj * L.nn, 6, L.nn, 0);
        add_entry(out, cap, n, "gen_transl.weight", L.wh + (int64_t)(nj + 6) * L.nn, 3, L.nn, 0);
        add_entry(out, cap, n, "gen_joint_conf.bias", L.bh, nj, 1, 0);
        add_entry(out, cap, n, "gen_rot.bias", L.bh + nj, 6, 1, 0);
        add_entry(out, cap, n, "gen_transl.bias", L.bh + nj + 6, 3, 1, 0);
    } else {
        add_entry(out, cap, n, "out_success.weight", L.wh, 1, L.nn, 0);
        add_entry(out, cap, n, "out_success.bias", L.bh, 1, 1, 0);
    }
    return n;
}

// ------------------------------------------------------------------------------------------------
// Workspace carving (identical in forward and backward so saved activations are found again)
// ------------------------------------------------------------------------------------------------
struct RbW {
    float *a13, *h, *a2, *y;
    uint8_t* keep;
    float *mean1, *rstd1, *mean2, *rstd2;
};
struct NetW {
    float* ivaru;                // [k_in] unbiased batch variance of the input (kept for DGG_REUSE_INPUT_NORM)
    float *X0, *imean, *irstd;   // X0 holds [prefix | x_hat]: the input BatchNorm's gamma/beta are folded into
                                 // the GEMMs (aff_scale / aff_shift), never applied to the stored operand
    float *aff_scale, *aff_shift;  // [k0]: [1.. | gamma], [0.. | beta]
    float *cs13[3];              // [n13] column sums of dY13, per ResBlock (read by the wgrad epilogue)
    float *ibn_part;             // [nrb][12][2][k0] per-tile partials of d gamma / d beta from the wgrad epilogue
    RbW rb[3];
    // scratch (not preserved between calls)
    float *slabs, *part, *part2, *sums, *dpre, *dh, *dX0, *dXa, *dXb;
    // per-ResBlock backward operands: the weight-gradient GEMMs may run on a side stream while the main
    // stream already back-propagates the next block, so they are not shared scratch
    float *dY13[3], *da2[3];
    float* dYs[3];               // dY13 of this pass + dY13 of the discriminator's other pass (Share::bwd_mode 2)
    float *slabs_w;              // split-K slabs of the fc2 weight gradient
    int64_t slab_floats;
    float* pre[3];               // forward pre-products (net_forward): block r's slabs of X0 W13_r[:, x0 columns]^T; NULL: none
    int pre_splits;              // K splits of each of them
    size_t bytes;
};
static bool share_wgrad() {  // diagnostics: DGG_SHARE_WGRAD=0: both passes of D form their own BPS weight gradient
    static const bool v = [] { const char* e = getenv("DGG_SHARE_WGRAD"); return !(e && e[0] == '0'); }();
    return v;
}
static bool obj_plain_fc2() {  // diagnostics: DGG_OBJ_PLAIN_FC2=0 sends the inference sweep through rb_tail's stage kernel
    static const bool v = [] { const char* e = getenv("DGG_OBJ_PLAIN_FC2"); return !(e && e[0] == '0'); }();
    return v;
}
static bool fuse_fc2() {  // diagnostics: DGG_FUSE_FC2=0 keeps bn_act / fc2 GEMM / reduce as separate launches
    static const bool v = [] { const char* e = getenv("DGG_FUSE_FC2"); return !(e && e[0] == '0'); }();
    return v;
}
struct Bump {
    char* base;
    size_t off;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct Seg {
    const float* p;  // forward operand (B x width, row pitch ld)
    int ld, width, wcol;
    bool is_x0;      // the [prefix | x_hat] operand: BN affine folded into the GEMMs
    float* dp;       // backward: gradient destination for columns [d_lo, d_hi) of this segment
    int dld, d_lo, d_hi, d_accum;
    bool d_split;    // the consumer of dp is the previous block's res_out_bwd, which can sum split-K slabs
    float* d_slabs;  // != NULL: leave columns [d_lo, d_hi) as d_ks split-K slabs here (pitch kPoseLd, B * kPoseLd apart)
    int d_ks;
};
static const int kPoseLd = 64;  // row pitch of the input-gradient slabs (a narrow column range of X0: the pose columns)
static const int kPoseKs = 6;

static int splits_used(int K, int ks) {  // what prob_split makes of a request for ks splits
    const int ksteps = (K + 31) / 32;
    const int per = (ksteps + ks - 1) / ks;
    return (K + per * 32 - 1) / (per * 32);
}

static const int kSlabCap = 24;  // upper bound on split-K slabs of one GEMM

// Split-K plan for the fused fc1||fc3 GEMM over `nseg` column segments.
static void plan13(int B, int n13, const int* widths, int nseg, int* splits) {
    int bm, bn;
    gemm_tile_shape(GEMM_NT, B, n13, bm, bn, 4096);
    const int tiles = ((B + bm - 1) / bm) * ((n13 + bn - 1) / bn);
    int total_kt = 0;
    for (int i = 0; i < nseg; ++i) total_kt += (widths[i] + 31) / 32;
    // DGG_SPLITK_TARGET is read once (thread-safe static initialisation); the default follows the current algorithm
    static const int env_target = [] {
        const char* e = getenv("DGG_SPLITK_TARGET");
        const int t = e ? atoi(e) : 0;
        return t < 0 ? 0 : t;
    }();
    // bf16x3 kernel: one workgroup per CU; measured best at 144-192 (other streams keep CUs)
    const int target = env_target > 0 ? env_target : (gemm_algo() >= 1 ? 192 : 512);
    int s_total = target / (tiles > 0 ? tiles : 1);
    if (s_total < 1) s_total = 1;
    if (s_total > 16) s_total = 16;
    int kt_per = (total_kt + s_total - 1) / s_total;
    if (kt_per < 4) kt_per = 4;
    for (int i = 0; i < nseg; ++i) {
        const int kt = (widths[i] + 31) / 32;
        int s = (kt + kt_per - 1) / kt_per;
        if (s < 1) s = 1;
        splits[i] = s;
    }
}

// Forward pre-products (round 6).  The [prefix | x_hat] segment of EVERY ResBlock's fc1 || fc3 product depends on nothing but
// the normalised input: the blocks' X0 products go out as ONE grouped launch at the head of the pass (nrb problems, the K
// splits chosen for ~192 work items in total), and a block's own launch is left with the previous block's 512 columns (and
// the few pose columns of a discriminator pass).  A launch of gemm_x5_kernel costs ~33 us whatever its size: one
// prologue / ramp / tail instead of nrb, and at small batches twice the k-steps per work item.
static bool pre_fwd_enabled() {   // diagnostics: DGG_PRE_FWD=0: every block multiplies its whole input itself (rounds 1-5)
    static const bool v = [] { const char* e = getenv("DGG_PRE_FWD"); return !(e && e[0] == '0'); }();
    return v;
}
static const int kPreSlabs = 8;
static int pre_splits_of(const NetL& L, int B) {
    // Small batches only (B <= 512): there a block's own launch has so few k-steps per work item (4-8 per wave) that it is
    // all prologue and tail, and one launch with twice the k-steps wins (config 2, B = 256: 0.762 -> 0.742 ms; config 4's
    // B = 512 step with encode 1.263 -> 1.232).  At B = 1024 the blocks' launches are long enough and the grouped one delays
    // the first block's tail: interleaved step 1.267 -> 1.300 ms, evaluator alone 0.502 -> 0.519 (profiles/r06_experiments.txt).
    if (!pre_fwd_enabled() || L.nrb < 2 || L.net == kNetResBlock || B > 512) return 0;
    const int tiles = ((B + 127) / 128) * ((L.n13 + 127) / 128);
    int s = 192 / (tiles * L.nrb);
    const int by_k = (L.in_bps / 32) / 4;   // >= 4 k-tiles of 32 per split
    if (s > by_k) s = by_k;
    if (s > kPreSlabs) s = kPreSlabs;
    return s < 1 ? 1 : s;
}

static void carve(const NetL& L, int B, void* base, NetW& W) {
    Bump b{reinterpret_cast<char*>(base), 0};
    const size_t Bz = (size_t)B;
    W.X0 = b.take<float>(Bz * L.ldx0);
    W.imean = b.take<float>(2 * (size_t)L.k_in);
    W.irstd = b.take<float>(2 * (size_t)L.k_in);
    W.ivaru = b.take<float>((size_t)L.k_in);
    W.aff_scale = b.take<float>((size_t)L.k0);
    W.aff_shift = b.take<float>((size_t)L.k0);
    for (int r = 0; r < L.nrb; ++r) W.cs13[r] = b.take<float>((size_t)L.n13);
    W.ibn_part = b.take<float>((size_t)L.nrb * 12 * 2 * L.k0);
    for (int r = 0; r < L.nrb; ++r) {
        RbW& R = W.rb[r];
        R.a13 = b.take<float>(Bz * L.n13);
        R.h = b.take<float>(Bz * L.hid);
        R.a2 = b.take<float>(Bz * L.nn);
        R.y = b.take<float>(Bz * L.nn);
        R.keep = b.take<uint8_t>(Bz * L.nn);
        R.mean1 = b.take<float>(2 * (size_t)L.hid);
        R.rstd1 = b.take<float>(2 * (size_t)L.hid);
        R.mean2 = b.take<float>(2 * (size_t)L.nn);
        R.rstd2 = b.take<float>(2 * (size_t)L.nn);
    }
    W.slab_floats = (int64_t)kSlabCap * B * L.n13;
    // the split plan never needs more than (target/tiles + nseg) slabs; cap the allocation by it
    {
        int widths[2] = {L.k0, L.nn};
        int sp[2];
        plan13(B, L.n13, widths, 2, sp);
        int64_t need = (int64_t)(sp[0] + sp[1] + 2) * B * L.n13;
        if (need < W.slab_floats) W.slab_floats = need;
        const int64_t need_w2 = (int64_t)8 * L.nn * L.hid;  // split-K slabs of the fc2 weight gradient
        if (W.slab_floats < need_w2) W.slab_floats = need_w2;
    }
    W.slabs = b.take<float>((size_t)W.slab_floats);
    const size_t nrb = (Bz + kRowBlock - 1) / kRowBlock + 2;
    size_t part = 3 * (size_t)L.n13;
    if (2 * (size_t)L.k_in > part) part = 2 * (size_t)L.k_in;
    if ((size_t)L.nh * (L.nn + 1) > part) part = (size_t)L.nh * (L.nn + 1);
    W.part = b.take<float>(nrb * part);
    W.part2 = b.take<float>(nrb * 2 * (size_t)L.n13);  // bn_bwd_apply's own partials while it reads W.part (ColFin)
    size_t sums = 2 * 3 * (size_t)L.n13;
    if (2 * 2 * (size_t)L.k_in > sums) sums = 2 * 2 * (size_t)L.k_in;
    W.sums = b.take<float>(sums);
    for (int r = 0; r < L.nrb; ++r) {
        W.dY13[r] = b.take<float>(Bz * L.n13);
        W.da2[r] = b.take<float>(Bz * L.nn);
        W.dYs[r] = L.net == DGG_NET_DISC ? b.take<float>(Bz * L.n13) : nullptr;
    }
    W.slabs_w = b.take<float>((size_t)8 * L.nn * L.hid);
    W.dpre = b.take<float>(Bz * L.nn);
    W.dh = b.take<float>(Bz * L.hid);
    W.dX0 = b.take<float>(Bz * L.ldx0);
    W.dXa = b.take<float>(Bz * L.nn);
    W.dXb = b.take<float>(Bz * L.nn);
    W.pre_splits = pre_splits_of(L, B);
    for (int r = 0; r < 3; ++r)
        W.pre[r] = (W.pre_splits > 0 && r < L.nrb) ? b.take<float>((size_t)(W.pre_splits + 1) * Bz * L.n13) : nullptr;
    W.bytes = (b.off + 255) & ~(size_t)255;
}

// ------------------------------------------------------------------------------------------------
// GEMM helpers
// ------------------------------------------------------------------------------------------------
static void prob_init(GemmProb& p, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N,
                      int K) {
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = Bm; p.C = C;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.splits = 0;
    p.kchunk = K;
}

static void prob_split(GemmProb& p, int splits, long long slab_stride) {
    const int ksteps = (p.K + 31) / 32;
    if (splits < 1) splits = 1;
    const int per = (ksteps + splits - 1) / splits;
    p.kchunk = per * 32;
    p.splits = (p.K + p.kchunk - 1) / p.kchunk;
    if (p.splits < 1) p.splits = 1;
    p.slab_stride = slab_stride;
}

// ------------------------------------------------------------------------------------------------
// Cross-stream plumbing: a per-thread pool of timing-disabled events (a wait captures the record that
// precedes it on the host, so round-robin reuse is safe) and the stream -> weight-gradient side stream map.
// ------------------------------------------------------------------------------------------------
static hipEvent_t next_event() {
    static thread_local hipEvent_t pool[64];
    static thread_local int made = 0, cur = 0;
    if (made < 64) {
        if (hipEventCreateWithFlags(&pool[made], hipEventDisableTiming) != hipSuccess) return nullptr;
        return pool[made++];
    }
    cur = (cur + 1) & 63;
    return pool[cur];
}

// Gradient-ready events (dgg_set_grad_events): per thread, keyed by the gradient buffer a backward writes
struct GradEvents {
    const float* grads;
    hipEvent_t ev[3];
    int n, accum_only;
};
static const int kGradEventSlots = 24;   // 3 networks x up to 8 engines per thread
static thread_local GradEvents g_grad_events[kGradEventSlots] = {};
static const GradEvents* grad_events_for(const float* grads) {
    if (!grads) return nullptr;
    for (int i = 0; i < kGradEventSlots; ++i)
        if (g_grad_events[i].grads == grads) return &g_grad_events[i];
    return nullptr;
}

static const int kAuxCap = 64;
struct AuxMap {
    hipStream_t key[kAuxCap], aux[kAuxCap];
    int owned[kAuxCap];   // > 0: library-created side stream, reference count of dgg_enable_wgrad_stream
    int n;
};
static thread_local AuxMap g_aux = {{}, {}, {}, 0};

static hipStream_t wgrad_stream_of(hipStream_t st) {
    for (int i = 0; i < g_aux.n; ++i)
        if (g_aux.key[i] == st) return g_aux.aux[i];
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// ResBlock forward (networks.py:43-56) on column segments of the (virtual) concatenated input.
// ------------------------------------------------------------------------------------------------
// The BPS columns of the discriminator's input — and their batch statistics, hence x_hat — are the same in the real
// and the fake pass of a D phase (dexgangrasp.py:788-789 feed the same bps_object twice): the partial product
// x_hat_bps W13[:, bps]^T of every ResBlock is computed once, by the real pass, and summed into both passes.
struct Share {
    int mode;             // 1: produce (real pass), 2: consume (fake pass)
    int ncols;            // leading columns of X0 that are shared
    float* slabs[3];      // per ResBlock: up to kShareSlabs slabs of B x n13
    int nslab[3];         // set by the producer
    hipEvent_t ready[3];  // two-stream form: recorded by the producer after its GEMM, awaited by the consumer
    bool events;
    // Backward: the same x_hat_bps also shares the WEIGHT GRADIENT of the BPS columns —
    //   dW13[:, bps] = dY_real^T x_hat_bps + dY_fake^T x_hat_bps = (dY_real + dY_fake)^T x_hat_bps
    // bwd_mode 1 (real pass): only the pose columns' weight gradient is formed, dY13 / cs13 stay in its workspace;
    // bwd_mode 2 (fake pass, runs after the real pass' backward): its dY13 producers also write dY13 + dY13_other, the
    // BPS columns take that sum (overwriting: nothing was written there), the pose columns accumulate as before.
    int bwd_mode;
    const float* dY13_other[3];
    const float* cs13_other[3];
    // Two-stream form of the backward (round 6; the D phase WITHOUT a look-ahead real pass — the drop-in order): the fake
    // pass' backward does not wait for the whole real backward, it trails it.  The real pass records bwd_ev[r] behind block
    // r's element-wise backward stages (its dY13 / column sums of block r are complete, its BatchNorm / bias gradients of
    // block r written, and — same stream, in order — the weight-gradient launch of block r + 1 enqueued before them done);
    // the fake pass waits for bwd_ev[nrb - 1] before its first backward kernel (its loss kernel reads the real pass' term)
    // and for bwd_ev[r - 1] before the weight-gradient launch of block r (r = 0: the real pass' end), which accumulates
    // onto what the real pass' launch wrote.  NULL events: the fake pass waited for the whole real pass (or shares a stream).
    hipEvent_t bwd_ev[3];
    hipEvent_t bwd_end;   // the real pass' end (its last weight-gradient launch and its input-BatchNorm finalize)
    bool bwd_fine;
};
static const int kShareSlabs = 16;  // plan13 never gives one piece more

// XShare (round 6): the BPS block is the same batch — same column statistics, same x_hat — in the generator's forward
// (networks.py:88-95), in D's real and fake pass (:148-154; dexgangrasp.py:788-789) and in the G phase's pass through the
// updated discriminator (:808).  It is normalised ONCE per step, by the real pass (the producer: off the chain, prefetched
// one step ahead), into shared rows `xs` = [x_hat_bps | pose columns of the fake pass | pad] of D's X0 pitch; the
// generator's operand [z | x_hat_bps] is the same memory seen `nprefix` floats earlier — its latent sits in the pad at the
// end of the previous shared row.  A consumer's input stage is one ibn_shared_kernel launch instead of ibn_stats +
// ibn_apply over 4096-4120 columns (and, for the fake pass, fake_pose_kernel).
struct XShare {
    int mode;                 // 1: producer, 2: consumer
    float* xs;                // the shared rows (D's view: row m at xs + m * ld)
    int ld;
    const float *src_mean, *src_rstd, *src_varu;   // consumer: the producer's batch statistics (its NetW.imean / irstd / ivaru)
    hipEvent_t ready;         // producer records it behind its ibn_apply; a consumer on another stream waits for it (NULL: none)
    bool record, wait;
    // the fake pass: pose columns from the generator's output rows
    const float* heads;
    int ldh;
    float *rot9, *transl, *joint;
};
static bool fine_bwd() {   // diagnostics: DGG_FINE_BWD=0: the fake pass' backward waits for the whole real pass (rounds 1-5)
    static const bool v = [] { const char* e = getenv("DGG_FINE_BWD"); return !(e && e[0] == '0'); }();
    return v;
}
static bool defer_wgrad_enabled() {   // diagnostics: DGG_DEFER_WGRAD=0: one weight-gradient launch per ResBlock (rounds 1-5)
    static const bool v = [] { const char* e = getenv("DGG_DEFER_WGRAD"); return !(e && e[0] == '0'); }();
    return v;
}
static bool xshare_enabled() {   // diagnostics: DGG_XSHARE=0: every pass normalises its own input (rounds 1-5)
    static const bool v = [] { const char* e = getenv("DGG_XSHARE"); return !(e && e[0] == '0'); }();
    return v;
}

struct Ctx {
    Share* share;
    const NetL* L;
    const float* P;   // params
    float* G;         // grads (backward)
    float* bn;        // running stats (forward)
    NetW W;
    int B;
    bool train;
    hipStream_t st;
    hipStream_t sw;   // side stream for weight-gradient GEMMs (NULL: same stream)
    // set by a block's dgrad when it left its output gradient as split-K slabs in W.slabs (consumed, and summed, by
    // the next rb_backward's first stage instead of the `dy` it is given)
    int dy_nslab = 0;
    long long dy_stride = 0;
    // Deferred weight gradients (round 6): nothing reads a block's weight gradient before the optimiser step, and every
    // operand of its launch lives in per-block buffers — so the blocks' launches are collected here and go out as ONE
    // grouped launch when the pass' last block is through (net_backward): one prologue / ramp / tail for 440-670
    // workgroups instead of two or three launches of 200-230.  Not with gradient-ready events (data parallel: the
    // buckets' all-reduces start per block) or a weight-gradient side stream.  (Share::bwd_fine: the fake pass' grouped
    // launch waits for the real pass' end, once, instead of per block.)
    bool defer_wgrad = false;
    GemmBatch dw;
    // forward pre-products (net_forward): block r's X0 product lies in pre_slabs[r] (pre_nslab[r] slabs) already
    bool pre = false;
    float* pre_slabs[3] = {nullptr, nullptr, nullptr};
    int pre_nslab[3] = {0, 0, 0};
};

static int rb_tail(Ctx& c, int r, int drop_mode, const float* keep_in, uint64_t seed, uint64_t seed_off);

static int rb_forward(Ctx& c, int r, const Seg* segs, int nseg, int drop_mode, const float* keep_in, uint64_t seed,
                      uint64_t seed_off) {
    const NetL& L = *c.L;
    const RbL& R = L.rb[r];
    RbW& A = c.W.rb[r];
    const int B = c.B;
    // 1. [a1 | a3] partial products: X_piece (B x width) * W13[:, wcol:wcol+width]^T -> split-K slabs.
    //    A piece is a column segment of the concatenated input; with a Share the [prefix | x_hat] segment is cut at
    //    share->ncols into the shared BPS piece and the pass' own pose piece.
    struct Piece { const float* p; int ld, width, wcol, x0_off; bool x0, shared, pre; };
    Piece pc[4];
    int np = 0;
    Share* sh = c.share;
    bool own_pre = false;   // this pass' own pre-product of the whole X0 segment is summed in below
    for (int i = 0; i < nseg; ++i) {
        if (sh && segs[i].is_x0 && L.nprefix == 0 && sh->ncols > 0 && sh->ncols < segs[i].width) {
            pc[np++] = Piece{segs[i].p, segs[i].ld, sh->ncols, segs[i].wcol, 0, true, true, c.pre};
            pc[np++] = Piece{segs[i].p + sh->ncols, segs[i].ld, segs[i].width - sh->ncols, segs[i].wcol + sh->ncols,
                             sh->ncols, true, false, false};
        } else {
            const bool pre = c.pre && segs[i].is_x0;
            own_pre = own_pre || pre;
            pc[np++] = Piece{segs[i].p, segs[i].ld, segs[i].width, segs[i].wcol, 0, segs[i].is_x0, false, pre};
        }
    }
    int widths[4], sp[4];
    for (int i = 0; i < np; ++i) widths[i] = pc[i].width;
    if (np > 0) plan13(B, L.n13, widths, np, sp);
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    const long long slab = (long long)B * L.n13;
    int nslab = 0, nq = 0;
    for (int i = 0; i < np; ++i) {
        float* dst = c.W.slabs + (size_t)nslab * slab;
        if (pc[i].pre) continue;          // multiplied at the head of the pass (net_forward's pre-products): summed in below
        if (pc[i].shared) {
            if (sh->mode == 2) continue;  // consumer: the producer's slabs are summed in below
            dst = sh->slabs[r];
        }
        GemmProb& q = gb.p[nq++];
        prob_init(q, pc[i].p, pc[i].ld, c.P + R.w13 + pc[i].wcol, R.fin, dst, L.n13, B, L.n13, pc[i].width);
        prob_split(q, sp[i], slab);
        if (pc[i].x0) {  // BatchNorm gamma/beta applied while the operand is staged
            q.a_scale = c.W.aff_scale + pc[i].x0_off;
            q.a_shift = c.W.aff_shift + pc[i].x0_off;
        }
        if (pc[i].shared) {
            if (q.splits > kShareSlabs) return DGG_ERR_WORKSPACE;
            sh->nslab[r] = q.splits;
        } else {
            nslab += q.splits;
        }
    }
    gb.nprob = nq;
    if ((int64_t)nslab * slab > c.W.slab_floats) return DGG_ERR_WORKSPACE;
    if (nq > 0) DGG_TRY(gemm_launch(GEMM_NT, gb, c.st));
    const bool shared_here = sh && np > nseg;
    if (shared_here && sh->events) {
        if (sh->mode == 1 && !c.pre && hipEventRecord(sh->ready[r], c.st) != hipSuccess) return DGG_ERR_LAUNCH;
        if (sh->mode == 2 && hipStreamWaitEvent(c.st, sh->ready[r], 0) != hipSuccess) return DGG_ERR_LAUNCH;
    }
    // 2. sum slabs (k order: shared BPS chunks / the pass' pre-product first) + bias; a1 -> statistics, a3 -> s = lrelu(a3)
    if (shared_here)
        DGG_TRY(reduce_bias_launch(sh->slabs[r], sh->nslab[r], slab, L.n13, c.P + R.b13, A.a13, L.n13, B, B, 1, L.n13,
                                   0, L.hid, L.hid, L.n13, c.W.part, c.st, 0, c.W.slabs, nslab));
    else if (own_pre)
        DGG_TRY(reduce_bias_launch(c.pre_slabs[r], c.pre_nslab[r], slab, L.n13, c.P + R.b13, A.a13, L.n13, B, B, 1, L.n13,
                                   0, L.hid, L.hid, L.n13, c.W.part, c.st, 0, c.W.slabs, nslab));
    else
        DGG_TRY(reduce_bias_launch(c.W.slabs, nslab, slab, L.n13, c.P + R.b13, A.a13, L.n13, B, B, 1, L.n13, 0, L.hid,
                                   L.hid, L.n13, c.W.part, c.st));
    return rb_tail(c, r, drop_mode, keep_in, seed, seed_off);
}

// ResBlock forward from the pre-activations a13 = [a1 | lrelu(a3)] on: bn1 + LeakyReLU, fc2, bn2 + residual + LeakyReLU
// [+ dropout] (networks.py:46-56).
static int rb_tail(Ctx& c, int r, int drop_mode, const float* keep_in, uint64_t seed, uint64_t seed_off) {
    const NetL& L = *c.L;
    const RbL& R = L.rb[r];
    RbW& A = c.W.rb[r];
    const int B = c.B;
    // 3 + 4. h = lrelu(bn1(a1)) and a2 = h W2^T + b2 (K = 256) in ONE launch: the GEMM finalizes bn1's statistics in its
    //    prologue (mean1 / rstd1 kept for the backward), normalises and activates its A operand while staging it (writing
    //    h out for the backward), adds the bias and emits bn2's row-block partials in its epilogue (GemmProb::bn_part).
    //    DGG_FUSE_FC2=0: the four-launch form (bn_act, GEMM, bias/statistics reduce).
    const int nrb_b = (B + kRowBlock - 1) / kRowBlock;
    float* part_bn2 = c.W.part;
    if (fuse_fc2() && L.hid <= 256) {
        GemmBatch g2;
        memset(&g2, 0, sizeof(g2));
        g2.nprob = 1;
        GemmProb& q = g2.p[0];
        prob_init(q, A.a13, L.n13, c.P + R.w2, L.hid, A.a2, L.nn, B, L.nn, L.hid);
        q.bn_part = c.W.part;
        q.bn_nrb = nrb_b;
        q.bn_train = c.train ? 1 : 0;
        q.bn_rows = B;
        q.bn_gamma = c.P + R.bn1w;
        q.bn_beta = c.P + R.bn1b;
        q.bn_run_mean = c.bn ? c.bn + R.rm1 : nullptr;
        q.bn_run_var = c.bn ? c.bn + R.rv1 : nullptr;
        q.bn_mean_out = A.mean1;
        q.bn_rstd_out = A.rstd1;
        q.a_out = A.h;
        q.c_bias = c.P + R.b2;
        q.c_part = c.W.part2;   // other workgroups are still reading W.part
        DGG_TRY(gemm_fc2_fused_launch(g2, c.st));
        part_bn2 = c.W.part2;
    } else {
        {
            BnFused fz{c.W.part, c.train ? 1 : 0, c.bn ? c.bn + R.rm1 : nullptr, c.bn ? c.bn + R.rv1 : nullptr, A.mean1,
                       A.rstd1};
            DGG_TRY(bn_act_launch(A.a13, L.n13, B, B, L.hid, A.mean1, A.rstd1, c.P + R.bn1w, c.P + R.bn1b, A.h, L.hid,
                                  c.st, &fz));
        }
        GemmBatch g2;
        memset(&g2, 0, sizeof(g2));
        g2.nprob = 1;
        prob_init(g2.p[0], A.h, L.hid, c.P + R.w2, L.hid, c.W.slabs, L.nn, B, L.nn, L.hid);
        const long long slab2 = (long long)B * L.nn;
        prob_split(g2.p[0], gemm_pick_splits(GEMM_NT, B, L.nn, L.hid, 0), slab2);
        DGG_TRY(gemm_launch(GEMM_NT, g2, c.st));
        DGG_TRY(reduce_bias_launch(c.W.slabs, g2.p[0].splits, slab2, L.nn, c.P + R.b2, A.a2, L.nn, B, B, 1, L.nn, 0,
                                   L.nn, 0, 0, c.W.part, c.st));
    }
    // 5. y = lrelu(s + bn2(a2)) [dropout]; bn2's statistics finalized inside
    {
        BnFused fz{part_bn2, c.train ? 1 : 0, c.bn ? c.bn + R.rm2 : nullptr, c.bn ? c.bn + R.rv2 : nullptr, A.mean2,
                   A.rstd2};
        DGG_TRY(res_out_launch(A.a13 + L.hid, L.n13, A.a2, L.nn, B, B, L.nn, A.mean2, A.rstd2, c.P + R.bn2w,
                               c.P + R.bn2b, drop_mode, keep_in, seed, seed_off, A.keep, A.y, L.nn, c.st, &fz, nullptr));
    }
    return DGG_OK;
}

// ------------------------------------------------------------------------------------------------
// ResBlock backward.  dy: gradient w.r.t. the block output y (B x 512).
// ------------------------------------------------------------------------------------------------
// Single-score head (D / E) folded into the last ResBlock's backward: d_score (B), the head's weights and the
// destinations of its parameter gradients.
struct HeadBwd {
    const float* d_score;   // (B) — or, n > 1 (the generator's pose heads, round 6): d_out (B, n)
    const float* w;         // (n, nn)
    float *gw, *gb;         // n == 1 only: the head's parameter gradients come out of the block's partial sums
    int n = 1;
};
// Multi-column heads (the generator's 21-24 pose heads): dWh = d_out^T y_last rides in the last ResBlock's
// weight-gradient launch as one more problem (4 tiles beside ~230) instead of a split-K GEMM + reduce of its own.
struct HeadsRide {
    const float* d_out;  // (B, nh)
    int nh;
    float* gwh;          // (nh, nn)
};

static int rb_backward(Ctx& c, int r, const Seg* segs, int nseg, const float* dy, bool had_dropout, bool accum,
                       bool skip_param, int* ibn_tiles, const HeadBwd* hb = nullptr, const HeadsRide* hr = nullptr) {
    const NetL& L = *c.L;
    const RbL& R = L.rb[r];
    RbW& A = c.W.rb[r];
    const int B = c.B;
    const int nrb = (B + kRowBlock - 1) / kRowBlock;
    float* G = c.G;
    auto gp = [&](int64_t off) -> float* { return skip_param ? nullptr : G + off; };
    const int acc = accum ? 1 : 0;
    float* dY13 = c.W.dY13[r];
    float* da2 = c.W.da2[r];
    float* cs13 = c.W.cs13[r];
    // 1. through the output LeakyReLU / dropout; split into the bn2 branch (dpre) and the fc3 branch (da3)
    const int nq1 = (hb && hb->n == 1 && !skip_param) ? 5 : 3;
    // shared BPS weight gradient of the discriminator's two passes (Share::bwd_mode)
    const Share* sh = c.share;
    const bool share_ok = sh && sh->bwd_mode && !skip_param && L.nprefix == 0 && sh->ncols > 0 && sh->ncols < L.k0;
    const bool share_r = share_ok && sh->bwd_mode == 1, share_f = share_ok && sh->bwd_mode == 2;
    // the fake pass of a shared gradient also writes dY13 + the other pass' dY13 (the shared weight gradient's operand)
    const SumOut so1 = {share_f ? sh->dY13_other[r] : nullptr, share_f ? c.W.dYs[r] : nullptr, L.n13};
    const SumOut so3 = {share_f ? sh->dY13_other[r] + L.hid : nullptr, share_f ? c.W.dYs[r] + L.hid : nullptr,
                        L.n13};
    const bool dy_slabs = !hb && c.dy_nslab > 1;
    DGG_TRY(res_out_bwd_launch(hb ? nullptr : (dy_slabs ? c.W.slabs : dy), L.nn, A.y, L.nn,
                               had_dropout ? A.keep : nullptr, A.a13 + L.hid, L.n13, A.a2, L.nn, A.mean2, A.rstd2, B, B,
                               L.nn, c.W.dpre, dY13 + L.hid, L.n13, c.W.part, c.st, hb ? hb->d_score : nullptr,
                               hb ? hb->w : nullptr, nq1, dy_slabs ? c.dy_nslab : 0, c.dy_stride,
                               share_f ? &so3 : nullptr, hb ? hb->n : 1));
    c.dy_nslab = 0;
    // fc1.bias / fc2.bias feed a train-mode BatchNorm: their gradient is EXACTLY zero (the batch-mean subtraction
    // removes any per-column constant; the reference stores fp32 rounding noise there) — cleared by this launch
    const bool zero_pre_bn = c.train && !skip_param && !accum;
    // 2. bn2 backward -> da2; its column sums are fc2.bias' gradient.  The column totals of stage 1's partials
    //    (d bn2.bias, d bn2.weight, the fc3 half of d fc13.bias [+ the head's gradients]) are formed inside this launch.
    {
        ColFin fin = {c.W.part, nq1, {gp(R.bn2b), gp(R.bn2w), gp(R.b13 + L.hid), nq1 == 5 ? hb->gw : nullptr,
                                      nq1 == 5 ? hb->gb : nullptr},
                      acc, cs13 + L.hid, zero_pre_bn ? G + R.b2 : nullptr, zero_pre_bn ? G + R.b13 : nullptr, L.nn,
                      L.hid};
        DGG_TRY(bn_bwd_apply_launch(c.W.dpre, L.nn, A.a2, L.nn, A.mean2, A.rstd2, c.P + R.bn2w, nullptr, c.train, B, B,
                                    L.nn, da2, L.nn, c.W.part2, c.st, &fin));
    }
    // fc2.bias feeds a BatchNorm: in train mode the batch-mean subtraction makes its gradient EXACTLY zero
    // (sum_m da2 = 0 analytically); the reference stores fp32 rounding noise there.  We store the exact
    // value in train mode and the real column sum in eval mode (running statistics).
    if (!skip_param) {
        if (!c.train)
            DGG_TRY(colsum_finalize_launch(c.W.part2, 1, nrb, 1, L.nn, nullptr, G + R.b2, nullptr, nullptr, acc, c.st));
    }
    // 3 + 4. fc2: dh = da2 W2 (K = 512), through lrelu', and bn1's backward partial sums per row block — one launch
    //    (fc2_bwd_stage_kernel, fc2_stage.hip; until round 5: a split-K GEMM leaving four slabs + act_bwd_kernel summing them)
    DGG_TRY(fc2_bwd_stage_launch(da2, L.nn, c.P + R.w2, L.hid, A.h, L.hid, A.a13, L.n13, A.mean1, A.rstd1, B, c.W.dh, L.hid,
                                 c.W.part, c.st));
    {
        ColFin fin = {c.W.part, 2, {gp(R.bn1b), gp(R.bn1w), nullptr, nullptr, nullptr}, acc, nullptr, nullptr, nullptr, 0, 0};
        DGG_TRY(bn_bwd_apply_launch(c.W.dh, L.hid, A.a13, L.n13, A.mean1, A.rstd1, c.P + R.bn1w, nullptr, c.train, B, B,
                                    L.hid, dY13, L.n13, c.W.part2, c.st, &fin, share_f ? &so1 : nullptr));
    }
    // fc1.bias: same argument as fc2.bias above; in eval mode the real column sum is also kept in cs13[0:hid]
    // for the BatchNorm-aware wgrad epilogue (train mode: exactly zero -> ep_cs0 = NULL)
    if (!c.train)
        DGG_TRY(colsum_finalize_launch(c.W.part2, 1, nrb, 1, L.hid, nullptr, gp(R.b13), nullptr, nullptr, acc, c.st, 0,
                                       0, cs13, nullptr));
    // 5. wgrad of [fc1;fc3]: dW13[:, wcol:] = dY13^T X_seg.  For the [prefix | x_hat] segment the epilogue
    //    turns T = dY13^T x_hat into dW = gamma T + beta cs and emits the partials of d gamma / d beta of the
    //    input BatchNorm, so dX0 is never needed for parameter gradients.
    //    The weight-gradient GEMMs have no consumer before the optimiser step: with a side stream configured
    //    (dgg_set_wgrad_stream) they run there, concurrently with the dgrad chain of the following blocks.
    if (share_ok && sh->bwd_fine) {
        // (see Share::bwd_ev) real pass: block r's element-wise stages are enqueued; fake pass: what its weight-gradient
        // launch of block r accumulates onto must have been written
        if (share_r && sh->bwd_ev[r] && hipEventRecord(sh->bwd_ev[r], c.st) != hipSuccess) return DGG_ERR_LAUNCH;
        if (share_f && !c.defer_wgrad) {   // (deferred: one wait for the real pass' end in front of the grouped launch)
            hipEvent_t e = r > 0 ? sh->bwd_ev[r - 1] : sh->bwd_end;
            if (e && hipStreamWaitEvent(c.st, e, 0) != hipSuccess) return DGG_ERR_LAUNCH;
        }
    }
    hipStream_t sw = c.st;
    if (!skip_param && c.sw) {
        hipEvent_t e = next_event();
        if (!e || hipEventRecord(e, c.st) != hipSuccess || hipStreamWaitEvent(c.sw, e, 0) != hipSuccess)
            return DGG_ERR_LAUNCH;
        sw = c.sw;
    }
    // fc2: dW2 = da2^T h (512 x 256, K = B).  It rides along as one more problem of the [fc1;fc3] weight-gradient launch
    // below — 8 tiles beside ~200, written (or accumulated) in place — unless that launch is full.
    int nx0 = share_f ? 2 : 1;           // problems the [prefix | x_hat] segment takes in the fp32 launch
    const bool fc2_rides = !skip_param && (nseg - 1 + nx0) + (hr ? 1 : 0) < kMaxProb;
    if (!skip_param && !fc2_rides) {  // separate: split-K slabs + reduce
        GemmBatch g;
        memset(&g, 0, sizeof(g));
        g.nprob = 1;
        prob_init(g.p[0], da2, L.nn, A.h, L.hid, c.W.slabs_w, L.hid, L.nn, L.hid, B);
        const long long slab = (long long)L.nn * L.hid;
        prob_split(g.p[0], 8, slab);
        DGG_TRY(gemm_launch(GEMM_TN, g, sw));
        DGG_TRY(reduce_bias_launch(c.W.slabs_w, g.p[0].splits, slab, L.hid, nullptr, G + R.w2, L.hid, L.nn, L.nn, 1,
                                   L.hid, 0, 0, 0, 0, nullptr, sw, acc));
    }
    if (!skip_param) {
        GemmBatch g;
        memset(&g, 0, sizeof(g));
        int np = 0;
        bool has_x0 = false;
        // the BatchNorm-aware epilogue of a problem over the columns [lo, ..) of the [prefix | x_hat] segment
        auto ep_x0 = [&](GemmProb& q, int wcol, int lo, const float* cs1b) {
            has_x0 = true;
            q.ep_gamma = c.W.aff_scale + lo;
            q.ep_beta = c.W.aff_shift + lo;
            q.ep_cs0 = c.train ? nullptr : cs13;
            q.ep_cs1 = cs13 + L.hid;
            q.ep_cs1b = cs1b;
            q.ep_split = L.hid;
            q.ep_w = c.P + R.w13 + wcol + lo;
            q.ep_ldw = R.fin;
            q.ep_part = c.W.ibn_part + (size_t)r * 12 * 2 * L.k0 + lo;
            q.ep_part_ld = L.k0;
        };
        // one problem per column range [lo, hi) of a segment; A: whose dY13 it takes
        auto add = [&](const Seg& sg, int lo, int hi, const float* Aop, int accumulate, const float* cs1b) {
            GemmProb& q = g.p[np++];
            prob_init(q, Aop, L.n13, sg.p + lo, sg.ld, G + R.w13 + sg.wcol + lo, R.fin, L.n13, hi - lo, B);
            q.accumulate = accumulate;
            if (sg.is_x0) ep_x0(q, sg.wcol, lo, cs1b);
        };
        for (int i = 0; i < nseg; ++i) {
            if (segs[i].is_x0 && share_r) {     // real pass: the pose columns only
                add(segs[i], sh->ncols, segs[i].width, dY13, acc, nullptr);
            } else if (segs[i].is_x0 && share_f) {     // fake pass: BPS columns on dY_real + dY_fake, pose columns on its own
                add(segs[i], 0, sh->ncols, c.W.dYs[r], 0, sh->cs13_other[r] + L.hid);
                add(segs[i], sh->ncols, segs[i].width, dY13, acc, nullptr);
            } else {
                add(segs[i], 0, segs[i].width, dY13, acc, nullptr);
            }
        }
        if (fc2_rides) {
            GemmProb& q = g.p[np++];
            prob_init(q, da2, L.nn, A.h, L.hid, G + R.w2, L.hid, L.nn, L.hid, B);
            q.accumulate = acc;
        }
        if (hr) {
            if (np >= kMaxProb) return DGG_ERR_INVALID;  // net_backward only asks when there is room
            GemmProb& q = g.p[np++];
            prob_init(q, hr->d_out, hr->nh, A.y, L.nn, hr->gwh, L.nn, hr->nh, L.nn, B);
            q.accumulate = acc;
        }
        g.nprob = np;
        if (np > 0) {
            // the tile shape gemm_launch will pick (the problem with the most work decides): the row-tile count of the
            // epilogue's per-tile partials
            int big = 0;
            for (int i = 1; i < np; ++i)
                if ((double)g.p[i].M * g.p[i].N * g.p[i].K > (double)g.p[big].M * g.p[big].N * g.p[big].K) big = i;
            int bm, bn;
            gemm_tile_shape(GEMM_TN, g.p[big].M, g.p[big].N, bm, bn, g.p[big].K);
            ibn_tiles[r] = (L.n13 + bm - 1) / bm;
            if (ibn_tiles[r] > 12 || (!has_x0 && L.net != kNetResBlock)) return DGG_ERR_INVALID;
            if (c.defer_wgrad && c.dw.nprob + np <= kMaxProb) {
                for (int i = 0; i < np; ++i) c.dw.p[c.dw.nprob++] = g.p[i];
            } else {
                if (c.defer_wgrad && share_f && sh->bwd_fine && sh->bwd_end && hipStreamWaitEvent(c.st, sh->bwd_end, 0) != hipSuccess)
                    return DGG_ERR_LAUNCH;   // (no room in the deferred launch: this one goes out now, behind the real pass' end)
                DGG_TRY(gemm_launch(GEMM_TN, g, sw));
            }
        }
    }
    // 6. dgrad: dX_seg[:, lo:hi] (+)= dY13 W13[:, wcol+lo : wcol+hi]
    {
        GemmBatch g;
        memset(&g, 0, sizeof(g));
        int np = 0;
        for (int i = 0; i < nseg; ++i) {
            if (!segs[i].dp || segs[i].d_hi <= segs[i].d_lo) continue;
            if (segs[i].d_slabs) {  // a few columns, 16 row tiles: K splits, summed by the input BatchNorm's backward
                prob_init(g.p[np], dY13, L.n13, c.P + R.w13 + segs[i].wcol + segs[i].d_lo, R.fin, segs[i].d_slabs,
                          kPoseLd, B, segs[i].d_hi - segs[i].d_lo, L.n13);
                prob_split(g.p[np], segs[i].d_ks, (long long)B * kPoseLd);
                ++np;
                continue;
            }
            prob_init(g.p[np], dY13, L.n13, c.P + R.w13 + segs[i].wcol + segs[i].d_lo, R.fin,
                      segs[i].dp + segs[i].d_lo, segs[i].dld, B, segs[i].d_hi - segs[i].d_lo, L.n13);
            g.p[np].accumulate = segs[i].d_accum;
            // B x 512 output, K = 768: 64 x 64 tiles alone fill half the chip with 24 k-tiles each — three K
            // splits, left as slabs that the previous block's first backward stage sums
            const int ks = 3;
            if (segs[i].d_split && !segs[i].d_accum && segs[i].d_lo == 0 && segs[i].dld == segs[i].d_hi &&
                (int64_t)ks * B * segs[i].dld <= c.W.slab_floats && L.n13 >= ks * 4 * 32) {
                g.p[np].C = c.W.slabs;
                prob_split(g.p[np], ks, (long long)B * segs[i].dld);
                c.dy_nslab = g.p[np].splits;
                c.dy_stride = (long long)B * segs[i].dld;
            }
            ++np;
        }
        if (np == 2 && g.p[1].N > g.p[0].N) { GemmProb t = g.p[0]; g.p[0] = g.p[1]; g.p[1] = t; }
        g.nprob = np;
        if (np > 0) DGG_TRY(gemm_launch(GEMM_NN, g, c.st));
    }
    return DGG_OK;
}

// ------------------------------------------------------------------------------------------------
// Whole-network forward / backward
// ------------------------------------------------------------------------------------------------
// The affine vectors the GEMMs fold into the [prefix | x_hat] operand are [1.. | gamma], [0.. | beta].  A network without
// prefix columns (D, E) needs no copies: its GEMMs read the input BatchNorm's gamma / beta in the parameter buffer itself (so
// a second forward with updated parameters — the G phase's pass through D — has nothing to rewrite).
static void aff_direct(const NetL& L, const float* params, NetW& W) {
    if (L.nprefix == 0 && L.k_in > 0) {
        W.aff_scale = const_cast<float*>(params + L.ibnw);
        W.aff_shift = const_cast<float*>(params + L.ibnb);
    }
}

static void make_segs(const NetL& L, NetW& W, int r, Seg* segs, int& nseg) {
    memset(segs, 0, 2 * sizeof(Seg));
    if (r == 0) {
        nseg = 1;
        segs[0].p = W.X0; segs[0].ld = L.ldx0; segs[0].width = L.k0; segs[0].wcol = 0; segs[0].is_x0 = true;
        return;
    }
    nseg = 2;
    const int ix0 = L.x1_first ? 1 : 0, ix1 = L.x1_first ? 0 : 1;
    segs[ix0].p = W.X0; segs[ix0].ld = L.ldx0; segs[ix0].width = L.k0; segs[ix0].wcol = L.x1_first ? L.nn : 0;
    segs[ix0].is_x0 = true;
    segs[ix1].p = W.rb[r - 1].y; segs[ix1].ld = L.nn; segs[ix1].width = L.nn; segs[ix1].wcol = L.x1_first ? 0 : L.k0;
}

static int net_forward(const NetL& L, const float* params, float* bn, void* ws, size_t ws_bytes, int B, unsigned flags,
                       const ColSrcs& srcs, const float* prefix, const float* drop_keep, uint64_t seed, float* out,
                       hipStream_t st, Share* share = nullptr, const XShare* xsh = nullptr) {
    if (!params || !ws || !out || B < 1) return DGG_ERR_INVALID;
    Ctx c;
    c.share = share;
    c.L = &L; c.P = params; c.G = nullptr; c.bn = bn; c.B = B; c.train = (flags & DGG_TRAIN) != 0; c.st = st;
    c.sw = nullptr;
    carve(L, B, ws, c.W);
    if (c.W.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    if (!bn) return DGG_ERR_INVALID;
    const bool reuse_norm = c.train && (flags & DGG_REUSE_INPUT_NORM) != 0;
    const bool consume = xsh && xsh->mode == 2;
    aff_direct(L, params, c.W);
    if (xsh && (!c.train || xsh->ld != L.ldx0 || !xsh->xs)) return DGG_ERR_INVALID;
    if (consume) c.W.X0 = xsh->xs - L.nprefix;   // [prefix | x_hat_bps | pose]: the shared rows, seen nprefix floats earlier
    if (c.train && !reuse_norm && !consume) DGG_TRY(ibn_stats_launch(srcs, B, 1, c.W.part, st));
    if (reuse_norm) {
        // x_hat, mean and rstd of this workspace are those of the previous train forward over these inputs
        if (!((flags & kRefreshElsewhere) && L.nprefix == 0))
            DGG_TRY(ibn_refresh_launch(c.W.imean, c.W.ivaru, bn + L.irm, bn + L.irv, params + L.ibnw, params + L.ibnb,
                                       L.nprefix ? c.W.aff_scale : nullptr, L.nprefix ? c.W.aff_shift : nullptr, L.nprefix,
                                       L.k_in, st));
    } else if (consume) {
        if (xsh->wait && xsh->ready && hipStreamWaitEvent(st, xsh->ready, 0) != hipSuccess) return DGG_ERR_LAUNCH;
        IbnShared q;
        memset(&q, 0, sizeof(q));
        q.n_alias = L.in_bps;
        q.src_mean = xsh->src_mean; q.src_rstd = xsh->src_rstd; q.src_varu = xsh->src_varu;
        q.mean = c.W.imean; q.rstd = c.W.irstd; q.varu = c.W.ivaru;
        q.run_mean = bn + L.irm; q.run_var = bn + L.irv;
        q.gamma = params + L.ibnw; q.beta = params + L.ibnb;
        q.aff_scale = L.nprefix ? c.W.aff_scale : nullptr; q.aff_shift = L.nprefix ? c.W.aff_shift : nullptr;
        q.nprefix = L.nprefix; q.prefix = prefix;
        q.X0 = c.W.X0; q.ldx = L.ldx0; q.B = B;
        q.npose = L.k_in - L.in_bps;
        if (q.npose > 0) {
            if (!xsh->heads) return DGG_ERR_INVALID;
            q.nj = q.npose - 12;
            q.heads = xsh->heads; q.ldh = xsh->ldh;
            q.rot9 = xsh->rot9; q.transl = xsh->transl; q.joint = xsh->joint;
        }
        DGG_TRY(ibn_shared_launch(q, st));
    } else {
        BnFused fz{c.W.part, c.train ? 1 : 0, bn + L.irm, bn + L.irv, c.W.imean, c.W.irstd, c.W.ivaru};
        const bool produce = xsh && xsh->mode == 1;
        DGG_TRY(ibn_apply_launch(srcs, B, B, c.W.imean, c.W.irstd, nullptr, nullptr, prefix, L.nprefix, c.W.X0, L.ldx0,
                                 params + L.ibnw, params + L.ibnb, L.nprefix ? c.W.aff_scale : nullptr,
                                 L.nprefix ? c.W.aff_shift : nullptr, st, &fz,
                                 produce ? xsh->xs : nullptr, produce ? xsh->ld : 0, produce ? L.in_bps : 0));
        if (produce && xsh->record && xsh->ready && hipEventRecord(xsh->ready, st) != hipSuccess) return DGG_ERR_LAUNCH;
    }
    // forward pre-products (see pre_splits_of): every block's X0 product in one grouped launch — the whole [prefix | x_hat]
    // segment, or, for the producer of a Share (D's real pass), its shared BPS columns, into the shared slabs
    if (c.W.pre_splits > 0 && !(share && share->mode == 2)) {
        const bool cut = share && share->mode == 1 && L.nprefix == 0 && share->ncols > 0 && share->ncols < L.k0;
        if (!share || cut) {
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            const long long slab = (long long)B * L.n13;
            for (int r = 0; r < L.nrb; ++r) {
                const RbL& R = L.rb[r];
                const int wcol = r == 0 ? 0 : (L.x1_first ? L.nn : 0);
                float* dst = cut ? share->slabs[r] : c.W.pre[r];
                GemmProb& q = g.p[g.nprob++];
                prob_init(q, c.W.X0, L.ldx0, params + R.w13 + wcol, R.fin, dst, L.n13, B, L.n13, cut ? share->ncols : L.k0);
                prob_split(q, c.W.pre_splits, slab);
                q.a_scale = c.W.aff_scale;
                q.a_shift = c.W.aff_shift;
                if (q.splits > (cut ? kShareSlabs : c.W.pre_splits + 1)) return DGG_ERR_WORKSPACE;
                c.pre_slabs[r] = dst;
                c.pre_nslab[r] = q.splits;
                if (cut) share->nslab[r] = q.splits;
            }
            DGG_TRY(gemm_launch(GEMM_NT, g, st));
            if (cut && share->events)
                for (int r = 0; r < L.nrb; ++r)
                    if (hipEventRecord(share->ready[r], st) != hipSuccess) return DGG_ERR_LAUNCH;
            c.pre = true;
        }
    }
    const bool drop = (L.net == DGG_NET_EVA) && c.train;
    for (int r = 0; r < L.nrb; ++r) {
        Seg segs[2];
        int nseg;
        make_segs(L, c.W, r, segs, nseg);
        // 1: masks given, 2: Philox with the seed passed by value, 3: Philox, `seed` is a device pointer to the seed
        const int mode = drop ? (drop_keep ? 1 : ((flags & DGG_SEED_ON_DEVICE) ? 3 : 2)) : 0;
        const float* keep = (drop && drop_keep) ? drop_keep + (size_t)r * B * L.nn : nullptr;
        DGG_TRY(rb_forward(c, r, segs, nseg, mode, keep, seed, (uint64_t)r * (uint64_t)B * L.nn));
    }
    const int sig = L.net != DGG_NET_GEN;
    DGG_TRY(heads_fwd_launch(c.W.rb[L.nrb - 1].y, L.nn, B, L.nn, params + L.wh, params + L.bh, L.nh, sig, out, L.nh,
                             c.W.slabs, st));
    return DGG_OK;
}

// restrict a column-source list to the columns [lo, hi) of the concat
static ColSrcs sub_srcs(const ColSrcs& S, int lo, int hi) {
    ColSrcs o;
    memset(&o, 0, sizeof(o));
    int base = 0;
    for (int i = 0; i < S.nsrc; ++i) {
        const int a = base, b = base + S.s[i].n;
        const int l = lo > a ? lo : a, h = hi < b ? hi : b;
        if (h > l) {
            o.s[o.nsrc].p = S.s[i].p + (l - a);
            o.s[o.nsrc].ld = S.s[i].ld;
            o.s[o.nsrc].n = h - l;
            o.total += h - l;
            ++o.nsrc;
        }
        base = b;
    }
    return o;
}

// d_out: gradient w.r.t. the heads' pre-activation outputs (B x nh).
// d_prefix (B x nprefix), d_in_lo/hi + d_in (B x (hi-lo)): input-gradient requests on the BN'd concat.
static int net_backward(const NetL& L, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                        unsigned flags, const ColSrcs& srcs, const float* d_out, float* d_prefix, int d_in_lo,
                        int d_in_hi, float* d_in, hipStream_t st, Share* share = nullptr, float* x0_alias = nullptr) {
    // x0_alias: the forward ran as an XShare consumer — its operand [prefix | x_hat] lies there (pitch L.ldx0), not in `ws`
    if (!params || !ws || !d_out || B < 1) return DGG_ERR_INVALID;
    const bool skip = (flags & DGG_SKIP_PARAM_GRADS) != 0;
    const bool accum = (flags & DGG_ACCUM_GRADS) != 0;
    if (!skip && !grads) return DGG_ERR_INVALID;
    Ctx c;
    c.share = share;
    c.L = &L; c.P = params; c.G = grads; c.bn = nullptr; c.B = B; c.train = (flags & DGG_TRAIN) != 0; c.st = st;
    c.sw = skip ? nullptr : wgrad_stream_of(st);
    if (c.sw == st) c.sw = nullptr;
    carve(L, B, ws, c.W);
    if (c.W.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    if (x0_alias) c.W.X0 = x0_alias;
    aff_direct(L, params, c.W);
    const bool want_in = d_in && d_in_hi > d_in_lo;
    // columns of X0 whose gradient is needed: only those for which the caller wants INPUT gradients —
    // the input BatchNorm's parameter gradients come out of the wgrad epilogue
    int x_lo = L.k0, x_hi = 0;
    if (d_prefix && L.nprefix > 0) { x_lo = 0; x_hi = L.nprefix; }
    if (want_in) {
        if (L.nprefix + d_in_lo < x_lo) x_lo = L.nprefix + d_in_lo;
        if (L.nprefix + d_in_hi > x_hi) x_hi = L.nprefix + d_in_hi;
    }
    // input gradients of a narrow column range only (the pose columns of D in the G phase): every block leaves its
    // contribution as split-K slabs inside the (otherwise unused) dX0 buffer — block after block, so that the
    // input BatchNorm's backward sums them in a fixed order — instead of 16 workgroups walking K = 768 three times
    const int pose_used = splits_used(L.n13, kPoseKs);
    const bool pose_slabs = want_in && !(d_prefix && L.nprefix > 0) && x_hi - x_lo <= kPoseLd && x_lo == L.nprefix + d_in_lo &&
                            L.n13 >= kPoseKs * 4 * 32 && (int64_t)L.nrb * pose_used * kPoseLd <= L.ldx0;
    int ibn_tiles[3] = {0, 0, 0};   // row tiles of block r's epilogue partials
    const bool drop = (L.net == DGG_NET_EVA) && c.train;
    const GradEvents* ge = skip ? nullptr : grad_events_for(grads);
    if (ge && (ge->n != L.nrb || (ge->accum_only && !accum))) ge = nullptr;
    // heads
    float* dy = c.W.dXa;
    float* other = c.W.dXb;
    HeadBwd hb{d_out, params + L.wh, skip ? nullptr : grads + L.wh, skip ? nullptr : grads + L.bh};
    const bool fold_head = L.nh == 1;  // the outer product d_score (x) w is formed inside the last block's backward
    // the last block's weight-gradient launch has room for the heads' dWh (problems: the input segments (<= 2) + fc2 = at
    // most 3 of 4; Share splits the X0 segment only for the discriminator, whose head folds)
    const bool heads_ride = !fold_head && !skip;
    HeadsRide hride{d_out, L.nh, skip ? nullptr : grads + L.wh};
    // several heads (the generator): dy = d_out Wh is formed inside the last block's first backward stage as well (round 6:
    // a K = 21 GEMM launch of its own until then); the weight gradient rides in that block's weight-gradient launch, the bias
    // gradient was formed with d_out (gen_loss_kernel) or takes one small launch
    memset(&c.dw, 0, sizeof(c.dw));
    // (fine-grained real / fake order, Share::bwd_fine: both passes defer — the per-block events cover the real pass'
    //  element-wise stages, which is what the fake pass' element-wise stages accumulate onto; the fake pass' grouped launch
    //  accumulates onto the real pass' grouped launch and waits for the real pass' END, once, in front of it)
    c.defer_wgrad = !skip && !ge && !c.sw && defer_wgrad_enabled();
    HeadBwd hbm{d_out, params + L.wh, nullptr, nullptr, L.nh};
    if (!fold_head && !skip && !(flags & kHeadBiasDone))
        DGG_TRY(heads_bias_grad_launch(d_out, L.nh, B, L.nh, grads + L.bh, accum ? 1 : 0, st));
    for (int r = L.nrb - 1; r >= 0; --r) {
        Seg segs[2];
        int nseg;
        make_segs(L, c.W, r, segs, nseg);
        for (int i = 0; i < nseg; ++i) {
            if (segs[i].p == c.W.X0) {
                if (x_hi > x_lo) {
                    segs[i].dp = c.W.dX0; segs[i].dld = L.ldx0; segs[i].d_lo = x_lo; segs[i].d_hi = x_hi;
                    segs[i].d_accum = (r != L.nrb - 1) ? 1 : 0;
                    if (pose_slabs) {
                        segs[i].d_slabs = c.W.dX0 + (size_t)(L.nrb - 1 - r) * pose_used * B * kPoseLd;
                        segs[i].d_ks = kPoseKs;
                    }
                }
            } else {
                segs[i].dp = other; segs[i].dld = L.nn; segs[i].d_lo = 0; segs[i].d_hi = L.nn; segs[i].d_accum = 0;
                segs[i].d_split = true;
            }
        }
        DGG_TRY(rb_backward(c, r, segs, nseg, dy, drop, accum, skip, ibn_tiles,
                            r == L.nrb - 1 ? (fold_head ? &hb : &hbm) : nullptr,
                            (heads_ride && r == L.nrb - 1) ? &hride : nullptr));
        // block r's parameter gradients (and, for the last block, the heads') are complete on the stream that ran its
        // weight-gradient GEMMs: a data-parallel caller starts that bucket's all-reduce behind this event
        if (ge && r >= 1 && hipEventRecord(ge->ev[r], c.sw ? c.sw : st) != hipSuccess) return DGG_ERR_LAUNCH;
        float* t = dy; dy = other; other = t;
    }
    if (c.dw.nprob > 0) {   // the pass' weight gradients, one grouped launch
        if (share && share->bwd_fine && share->bwd_mode == 2 && share->bwd_end &&
            hipStreamWaitEvent(st, share->bwd_end, 0) != hipSuccess)
            return DGG_ERR_LAUNCH;
        DGG_TRY(gemm_launch(GEMM_TN, c.dw, st));
    }
    // input BatchNorm backward: parameter gradients = sum over ResBlocks and row tiles of the epilogue partials
    if (c.sw) {  // join: everything queued after this call on `st` sees the complete gradient buffer
        hipEvent_t e = next_event();
        if (!e || hipEventRecord(e, c.sw) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess)
            return DGG_ERR_LAUNCH;
    }
    if (!skip) {
        // shared BPS weight gradient (Share::bwd_mode, same conditions as in rb_backward): the real pass produced
        // partials for the pose columns only; the fake pass overwrites the BPS columns and accumulates the rest
        const bool share_ok = share && share->bwd_mode && L.nprefix == 0 && share->ncols > 0 && share->ncols < L.k0;
        // part layout per block: [tiles][2][k0], blocks 12 * 2 * k0 apart; q0 -> d gamma, q1 -> d beta.  One launch sums
        // the blocks (groups) of a column range when they all used the same tile count — they do within one pass.
        // fin(c_lo, ncols, tiles, noacc): BatchNorm columns [c_lo, c_lo + ncols); the first `noacc` of them are overwritten
        // even when accumulating
        auto fin = [&](int c_lo, int ncols, const int* tiles, int noacc) -> int {
            if (ncols <= 0) return DGG_OK;
            bool same = true;
            for (int r = 1; r < L.nrb; ++r) same = same && tiles[r] == tiles[0];
            if (same) {
                if (tiles[0] <= 0) return DGG_ERR_INVALID;
                return colsum_finalize_launch(c.W.ibn_part, L.nrb, tiles[0], 2, ncols, nullptr, grads + L.ibnw + c_lo,
                                              grads + L.ibnb + c_lo, nullptr, accum ? 1 : 0, st, L.k0, L.nprefix + c_lo,
                                              nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr,
                                              noacc, (long long)12 * 2 * L.k0);
            }
            for (int r = 0; r < L.nrb; ++r) {
                if (tiles[r] <= 0) return DGG_ERR_INVALID;
                DGG_TRY(colsum_finalize_launch(c.W.ibn_part + (size_t)r * 12 * 2 * L.k0, 1, tiles[r], 2, ncols, nullptr,
                                               grads + L.ibnw + c_lo, grads + L.ibnb + c_lo, nullptr,
                                               (accum || r > 0) ? 1 : 0, st, L.k0, L.nprefix + c_lo, nullptr, nullptr,
                                               nullptr, 0, nullptr, 0, nullptr, nullptr, r == 0 ? noacc : 0));
            }
            return DGG_OK;
        };
        const int c_first = (share_ok && share->bwd_mode == 1) ? share->ncols : 0;
        if (share_ok && share->bwd_mode == 2) {   // one launch: the BPS columns overwritten, the pose columns accumulated
            DGG_TRY(fin(0, L.k_in, ibn_tiles, share->ncols));
        } else {
            DGG_TRY(fin(c_first, L.k_in - c_first, ibn_tiles, 0));
        }
    }
    // everything else — block 0 and the input BatchNorm — is complete here
    if (ge && hipEventRecord(ge->ev[0], st) != hipSuccess) return DGG_ERR_LAUNCH;
    if (want_in) {  // input gradients for the requested columns through the (train: batch-statistics) BN backward
        ColSrcs sub = sub_srcs(srcs, d_in_lo, d_in_hi);
        const float* dsrc = c.W.dX0;
        int dld = L.ldx0, dcol = L.nprefix + d_in_lo, nslab = 0;
        long long dstride = 0;
        if (pose_slabs) { dld = kPoseLd; dcol = 0; nslab = L.nrb * pose_used; dstride = (long long)B * kPoseLd; }
        if (c.train) {
            // the statistics pass also leaves the slab sum, compact, in dXa (free: the block chain is done)
            float* dsum = nslab > 1 ? c.W.dXa : nullptr;
            DGG_TRY(ibn_bwd_stats_launch(sub, dsrc, dld, dcol, c.W.imean + d_in_lo, c.W.irstd + d_in_lo, B, 1, c.W.part, st,
                                         nslab, dstride, dsum));
            if (dsum) { dsrc = dsum; dld = sub.total; dcol = 0; nslab = 0; }
        }
        // the column sums of the statistics pass are formed inside the apply kernel (no finalize launch)
        DGG_TRY(ibn_bwd_apply_launch(sub, dsrc, dld, dcol, c.W.imean + d_in_lo, c.W.irstd + d_in_lo,
                                     params + L.ibnw + d_in_lo, nullptr, c.train, B, B, 0, sub.total, d_in, sub.total, st,
                                     c.train ? c.W.part : nullptr, nslab, dstride));
    }
    if (d_prefix && L.nprefix > 0) DGG_TRY(copy_cols_launch(c.W.dX0, L.ldx0, d_prefix, L.nprefix, B, L.nprefix, st));
    return DGG_OK;
}

static ColSrcs srcs_gen(const dgg_cfg* cfg, const float* bps) {
    ColSrcs s;
    memset(&s, 0, sizeof(s));
    s.nsrc = 1;
    s.s[0].p = bps; s.s[0].ld = cfg->in_bps; s.s[0].n = cfg->in_bps;
    s.total = cfg->in_bps;
    return s;
}

static ColSrcs srcs_pose(const dgg_cfg* cfg, const float* bps, const float* rot9, const float* transl,
                         const float* joint) {
    ColSrcs s;
    memset(&s, 0, sizeof(s));
    s.s[0].p = bps; s.s[0].ld = cfg->in_bps; s.s[0].n = cfg->in_bps;
    s.s[1].p = rot9; s.s[1].ld = 9; s.s[1].n = 9;
    s.s[2].p = transl; s.s[2].ld = 3; s.s[2].n = 3;
    s.nsrc = 3;
    s.total = cfg->in_bps + 12;
    if (joint) {
        s.s[3].p = joint; s.s[3].ld = cfg->n_hand_joints; s.s[3].n = cfg->n_hand_joints;
        s.nsrc = 4;
        s.total += cfg->n_hand_joints;
    }
    return s;
}

// ------------------------------------------------------------------------------------------------
// Inference over many objects, factorised per object (BASELINE config 5).  generate_grasps / filter_grasps
// tile ONE BPS row over the n samples of an object (dexgangrasp.py:495-514, :406-411): in eval mode (running
// statistics: rows do not interact) the BPS part of every fused fc1||fc3 product, bn(bps) W13[:, bps]^T + b13, is one row
// per OBJECT.  It is computed once per object (M = objects instead of objects x n: 200x fewer MACs at n = 200) and
// every sample adds only its own few columns (generator: Z, 5; evaluator: rot9 | transl, 12) and, from the second
// block on, the previous block's 512 outputs.  The (objects x n) x 4.1k input block is never materialised.
// ------------------------------------------------------------------------------------------------
struct ObjW {
    float *Xobj, *imean, *irstd, *aff_scale, *aff_shift, *cobj[3], *oslabs;
    NetW S;  // sample-level buffers in NetW form (rb[r] share a13 / h / a2; y alternates)
    int osplits;
    size_t bytes;
};

static void carve_objects(const NetL& L, int O, int n, void* base, ObjW& W) {
    Bump b{reinterpret_cast<char*>(base), 0};
    const size_t Oz = (size_t)O, Bz = (size_t)O * (size_t)n;
    const int kb = (int)align4(L.in_bps);
    W.Xobj = b.take<float>(Oz * kb);
    W.imean = b.take<float>(2 * (size_t)L.in_bps);
    W.irstd = b.take<float>(2 * (size_t)L.in_bps);
    W.aff_scale = b.take<float>((size_t)L.in_bps);
    W.aff_shift = b.take<float>((size_t)L.in_bps);
    for (int r = 0; r < 3; ++r) W.cobj[r] = r < L.nrb ? b.take<float>(Oz * L.n13) : nullptr;
    {
        int widths[1] = {L.in_bps}, sp[1];
        plan13(O, L.n13, widths, 1, sp);
        W.osplits = sp[0];
    }
    W.oslabs = b.take<float>((size_t)(W.osplits + 1) * Oz * L.n13);
    memset(&W.S, 0, sizeof(W.S));
    float* a13 = b.take<float>(Bz * L.n13);
    float* h = b.take<float>(Bz * L.hid);
    float* a2 = b.take<float>(Bz * L.nn);
    float* y[2] = {b.take<float>(Bz * L.nn), b.take<float>(Bz * L.nn)};
    float* m1 = b.take<float>(2 * (size_t)L.hid);
    float* r1 = b.take<float>(2 * (size_t)L.hid);
    float* m2 = b.take<float>(2 * (size_t)L.nn);
    float* r2 = b.take<float>(2 * (size_t)L.nn);
    for (int r = 0; r < L.nrb; ++r) {
        RbW& R = W.S.rb[r];
        R.a13 = a13; R.h = h; R.a2 = a2; R.y = y[r & 1]; R.keep = nullptr;
        R.mean1 = m1; R.rstd1 = r1; R.mean2 = m2; R.rstd2 = r2;
    }
    // split-K slabs of the sample-level GEMMs: the previous block's product (B x 768) / fc2 (B x 512), few splits at
    // these row counts; heads scratch
    int wd[1] = {L.nn}, sp[1];
    plan13((int)Bz, L.n13, wd, 1, sp);
    int64_t need = (int64_t)(sp[0] + 1) * (int64_t)Bz * L.n13;
    const int64_t need2 = (int64_t)(gemm_pick_splits(GEMM_NT, (int)Bz, L.nn, L.hid, 0) + 1) * (int64_t)Bz * L.nn;
    if (need2 > need) need = need2;
    const int64_t needh = (int64_t)8 * (int64_t)Bz * (L.nh > 1 ? L.nh : 1);
    if (needh > need) need = needh;
    W.S.slab_floats = need;
    W.S.slabs = b.take<float>((size_t)need);
    const size_t nrb = (Bz + kRowBlock - 1) / kRowBlock + 2;
    W.S.part = b.take<float>(nrb * 2 * (size_t)L.n13);
    W.S.part2 = b.take<float>(nrb * 2 * (size_t)L.n13);  // rb_tail: the fc2 GEMM's bn2 partials (it still reads part)
    W.bytes = (b.off + 255) & ~(size_t)255;
}

// small: the per-sample raw columns (generator: Z; evaluator: rot9 | transl), `out`: heads (O*n, nh).
static int objects_forward(const NetL& L, const float* params, float* bn, void* ws, size_t ws_bytes, int O, int n,
                           const float* bps_obj, const ColSrcs& small, float* out, hipStream_t st) {
    if (!params || !bn || !ws || !bps_obj || !out || O < 1 || n < 1) return DGG_ERR_INVALID;
    if ((int64_t)O * n > (int64_t)1 << 24) return DGG_ERR_INVALID;
    ObjW W;
    carve_objects(L, O, n, ws, W);
    if (W.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    const int B = O * n;
    const int kb = (int)align4(L.in_bps);
    const bool small_first = L.nprefix > 0;              // generator: X0 = [Z | bps]; evaluator: X0 = [bps | pose]
    const int bps_off = small_first ? L.nprefix : 0;     // first BPS column inside the X0 segment
    const int small_off = small_first ? 0 : L.in_bps;
    if (small.total != (small_first ? L.nprefix : L.k_in - L.in_bps)) return DGG_ERR_INVALID;
    // 1. x_hat of the objects' BPS rows (running statistics) + the affine vectors the GEMM folds in
    {
        ColSrcs sb;
        memset(&sb, 0, sizeof(sb));
        sb.nsrc = 1; sb.s[0].p = bps_obj; sb.s[0].ld = L.in_bps; sb.s[0].n = L.in_bps; sb.total = L.in_bps;
        BnFused fz{W.S.part, 0, bn + L.irm, bn + L.irv, W.imean, W.irstd};
        DGG_TRY(ibn_apply_launch(sb, O, O, W.imean, W.irstd, nullptr, nullptr, nullptr, 0, W.Xobj, kb, params + L.ibnw,
                                 params + L.ibnb, W.aff_scale, W.aff_shift, st, &fz));
    }
    auto wcol_x0 = [&](int r) { return r == 0 ? 0 : (L.x1_first ? L.nn : 0); };
    // 2. per object and ResBlock: cobj = bn(bps) W13[:, bps]^T + b13
    for (int r = 0; r < L.nrb; ++r) {
        const RbL& R = L.rb[r];
        GemmBatch g;
        memset(&g, 0, sizeof(g));
        g.nprob = 1;
        prob_init(g.p[0], W.Xobj, kb, params + R.w13 + wcol_x0(r) + bps_off, R.fin, W.oslabs, L.n13, O, L.n13, L.in_bps);
        prob_split(g.p[0], W.osplits, (long long)O * L.n13);
        g.p[0].a_scale = W.aff_scale;
        g.p[0].a_shift = W.aff_shift;
        DGG_TRY(gemm_launch(GEMM_NT, g, st));
        DGG_TRY(reduce_bias_launch(W.oslabs, g.p[0].splits, (long long)O * L.n13, L.n13, params + R.b13, W.cobj[r], L.n13,
                                   O, O, 1, L.n13, 0, 0, 0, 0, nullptr, st));
    }
    // 3. per sample
    Ctx c;
    c.share = nullptr;
    c.L = &L; c.P = params; c.G = nullptr; c.bn = bn; c.W = W.S; c.B = B; c.train = false; c.st = st; c.sw = nullptr;
    const float* sg = small_first ? nullptr : params + L.ibnw + L.in_bps;
    const float* sbeta = small_first ? nullptr : params + L.ibnb + L.in_bps;
    const float* srm = small_first ? nullptr : bn + L.irm + L.in_bps;
    const float* srv = small_first ? nullptr : bn + L.irv + L.in_bps;
    for (int r = 0; r < L.nrb; ++r) {
        const RbL& R = L.rb[r];
        int nslab = 0;
        const long long slab = (long long)B * L.n13;
        if (r > 0) {  // the previous block's 512 outputs: X1 W13[:, x1]^T
            int wd[1] = {L.nn}, sp[1];
            plan13(B, L.n13, wd, 1, sp);
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            g.nprob = 1;
            prob_init(g.p[0], c.W.rb[r - 1].y, L.nn, params + R.w13 + (L.x1_first ? 0 : L.k0), R.fin, c.W.slabs, L.n13, B,
                      L.n13, L.nn);
            prob_split(g.p[0], sp[0], slab);
            if ((int64_t)g.p[0].splits * slab > c.W.slab_floats) return DGG_ERR_WORKSPACE;
            DGG_TRY(gemm_launch(GEMM_NT, g, st));
            nslab = g.p[0].splits;
        }
        // inference tail: expand_rows leaves h = lrelu(bn1(a1)) (running statistics) for the first `hid` columns, so fc2
        // is a plain 128 x 128-tile product on the bf16x3 kernel (K = 256, no split: B is large) and its bias goes into
        // the output stage — against rb_tail's f32-MFMA stage kernel, whose statistics work an eval pass does not need
        RbW& A = c.W.rb[r];
        const bool plain = obj_plain_fc2();
        DGG_TRY(expand_rows_launch(W.cobj[r], L.n13, n, c.W.slabs, nslab, slab, L.n13, small, sg, sbeta, srm, srv,
                                   params + R.w13 + wcol_x0(r) + small_off, R.fin, A.a13, L.n13, B, L.n13, L.hid, L.n13,
                                   st, plain ? A.h : nullptr, L.hid, L.hid, params + R.bn1w, params + R.bn1b,
                                   bn + R.rm1, bn + R.rv1));
        if (!plain) {
            DGG_TRY(rb_tail(c, r, 0, nullptr, 0, 0));
            continue;
        }
        {
            GemmBatch g2;
            memset(&g2, 0, sizeof(g2));
            g2.nprob = 1;
            prob_init(g2.p[0], A.h, L.hid, params + R.w2, L.hid, A.a2, L.nn, B, L.nn, L.hid);
            DGG_TRY(gemm_launch(GEMM_NT, g2, st));
        }
        {
            BnFused fz{c.W.part, 0, bn + R.rm2, bn + R.rv2, A.mean2, A.rstd2};
            DGG_TRY(res_out_launch(A.a13 + L.hid, L.n13, A.a2, L.nn, B, B, L.nn, A.mean2, A.rstd2, params + R.bn2w,
                                   params + R.bn2b, 0, nullptr, 0, 0, nullptr, A.y, L.nn, st, &fz, params + R.b2));
        }
    }
    const int sig = L.net != DGG_NET_GEN;
    DGG_TRY(heads_fwd_launch(c.W.rb[L.nrb - 1].y, L.nn, B, L.nn, params + L.wh, params + L.bh, L.nh, sig, out, L.nh,
                             c.W.slabs, st));
    return DGG_OK;
}

// ------------------------------------------------------------------------------------------------
// Training phases (dexgangrasp.py:739-823)
// ------------------------------------------------------------------------------------------------
struct StepW {
    void *ws_g, *ws_dr, *ws_df;
    size_t g_bytes, d_bytes;
    float *heads, *rfake, *f_transl, *f_joint, *p_real, *p_fake, *p_gen, *dl_real, *dl_fake, *dl_gen, *d_pose,
        *d_heads;
    float* dshare[3];  // per ResBlock of D: split-K slabs of the BPS-column product shared by the real and fake pass
    float* bn_snap;    // D's running statistics as they were before a prefetched real pass advanced them (dropped pass: restored)
    float* gl_part;    // gen_loss_kernel's per-workgroup partial sums (gen_loss_scratch_floats)
    unsigned* ticket;  // GenLossFin::ticket: cleared by the G phase's loss-gradient launch, drawn by gen_loss_kernel's workgroups
    int d_nrb, d_n13, d_nbn;
    // XShare: two generations of the shared rows (the real pass of step k + 1 is prefetched into one while step k's
    // generator backward still reads the other), B x xs_ld floats each behind 64 floats of headroom (the generator's view
    // of row 0 starts nprefix floats in front of it); xs_ok: this configuration can share (see carve_step)
    float* xs[2];
    int xs_ld;
    bool xs_ok;
    size_t bytes;
};

static int carve_step(const dgg_cfg* cfg, int B, void* base, StepW& S) {
    NetL Lg, Ld;
    DGG_TRY(build_layout(DGG_NET_GEN, cfg, Lg));
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, Ld));
    NetW t;
    carve(Lg, B, nullptr, t);
    S.g_bytes = t.bytes;
    carve(Ld, B, nullptr, t);
    S.d_bytes = t.bytes;
    Bump b{reinterpret_cast<char*>(base), 0};
    S.ws_g = b.take<char>(S.g_bytes);
    S.ws_dr = b.take<char>(S.d_bytes);
    S.ws_df = b.take<char>(S.d_bytes);
    const size_t Bz = (size_t)B;
    const int nj = cfg->n_hand_joints;
    S.heads = b.take<float>(Bz * (nj + 9));
    S.rfake = b.take<float>(Bz * 9);
    S.f_transl = b.take<float>(Bz * 3);
    S.f_joint = b.take<float>(Bz * nj);
    S.p_real = b.take<float>(Bz);
    S.p_fake = b.take<float>(Bz);
    S.p_gen = b.take<float>(Bz);
    S.dl_real = b.take<float>(Bz);
    S.dl_fake = b.take<float>(Bz);
    S.dl_gen = b.take<float>(Bz);
    S.d_pose = b.take<float>(Bz * (12 + nj));
    S.d_heads = b.take<float>(Bz * (nj + 9));
    S.d_nrb = Ld.nrb;
    S.d_n13 = Ld.n13;
    S.d_nbn = (int)Ld.nbn;
    S.bn_snap = b.take<float>((size_t)Ld.nbn);
    S.gl_part = b.take<float>((size_t)gen_loss_scratch_floats(cfg, B));
    S.ticket = b.take<unsigned>(1);
    for (int r = 0; r < 3; ++r) {
        S.dshare[r] = nullptr;
        if (r >= Ld.nrb) continue;
        // as rb_forward cuts the pieces: [BPS | pose] (+ the previous block's output)
        int widths[3] = {cfg->in_bps, Ld.k0 - cfg->in_bps, Ld.nn}, sp[3];
        plan13(B, Ld.n13, widths, r == 0 ? 2 : 3, sp);
        const int pre_s = pre_splits_of(Ld, B) + 1;   // (the real pass' pre-product launch may cut the BPS piece otherwise)
        S.dshare[r] = b.take<float>((size_t)(sp[0] > pre_s ? sp[0] : pre_s) * Bz * Ld.n13);
    }
    // XShare needs: the generator's and the discriminator's operands on ONE row pitch, room for the latent in the pad
    // behind a discriminator row, the fake pose in ibn_shared_kernel's one instantiation (12 hand joints: the shipped
    // configuration), a batch whose fake pose one workgroup holds in LDS, a row per thread (B <= 1024); anything else
    // normalises per pass as before
    S.xs_ld = Ld.ldx0;
    S.xs_ok = xshare_enabled() && Lg.ldx0 == Ld.ldx0 && Lg.in_bps == Ld.in_bps && Ld.ldx0 - Ld.k0 >= Lg.nprefix &&
              Lg.nprefix <= 64 && cfg->n_hand_joints == 12 && B >= 2 && B <= 1024;
    for (int i = 0; i < 2; ++i) {
        float* base_ = b.take<float>(64 + Bz * Ld.ldx0);
        S.xs[i] = base_ ? base_ + 64 : nullptr;
    }
    S.bytes = (b.off + 255) & ~(size_t)255;
    return DGG_OK;
}

}  // namespace dgg

using namespace dgg;

// =================================================================================================
extern "C" {

const char* dgg_version(void) { return "dgg 0.2 (gfx950)"; }
const char* dgg_last_error(void) { return g_err; }

int dgg_table(int net, const dgg_cfg* cfg, dgg_entry* out, int cap) {
    NetL L;
    int rc = build_layout(net, cfg, L);
    if (rc) return rc;
    return fill_table(L, cfg, out, cap);
}

int64_t dgg_param_floats(int net, const dgg_cfg* cfg) {
    NetL L;
    if (build_layout(net, cfg, L)) return DGG_ERR_INVALID;
    return L.nparam;
}

int64_t dgg_bnstat_floats(int net, const dgg_cfg* cfg) {
    NetL L;
    if (build_layout(net, cfg, L)) return DGG_ERR_INVALID;
    return L.nbn;
}

int64_t dgg_workspace_bytes(int net, const dgg_cfg* cfg, int batch) {
    NetL L;
    if (batch < 1 || build_layout(net, cfg, L)) return DGG_ERR_INVALID;
    NetW W;
    carve(L, batch, nullptr, W);
    return (int64_t)W.bytes;
}

int dgg_bps_encode(const float* clouds, int B, int N, const float* basis, int P, float* dists, int32_t* ids,
                   void* stream) {
    return bps_encode_launch(clouds, B, N, basis, P, dists, ids, nullptr, 0, (hipStream_t)stream);
}
int64_t dgg_bps_workspace_bytes(int B, int N, int P) { return bps_workspace_bytes(B, N, P); }
int dgg_bps_encode_ws(const float* clouds, int B, int N, const float* basis, int P, float* dists, int32_t* ids, void* ws,
                      size_t ws_bytes, void* stream) {
    return bps_encode_launch(clouds, B, N, basis, P, dists, ids, ws, ws_bytes, (hipStream_t)stream);
}
int dgg_set_bps_algo(int algo) {
    if (algo != DGG_BPS_AUTO && algo != DGG_BPS_SCAN) return DGG_ERR_INVALID;
    bps_set_algo(algo);
    return DGG_OK;
}
int dgg_get_bps_algo(void) { return bps_algo(); }

int dgg_gen_forward(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                    unsigned flags, const float* z, const float* bps, float* out_heads, void* stream) {
    NetL L;
    flags &= kPublicFlags;   // internal bits (kHeadBiasDone) cannot come in through the C-ABI
    DGG_TRY(build_layout(DGG_NET_GEN, cfg, L));
    if (!z || !bps) return DGG_ERR_INVALID;
    return net_forward(L, params, bnstats, ws, ws_bytes, B, flags, srcs_gen(cfg, bps), z, nullptr, 0, out_heads,
                       (hipStream_t)stream);
}

int dgg_gen_backward(const dgg_cfg* cfg, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                     unsigned flags, const float* z, const float* bps, const float* d_heads, float* d_z, float* d_bps,
                     void* stream) {
    NetL L;
    flags &= kPublicFlags;   // internal bits (kHeadBiasDone) cannot come in through the C-ABI
    DGG_TRY(build_layout(DGG_NET_GEN, cfg, L));
    if (!bps) return DGG_ERR_INVALID;
    (void)z;
    return net_backward(L, params, grads, ws, ws_bytes, B, flags, srcs_gen(cfg, bps), d_heads, d_z, 0,
                        d_bps ? L.k_in : 0, d_bps, (hipStream_t)stream);
}

int dgg_disc_forward(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                     unsigned flags, const float* bps, const float* rot9, const float* transl, const float* joint,
                     float* out_p, void* stream) {
    NetL L;
    flags &= kPublicFlags;   // internal bits (kHeadBiasDone) cannot come in through the C-ABI
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, L));
    if (!bps || !rot9 || !transl || !joint) return DGG_ERR_INVALID;
    return net_forward(L, params, bnstats, ws, ws_bytes, B, flags, srcs_pose(cfg, bps, rot9, transl, joint), nullptr,
                       nullptr, 0, out_p, (hipStream_t)stream);
}

int dgg_disc_backward(const dgg_cfg* cfg, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                      unsigned flags, const float* bps, const float* rot9, const float* transl, const float* joint,
                      const float* d_logit, float* d_pose, void* stream) {
    NetL L;
    flags &= kPublicFlags;   // internal bits (kHeadBiasDone) cannot come in through the C-ABI
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, L));
    if (!bps || !rot9 || !transl || !joint) return DGG_ERR_INVALID;
    const bool want = (flags & DGG_NEED_INPUT_GRADS) != 0;
    if (want && !d_pose) return DGG_ERR_INVALID;
    return net_backward(L, params, grads, ws, ws_bytes, B, flags, srcs_pose(cfg, bps, rot9, transl, joint), d_logit,
                        nullptr, want ? cfg->in_bps : 0, want ? L.k_in : 0, want ? d_pose : nullptr,
                        (hipStream_t)stream);
}

int dgg_eva_forward(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                    unsigned flags, const float* bps, const float* rot9, const float* transl, const float* drop_keep,
                    uint64_t seed, float* out_p, void* stream) {
    NetL L;
    flags &= kPublicFlags;   // internal bits (kHeadBiasDone) cannot come in through the C-ABI
    DGG_TRY(build_layout(DGG_NET_EVA, cfg, L));
    if (!bps || !rot9 || !transl) return DGG_ERR_INVALID;
    return net_forward(L, params, bnstats, ws, ws_bytes, B, flags, srcs_pose(cfg, bps, rot9, transl, nullptr), nullptr,
                       drop_keep, seed, out_p, (hipStream_t)stream);
}

int dgg_eva_backward(const dgg_cfg* cfg, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                     unsigned flags, const float* bps, const float* rot9, const float* transl, const float* d_logit,
                     float* d_pose, void* stream) {
    NetL L;
    flags &= kPublicFlags;   // internal bits (kHeadBiasDone) cannot come in through the C-ABI
    DGG_TRY(build_layout(DGG_NET_EVA, cfg, L));
    if (!bps || !rot9 || !transl) return DGG_ERR_INVALID;
    const bool want = (flags & DGG_NEED_INPUT_GRADS) != 0;
    if (want && !d_pose) return DGG_ERR_INVALID;
    return net_backward(L, params, grads, ws, ws_bytes, B, flags, srcs_pose(cfg, bps, rot9, transl, nullptr), d_logit,
                        nullptr, want ? cfg->in_bps : 0, want ? L.k_in : 0, want ? d_pose : nullptr,
                        (hipStream_t)stream);
}

int64_t dgg_objects_workspace_bytes(int net, const dgg_cfg* cfg, int n_obj, int n_per_obj) {
    NetL L;
    if (n_obj < 1 || n_per_obj < 1 || build_layout(net, cfg, L) || net == DGG_NET_DISC) return DGG_ERR_INVALID;
    ObjW W;
    carve_objects(L, n_obj, n_per_obj, nullptr, W);
    return (int64_t)W.bytes;
}

int dgg_gen_forward_objects(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes,
                            int n_obj, int n_per_obj, const float* z, const float* bps_obj, float* out_heads,
                            void* stream) {
    NetL L;
    DGG_TRY(build_layout(DGG_NET_GEN, cfg, L));
    if (!z) return DGG_ERR_INVALID;
    ColSrcs sm;
    memset(&sm, 0, sizeof(sm));
    sm.nsrc = 1; sm.s[0].p = z; sm.s[0].ld = cfg->latent_d; sm.s[0].n = cfg->latent_d; sm.total = cfg->latent_d;
    return objects_forward(L, params, bnstats, ws, ws_bytes, n_obj, n_per_obj, bps_obj, sm, out_heads, (hipStream_t)stream);
}

int dgg_eva_forward_objects(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes,
                            int n_obj, int n_per_obj, const float* bps_obj, const float* rot9, const float* transl,
                            float* out_p, void* stream) {
    NetL L;
    DGG_TRY(build_layout(DGG_NET_EVA, cfg, L));
    if (!rot9 || !transl) return DGG_ERR_INVALID;
    ColSrcs sm;
    memset(&sm, 0, sizeof(sm));
    sm.nsrc = 2;
    sm.s[0].p = rot9; sm.s[0].ld = 9; sm.s[0].n = 9;
    sm.s[1].p = transl; sm.s[1].ld = 3; sm.s[1].n = 3;
    sm.total = 12;
    return objects_forward(L, params, bnstats, ws, ws_bytes, n_obj, n_per_obj, bps_obj, sm, out_p, (hipStream_t)stream);
}

int dgg_ortho6d_forward(const float* o6, int ld_o6, int B, float* rot9, void* stream) {
    if (!o6 || !rot9 || B < 1 || ld_o6 < 6) return DGG_ERR_INVALID;
    return ortho6d_fwd_launch(o6, ld_o6, B, rot9, (hipStream_t)stream);
}

int dgg_ortho6d_backward(const float* o6, int ld_o6, int B, const float* d_rot9, float* d_o6, int ld_do6,
                         void* stream) {
    if (!o6 || !d_rot9 || !d_o6 || B < 1 || ld_o6 < 6 || ld_do6 < 6) return DGG_ERR_INVALID;
    return ortho6d_bwd_launch(o6, ld_o6, B, d_rot9, d_o6, ld_do6, (hipStream_t)stream);
}

int dgg_bce(const float* p, const float* labels, float target_const, int B, float weight, float loss_scale,
            float* loss_out, float* d_logit, void* stream) {
    if (!p || !loss_out || B < 1) return DGG_ERR_INVALID;
    return bce_launch(p, labels, target_const, B, weight, loss_scale, loss_out, d_logit, (hipStream_t)stream);
}

int dgg_adam_l2(float* params, const float* grads, float* m, float* v, int64_t n, int step, double lr, double beta1,
                double beta2, double eps, double weight_decay, double grad_scale, void* stream) {
    if (!params || !grads || !m || !v) return DGG_ERR_INVALID;
    return adam_launch(params, grads, m, v, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                       (hipStream_t)stream);
}

int dgg_gen_loss(const dgg_cfg* cfg, const dgg_hparams* hp, int B, const float* heads, const float* gt_rot9,
                 const float* gt_transl, const float* gt_joint, const float* fake_score, const float* d_pose,
                 float* losses, float* d_heads, void* stream) {
    if (!cfg_ok(cfg) || !hp || !heads || !gt_rot9 || !gt_transl || !gt_joint || !losses || B < 1)
        return DGG_ERR_INVALID;
    return gen_loss_launch(cfg, hp, B, heads, gt_rot9, gt_transl, gt_joint, fake_score, d_pose, losses, d_heads,
                           (hipStream_t)stream);
}

int dgg_translate_along_axis(float* transl, const float* rot9, int B, int axis, float offset, void* stream) {
    if (!transl || !rot9 || B < 1 || axis < 0 || axis > 2) return DGG_ERR_INVALID;
    return translate_axis_launch(transl, rot9, B, axis, offset, (hipStream_t)stream);
}

int64_t dgg_cgan_workspace_bytes(const dgg_cfg* cfg, int batch) {
    if (batch < 1) return DGG_ERR_INVALID;
    StepW S;
    if (carve_step(cfg, batch, nullptr, S)) return DGG_ERR_INVALID;
    return (int64_t)S.bytes;
}

static void aux_remove(int i) {
    g_aux.key[i] = g_aux.key[g_aux.n - 1];
    g_aux.aux[i] = g_aux.aux[g_aux.n - 1];
    g_aux.owned[i] = g_aux.owned[g_aux.n - 1];
    --g_aux.n;
}

int dgg_set_wgrad_stream(void* stream, void* wgrad_stream) {
    hipStream_t k = (hipStream_t)stream, a = (hipStream_t)wgrad_stream;
    if (a && a == k) return DGG_ERR_INVALID;
    for (int i = 0; i < g_aux.n; ++i)
        if (g_aux.key[i] == k) {
            if (g_aux.owned[i]) return DGG_ERR_INVALID;  // managed by dgg_enable_wgrad_stream
            if (a) { g_aux.aux[i] = a; return DGG_OK; }
            aux_remove(i);
            return DGG_OK;
        }
    if (!a) return DGG_OK;
    if (g_aux.n >= kAuxCap) return DGG_ERR_INVALID;
    g_aux.key[g_aux.n] = k;
    g_aux.aux[g_aux.n] = a;
    g_aux.owned[g_aux.n] = 0;
    ++g_aux.n;
    return DGG_OK;
}

int dgg_enable_wgrad_stream(void* stream, int enable) {
    hipStream_t k = (hipStream_t)stream;
    for (int i = 0; i < g_aux.n; ++i)
        if (g_aux.key[i] == k) {
            if (!g_aux.owned[i]) return DGG_ERR_INVALID;  // an explicit mapping exists
            if (enable) { ++g_aux.owned[i]; return DGG_OK; }
            if (--g_aux.owned[i] == 0) {
                hipStream_t dead = g_aux.aux[i];
                aux_remove(i);
                if (hipStreamDestroy(dead) != hipSuccess) return DGG_ERR_LAUNCH;  // pending work completes first
            }
            return DGG_OK;
        }
    if (!enable) return DGG_OK;
    if (g_aux.n >= kAuxCap) return DGG_ERR_INVALID;
    hipStream_t a;
    if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) return DGG_ERR_LAUNCH;
    g_aux.key[g_aux.n] = k;
    g_aux.aux[g_aux.n] = a;
    g_aux.owned[g_aux.n] = 1;
    ++g_aux.n;
    return DGG_OK;
}

// ---- pipelining across steps -------------------------------------------------------------------
// The real-batch pass of the NEXT D phase needs only D's parameters after this step's Adam and D's BatchNorm running
// statistics after this step's D forward #3 (dexgangrasp.py:788 follows :808 of the previous call): it can be
// enqueued on the side stream while this step's G phase is still running (dgg_cgan_d_real_prefetch).  The matching
// dgg_cgan_d_phase2 call then finds it here and launches only the generator / fake-batch half.
struct PendingReal {
    const void* ws;       // key: the step workspace this state belongs to (NULL = free slot)
    bool valid;           // a real pass has been prefetched for (B, bps, rot9, transl, joint)
    int B;
    int step_B;           // batch size of the step whose G phase left the D-free mark (the workspace is carved for it)
    const float *bps, *rot9, *transl, *joint;
    const float* d_params;  // the discriminator the prefetched pass was computed with
    float* d_bn;            // its running statistics (advanced by the pass) and their state before it
    const float* bn_snap;
    int nbn;
    hipEvent_t ev[7];     // fwd_done, bwd_done, 3 x share.ready, D-free, (one spare); created with the slot
    bool events_ok, dfree_set;
    Share share;
    int xpar;             // XShare: the generation of shared rows the prefetched real pass wrote
    int step_xpar;        // ... and the one the step in flight (D phase -> G phase) uses
    bool step_set;        // a D phase has run on this workspace since the slot was (re)assigned
    unsigned long long stamp;
};
static const int kPendingSlots = 16;  // step workspaces in use on one thread (a G phase finds its D phase's state here)
static thread_local PendingReal g_pending[kPendingSlots] = {};
static thread_local unsigned long long g_pending_clock = 0;
static const int kLossRealNext = 12;  // losses[] slot of a prefetched real pass' BCE term

// the slot of workspace `ws`; with create: a free slot or the least recently used one is (re)assigned
static PendingReal* pending_slot(const void* ws, bool create) {
    PendingReal* lru = &g_pending[0];
    for (int i = 0; i < kPendingSlots; ++i) {
        PendingReal& p = g_pending[i];
        if (p.ws == ws) { p.stamp = ++g_pending_clock; return &p; }
        if (p.stamp < lru->stamp) lru = &p;
    }
    if (!create) return nullptr;
    if (!lru->events_ok) {
        for (int i = 0; i < 7; ++i)
            if (hipEventCreateWithFlags(&lru->ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        lru->events_ok = true;
    }
    lru->ws = ws;
    lru->valid = false;
    lru->dfree_set = false;
    lru->step_B = 0;
    lru->xpar = lru->step_xpar = 0;
    lru->step_set = false;
    lru->stamp = ++g_pending_clock;
    return lru;
}

// D(real) forward (producing the shared BPS slabs), BCE(real, 1), D backward(real) overwriting d_grads, all on `sb`
// (dexgangrasp.py:788, :215-218, :793)
static int real_pass(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, const StepW& S, const NetL& Ld,
                     int B, const float* bps, const float* rot9, const float* transl, const float* joint,
                     float* loss_slot, hipStream_t sb, Share& share, hipEvent_t fwd_done, hipEvent_t bwd_done,
                     int xpar = -1, hipEvent_t x_ready = nullptr) {
    share.mode = 1;
    XShare xp;
    memset(&xp, 0, sizeof(xp));
    if (xpar >= 0) {   // the BPS block's x_hat also goes to generation `xpar` of the shared rows (XShare producer)
        xp.mode = 1; xp.xs = S.xs[xpar]; xp.ld = S.xs_ld; xp.ready = x_ready; xp.record = x_ready != nullptr;
    }
    DGG_TRY(net_forward(Ld, s->d_params, s->d_bn, S.ws_dr, S.d_bytes, B, DGG_TRAIN,
                        srcs_pose(cfg, bps, rot9, transl, joint), nullptr, nullptr, 0, S.p_real, sb, &share,
                        xpar >= 0 ? &xp : nullptr));
    if (fwd_done && hipEventRecord(fwd_done, sb) != hipSuccess) return DGG_ERR_LAUNCH;
    DGG_TRY(bce_launch(S.p_real, nullptr, 1.f, B, (float)hp->bce_weight, 0.5f, loss_slot, S.dl_real, sb));
    // (bwd_mode 1: the weight gradient of the BPS columns is left to the fake pass, which takes dY_real + dY_fake)
    share.bwd_mode = share_wgrad() ? 1 : 0;
    DGG_TRY(net_backward(Ld, s->d_params, s->d_grads, S.ws_dr, S.d_bytes, B, DGG_TRAIN, srcs_pose(cfg, bps, rot9, transl, joint),
                         S.dl_real, nullptr, 0, 0, nullptr, sb, &share));
    if (bwd_done && hipEventRecord(bwd_done, sb) != hipSuccess) return DGG_ERR_LAUNCH;
    return DGG_OK;
}

int dgg_cgan_d_real_prefetch(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws,
                             size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                             const float* joint, float* losses, void* stream2) {
    if (!s || !hp || !ws || !bps || !rot9 || !transl || !joint || !losses || !stream2) return DGG_ERR_INVALID;
    StepW S;
    DGG_TRY(carve_step(cfg, B, ws, S));
    if (S.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    hipStream_t sb = (hipStream_t)stream2;
    PendingReal* Pp = pending_slot(ws, true);
    if (!Pp) return DGG_ERR_LAUNCH;
    PendingReal& P = *Pp;
    // The workspace is carved by batch size: a prefetch for another B would lay its real-pass regions over the
    // generator / fake-pass regions the running G phase is still reading.  Refuse it (the caller simply skips the
    // prefetch for a ragged last batch).
    if (P.dfree_set && P.step_B != B) {
        snprintf(g_err, sizeof(g_err), "dgg_cgan_d_real_prefetch: batch %d differs from the running step's %d", B,
                 P.step_B);
        return DGG_ERR_INVALID;
    }
    // D's running statistics must already hold this step's third forward, its parameters this step's Adam update:
    // both precede the D-free mark that dgg_cgan_g_phase (same workspace) leaves on its stream
    if (P.dfree_set && hipStreamWaitEvent(sb, P.ev[5], 0) != hipSuccess) return DGG_ERR_LAUNCH;
    NetL Ld;
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, Ld));
    memset(&P.share, 0, sizeof(P.share));
    P.share.ncols = cfg->in_bps;
    P.share.events = true;
    for (int r = 0; r < S.d_nrb; ++r) {
        P.share.slabs[r] = S.dshare[r];
        P.share.ready[r] = P.ev[2 + r];
    }
    P.valid = false;
    // the pass advances D's running statistics one batch early: keep what they were, so that a pass that is not picked
    // up (another batch follows, a serial entry point, a checkpoint: dgg_cgan_drop_prefetch) leaves no trace
    DGG_TRY(copy_cols_launch(s->d_bn, S.d_nbn, S.bn_snap, S.d_nbn, 1, S.d_nbn, sb));
    // XShare: the step in flight reads generation step_xpar of the shared rows until its generator backward ends — this
    // pass writes the other one
    P.xpar = P.step_xpar ^ 1;
    DGG_TRY(real_pass(cfg, s, hp, S, Ld, B, bps, rot9, transl, joint, losses + kLossRealNext, sb, P.share, P.ev[0],
                      P.ev[1], S.xs_ok ? P.xpar : -1, S.xs_ok ? P.ev[6] : nullptr));
    P.B = B; P.bps = bps; P.rot9 = rot9; P.transl = transl; P.joint = joint;
    P.d_params = s->d_params;
    P.d_bn = s->d_bn; P.bn_snap = S.bn_snap; P.nbn = S.d_nbn;
    P.valid = true;
    return DGG_OK;
}

int dgg_cgan_d_phase2(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws, size_t ws_bytes,
                      int B, const float* bps, const float* rot9, const float* transl, const float* joint,
                      const float* z, float* losses, void* stream, void* stream2) {
    if (!s || !hp || !ws || !bps || !rot9 || !transl || !joint || !z || !losses) return DGG_ERR_INVALID;
    StepW S;
    DGG_TRY(carve_step(cfg, B, ws, S));
    if (S.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    const int nj = cfg->n_hand_joints;
    const int nh = nj + 9;
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sb = stream2 ? (hipStream_t)stream2 : st;  // the real-batch pass of D
    const bool fork = stream2 && stream2 != stream;
    NetL Ld;
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, Ld));
    // a real pass prefetched for exactly these buffers? (anything else: it is dropped and redone)
    PendingReal* Pp = pending_slot(ws, S.xs_ok);   // (XShare: the slot also carries the step's generation of shared rows)
    if (S.xs_ok && !Pp) return DGG_ERR_LAUNCH;
    const bool pre = fork && Pp && Pp->valid && Pp->B == B && Pp->bps == bps && Pp->rot9 == rot9 &&
                     Pp->transl == transl && Pp->joint == joint && Pp->d_params == s->d_params;
    if (Pp && Pp->valid && !pre) {   // dropped: undo its running-statistics update once it has run (its gradients are overwritten)
        if (hipStreamWaitEvent(st, Pp->ev[1], 0) != hipSuccess) return DGG_ERR_LAUNCH;
        DGG_TRY(copy_cols_launch(Pp->bn_snap, Pp->nbn, Pp->d_bn, Pp->nbn, 1, Pp->nbn, st));
    }
    if (Pp) Pp->valid = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    Share share;
    int real_slot = DGG_L_BCE_REAL;
    // XShare: which generation of the shared rows this step uses — the one a picked-up prefetch wrote, else the one the
    // previous step did not use — and the event behind the producer's ibn_apply
    int xpar = -1;
    hipEvent_t x_ready = nullptr;
    if (S.xs_ok) {
        xpar = pre ? Pp->xpar : (Pp->step_xpar ^ 1);
        Pp->step_xpar = xpar;
        Pp->step_set = true;
        x_ready = pre ? Pp->ev[6] : (fork ? next_event() : nullptr);
        if (!pre && fork && !x_ready) return DGG_ERR_LAUNCH;
    }
    if (pre) {
        share = Pp->share;
        ev[1] = Pp->ev[0];
        ev[2] = Pp->ev[1];
        real_slot = kLossRealNext;
    } else {
        if (fork) {
            for (int i = 0; i < 3; ++i)
                if (!(ev[i] = next_event())) return DGG_ERR_LAUNCH;
            if (hipEventRecord(ev[0], st) != hipSuccess || hipStreamWaitEvent(sb, ev[0], 0) != hipSuccess)
                return DGG_ERR_LAUNCH;
        }
        // the BPS-column partial products of D's ResBlocks are computed by the real pass and reused by the fake pass
        memset(&share, 0, sizeof(share));
        share.ncols = cfg->in_bps;
        share.events = fork;
        for (int r = 0; r < S.d_nrb; ++r) {
            share.slabs[r] = S.dshare[r];
            if (fork && !(share.ready[r] = next_event())) return DGG_ERR_LAUNCH;
        }
        // two streams, no look-ahead pass (the drop-in order): the real pass is the LONGER of the two chains that run side by
        // side (its forward and backward against the generator's and the fake pass' forward), so the fake pass' backward
        // trails it block by block instead of waiting for its end (Share::bwd_ev)
        if (fork && share_wgrad() && fine_bwd()) {
            share.bwd_fine = true;
            share.bwd_end = ev[2];
            for (int r = 0; r < S.d_nrb; ++r)
                if (!(share.bwd_ev[r] = next_event())) return DGG_ERR_LAUNCH;
        }
        // ---- stream B: the real-batch pass
        DGG_TRY(real_pass(cfg, s, hp, S, Ld, B, bps, rot9, transl, joint, losses + DGG_L_BCE_REAL, sb, share, ev[1],
                          ev[2], xpar, x_ready));
    }
    // ---- stream A: :772-776 generator forward (train mode, activations kept for the G phase), R = ortho6d(rot_6D)
    XShare xc;   // consumer side of the shared rows: the generator, then the fake pass
    memset(&xc, 0, sizeof(xc));
    if (xpar >= 0) {
        NetW Wr;
        carve(Ld, B, S.ws_dr, Wr);
        xc.mode = 2; xc.xs = S.xs[xpar]; xc.ld = S.xs_ld;
        xc.src_mean = Wr.imean; xc.src_rstd = Wr.irstd; xc.src_varu = Wr.ivaru;
        xc.ready = x_ready; xc.wait = x_ready != nullptr;
        NetL Lg;
        DGG_TRY(build_layout(DGG_NET_GEN, cfg, Lg));
        DGG_TRY(net_forward(Lg, s->g_params, s->g_bn, S.ws_g, S.g_bytes, B, DGG_TRAIN, srcs_gen(cfg, bps), z, nullptr, 0,
                            S.heads, st, nullptr, &xc));
        // (the fake pose — R = ortho6d(rot_6D), contiguous transl / joint — is written by the fake pass' ibn_shared launch)
        xc.wait = false;
        xc.heads = S.heads; xc.ldh = nh; xc.rot9 = S.rfake; xc.transl = S.f_transl; xc.joint = S.f_joint;
    } else {
        DGG_TRY(dgg_gen_forward(cfg, s->g_params, s->g_bn, S.ws_g, S.g_bytes, B, DGG_TRAIN, z, bps, S.heads, stream));
        // R = ortho6d(rot_6D); fake transl / joint live inside the heads buffer with pitch nh: contiguous copies for D
        DGG_TRY(fake_pose_launch(S.heads, nh, nj, B, S.rfake, S.f_transl, S.f_joint, st));
    }
    // :789 D(fake.detach()): its BatchNorm running-statistic updates must follow those of the real pass
    if (fork && hipStreamWaitEvent(st, ev[1], 0) != hipSuccess) return DGG_ERR_LAUNCH;
    share.mode = 2;
    DGG_TRY(net_forward(Ld, s->d_params, s->d_bn, S.ws_df, S.d_bytes, B, DGG_TRAIN,
                        srcs_pose(cfg, bps, S.rfake, S.f_transl, S.f_joint), nullptr, nullptr, 0, S.p_fake, st, &share,
                        xpar >= 0 ? &xc : nullptr));
    // :792-793 the fake pass accumulates onto the real pass' gradients (and the loss terms meet: the BCE kernel of the fake
    // pass also forms the total — one launch less on the chain than a combine kernel of its own)
    // (fine-grained form: behind the real pass' first block-backward stages — its loss term is written by then — and its
    //  weight-gradient launches are waited for one by one in rb_backward)
    if (fork && hipStreamWaitEvent(st, share.bwd_fine ? share.bwd_ev[S.d_nrb - 1] : ev[2], 0) != hipSuccess) return DGG_ERR_LAUNCH;
    DGG_TRY(bce_launch(S.p_fake, nullptr, 0.f, B, (float)hp->bce_weight, 0.5f, losses + DGG_L_BCE_FAKE, S.dl_fake, st, losses,
                       real_slot));
    // the real pass (bwd_mode 1) kept dY13 / cs13 of every block in its workspace: the BPS columns' weight gradient is
    // formed here on dY_real + dY_fake
    if (share.bwd_mode == 1) {
        NetW Wr;
        carve(Ld, B, S.ws_dr, Wr);
        for (int r = 0; r < Ld.nrb; ++r) {
            share.dY13_other[r] = Wr.dY13[r];
            share.cs13_other[r] = Wr.cs13[r];
        }
        share.bwd_mode = 2;
    } else {
        share.bwd_mode = 0;
    }
    DGG_TRY(net_backward(Ld, s->d_params, s->d_grads, S.ws_df, S.d_bytes, B, DGG_TRAIN | DGG_ACCUM_GRADS,
                         srcs_pose(cfg, bps, S.rfake, S.f_transl, S.f_joint), S.dl_fake, nullptr, 0, 0, nullptr, st, &share,
                         xpar >= 0 ? S.xs[xpar] : nullptr));
    return DGG_OK;
}

int dgg_cgan_d_phase(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws, size_t ws_bytes,
                     int B, const float* bps, const float* rot9, const float* transl, const float* joint,
                     const float* z, float* losses, void* stream) {
    return dgg_cgan_d_phase2(cfg, s, hp, ws, ws_bytes, B, bps, rot9, transl, joint, z, losses, stream, nullptr);
}

// The tail of a G phase (cGAN and WGAN): :272-307 reconstruction losses + total — d_heads combines the MSE gradients with D's
// input gradient S.d_pose — then :814-815 zero_grad + backward through G.  The loss launch also forms the heads' bias
// gradient and finishes its own sums (GenLossFin; S.ticket was cleared by the phase's loss-gradient launch).
static int gen_loss_and_backward(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, StepW& S, int B,
                                 const float* bps, const float* rot9, const float* transl, const float* joint, const float* z,
                                 float* losses, int wass, hipStream_t st, int xpar = -1) {
    NetL Lg;
    DGG_TRY(build_layout(DGG_NET_GEN, cfg, Lg));
    const bool spread = gen_loss_spread(B);   // gen_loss_launch's own condition for the 64-rows-per-workgroup form (scratch and ticket are given)
    GenLossFin fin;
    fin.ticket = S.ticket;
    fin.gbh = spread ? s->g_grads + Lg.bh : nullptr;
    DGG_TRY(gen_loss_launch(cfg, hp, B, S.heads, rot9, transl, joint, S.p_gen, S.d_pose, losses, S.d_heads, st, S.gl_part,
                            wass, &fin));
    (void)z;
    return net_backward(Lg, s->g_params, s->g_grads, S.ws_g, S.g_bytes, B, DGG_TRAIN | (spread ? kHeadBiasDone : 0u),
                        srcs_gen(cfg, bps), S.d_heads, nullptr, 0, 0, nullptr, st, nullptr,
                        xpar >= 0 ? S.xs[xpar] - Lg.nprefix : nullptr);
}

int dgg_cgan_g_phase(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws, size_t ws_bytes,
                     int B, const float* bps, const float* rot9, const float* transl, const float* joint,
                     const float* z, float* losses, void* stream) {
    if (!s || !hp || !ws || !bps || !rot9 || !transl || !joint || !z || !losses) return DGG_ERR_INVALID;
    StepW S;
    DGG_TRY(carve_step(cfg, B, ws, S));
    if (S.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // :808 D forward #3 with the updated discriminator on the (non-detached) fake grasp; the packed
    // fake transl / joint copies written by the D phase are still valid (same generator output)
    float* g_transl = S.f_transl;
    float* g_joint = S.f_joint;
    // (no weight gradient follows this pass: DGG_SKIP_PARAM_GRADS spares the forward the operands only a wgrad reads)
    // Same inputs as the D phase's fake-batch pass, same workspace: the input normalisation is reused.
    int xpar = -1;
    NetL Ld;
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, Ld));
    if (S.xs_ok) {   // XShare: the D phase of this step left its generation of shared rows in the workspace's slot
        PendingReal* Ps = pending_slot(ws, false);
        if (!Ps || !Ps->step_set) {
            set_error_text("dgg_cgan_g_phase: no D phase has run on this workspace (or its state was evicted: more than %d "
                           "step workspaces in use on this thread)", kPendingSlots);
            return DGG_ERR_INVALID;
        }
        xpar = Ps->step_xpar;
        XShare xc;
        memset(&xc, 0, sizeof(xc));
        xc.mode = 2; xc.xs = S.xs[xpar]; xc.ld = S.xs_ld;
        DGG_TRY(net_forward(Ld, s->d_params, s->d_bn, S.ws_df, S.d_bytes, B,
                            DGG_TRAIN | DGG_SKIP_PARAM_GRADS | DGG_REUSE_INPUT_NORM | kRefreshElsewhere,
                            srcs_pose(cfg, bps, S.rfake, g_transl, g_joint), nullptr, nullptr, 0, S.p_gen, st, nullptr, &xc));
    } else {
        DGG_TRY(dgg_disc_forward(cfg, s->d_params, s->d_bn, S.ws_df, S.d_bytes, B,
                                 DGG_TRAIN | DGG_SKIP_PARAM_GRADS | DGG_REUSE_INPUT_NORM, bps, S.rfake, g_transl, g_joint,
                                 S.p_gen, stream));
    }
    // :293-296 BCE(fake_score, 1) and its gradient; backward through D for the pose inputs only.  (XShare: the third advance
    // of D's running statistics — same batch statistics as the fake pass — rides in this launch: RunUpd.)
    RunUpd ru = {nullptr, nullptr, nullptr, nullptr, 0};
    if (xpar >= 0) {
        NetW Wf;
        carve(Ld, B, S.ws_df, Wf);
        ru.mean = Wf.imean; ru.var_u = Wf.ivaru; ru.run_mean = s->d_bn + Ld.irm; ru.run_var = s->d_bn + Ld.irv; ru.n = Ld.k_in;
    }
    DGG_TRY(bce_launch(S.p_gen, nullptr, 1.f, B, (float)hp->bce_weight, 1.0f, losses + DGG_L_GEN_FAKE, S.dl_gen, st, nullptr,
                       DGG_L_BCE_REAL, S.ticket, nullptr, xpar >= 0 ? &ru : nullptr));
    // D-free mark: from here on nothing in this step writes D's parameters or running statistics
    if (PendingReal* Pp = pending_slot(ws, true)) {
        if (hipEventRecord(Pp->ev[5], st) != hipSuccess) return DGG_ERR_LAUNCH;
        Pp->dfree_set = true;
        Pp->step_B = B;
    } else {
        return DGG_ERR_LAUNCH;
    }
    DGG_TRY(net_backward(Ld, s->d_params, nullptr, S.ws_df, S.d_bytes, B,
                         DGG_TRAIN | DGG_SKIP_PARAM_GRADS | DGG_NEED_INPUT_GRADS, srcs_pose(cfg, bps, S.rfake, g_transl, g_joint),
                         S.dl_gen, nullptr, cfg->in_bps, Ld.k_in, S.d_pose, st, nullptr, xpar >= 0 ? S.xs[xpar] : nullptr));
    return gen_loss_and_backward(cfg, s, hp, S, B, bps, rot9, transl, joint, z, losses, 0, st, xpar);
}

int dgg_cgan_step(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws, size_t ws_bytes, int B,
                  const float* bps, const float* rot9, const float* transl, const float* joint, const float* z,
                  int step_d, int step_g, int is_train_gen, float* losses, void* stream) {
    DGG_TRY(dgg_cgan_d_phase(cfg, s, hp, ws, ws_bytes, B, bps, rot9, transl, joint, z, losses, stream));
    const int64_t nd = dgg_param_floats(DGG_NET_DISC, cfg), ng = dgg_param_floats(DGG_NET_GEN, cfg);
    DGG_TRY(dgg_adam_l2(s->d_params, s->d_grads, s->d_m, s->d_v, nd, step_d, hp->lr_dis, hp->beta1, hp->beta2, hp->eps,
                        hp->weight_decay, hp->grad_scale, stream));
    if (is_train_gen) {
        DGG_TRY(dgg_cgan_g_phase(cfg, s, hp, ws, ws_bytes, B, bps, rot9, transl, joint, z, losses, stream));
        DGG_TRY(dgg_adam_l2(s->g_params, s->g_grads, s->g_m, s->g_v, ng, step_g, hp->lr_gen, hp->beta1, hp->beta2,
                            hp->eps, hp->weight_decay, hp->grad_scale, stream));
    }
    return DGG_OK;
}

int dgg_eva_phase(const dgg_cfg* cfg, const float* e_params, float* e_grads, float* e_bn, const dgg_hparams* hp,
                  void* ws, size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                  const float* label, const float* drop_keep, uint64_t seed, float* losses, void* stream) {
    if (!hp || !ws || !label || !losses) return DGG_ERR_INVALID;
    NetL L;
    DGG_TRY(build_layout(DGG_NET_EVA, cfg, L));
    NetW W;
    carve(L, B, nullptr, W);
    const size_t extra = ((size_t)B * 2 * sizeof(float) + 255) & ~(size_t)255;
    if (W.bytes + extra > ws_bytes) return DGG_ERR_WORKSPACE;
    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + W.bytes);
    float* dl = p + B;
    hipStream_t st = (hipStream_t)stream;
    DGG_TRY(dgg_eva_forward(cfg, e_params, e_bn, ws, W.bytes, B, DGG_TRAIN, bps, rot9, transl, drop_keep, seed, p,
                            stream));
    DGG_TRY(bce_launch(p, label, 0.f, B, (float)hp->bce_weight, 1.0f, losses + DGG_L_TOTAL_EVA, dl, st, nullptr, DGG_L_BCE_REAL,
                       nullptr, losses + DGG_L_BCE_EVA));
    DGG_TRY(dgg_eva_backward(cfg, e_params, e_grads, ws, W.bytes, B, DGG_TRAIN, bps, rot9, transl, dl, nullptr, stream));
    return DGG_OK;
}

// ---- hidden-state management -------------------------------------------------------------------
int dgg_cgan_drop_prefetch(void* ws, void* stream) {
    PendingReal* Pp = ws ? pending_slot(ws, false) : nullptr;
    if (!Pp || !Pp->valid) return DGG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipStreamWaitEvent(st, Pp->ev[1], 0) != hipSuccess) return DGG_ERR_LAUNCH;
    DGG_TRY(copy_cols_launch(Pp->bn_snap, Pp->nbn, Pp->d_bn, Pp->nbn, 1, Pp->nbn, st));
    Pp->valid = false;
    return DGG_OK;
}

int dgg_cgan_release(void* ws) {
    for (int i = 0; i < kPendingSlots; ++i)
        if (g_pending[i].ws == ws && ws) {
            g_pending[i].ws = nullptr;
            g_pending[i].valid = false;
            g_pending[i].dfree_set = false;
            g_pending[i].step_B = 0;
            g_pending[i].stamp = 0;
        }
    return DGG_OK;
}

int dgg_reset(void) {
    memset(g_grad_events, 0, sizeof(g_grad_events));
    gemm_set_side_work(0);
    for (int i = 0; i < kPendingSlots; ++i) {
        g_pending[i].ws = nullptr;
        g_pending[i].valid = false;
        g_pending[i].dfree_set = false;
        g_pending[i].step_B = 0;
        g_pending[i].stamp = 0;
    }
    int rc = DGG_OK;
    for (int i = g_aux.n - 1; i >= 0; --i) {
        if (g_aux.owned[i]) {
            if (hipStreamSynchronize(g_aux.aux[i]) != hipSuccess || hipStreamDestroy(g_aux.aux[i]) != hipSuccess)
                rc = DGG_ERR_LAUNCH;
        }
    }
    g_aux.n = 0;
    return rc;
}

int64_t dgg_eva_keep_offset(const dgg_cfg* cfg, int batch, int r) {
    NetL L;
    if (batch < 1 || build_layout(DGG_NET_EVA, cfg, L) || r < 0 || r >= L.nrb) return DGG_ERR_INVALID;
    // carve against a fake (never dereferenced) base to read the offset back
    NetW W;
    char* fake = reinterpret_cast<char*>((uintptr_t)1 << 40);
    carve(L, batch, fake, W);
    return (int64_t)(reinterpret_cast<char*>(W.rb[r].keep) - fake);
}

int64_t dgg_ws_offset(int net, const dgg_cfg* cfg, int batch, int r, int what) {
    NetL L;
    if (batch < 1 || build_layout(net, cfg, L) || r < 0 || r >= L.nrb || what < 0 || what > 3) return DGG_ERR_INVALID;
    NetW W;
    char* fake = reinterpret_cast<char*>((uintptr_t)1 << 40);
    carve(L, batch, fake, W);
    const RbW& R = W.rb[r];
    const void* p = what == 0 ? (const void*)R.a13 : what == 1 ? (const void*)R.h : what == 2 ? (const void*)R.y : (const void*)R.keep;
    return (int64_t)(reinterpret_cast<const char*>(p) - fake);
}

int64_t dgg_cgan_ws_offset(const dgg_cfg* cfg, int batch, int which) {
    if (batch < 1 || which < 0 || which > 2) return DGG_ERR_INVALID;
    StepW S;
    char* fake = reinterpret_cast<char*>((uintptr_t)1 << 40);
    if (carve_step(cfg, batch, fake, S)) return DGG_ERR_INVALID;
    const void* p = which == 0 ? S.ws_g : which == 1 ? S.ws_dr : S.ws_df;
    return (int64_t)(reinterpret_cast<const char*>(p) - fake);
}

// ---- A2: stand-alone ResBlock (networks.py:43-56, final_nl = True) ------------------------------
int64_t dgg_resblock_param_floats(int fin) {
    NetL L;
    if (build_rb_layout(fin, L)) return DGG_ERR_INVALID;
    return L.nparam;
}
int64_t dgg_resblock_bnstat_floats(int fin) {
    NetL L;
    if (build_rb_layout(fin, L)) return DGG_ERR_INVALID;
    return L.nbn;
}
int dgg_resblock_table(int fin, dgg_entry* out, int cap) {
    NetL L;
    int rc = build_rb_layout(fin, L);
    if (rc) return rc;
    return fill_table(L, nullptr, out, cap);
}
int64_t dgg_resblock_workspace_bytes(int fin, int batch) {
    NetL L;
    if (batch < 1 || build_rb_layout(fin, L)) return DGG_ERR_INVALID;
    NetW W;
    carve(L, batch, nullptr, W);
    return (int64_t)W.bytes;
}

int dgg_resblock_forward(int fin, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                         unsigned flags, const float* x, int ldx, float* y, void* stream) {
    NetL L;
    DGG_TRY(build_rb_layout(fin, L));
    if (!params || !bnstats || !ws || !x || !y || B < 1 || ldx < fin) return DGG_ERR_INVALID;
    Ctx c;
    c.share = nullptr;
    c.L = &L; c.P = params; c.G = nullptr; c.bn = bnstats; c.B = B; c.train = (flags & DGG_TRAIN) != 0;
    c.st = (hipStream_t)stream; c.sw = nullptr;
    if (c.train && B < 2) return DGG_ERR_INVALID;
    carve(L, B, ws, c.W);
    if (c.W.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    Seg seg;
    memset(&seg, 0, sizeof(seg));
    seg.p = x; seg.ld = ldx; seg.width = fin; seg.wcol = 0; seg.is_x0 = false;
    DGG_TRY(rb_forward(c, 0, &seg, 1, 0, nullptr, 0, 0));
    return copy_cols_launch(c.W.rb[0].y, L.nn, y, L.nn, B, L.nn, c.st);
}

int dgg_resblock_backward(int fin, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                          unsigned flags, const float* x, int ldx, const float* dy, float* dx, void* stream) {
    NetL L;
    DGG_TRY(build_rb_layout(fin, L));
    const bool skip = (flags & DGG_SKIP_PARAM_GRADS) != 0;
    if (!params || !ws || !x || !dy || B < 1 || ldx < fin || (!skip && !grads)) return DGG_ERR_INVALID;
    Ctx c;
    c.share = nullptr;
    c.L = &L; c.P = params; c.G = grads; c.bn = nullptr; c.B = B; c.train = (flags & DGG_TRAIN) != 0;
    c.st = (hipStream_t)stream; c.sw = nullptr;
    carve(L, B, ws, c.W);
    if (c.W.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    Seg seg;
    memset(&seg, 0, sizeof(seg));
    seg.p = x; seg.ld = ldx; seg.width = fin; seg.wcol = 0; seg.is_x0 = false;
    if (dx) { seg.dp = dx; seg.dld = fin; seg.d_lo = 0; seg.d_hi = fin; seg.d_accum = 0; }
    int ibn_tiles[3] = {0, 0, 0};
    return rb_backward(c, 0, &seg, 1, dy, false, (flags & DGG_ACCUM_GRADS) != 0, skip, ibn_tiles);
}

// ---- N3: WGAN-GP (dexgangrasp.py:149-211, 234-270) ----------------------------------------------
// second_order.hip explains the method: (1) forward + input-gradient backward on the interpolates -> g, v = gain dGP/dg;
// (2) tangent forward of v; (3) one reverse sweep of F = sum_b sigmoid'(l_b) ldot_b through the primal + tangent graph.
namespace dgg {
struct GpW {
    void* ws_di;
    float *xi_rot, *xi_tr, *xi_jt, *p_i, *dl_i, *g, *v, *xhd0, *tu0, *tu0bar, *c1, *c2, *cs;
    float* gp_part;   // per-workgroup partial sums of the penalty (gp_value_launch -> wgan_combine_launch)
    float *ta13[2], *th[2], *ta2[2], *ty[2];
    float *A13bar, *A13dbar, *a2bar, *a2dbar, *hbar, *hdbar, *pbar, *pdbar, *ybar, *ydbar;
    size_t bytes;
};

static int carve_wgan(const dgg_cfg* cfg, int B, void* base, StepW& S, GpW& W) {
    DGG_TRY(carve_step(cfg, B, base, S));
    NetL Ld;
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, Ld));
    if (Ld.nrb != 2) return DGG_ERR_UNSUPPORTED;
    Bump b{reinterpret_cast<char*>(base), S.bytes};
    const size_t Bz = (size_t)B;
    const int nj = cfg->n_hand_joints, np = 12 + nj;
    W.ws_di = b.take<char>(S.d_bytes);
    W.xi_rot = b.take<float>(Bz * 9);
    W.xi_tr = b.take<float>(Bz * 3);
    W.xi_jt = b.take<float>(Bz * nj);
    W.p_i = b.take<float>(Bz);
    W.dl_i = b.take<float>(Bz);
    W.g = b.take<float>(Bz * np);
    W.v = b.take<float>(Bz * np);
    W.xhd0 = b.take<float>(Bz * np);
    W.tu0 = b.take<float>(Bz * np);
    W.tu0bar = b.take<float>(Bz * np);
    W.c1 = b.take<float>(Bz);
    W.c2 = b.take<float>(Bz);
    W.cs = b.take<float>((size_t)Ld.n13);
    W.gp_part = b.take<float>((size_t)gp_value_parts(B));
    for (int r = 0; r < 2; ++r) {
        W.ta13[r] = b.take<float>(Bz * Ld.n13);
        W.th[r] = b.take<float>(Bz * Ld.hid);
        W.ta2[r] = b.take<float>(Bz * Ld.nn);
        W.ty[r] = b.take<float>(Bz * Ld.nn);
    }
    W.A13bar = b.take<float>(Bz * Ld.n13);
    W.A13dbar = b.take<float>(Bz * Ld.n13);
    W.a2bar = b.take<float>(Bz * Ld.nn);
    W.a2dbar = b.take<float>(Bz * Ld.nn);
    W.hbar = b.take<float>(Bz * Ld.hid);
    W.hdbar = b.take<float>(Bz * Ld.hid);
    W.pbar = b.take<float>(Bz * Ld.nn);
    W.pdbar = b.take<float>(Bz * Ld.nn);
    W.ybar = b.take<float>(Bz * Ld.nn);
    W.ydbar = b.take<float>(Bz * Ld.nn);
    W.bytes = (b.off + 255) & ~(size_t)255;
    return DGG_OK;
}

static int gemm1(int mode, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                 int accumulate, hipStream_t st) {
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    g.nprob = 1;
    prob_init(g.p[0], A, lda, Bm, ldb, C, ldc, M, N, K);
    g.p[0].accumulate = accumulate;
    return gemm_launch(mode, g, st);
}

// The gradient-penalty term: adds gain * dGP/dtheta to d_grads; the penalty's per-workgroup partial sums go to Q.gp_part
// (wgan_combine_launch forms GP in losses[DGG_L_GRAD_PENALTY] from them).
static int gradient_penalty(const dgg_cfg* cfg, const dgg_gan_state* s, const StepW& S, GpW& Q, int B, const float* bps,
                            const float* rot9, const float* transl, const float* joint, const float* alphas, float gain,
                            float* losses, hipStream_t st, Share* share) {
    NetL L;
    DGG_TRY(build_layout(DGG_NET_DISC, cfg, L));
    const int nj = cfg->n_hand_joints, np = 12 + nj, ib = cfg->in_bps;
    const float* P = s->d_params;
    float* G = s->d_grads;
    // (1) interpolates (:149-158), D forward in train mode (its running statistics advance: the step's third D pass),
    //     g = d sum(score) / d(rot, transl, joint)  (:160-172)
    DGG_TRY(gp_interp_launch(alphas, rot9, transl, joint, S.rfake, S.f_transl, S.f_joint, B, nj, Q.xi_rot, Q.xi_tr,
                             Q.xi_jt, st));
    // (share: the interpolates sit on the same BPS batch as the real / fake passes — that part of the fc1 || fc3 products is
    //  read from the real pass' slabs, mode 2, as the fake pass does)
    if (share) {
        share->mode = 2;
        DGG_TRY(net_forward(L, P, s->d_bn, Q.ws_di, S.d_bytes, B, DGG_TRAIN, srcs_pose(cfg, bps, Q.xi_rot, Q.xi_tr, Q.xi_jt),
                            nullptr, nullptr, 0, Q.p_i, st, share));
    } else {
        DGG_TRY(dgg_disc_forward(cfg, P, s->d_bn, Q.ws_di, S.d_bytes, B, DGG_TRAIN, bps, Q.xi_rot, Q.xi_tr, Q.xi_jt, Q.p_i,
                                 st));
    }
    DGG_TRY(sig_grad_launch(Q.p_i, B, Q.dl_i, st));
    DGG_TRY(dgg_disc_backward(cfg, P, nullptr, Q.ws_di, S.d_bytes, B,
                              DGG_TRAIN | DGG_SKIP_PARAM_GRADS | DGG_NEED_INPUT_GRADS, bps, Q.xi_rot, Q.xi_tr, Q.xi_jt,
                              Q.dl_i, Q.g, st));
    DGG_TRY(gp_value_launch(Q.g, B, nj, gain, Q.gp_part, Q.v, st));
    // (2) tangent forward of v next to the stored activations of that forward
    NetW W;
    carve(L, B, Q.ws_di, W);
    aff_direct(L, P, W);
    DGG_TRY(bn_jvp_launch(W.X0 + ib, L.ldx0, Q.v, np, nullptr, W.irstd + ib, P + L.ibnw + ib, nullptr, 0, nullptr, 0, B, np,
                          Q.tu0, np, Q.xhd0, np, 1, st));
    for (int r = 0; r < 2; ++r) {
        const RbL& R = L.rb[r];
        RbW& A = W.rb[r];
        DGG_TRY(gemm1(GEMM_NT, Q.tu0, np, P + R.w13 + ib, R.fin, Q.ta13[r], L.n13, B, L.n13, np, 0, st));
        if (r > 0)
            DGG_TRY(gemm1(GEMM_NT, Q.ty[r - 1], L.nn, P + R.w13 + L.k0, R.fin, Q.ta13[r], L.n13, B, L.n13, L.nn, 1, st));
        DGG_TRY(mask_inplace_launch(Q.ta13[r] + L.hid, L.n13, A.a13 + L.hid, L.n13, B, L.nn, st));
        DGG_TRY(bn_jvp_launch(A.a13, L.n13, Q.ta13[r], L.n13, A.mean1, A.rstd1, P + R.bn1w, nullptr, 0, A.h, L.hid, B, L.hid,
                              Q.th[r], L.hid, nullptr, 0, 0, st));
        DGG_TRY(gemm1(GEMM_NT, Q.th[r], L.hid, P + R.w2, L.hid, Q.ta2[r], L.nn, B, L.nn, L.hid, 0, st));
        DGG_TRY(bn_jvp_launch(A.a2, L.nn, Q.ta2[r], L.nn, A.mean2, A.rstd2, P + R.bn2w, Q.ta13[r] + L.hid, L.n13, A.y, L.nn, B,
                              L.nn, Q.ty[r], L.nn, nullptr, 0, 0, st));
    }
    DGG_TRY(head2_launch(Q.ty[1], L.nn, P + L.wh, Q.p_i, B, L.nn, Q.c1, Q.c2, st));
    // (3) reverse sweep of F: primal adjoints (bar) and tangent adjoints (dbar) side by side; everything accumulates
    DGG_TRY(head2_grads_launch(W.rb[1].y, L.nn, Q.ty[1], L.nn, Q.c1, Q.c2, B, L.nn, G + L.wh, G + L.bh, st));
    int ibn_tiles[2] = {0, 0};
    for (int r = 1; r >= 0; --r) {
        const RbL& R = L.rb[r];
        RbW& A = W.rb[r];
        const bool last = r == 1;
        DGG_TRY(out2_bwd_launch(last ? nullptr : Q.ybar, last ? nullptr : Q.ydbar, L.nn, last ? Q.c1 : nullptr,
                                last ? Q.c2 : nullptr, P + L.wh, A.y, L.nn, A.a13 + L.hid, L.n13, B, L.nn, Q.pbar, Q.pdbar,
                                L.nn, Q.A13bar + L.hid, Q.A13dbar + L.hid, L.n13, st));
        DGG_TRY(bn_second_bwd_launch(A.a2, L.nn, Q.ta2[r], L.nn, A.mean2, A.rstd2, P + R.bn2w, Q.pbar, Q.pdbar, L.nn, nullptr,
                                     0, B, L.nn, Q.a2bar, Q.a2dbar, L.nn, G + R.bn2w, G + R.bn2b, st));
        // fc2: a2 = h W2^T + b2, a2dot = hdot W2^T  (b2 feeds a train-mode BatchNorm: its gradient is exactly zero).
        // Both adjoint streams through W2 in one launch; the two weight-gradient products dW2 += a2bar^T h and
        // dW2 += a2dbar^T hdot ride in the block's primal / tangent weight-gradient launches below (alone they were 32
        // workgroups for 33 us each).
        {
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            g.nprob = 2;
            prob_init(g.p[0], Q.a2bar, L.nn, P + R.w2, L.hid, Q.hbar, L.hid, B, L.hid, L.nn);
            prob_init(g.p[1], Q.a2dbar, L.nn, P + R.w2, L.hid, Q.hdbar, L.hid, B, L.hid, L.nn);
            DGG_TRY(gemm_launch(GEMM_NN, g, st));
        }
        DGG_TRY(bn_second_bwd_launch(A.a13, L.n13, Q.ta13[r], L.n13, A.mean1, A.rstd1, P + R.bn1w, Q.hbar, Q.hdbar, L.hid, A.h,
                                     L.hid, B, L.hid, Q.A13bar, Q.A13dbar, L.n13, G + R.bn1w, G + R.bn1b, st));
        // fc3.bias (the fc1.bias part is exactly zero) and the column sums the BatchNorm-aware wgrad epilogue needs
        DGG_TRY(colsum_prod_launch(Q.A13bar + L.hid, L.n13, nullptr, 0, B, L.nn, Q.cs + L.hid, 0, st));
        DGG_TRY(colsum_prod_launch(Q.A13bar + L.hid, L.n13, nullptr, 0, B, L.nn, G + R.b13 + L.hid, 1, st));
        {  // [fc1;fc3] weights, primal stream: the x_hat segment through the epilogue that also yields d gamma0 / d beta0
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            g.nprob = r > 0 ? 2 : 1;
            prob_init(g.p[0], Q.A13bar, L.n13, W.X0, L.ldx0, G + R.w13, R.fin, L.n13, L.k0, B);
            g.p[0].accumulate = 1;
            g.p[0].ep_gamma = W.aff_scale;
            g.p[0].ep_beta = W.aff_shift;
            g.p[0].ep_cs0 = nullptr;
            g.p[0].ep_cs1 = Q.cs + L.hid;
            g.p[0].ep_split = L.hid;
            g.p[0].ep_w = P + R.w13;
            g.p[0].ep_ldw = R.fin;
            g.p[0].ep_part = W.ibn_part + (size_t)r * 12 * 2 * L.k0;
            if (r > 0) {
                prob_init(g.p[1], Q.A13bar, L.n13, W.rb[r - 1].y, L.nn, G + R.w13 + L.k0, R.fin, L.n13, L.nn, B);
                g.p[1].accumulate = 1;
            }
            {   // rider: dW2 += a2bar^T h
                GemmProb& q = g.p[g.nprob++];
                prob_init(q, Q.a2bar, L.nn, A.h, L.hid, G + R.w2, L.hid, L.nn, L.hid, B);
                q.accumulate = 1;
            }
            int bm, bn;
            gemm_tile_shape(GEMM_TN, g.p[0].M, g.p[0].N, bm, bn, g.p[0].K);
            ibn_tiles[r] = (L.n13 + bm - 1) / bm;
            if (ibn_tiles[r] > 12) return DGG_ERR_INVALID;
            DGG_TRY(gemm_launch(GEMM_TN, g, st));
        }
        // tangent stream: only the pose columns of x_hat (and the previous block) carry a tangent.  One launch, after the
        // primal one (both accumulate into the same entries): dW13[:, pose] += A13dbar^T u0dot, dW13[:, previous block] +=
        // A13dbar^T ydot, dW2 += a2dbar^T hdot
        {
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            int nq = 0;
            prob_init(g.p[nq], Q.A13dbar, L.n13, Q.tu0, np, G + R.w13 + ib, R.fin, L.n13, np, B);
            g.p[nq++].accumulate = 1;
            if (r > 0) {
                prob_init(g.p[nq], Q.A13dbar, L.n13, Q.ty[r - 1], L.nn, G + R.w13 + L.k0, R.fin, L.n13, L.nn, B);
                g.p[nq++].accumulate = 1;
            }
            prob_init(g.p[nq], Q.a2dbar, L.nn, Q.th[r], L.hid, G + R.w2, L.hid, L.nn, L.hid, B);
            g.p[nq++].accumulate = 1;
            g.nprob = nq;
            DGG_TRY(gemm_launch(GEMM_TN, g, st));
        }
        if (r > 0) {   // both adjoint streams into the previous block's output, one launch
            GemmBatch g;
            memset(&g, 0, sizeof(g));
            g.nprob = 2;
            prob_init(g.p[0], Q.A13bar, L.n13, P + R.w13 + L.k0, R.fin, Q.ybar, L.nn, B, L.nn, L.n13);
            prob_init(g.p[1], Q.A13dbar, L.n13, P + R.w13 + L.k0, R.fin, Q.ydbar, L.nn, B, L.nn, L.n13);
            DGG_TRY(gemm_launch(GEMM_NN, g, st));
        }
        // (a narrow product: 16 workgroups walking 24 k-tiles on gemm_f32_kernel, 27.6 us — one 32-row tile per workgroup with
        //  the waves splitting K instead, fc2_stage.hip)
        if (np <= 32 && L.n13 % 64 == 0 && L.n13 <= 1024)
            DGG_TRY(narrow_nn_launch(Q.A13dbar, L.n13, P + R.w13 + ib, R.fin, Q.tu0bar, np, B, np, L.n13, last ? 0 : 1, st));
        else
            DGG_TRY(gemm1(GEMM_NN, Q.A13dbar, L.n13, P + R.w13 + ib, R.fin, Q.tu0bar, np, B, np, L.n13, last ? 0 : 1, st));
    }
    // input BatchNorm: d gamma0 / d beta0 from the primal epilogue partials, d gamma0 (pose columns) from the tangent
    for (int r = 0; r < 2; ++r)
        DGG_TRY(colsum_finalize_launch(W.ibn_part + (size_t)r * 12 * 2 * L.k0, 1, ibn_tiles[r], 2, L.k_in, nullptr, G + L.ibnw,
                                       G + L.ibnb, nullptr, 1, st, L.k0, L.nprefix));
    DGG_TRY(colsum_prod_launch(Q.tu0bar, np, Q.xhd0, np, B, np, G + L.ibnw + ib, 1, st));
    return DGG_OK;
}
}  // namespace dgg

int64_t dgg_wgan_workspace_bytes(const dgg_cfg* cfg, int batch) {
    if (batch < 2) return DGG_ERR_INVALID;
    StepW S;
    GpW Q;
    if (carve_wgan(cfg, batch, nullptr, S, Q)) return DGG_ERR_INVALID;
    return (int64_t)Q.bytes;
}

int dgg_wgan_d_phase(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws, size_t ws_bytes, int B,
                     const float* bps, const float* rot9, const float* transl, const float* joint, const float* z,
                     const float* alphas, double penalty_gain, float* losses, void* stream) {
    if (!s || !hp || !ws || !bps || !rot9 || !transl || !joint || !z || !alphas || !losses || B < 2) return DGG_ERR_INVALID;
    StepW S;
    GpW Q;
    DGG_TRY(carve_wgan(cfg, B, ws, S, Q));
    if (Q.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    const int nj = cfg->n_hand_joints, nh = nj + 9;
    hipStream_t st = (hipStream_t)stream;
    // :772-776 generator forward (train, activations kept for the G phase), R = ortho6d(rot_6D)
    DGG_TRY(dgg_gen_forward(cfg, s->g_params, s->g_bn, S.ws_g, S.g_bytes, B, DGG_TRAIN, z, bps, S.heads, stream));
    DGG_TRY(fake_pose_launch(S.heads, nh, nj, B, S.rfake, S.f_transl, S.f_joint, st));
    // :788-789 D(real), D(fake.detach()); :199-205 -mean(real) + mean(fake).  The two passes share what the cGAN D phase
    // shares (dgg_cgan_d_phase2, one stream): the BPS columns are the same batch in both, so their part of every fc1 || fc3
    // product is formed once, by the real pass, and their weight gradient once, by the fake pass, on dY_real + dY_fake.
    // (The order of the forwards — real, then fake — is the reference's: D's running statistics see the same sequence.)
    Share share;
    {
        NetL Ld;
        DGG_TRY(build_layout(DGG_NET_DISC, cfg, Ld));
        memset(&share, 0, sizeof(share));
        share.ncols = cfg->in_bps;
        share.events = false;
        for (int r = 0; r < S.d_nrb; ++r) share.slabs[r] = S.dshare[r];
        share.mode = 1;
        DGG_TRY(net_forward(Ld, s->d_params, s->d_bn, S.ws_dr, S.d_bytes, B, DGG_TRAIN, srcs_pose(cfg, bps, rot9, transl, joint),
                            nullptr, nullptr, 0, S.p_real, st, &share));
        DGG_TRY(wass_launch(S.p_real, B, -1.f, losses + DGG_L_BCE_REAL, S.dl_real, st));
        share.bwd_mode = share_wgrad() ? 1 : 0;
        DGG_TRY(net_backward(Ld, s->d_params, s->d_grads, S.ws_dr, S.d_bytes, B, DGG_TRAIN,
                             srcs_pose(cfg, bps, rot9, transl, joint), S.dl_real, nullptr, 0, 0, nullptr, st, &share));
        share.mode = 2;
        DGG_TRY(net_forward(Ld, s->d_params, s->d_bn, S.ws_df, S.d_bytes, B, DGG_TRAIN,
                            srcs_pose(cfg, bps, S.rfake, S.f_transl, S.f_joint), nullptr, nullptr, 0, S.p_fake, st, &share));
        DGG_TRY(wass_launch(S.p_fake, B, 1.f, losses + DGG_L_BCE_FAKE, S.dl_fake, st));
        if (share.bwd_mode == 1) {
            NetW Wr;
            carve(Ld, B, S.ws_dr, Wr);
            for (int r = 0; r < Ld.nrb; ++r) {
                share.dY13_other[r] = Wr.dY13[r];
                share.cs13_other[r] = Wr.cs13[r];
            }
            share.bwd_mode = 2;
        } else {
            share.bwd_mode = 0;
        }
        DGG_TRY(net_backward(Ld, s->d_params, s->d_grads, S.ws_df, S.d_bytes, B, DGG_TRAIN | DGG_ACCUM_GRADS,
                             srcs_pose(cfg, bps, S.rfake, S.f_transl, S.f_joint), S.dl_fake, nullptr, 0, 0, nullptr, st, &share));
    }
    // :174-197 gradient penalty and its parameter gradient
    DGG_TRY(gradient_penalty(cfg, s, S, Q, B, bps, rot9, transl, joint, alphas, (float)penalty_gain, losses, st, &share));
    return wgan_combine_launch(losses, (float)penalty_gain, DGG_L_GRAD_PENALTY, Q.gp_part, B, st);
}

int dgg_wgan_g_phase(const dgg_cfg* cfg, const dgg_gan_state* s, const dgg_hparams* hp, void* ws, size_t ws_bytes, int B,
                     const float* bps, const float* rot9, const float* transl, const float* joint, const float* z,
                     float* losses, void* stream) {
    if (!s || !hp || !ws || !bps || !rot9 || !transl || !joint || !z || !losses || B < 2) return DGG_ERR_INVALID;
    StepW S;
    GpW Q;
    DGG_TRY(carve_wgan(cfg, B, ws, S, Q));
    if (Q.bytes > ws_bytes) return DGG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // :808 D forward with the updated discriminator; :258 gen_loss_fake = -mean(fake_score)
    DGG_TRY(dgg_disc_forward(cfg, s->d_params, s->d_bn, S.ws_df, S.d_bytes, B, DGG_TRAIN, bps, S.rfake, S.f_transl, S.f_joint,
                             S.p_gen, stream));
    DGG_TRY(wass_launch(S.p_gen, B, -1.f, nullptr, S.dl_gen, st, S.ticket));
    DGG_TRY(dgg_disc_backward(cfg, s->d_params, nullptr, S.ws_df, S.d_bytes, B,
                              DGG_TRAIN | DGG_SKIP_PARAM_GRADS | DGG_NEED_INPUT_GRADS, bps, S.rfake, S.f_transl, S.f_joint,
                              S.dl_gen, S.d_pose, stream));
    return gen_loss_and_backward(cfg, s, hp, S, B, bps, rot9, transl, joint, z, losses, 1, st);
}

// ---- HIP-graph-friendly forms: everything that changes from step to step lives in device memory ---------------
int dgg_adam_l2_graph(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step_dev, float* dyn2,
                      double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                      void* stream) {
    if (!params || !grads || !m || !v || !step_dev || !dyn2) return DGG_ERR_INVALID;
    return adam_graph_launch(params, grads, m, v, n, step_dev, dyn2, lr, beta1, beta2, eps, weight_decay, grad_scale,
                             (hipStream_t)stream);
}

int dgg_eva_phase_graph(const dgg_cfg* cfg, const float* e_params, float* e_grads, float* e_bn, const dgg_hparams* hp,
                        void* ws, size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                        const float* label, uint64_t* seed_dev, float* losses, void* stream) {
    if (!hp || !ws || !label || !losses || !seed_dev) return DGG_ERR_INVALID;
    NetL L;
    DGG_TRY(build_layout(DGG_NET_EVA, cfg, L));
    NetW W;
    carve(L, B, nullptr, W);
    const size_t extra = ((size_t)B * 2 * sizeof(float) + 255) & ~(size_t)255;
    if (W.bytes + extra > ws_bytes) return DGG_ERR_WORKSPACE;
    float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + W.bytes);
    float* dl = p + B;
    hipStream_t st = (hipStream_t)stream;
    DGG_TRY(dgg_eva_forward(cfg, e_params, e_bn, ws, W.bytes, B, DGG_TRAIN | DGG_SEED_ON_DEVICE, bps, rot9, transl, nullptr,
                            (uint64_t)(uintptr_t)seed_dev, p, stream));
    DGG_TRY(bce_launch(p, label, 0.f, B, (float)hp->bce_weight, 1.0f, losses + DGG_L_TOTAL_EVA, dl, st, nullptr, DGG_L_BCE_REAL,
                       nullptr, losses + DGG_L_BCE_EVA));
    DGG_TRY(dgg_eva_backward(cfg, e_params, e_grads, ws, W.bytes, B, DGG_TRAIN, bps, rot9, transl, dl, nullptr, stream));
    return seed_bump_launch(seed_dev, st);  // the next replay draws other masks
}

int dgg_gemm_occupancy(int mode, int bm, int bn) {
    return gemm_algo() == 1 && bm == 128 && bn == 128 ? 1 : gemm_occupancy(mode, bm, bn);   // (gemm_x5_kernel: 130 KB of LDS, one per CU)
}

int dgg_set_gemm_algo(int algo) {
    if (algo != DGG_GEMM_F32 && algo != DGG_GEMM_BF16X3 && algo != DGG_GEMM_BF16) return DGG_ERR_INVALID;
    gemm_set_algo(algo);
    return DGG_OK;
}
int dgg_get_gemm_algo(void) { return gemm_algo(); }

int dgg_gemm_f32(int mode, int M, int N, int K, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc,
                 int splits, float* slabs, int accumulate, void* stream) {
    if (!A || !Bm || (!C && splits < 1) || M < 1 || N < 1 || K < 1 || mode < 0 || mode > 2) return DGG_ERR_INVALID;
    GemmBatch g;
    memset(&g, 0, sizeof(g));
    g.nprob = 1;
    hipStream_t st = (hipStream_t)stream;
    if (splits >= 1) {
        if (!slabs || accumulate) return DGG_ERR_INVALID;
        prob_init(g.p[0], A, lda, Bm, ldb, slabs, N, M, N, K);
        prob_split(g.p[0], splits, (long long)M * N);
        DGG_TRY(gemm_launch(mode, g, st));
        if (!C) return DGG_OK;  // slabs only (bench: times the MFMA kernel alone)
        return reduce_bias_launch(slabs, g.p[0].splits, (long long)M * N, N, nullptr, C, ldc, M, M, 1, N, 0, 0, 0, 0,
                                  nullptr, st);
    }
    prob_init(g.p[0], A, lda, Bm, ldb, C, ldc, M, N, K);
    g.p[0].accumulate = accumulate;
    return gemm_launch(mode, g, st);
}

int dgg_gemm_route(int mode, int M, int N, int K, int lda, int ldb, int ldc, int splits) {
    if (M < 1 || N < 1 || K < 1 || mode < 0 || mode > 2 || splits < 0) return DGG_ERR_INVALID;
    return gemm_route_single(mode, M, N, K, lda, ldb, ldc, splits);
}

int dgg_set_side_work(int on) {
    gemm_set_side_work(on);
    return DGG_OK;
}

int dgg_set_grad_events(const float* grads, void* const* events, int n, int accum_only) {
    if (!grads) {  // clear the calling thread's registrations
        memset(g_grad_events, 0, sizeof(g_grad_events));
        return DGG_OK;
    }
    int slot = -1;
    for (int i = 0; i < kGradEventSlots; ++i)
        if (g_grad_events[i].grads == grads) slot = i;
    if (!events || n <= 0) {  // unregister
        if (slot >= 0) memset(&g_grad_events[slot], 0, sizeof(GradEvents));
        return DGG_OK;
    }
    if (n > 3) return DGG_ERR_INVALID;
    for (int i = 0; i < kGradEventSlots && slot < 0; ++i)
        if (!g_grad_events[i].grads) slot = i;
    if (slot < 0) return DGG_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!events[i]) return DGG_ERR_INVALID;
        g_grad_events[slot].ev[i] = (hipEvent_t)events[i];
    }
    g_grad_events[slot].grads = grads;
    g_grad_events[slot].n = n;
    g_grad_events[slot].accum_only = accum_only;
    return DGG_OK;
}


}  // extern "C"
