/* dgg.h — C-ABI of libdgg.so: the MI355X (gfx950) implementation of the DexGANGrasp hot path.
 *
 * The reference (david-s-martinez/Dex-GAN-Grasp) is pure Python on PyTorch-CUDA and has no FFI
 * layer of its own: its "operator API" for this path is the Python classes
 *   DexGanGrasp/models/networks.py:26-338   (ResBlock, Generator, Discriminator, DexGANGrasp, DexEvaluator)
 *   DexGanGrasp/models/dexgangrasp.py:65-823 (DexGanGrasp wrapper: losses, update_*, generate/filter)
 *   DexGanGrasp/data/bps_encoder.py:12-55    (BPSEncoder -> bps_torch.encode)
 * Each entry point below names the reference interface it replaces; the Python mirror in
 * dex_gan_grasp_amd/ binds them with ctypes (INTEGRATION.md shows the stub a maintainer adds).
 *
 * Conventions:
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked `host`;
 *  - the library never allocates or frees caller memory: parameters, gradients, Adam moments, BN
 *    running statistics and workspaces are caller-owned buffers whose sizes the *_floats / *_bytes
 *    queries return; layouts of the flat buffers are described by dgg_table();
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the
 *    null stream); nothing synchronises with the host;
 *  - return value: DGG_OK (0) or a negative dgg_status; no exceptions cross the boundary;
 *  - re-entrant across streams with distinct workspaces; not thread-safe on one workspace.
 *  - all tensors fp32 row-major contiguous unless a leading dimension is given;
 *  - batch limit of every forward / backward / phase / step entry: **B <= 260 111 rows per call**.  Two 32-bit byte offsets
 *    set it: the GEMM epilogues address their outputs (rows * pitch * 4 < 2 GiB: B <= 699 050 at the widest output pitch,
 *    768 floats) and the bf16x3 kernels address a k-contiguous OPERAND's rows (rows * pitch * 4 < 4 GiB: the normalised
 *    input block X0 has a pitch of 4128 floats -> B <= 260 111; the per-object inference entries stage O rows of it and
 *    O * n rows of 512 / 768 floats: O * n <= 699 050).  A larger batch returns DGG_ERR_INVALID and dgg_last_error()
 *    says which product and how many rows fit; callers chunk their rows (the reference's own batches are 32 .. 5012).
 *
 * Hidden state (everything else is stateless).  The library keeps, PER CALLING THREAD:
 *  (a) a pool of 64 timing-disabled HIP events used for its internal fork/join between streams;
 *  (b) the stream -> weight-gradient side stream map of dgg_set_wgrad_stream / dgg_enable_wgrad_stream
 *      (the latter creates and owns the side streams, reference-counted);
 *  (c) up to 4 "pending real pass" slots of dgg_cgan_d_real_prefetch, keyed by the raw step-workspace pointer
 *      (6 events each) and matched against (B, input pointers, d_params) by the next dgg_cgan_d_phase2.  A prefetched
 *      real pass also leaves its dY13 / column sums in the workspace for the fake pass of that dgg_cgan_d_phase2, which
 *      forms the weight gradient of the shared BPS columns on dY_real + dY_fake: until that call has run, the BPS
 *      columns of d_grads' W13 / input-BatchNorm entries are NOT the real pass' gradient (nothing is written there);
 *  (d) up to 24 registrations of gradient-ready events (dgg_set_grad_events), keyed by the gradient buffer.
 * Consequences: use one thread per engine (or call from one thread only); a workspace that is freed while a slot
 * refers to it must be released first with dgg_cgan_release(ws) — a new allocation at the same address with the same
 * input pointers would otherwise match a stale slot; dgg_reset() drops (b), (c) and (d) for the calling thread
 * (synchronising and destroying the library-owned side streams).  (a) lives until the thread exits.
 *
 * Arithmetic: the large GEMMs use the "bf16x3" scheme by default (see DGG_GEMM_BF16X3 below): fp32-accurate on
 * finite operands; an operand element that is +-Inf yields NaN in every output it touches (x - bf16(x) = Inf - Inf),
 * where an fp32 fma chain would give +-Inf (or NaN, depending on the other operand).  NaN operands give NaN either way.
 */
#ifndef DGG_H
#define DGG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DGG_OK = 0,
    DGG_ERR_INVALID = -1,   /* bad argument (null pointer, non-positive size, unsupported config) */
    DGG_ERR_LAUNCH = -2,    /* hipLaunch / runtime error; dgg_last_error() has the text */
    DGG_ERR_WORKSPACE = -3, /* workspace too small for the requested batch */
    DGG_ERR_UNSUPPORTED = -4
} dgg_status;

typedef enum { DGG_NET_GEN = 0, DGG_NET_DISC = 1, DGG_NET_EVA = 2 } dgg_net;

/* Network hyper-parameters (config_dexgangrasp.yaml:36,42; networks.py:27,61-62). */
typedef struct {
    int32_t latent_d;      /* cfg["latentD"] = 5 */
    int32_t n_hand_joints; /* cfg["n_hand_joints"] = 12 */
    int32_t in_bps;        /* 4096 */
    int32_t n_neurons;     /* 512 (ResBlock Fout) */
    int32_t hidden;        /* 256 (ResBlock n_neurons) */
} dgg_cfg;

/* One entry of a flat-buffer layout table.  kind: 0 = trainable parameter (offset into the params /
 * grads / adam_m / adam_v buffers), 1 = BatchNorm running statistic (offset into the bnstats
 * buffer).  `name` is the reference's state_dict key relative to the module
 * (e.g. "gen_rb1.fc1.weight", networks.py:71-77). */
typedef struct {
    char name[64];
    int64_t offset; /* in floats */
    int32_t rows;
    int32_t cols;   /* 1-D tensors: rows = n, cols = 1 */
    int32_t kind;
    int32_t pad_;
} dgg_entry;

const char* dgg_version(void);
const char* dgg_last_error(void);

/* Layout queries ------------------------------------------------------------------------------ */
/* Fills up to `cap` entries; returns the number of entries of the table (or <0). */
int dgg_table(int net, const dgg_cfg* cfg, dgg_entry* out /*host*/, int cap);
int64_t dgg_param_floats(int net, const dgg_cfg* cfg);   /* size of params/grads/m/v buffers */
int64_t dgg_bnstat_floats(int net, const dgg_cfg* cfg);  /* size of the running-stat buffer */
int64_t dgg_workspace_bytes(int net, const dgg_cfg* cfg, int batch); /* activations + scratch */

/* A1 — replaces bps_torch.encode as called at data/bps_encoder.py:33 and
 * dexgangrasp_real_robot.py:139.  clouds (B,N,3), basis (P,3) -> dists (B,P), ids (B,P) (ids may
 * be NULL).  id = argmin_j |b - x_j|^2 with strict '<' (lowest j wins ties); dist = sqrt of it.
 * Every shape gives the same bits (d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx))); what differs is the kernel
 * (csrc/bps_encode.hip):
 *   - clouds of 64 <= N <= 2560 points: the matrix-core path — a two-plane fp16 coarse pass on
 *     v_mfma_f32_32x32x16_f16 SELECTS candidate blocks of 16 points, the contract's arithmetic on the candidates
 *     DECIDES (16-byte aligned clouds with N % 4 == 0 load four points as three 16-byte vectors, others point by point);
 *   - dgg_bps_encode_ws with a workspace of dgg_bps_workspace_bytes(B, N, P): few clouds of many points (the
 *     reference's own call shape: ONE cloud of 15k-73k points, dexgangrasp_real_robot.py:133-141) and clouds of more
 *     than 2560 points take the same path chunk by chunk — a workgroup per (chunk of 256..1536 points, 1024 basis
 *     points), each with its own power-of-two scale, (d2 bits, index) per chunk and basis point into the workspace,
 *     a second launch takes the 64-bit minimum over the chunks = the lowest-index rule over the whole cloud;
 *   - without a workspace such clouds are scanned by the vector ALU as ranges that meet in the output arrays
 *     (atomicMin, two passes); fewer than 64 points: the plain scan.
 * A NaN point never wins; a basis point that nothing is closer to than +inf gets id 0, dist +inf.
 * ws == NULL: exactly dgg_bps_encode.  ws must be 8-byte aligned; a ws smaller than the query falls back to the
 * workspace-free kernels (never an error: the answer does not depend on the route). */
int dgg_bps_encode(const float* clouds, int B, int N, const float* basis, int P, float* dists,
                   int32_t* ids, void* stream);
int64_t dgg_bps_workspace_bytes(int B, int N, int P);   /* 0: this shape needs none */
int dgg_bps_encode_ws(const float* clouds, int B, int N, const float* basis, int P, float* dists,
                      int32_t* ids, void* ws, size_t ws_bytes, void* stream);
/* Process-wide route override (A/B timing and the regression tests that compare the matrix-core path with the scan
 * output by output): DGG_BPS_SCAN = the vector-ALU scan for every shape.  DGG_BPS=valu in the environment selects it
 * at start-up. */
enum { DGG_BPS_AUTO = 0, DGG_BPS_SCAN = 1 };
int dgg_set_bps_algo(int algo);
int dgg_get_bps_algo(void);

/* Flags */
#define DGG_TRAIN 1u            /* BatchNorm batch statistics + running-stat update, dropout on */
#define DGG_ACCUM_GRADS 2u      /* backward: add into grads instead of overwriting */
#define DGG_SKIP_PARAM_GRADS 4u /* backward: input gradients only (G-phase pass through D).  Optional on the FORWARD
                                 * of such a pass: it then skips preparing the operands only a weight-gradient pass
                                 * reads — a backward WITHOUT this flag must not follow a forward WITH it */
#define DGG_NEED_INPUT_GRADS 8u /* backward: also produce gradients w.r.t. the network inputs */
#define DGG_REUSE_INPUT_NORM 32u /* forward (train): the workspace still holds the normalised input of the previous TRAIN
                                  * forward over the very same inputs (dgg_cgan_g_phase after dgg_cgan_d_phase2): only
                                  * the running statistics advance and the affine follows the current gamma / beta */
#define DGG_SEED_ON_DEVICE 16u  /* dgg_eva_forward: `seed` holds a device pointer (uint64_t*) to the dropout seed */

/* A2 — ResBlock.forward(x, final_nl=True) on its own, networks.py:43-56 (Fin != Fout: fc3 present, as at every
 * call site of the reference).  One flat parameter buffer [fc1.weight ; fc3.weight | fc1.bias ; fc3.bias |
 * bn1.weight | bn1.bias | fc2.weight | fc2.bias | bn2.weight | bn2.bias] and one running-statistics buffer
 * [bn1.running_mean | bn1.running_var | bn2.running_mean | bn2.running_var], offsets from dgg_resblock_table
 * (names as in the module's state_dict: "fc1.weight", ...).  x (B, fin) with row stride ldx -> y (B, 512).
 * Backward (autograd of :43-56): dy (B,512) -> grads (layout of params; DGG_ACCUM_GRADS / DGG_SKIP_PARAM_GRADS
 * honoured) and dx (B, fin) if non-NULL; must follow a forward on the same workspace. */
int64_t dgg_resblock_param_floats(int fin);
int64_t dgg_resblock_bnstat_floats(int fin);
int dgg_resblock_table(int fin, dgg_entry* out /*host*/, int cap);
int64_t dgg_resblock_workspace_bytes(int fin, int batch);
int dgg_resblock_forward(int fin, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                         unsigned flags, const float* x, int ldx, float* y, void* stream);
int dgg_resblock_backward(int fin, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                          unsigned flags, const float* x, int ldx, const float* dy, float* dx, void* stream);

/* A3 — Generator.forward(Zin, bps_object), networks.py:88-103.
 * out_heads (B,21) = [joint_conf(12) | rot_6D(6) | transl(3)] (column blocks of one buffer). */
int dgg_gen_forward(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes,
                    int B, unsigned flags, const float* z, const float* bps, float* out_heads, void* stream);
/* Backward of the above (autograd of networks.py:88-103).  d_heads (B,21).  d_z (B,latent) and
 * d_bps (B,4096) may be NULL. */
int dgg_gen_backward(const dgg_cfg* cfg, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                     unsigned flags, const float* z, const float* bps, const float* d_heads, float* d_z,
                     float* d_bps, void* stream);

/* A4 — Discriminator.forward(data), networks.py:132-159: inputs bps (B,4096), rot (B,9) row-major
 * of rot_matrix.view(B,-1), transl (B,3), joint_conf (B,nj) -> p (B,1) after sigmoid. */
int dgg_disc_forward(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                     unsigned flags, const float* bps, const float* rot9, const float* transl,
                     const float* joint, float* out_p, void* stream);
/* d_logit (B,1) is the gradient w.r.t. the pre-sigmoid output.  d_pose (B,12+nj) =
 * [d rot9 | d transl | d joint] is written when DGG_NEED_INPUT_GRADS (else may be NULL). */
int dgg_disc_backward(const dgg_cfg* cfg, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                      unsigned flags, const float* bps, const float* rot9, const float* transl,
                      const float* joint, const float* d_logit, float* d_pose, void* stream);

/* A5 — DexEvaluator.forward(data), networks.py:315-338.  Dropout (p = 0.3, three sites): in train
 * mode either `drop_keep` (3,B,512) floats in {0,1} is given (parity tests) or, if NULL, masks are
 * drawn in-kernel from Philox4x32-10 keyed by (seed, element index). */
int dgg_eva_forward(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes, int B,
                    unsigned flags, const float* bps, const float* rot9, const float* transl,
                    const float* drop_keep, uint64_t seed, float* out_p, void* stream);
int dgg_eva_backward(const dgg_cfg* cfg, const float* params, float* grads, void* ws, size_t ws_bytes, int B,
                     unsigned flags, const float* bps, const float* rot9, const float* transl,
                     const float* d_logit, float* d_pose /* (B,12) or NULL */, void* stream);
/* Byte offset, inside an evaluator workspace for `batch` rows, of the dropout keep mask (batch x 512 bytes in
 * {0,1}) that the last train-mode forward wrote for dropout site r (0..2) and that the backward reads.  Lets a
 * caller (tests; a checkpoint of an in-flight step) read back the masks drawn in-kernel.  <0 on bad arguments. */
int64_t dgg_eva_keep_offset(const dgg_cfg* cfg, int batch, int r);
/* Byte offsets of the activations a train-mode forward of network `net` leaves in its workspace for ResBlock r and
 * that its backward reads (networks.py:43-56): what = 0: a13 = [a1 | lrelu(fc3(x))] (batch x 768 floats), 1: h =
 * lrelu(bn1(fc1(x))) (batch x 256), 2: y = the block output after the final LeakyReLU [and dropout] (batch x 512),
 * 3: the dropout keep bytes (= dgg_eva_keep_offset).  The SIGN of an element of (0 / columns 256.., 1, 2) is the side
 * of the LeakyReLU kink the backward takes for it: parity tests read it back so that a CPU evaluation takes the same
 * side where a pre-activation lies within rounding distance of zero (tests/helpers.py).  <0 on bad arguments. */
int64_t dgg_ws_offset(int net, const dgg_cfg* cfg, int batch, int r, int what);
/* Byte offset, inside a dgg_cgan_workspace_bytes(cfg, batch) workspace, of the network workspaces the cGAN phases use:
 * which = 0: generator (forward of the D phase, read by the G phase's backward), 1: discriminator, real-batch pass,
 * 2: discriminator, fake-batch pass of the D phase — overwritten by the G phase's pass through the updated
 * discriminator (dexgangrasp.py:808).  <0 on bad arguments. */
int64_t dgg_cgan_ws_offset(const dgg_cfg* cfg, int batch, int which);

/* Config 5 (dexgangrasp_offline.py:147-162 over many objects) — eval-mode forward of O objects x n samples each,
 * factorised per object: DexGanGrasp.generate_grasps / filter_grasps tile ONE BPS row over an object's n samples
 * (dexgangrasp.py:495-514, :406-411), so the BPS part of every fused fc1||fc3 product is computed once per object
 * (M = O rows instead of O*n) and each sample adds only its own columns.  Same results as dgg_gen_forward /
 * dgg_eva_forward (flags = 0) on the tiled rows, up to the summation order.  bps_obj (O, 4096); per-sample inputs
 * have O*n rows, sample s of object o at row o*n + s: z (O*n, latent); rot9 (O*n, 9), transl (O*n, 3).
 * Outputs: out_heads (O*n, nj+9) as dgg_gen_forward; out_p (O*n, 1).  Workspace: dgg_objects_workspace_bytes. */
int64_t dgg_objects_workspace_bytes(int net, const dgg_cfg* cfg, int n_obj, int n_per_obj);
int dgg_gen_forward_objects(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes,
                            int n_obj, int n_per_obj, const float* z, const float* bps_obj, float* out_heads,
                            void* stream);
int dgg_eva_forward_objects(const dgg_cfg* cfg, const float* params, float* bnstats, void* ws, size_t ws_bytes,
                            int n_obj, int n_per_obj, const float* bps_obj, const float* rot9, const float* transl,
                            float* out_p, void* stream);

/* A6 — utils.rot_matrix_from_ortho6d, utils/utils.py:380-394 (+ normalize_vector :323-332,
 * cross_product :68-78).  o6 (B,6) with row stride ld_o6 -> R (B,9) row-major, columns x,y,z. */
int dgg_ortho6d_forward(const float* o6, int ld_o6, int B, float* rot9, void* stream);
int dgg_ortho6d_backward(const float* o6, int ld_o6, int B, const float* d_rot9, float* d_o6, int ld_do6,
                         void* stream);

/* A7/A9 — torch.nn.BCELoss(reduction='mean') as used at dexgangrasp.py:142-147, 213-232, 293-296.
 * p (B,1); target: `labels` (B) if non-NULL else the constant `target_const`.
 * loss_out[0] = weight * mean BCE (log clamped at -100).  If d_logit != NULL it receives
 * d(loss_scale * loss)/d(logit) through the sigmoid (BCELoss backward divides by
 * max((1-p)p, 1e-12), as torch does). */
int dgg_bce(const float* p, const float* labels, float target_const, int B, float weight, float loss_scale,
            float* loss_out, float* d_logit, void* stream);

/* A12 — torch.optim.Adam as configured at dexgangrasp.py:97-112 (L2-coupled weight decay):
 * g += wd*theta; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * theta -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  `step` is t (1-based).  Hyper-parameters
 * are doubles, as in torch: 1-beta and the bias corrections are formed in double on the host and only
 * then rounded to the fp32 the kernel computes in (1.f - 0.999f would be off by 1.3e-5). */
int dgg_adam_l2(float* params, const float* grads, float* m, float* v, int64_t n, int step, double lr,
                double beta1, double beta2, double eps, double weight_decay, double grad_scale, void* stream);

/* A10 — DexGanGrasp.update_dexgangrasp, dexgangrasp.py:757-823, split at the two points where a
 * data-parallel run all-reduces gradients (between backward and Adam).
 *
 * Hyper-parameters of one step. */
typedef struct {
    double lr_gen, lr_dis, lr_eva; /* cfg lr_gen / lr_dis / lr */
    double beta1, beta2, eps, weight_decay;
    double bce_weight;
    double transl_coef, rot_coef, conf_coef; /* 100, 1, 10 (dexgangrasp.py:92-94) */
    double grad_scale; /* 1/world_size after a sum all-reduce, else 1 */
} dgg_hparams;

/* State of the GAN pair (device pointers, layouts from dgg_table). */
typedef struct {
    float *g_params, *g_grads, *g_m, *g_v, *g_bn;
    float *d_params, *d_grads, *d_m, *d_v, *d_bn;
} dgg_gan_state;

int64_t dgg_cgan_workspace_bytes(const dgg_cfg* cfg, int batch);

/* Losses written by the phases: indices into a device float[16]. */
enum {
    DGG_L_TOTAL_DISC = 0, DGG_L_BCE_REAL = 1, DGG_L_BCE_FAKE = 2, DGG_L_GEN_FAKE = 3,
    DGG_L_TRANSL = 4, DGG_L_ROT = 5, DGG_L_CONF = 6, DGG_L_TOTAL_GEN = 7,
    DGG_L_TOTAL_EVA = 8, DGG_L_BCE_EVA = 9,
    DGG_L_GRAD_PENALTY = 10 /* WGAN-GP: the gradient penalty (before the gain) */
};

/* Optional concurrency hint (per calling thread): while set, every *_backward / phase call enqueued on `stream`
 * launches its weight-gradient GEMMs (dW13, dW2 — no consumer before the optimiser step) on `wgrad_stream`,
 * overlapping the dgrad chain; the call joins `wgrad_stream` back into `stream` before it returns, so callers
 * keep single-stream semantics.  wgrad_stream == NULL removes the mapping.  Results are bit-identical.
 * `wgrad_stream` must be used for nothing else (in particular not as a compute stream that has a side stream). */
int dgg_set_wgrad_stream(void* stream, void* wgrad_stream);
/* Same, with a side stream created and owned by the library (reference-counted per `stream`: every enable needs
 * one disable).  Prefer this form: a caller-provided side stream that is ALSO some compute stream with a side stream
 * of its own (easy to get from a pooled stream allocator) can close a wait cycle between the two. */
int dgg_enable_wgrad_stream(void* stream, int enable);
/* Stream capture (hipStreamBeginCapture on `stream`): every call of this library may be captured.  ROCm 7.2 limits
 * the shape: a stream that was itself forked into the capture must not wait on an event of ANOTHER captured stream
 * (hip::Stream::EndCapture then recurses without end), so map weight-gradient side streams only to the capturing
 * stream itself — second compute streams (stream2 below, a caller's evaluator stream) run their dW GEMMs in line. */

/* Side work (per calling thread, until switched off again): the calls that follow enqueue work that runs BESIDE a
 * latency-critical chain on another stream — the evaluator update and the prefetched real-batch pass beside the cGAN
 * chain in engine.TrainEngine.joint_step.  Their large forward products are then launched with half as many workgroups,
 * each walking two work items: half the chip for twice as long instead of every CU, so that the chain's ~100 short
 * dependent kernels always find free CUs (-3 % per step).  Same work items, same sums: results are bit-identical.
 * on: 0 = off; 1 = half the workgroups; n > 1 = 1 / (n + 1) of them (round 6: the look-ahead real pass, which runs beside
 * the G phase's large launches, takes a third; the evaluator update a half). */
int dgg_set_side_work(int on);

/* Gradient-ready events for data-parallel callers (per calling thread).  While `events` are registered for a gradient
 * buffer, every backward / phase call that writes parameter gradients into exactly that buffer records, in stream
 * order, events[r] (hipEvent_t, r = n-1 .. 1) as soon as ResBlock r's gradients — for the last block also the heads' —
 * are complete, and events[0] when the rest (input BatchNorm, block 0) is: the flat layout puts these ranges back to back
 * (dgg_table: [input BN | block 0 | block 1 | .. | heads]), so an all-reduce of the later blocks' range can run while the
 * backward of the earlier blocks is still in flight.  n must equal the network's number of ResBlocks (else nothing is
 * recorded).  accum_only != 0: only calls with DGG_ACCUM_GRADS record (the discriminator's gradients are final after
 * the fake-batch pass, which accumulates onto the real-batch pass).  events == NULL or n == 0 unregisters `grads`;
 * grads == NULL unregisters everything. */
int dgg_set_grad_events(const float* grads, void* const* events, int n, int accum_only);

/* Steps 1-8 of SURVEY.md 3.1 up to (not including) optim_D.step: G forward (train), ortho6d,
 * D forward on real then on fake.detach(), discriminator loss, D backward (both) -> st->d_grads.
 * Inputs: bps (B,4096), rot9 (B,9), transl (B,3), joint (B,nj) = the real batch; z (B,latent). */
int dgg_cgan_d_phase(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws,
                     size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                     const float* joint, const float* z, float* losses, void* stream);
/* Same, with the real-batch pass of D (forward, BCE, backward) enqueued on `stream2` so that it overlaps the
 * generator forward and the fake-batch pass on `stream` (fork/join with HIP events; BatchNorm running statistics
 * still advance real-then-fake).  stream2 == NULL or == stream: identical to dgg_cgan_d_phase. */
int dgg_cgan_d_phase2(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws,
                      size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                      const float* joint, const float* z, float* losses, void* stream, void* stream2);
/* Pipelining across steps (optional): enqueue on `stream2` the real-batch pass (forward, BCE, backward) of the NEXT
 * D phase — its inputs must already be resident — while the current step's G phase is still running; it starts
 * once this thread's last dgg_cgan_g_phase on the same workspace has passed its D forward (D's parameters and
 * BatchNorm running statistics are then final for this step).  The next dgg_cgan_d_phase2 call on this thread with
 * the same workspace, batch size, input pointers, d_params and a stream2 picks it up and runs only the generator /
 * fake-batch half; with anything else the prefetched pass is dropped and redone — without a trace: the pass' update of
 * D's BatchNorm running statistics is undone from a snapshot taken in front of it (the step workspace holds it), its
 * gradients are overwritten by the redone pass.  Until the pass is picked up or dropped, d_bn and d_grads hold the NEXT
 * batch's real-pass contribution: a caller that reads them in between (checkpoint, evaluation forward) calls
 * dgg_cgan_drop_prefetch first.  B must equal the batch size of the step in flight on this workspace
 * (the workspace is carved per batch size): otherwise DGG_ERR_INVALID, nothing is enqueued.  Results are
 * bit-identical to the unpipelined order.  Uses losses[12] as scratch; up to 4 workspaces per thread can be in
 * flight. */
int dgg_cgan_d_real_prefetch(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws,
                             size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                             const float* joint, float* losses, void* stream2);
/* Drop a prefetched real pass of workspace `ws` that has not been picked up: `stream` waits for it and restores D's
 * running statistics to what they were before it (d_grads keeps the dropped pass' values until the next D phase
 * overwrites them).  The next dgg_cgan_d_phase2 then runs its own real pass.  No-op without a pending pass. */
int dgg_cgan_drop_prefetch(void* ws, void* stream);
/* Forget the pending-prefetch slot keyed by workspace `ws` (call before freeing / reusing that memory). */
int dgg_cgan_release(void* ws);
/* Drop this thread's hidden state (see the header comment): pending prefetches and the wgrad side-stream map;
 * library-owned side streams are synchronised and destroyed. */
int dgg_reset(void);
/* Steps 9-12 up to (not including) optim_G.step: D forward #3 with the updated discriminator,
 * generator loss, backward through D (input gradients only) and G -> st->g_grads.
 * Must follow dgg_cgan_d_phase on the same workspace (reuses the saved generator activations). */
int dgg_cgan_g_phase(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws,
                     size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                     const float* joint, const float* z, float* losses, void* stream);
/* N3 — the WGAN-GP variant of the two phases (cfg["is_wgan"]; dexgangrasp.py:149-211, 234-270).
 * D phase: total_loss_disc = -mean(real_score) + mean(fake_score) + penalty_gain * GP (slots DGG_L_BCE_REAL /
 * DGG_L_BCE_FAKE hold the two means, DGG_L_GRAD_PENALTY the penalty), gradients -> st->d_grads.  The penalty is
 * evaluated on interpolates alpha * real + (1 - alpha) * fake with the per-sample weights alphas (B,3) =
 * [rot | transl | joint_conf] — the reference draws them with torch.randn (:151-153), the caller supplies them; its
 * parameter gradient needs a second derivative of the discriminator (tangent forward + one reverse sweep,
 * csrc/second_order.hip).  The discriminator's BatchNorm running statistics advance three times here (real, fake,
 * interpolates) and once more in the G phase.  G phase: gen_loss_fake = -mean(fake_score) (:258).
 * Workspace: dgg_wgan_workspace_bytes (a superset of the cGAN step's).  Single stream; B >= 2. */
int64_t dgg_wgan_workspace_bytes(const dgg_cfg* cfg, int batch);
int dgg_wgan_d_phase(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws, size_t ws_bytes, int B,
                     const float* bps, const float* rot9, const float* transl, const float* joint, const float* z,
                     const float* alphas, double penalty_gain, float* losses, void* stream);
int dgg_wgan_g_phase(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws, size_t ws_bytes, int B,
                     const float* bps, const float* rot9, const float* transl, const float* joint, const float* z,
                     float* losses, void* stream);
/* Convenience single-GPU step: d_phase, Adam(D), [g_phase, Adam(G)] (is_train_gen). */
int dgg_cgan_step(const dgg_cfg* cfg, const dgg_gan_state* st, const dgg_hparams* hp, void* ws, size_t ws_bytes,
                  int B, const float* bps, const float* rot9, const float* transl, const float* joint,
                  const float* z, int step_d, int step_g, int is_train_gen, float* losses, void* stream);

/* A11 — DexGanGrasp.update_dexevaluator, dexgangrasp.py:739-755: forward (train), BCE, backward
 * -> e_grads (no Adam: the caller all-reduces, then dgg_adam_l2).  label (B) floats in {0,1}. */
int dgg_eva_phase(const dgg_cfg* cfg, const float* e_params, float* e_grads, float* e_bn, const dgg_hparams* hp,
                  void* ws, size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                  const float* label, const float* drop_keep, uint64_t seed, float* losses, void* stream);

/* HIP-graph-friendly forms of the two entry points whose arguments change every step (a captured launch replays its
 * arguments): the optimiser's step number and the dropout seed live in DEVICE memory and are advanced by the calls.
 * dgg_adam_l2_graph: step_dev[0] += 1, then the update of dgg_adam_l2 with t = step_dev[0] (bias corrections formed in
 * double on the device; dyn2: 2 floats of scratch).  dgg_eva_phase_graph: dgg_eva_phase with in-kernel masks drawn
 * from seed_dev[0], which is then advanced.  With these a whole gen+eval step (engine.TrainEngine.joint_step) is
 * captured once into a HIP graph and replayed. */
int dgg_adam_l2_graph(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step_dev, float* dyn2,
                      double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                      void* stream);
int dgg_eva_phase_graph(const dgg_cfg* cfg, const float* e_params, float* e_grads, float* e_bn, const dgg_hparams* hp,
                        void* ws, size_t ws_bytes, int B, const float* bps, const float* rot9, const float* transl,
                        const float* label, uint64_t* seed_dev, float* losses, void* stream);

/* Generator-loss pieces (A8, dexgangrasp.py:272-307 + losses.py:61-76), exposed for the parity
 * tests and for the autograd path: heads (B,21) -> losses[DGG_L_GEN_FAKE..DGG_L_TOTAL_GEN] and
 * d_heads (B,21) given the discriminator's input gradient d_pose (B,12+nj) (may be NULL). */
int dgg_gen_loss(const dgg_cfg* cfg, const dgg_hparams* hp, int B, const float* heads, const float* gt_rot9,
                 const float* gt_transl, const float* gt_joint, const float* fake_score, const float* d_pose,
                 float* losses, float* d_heads, void* stream);

/* A13 helper — utils.translate_along_axis (utils/utils.py:19-48) on device:
 * transl[i] -= R[i][:, axis] * offset. */
int dgg_translate_along_axis(float* transl, const float* rot9, int B, int axis, float offset, void* stream);

/* Plain GEMM entry used by the unit tests of the MFMA kernel (not part of the reference surface):
 * mode 0: C[M,N] = A[M,K] * B[N,K]^T;  1: C[M,N] = A[M,K] * B[K,N];  2: C[M,N] = A[K,M]^T * B[K,N].
 * splits >= 1 (split-K through `slabs`, S*M*N floats; reduced into C, or left as slabs when C is
 * NULL); accumulate: C += . */
/* GEMM arithmetic (process-wide; default DGG_GEMM_BF16X3; DGG_GEMM_ALGO=f32 / bf16 in the environment select the others):
 *   DGG_GEMM_F32    : v_mfma_f32_32x32x2_f32, a k-ordered fp32 fma chain (157 TFLOP/s peak on MI355X);
 *   DGG_GEMM_BF16X3 : each fp32 operand split exactly into three bf16 pieces, the six leading partial products
 *                     accumulated in fp32 on v_mfma_f32_32x32x16_bf16 — error below the fp32 accumulation's own
 *                     rounding, 6/16 of the f32-MFMA cost.  Kernels: csrc/gemm_x5.hip (one wave per 128 x 128
 *                     accumulator block, the workgroup's four waves split K; every product of a train step), csrc/
 *                     gemm_x4.hip (loader / consumer waves; launches of >= 1024 workgroups: the inference sweep);
 *   DGG_GEMM_BF16   : plain bf16 (BASELINE config 3, "bf16 MFMA fc stacks"; never the default, never the headline dtype):
 *                     the large (128 x 128-tile) products take only the hi plane — fp32 master operands rounded to bf16
 *                     while they are staged, ONE product, fp32 accumulate; the K = 256 products, BatchNorm, losses and
 *                     Adam stay fp32.  Measured against the fp32 goldens: tests/test_gpu_steps.py::test_bf16_mode_* state the
 *                     tolerances (scores ~1e-2, losses ~1e-2 relative) — outside the 1e-4 contract by construction.
 * (Value 2 was the pre-split "planes" path of rounds 3-4: faster kernels, slower step; removed in round 5 together with
 *  its producers and the first in-register-split kernel, gemm_x3.) */
enum { DGG_GEMM_F32 = 0, DGG_GEMM_BF16X3 = 1, DGG_GEMM_BF16 = 3 };
int dgg_set_gemm_algo(int algo);
int dgg_get_gemm_algo(void);
/* Diagnostic: workgroups of the GEMM variant (mode, tile) resident per CU according to the HIP occupancy API. */
int dgg_gemm_occupancy(int mode, int bm, int bn);
/* Diagnostic / test entry of the fp32 GEMM family (gemm_f32 / gemm_x5 / gemm_x4 kernels): mode 0 NT (A [M][K], B [N][K]),
 * 1 NN (B [K][N]), 2 TN (A [K][M], B [K][N]); splits >= 1: split-K partial products into `slabs` [splits][M][N]
 * (summed into C too when C != NULL).  Operand bases and pitches need 4-byte alignment only.  PADDING: the bf16x3
 * kernel (gemm_x5_kernel, the default for 128 x 128 tiles) stages operands in 16-byte chunks.  Along K it never leaves
 * the operand (a partial last k-step is moved back to end at K and the overlap masked); a k-STRIDED operand ([K][rows]:
 * B of mode 1, A and B of mode 2) is fetched in groups of 4 rows, clamped, not masked: on the last group of every k-row
 * it reads up to 12 bytes past the last element of that k-row (never used in a result).  Inside the library every
 * operand lies in a larger buffer (flat parameter buffers, workspaces); a caller-owned k-strided A / B whose last k-row
 * ends exactly at the end of its allocation with a row count % 4 != 0 must be padded by 12 bytes — or run with
 * DGG_GEMM_ALGO=f32, whose tails are fetched element by element.  Problems with K < 64 (or split-K chunks that short) run
 * on the f32-MFMA kernel.  SIZE: the epilogues address C through 32-bit byte offsets — M * ldc * 4 must stay below 2 GiB —
 * and the bf16x3 kernels a k-contiguous operand's rows — M * lda * 4 (modes 0, 1) and N * ldb * 4 (mode 0) must stay below
 * 4 GiB (DGG_ERR_INVALID otherwise on every route, dgg_last_error() says so; the library's own products are 50 000 x 768 at most). */
int dgg_gemm_f32(int mode, int M, int N, int K, const float* A, int lda, const float* Bm, int ldb, float* C,
                 int ldc, int splits, float* slabs, int accumulate, void* stream);
/* Which kernel dgg_gemm_f32 — and every internal product of that shape launched alone — runs on under the current
 * arithmetic; nothing is launched, no pointer is read (host-only: usable without a GPU).  Rules: tiles of 128 x 128 are
 * chosen where they fill the chip (alone, or with split-K in mode 0); on them DGG_GEMM_BF16X3 takes gemm_x4_kernel from 1024
 * workgroups up and gemm_x5_kernel otherwise, provided K >= 64 (and every split-K chunk >= 64, the last one >= 16); everything
 * else, and all of DGG_GEMM_F32, runs on the f32-MFMA kernels; DGG_GEMM_BF16 uses gemm_x5_kernel for the same shapes.
 * Returns DGG_KERNEL_* or a negative dgg_status (e.g. the 2 GiB limit of C, the 4 GiB limit of a k-contiguous operand). */
enum { DGG_KERNEL_F32 = 0, DGG_KERNEL_X4 = 4, DGG_KERNEL_X5 = 5 };
int dgg_gemm_route(int mode, int M, int N, int K, int lda, int ldb, int ldc, int splits);

#ifdef __cplusplus
}
#endif
#endif /* DGG_H */
