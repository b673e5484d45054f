/* dsdf.h -- C ABI of libdsdf_hip.so: the DeepSDF auto-decoder training step on MI355X (gfx950).
 *
 * The reference (mkofler96/DeepSDF) has NO C/FFI boundary for this path: its hot loop is inline Python
 * (train_deep_sdf.py:481-545) over torch ops.  This header is therefore the boundary that a binding of
 * that loop would target; every entry point names the reference lines it replaces.  The Python side of
 * this repo (deepsdf_amd/_lib.py) binds it with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (DSDF_E_*); dsdf_last_error() returns a
 *     thread-local message.  Nothing throws, nothing allocates device memory, nothing synchronises the
 *     device: work is enqueued on the caller's stream (hipStream_t passed as void*).
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor kept alive by the caller),
 *     16-byte aligned, unless marked [host].
 *   - all tensors are row-major fp32; index arrays are int64.
 *   - decoder parameters, their gradients and Adam moments live in flat "arenas" whose layout is the
 *     reference module's named_parameters() order (dsdf_param_layout).
 */
#ifndef DSDF_H
#define DSDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSDF_MAX_LAYERS 16
#define DSDF_ABI_VERSION 15
#define DSDF_MAX_BUCKETS 8

enum {
  DSDF_OK = 0,
  DSDF_E_INVALID = -1,   /* bad argument (shape, alignment, unsupported NetworkSpecs variant) */
  DSDF_E_WORKSPACE = -2, /* workspace too small */
  DSDF_E_LAUNCH = -3     /* a HIP call failed */
};

/* Architecture of deep_sdf/networks/deep_sdf_decoder.py:10-73 after its layer-size arithmetic (:29-48).
 * Layer l is Linear(in_dim[l] -> out_dim[l]); hidden layers apply ReLU (+dropout); the last layer
 * (out_dim == 1) is followed by tanh (twice if use_tanh), :94-95,108-109. */
typedef struct DsdfNet {
  int32_t n_layers;                 /* number of Linear layers (= len(dims)+1) */
  int32_t latent_size;              /* L  (CodeLength) */
  int32_t geom_dim;                 /* G  (geom_dimension) */
  int32_t in_dim[DSDF_MAX_LAYERS];
  int32_t out_dim[DSDF_MAX_LAYERS];
  uint32_t weight_norm_mask;        /* bit l: layer l is weight-normed (g = original0, v = original1) */
  uint32_t dropout_mask;            /* bit l: F.dropout after layer l's ReLU (:105-106) */
  uint32_t skip_mask;               /* bit l: layer l's input is [x || x0]  (latent_in, :88-89) */
  float dropout_p;
  int32_t use_tanh;
  int32_t fwd_bf16;                 /* BASELINE config 5: hidden-layer forward GEMMs take bf16 inputs (weights and layer inputs
                                       rounded to nearest-even), fp32 accumulate on v_mfma_f32_32x32x16_bf16; the output layer,
                                       the backward pass, master weights and Adam stay fp32.  Needs every width <= 512. */
  /* Decoder variants no shipped spec uses (layer-by-layer kernels, general mode; not with fwd_bf16): */
  int32_t latent_dropout;           /* deep_sdf_decoder.py:79-82: in training, layer 0 sees F.dropout(latent, 0.2) (the skip layer
                                       still concatenates the original input); mask = the dropout hash under dropout_key[15] */
  int32_t xyz_in_all;               /* :90-91: every layer l >= 1 that is not a latent_in layer takes [x || xyz]
                                       (in_dim[l] = out_dim[l-1] + geom_dim) */
  uint32_t ln_param_mask;           /* :60-65 (norm_layers WITHOUT weight_norm): bit l: a bn{l} = nn.LayerNorm(out_dim[l]) module exists
                                       (parameters bn{l}.weight, bn{l}.bias right after lin{l}.weight, lin{l}.bias); forward applies it
                                       between the Linear and the ReLU of every HIDDEN layer that has one (:97-103) */
  int32_t gemm_split;               /* opt-in: the hidden-layer GEMMs of the three MFMA-bound kernels (forward, backward dX chain, dW) run on the bf16 matrix
                                       pipe with every fp32 operand cut into three bf16 terms (6 of the 9 cross products, fp32 accumulate):
                                       fp32 accuracy (same parity tolerances), 2.7 x the MFMA rate.  Needs every width <= 512; not with the
                                       variants above.  Together with fwd_bf16: the bf16 forward as it is, the backward dX chain in split
                                       and dW in split mode.  Everything else is unchanged. */
} DsdfNet;

/* Offsets (in floats) of every parameter tensor inside the decoder arena, named_parameters() order:
 * weight-normed layer: bias, g [out,1], v [out,in];  plain layer: weight [out,in], bias. */
typedef struct DsdfParamLayout {
  int64_t total;
  int64_t bias_off[DSDF_MAX_LAYERS];
  int64_t g_off[DSDF_MAX_LAYERS];   /* -1 for plain layers */
  int64_t v_off[DSDF_MAX_LAYERS];   /* v (weight-normed) or weight (plain) */
  int64_t ln_w_off[DSDF_MAX_LAYERS]; /* bn{l}.weight (LayerNorm gamma), -1 if layer l has no bn module */
  int64_t ln_b_off[DSDF_MAX_LAYERS]; /* bn{l}.bias   (LayerNorm beta) */
} DsdfParamLayout;

/* Batch of one optimiser (sub-)step, train_deep_sdf.py:483-501.  Points are grouped in R contiguous
 * runs ("segments"), run r = points [seg_offset[r], seg_offset[r+1]) all of scene seg_scene[r]; this is
 * the layout `indices.unsqueeze(-1).repeat(1, S)` (+ torch.chunk) always produces. */
typedef struct DsdfBatch {
  const int64_t* seg_scene;   /* [R]   row of the latent table */
  const int64_t* seg_offset;  /* [R+1] seg_offset[0] = 0, seg_offset[R] = n_points */
  int64_t n_segments;         /* R */
  const float* xyz;           /* [n_points, G] */
  const float* sdf_gt;        /* [n_points]  (unclamped; clamped inside, :493) or NULL for inference */
  int64_t n_points;           /* N of this chunk */
  int64_t n_norm;             /* loss normaliser: the FULL step's point count, also across ranks (:519) */
  int64_t row_offset;         /* index of this chunk's first point inside the step (dropout hash) */
  int64_t seg_len;            /* > 0: EVERY segment has exactly this many points (the reference's B x S layout, also per
                                 --batch_split chunk when chunks hold whole scenes); 0: irregular.  A multiple of 64
                                 selects SEGMENT MODE: the per-scene latent products are computed once per scene instead
                                 of once per point (same results up to fp32 summation order). */
} DsdfBatch;

typedef struct DsdfLossCfg {
  float clamp_dist;           /* ClampingDistance delta (:335,493,517) */
  float reg_coef;             /* CodeRegularizationLambda * min(1, epoch/100), 0 disables (:523-527) */
  float code_bound;           /* CodeBound (Embedding max_norm, :343,385); <= 0 disables the renorm */
  int32_t training;           /* 1: dropout active (decoder.train(), :477) */
  int32_t frozen_decoder;     /* 1: skip the decoder's weight gradients (latent-only optimisation, config 4); grads untouched */
  uint32_t dropout_key[DSDF_MAX_LAYERS]; /* [host-computed] per-layer hash keys (oracle: dropout_layer_key) */
  int32_t dw_phase;           /* data-parallel steps that exchange the decoder gradient in K = dw_buckets >= 2 buckets (replaces
                                 nn.DataParallel's reduce, train_deep_sdf.py:353), one call per bucket: 0 = the whole backward in this call
                                 (dw_buckets <= 1); 1 = everything except the weight gradients of the layers below bucket 0 -- on
                                 return (stream order) the arena holds the gradients of bucket 0 (the LAST layers), whose all-reduce
                                 can start; p in 2..K = only the weight gradients of bucket p - 1, from the activations / dP the
                                 phase-1 call left in `ws` (same net, batch, workspace and dw_buckets; nothing else may touch `ws` in
                                 between).  Buckets and their arena ranges: dsdf_grad_buckets.  Phases need the fused kernels
                                 (dsdf_dw_phase_supported), a trainable decoder and accumulate = 0. */
  int32_t dw_buckets;         /* K of dw_phase (0 or 1: no buckets); workspace: dsdf_workspace_bytes_buckets when K > 2 */
} DsdfLossCfg;

typedef struct DsdfAdamCfg {
  int64_t step;               /* 1-based, shared by every tensor (torch/optim/adam.py) */
  float lr_decoder, lr_latent;/* LearningRateSchedule[0], [1] at this epoch (:315-318) */
  float beta1, beta2, eps;    /* torch defaults 0.9, 0.999, 1e-8 (:400) */
  const float* grad_scale;    /* optional device scalar multiplying decoder grads (grad clipping), or NULL */
} DsdfAdamCfg;

/* ---- introspection ------------------------------------------------------------------------------ */
int dsdf_abi_version(void);
const char* dsdf_last_error(void);
int dsdf_param_layout(const DsdfNet* net, DsdfParamLayout* out);          /* [host] */
int dsdf_packed_floats(const DsdfNet* net, int64_t* n_floats);            /* [host] size of the packed-weight buffer */
int dsdf_workspace_bytes(const DsdfNet* net, int64_t n_points, int64_t n_segments, size_t* bytes); /* [host] train/module */
int dsdf_decode_workspace_bytes(const DsdfNet* net, int64_t n_points, size_t* bytes);                /* [host] dsdf_decode */

/* [host] workspace of a step that runs its backward in n_buckets phases (DsdfLossCfg.dw_buckets): the split-K slabs of a
 * bucket's weight-gradient launch are finer than the whole launch's.  dsdf_workspace_bytes covers n_buckets <= 2. */
int dsdf_workspace_bytes_buckets(const DsdfNet* net, int64_t n_points, int64_t n_segments, int32_t n_buckets, size_t* bytes);

/* [host] 1 if this net's training step can run its backward in phases (the fused kernels take it; in this process: the
 * DSDF_NO_FUSED switch counts), 0 if not (a caller then exchanges the gradient in ONE piece), < 0 for an invalid net. */
int dsdf_dw_phase_supported(const DsdfNet* net);

/* [host] the K = n_buckets (2..DSDF_MAX_BUCKETS) gradient buckets of DsdfLossCfg.dw_phase, in the order the backward
 * finishes them (last layers first): bucket b = layers [first_layer[b], first_layer[b - 1]) (b = 0: up to the last layer)
 * = arena floats [arena_off[b + 1], arena_off[b]); arena_off has K + 1 entries, arena_off[0] = total, arena_off[K] = 0.
 * A net with fewer layers than buckets leaves trailing buckets empty. */
int dsdf_grad_buckets(const DsdfNet* net, int32_t n_buckets, int32_t* first_layer, int64_t* arena_off);

/* ---- weights -------------------------------------------------------------------------------------
 * W = g * v / ||v||_row for weight-normed layers (torch._weight_norm via parametrizations.weight_norm,
 * deep_sdf_decoder.py:50-55), plain copy otherwise; written as W [out,in] and W^T [in,out] in padded,
 * MFMA-friendly layouts (row-major W and W^T, plus fragment-ordered copies for the fused kernels).  `packed` must be
 * ZERO-INITIALISED once by the caller (padding tiles are never written).  Must be called after every parameter
 * change (dsdf_adam_step does it itself). */
int dsdf_materialize_weights(const DsdfNet* net, const float* params, float* packed, void* stream);

/* ---- inference: deep_sdf/utils.py:54-65 decode_sdf / Decoder.forward in eval mode ------------------
 * input [n, L+G] (latent first, xyz last) with row stride ld_in floats -> sdf [n]. */
int dsdf_decode(const DsdfNet* net, const float* packed, const float* params, const float* input, int64_t ld_in,
                int64_t n, float* sdf_out, void* ws, size_t ws_bytes, void* stream);

/* deep_sdf/utils.py:54-65 decode_sdf with ONE latent vector for all query points (what every caller does: mesh.py:61-70,
 * 262-271): sdf = Decoder.eval()([latent.expand(n) || xyz]).  The [n, L+G] input is never built: W[:, latent] latent is
 * computed once and enters the forward as the accumulators' initial value (segment mode).  latent [L],
 * xyz [n, G], sdf_out [n]; workspace: dsdf_decode_workspace_bytes.  Needs the fp32 fused forward (widths <= 512,
 * geom_dim <= 4, at least two hidden layers), otherwise DSDF_E_INVALID: use dsdf_decode. */
int dsdf_decode_latent(const DsdfNet* net, const float* packed, const float* params, const float* latent, const float* xyz,
                       int64_t n, float* sdf_out, void* ws, size_t ws_bytes, void* stream);
/* [host] 1 if dsdf_decode_latent accepts this net (in this process: the DSDF_NO_FUSED switch counts), 0 if the caller has
 * to build the [n, L+G] input and use dsdf_decode (variants on the layer-by-layer kernels: xyz_in_all, latent_dropout,
 * LayerNorm; widths > 512; geom_dim > 4; fewer than two hidden layers), < 0 for an invalid net.  The ONE definition of
 * that condition: deep_sdf.utils.decode_sdf (deep_sdf/utils.py:54-65) asks it instead of restating it. */
int dsdf_decode_latent_supported(const DsdfNet* net);

/* ---- module path: Decoder.forward / autograd backward on an explicit input (plugin seam,
 * train_deep_sdf.py:275,514).  forward keeps activations in ws; backward consumes them. */
int dsdf_module_forward(const DsdfNet* net, const float* packed, const float* params, const float* input,
                        int64_t ld_in, int64_t n, int32_t training, const uint32_t* dropout_key /*[host]*/,
                        float* sdf_out, void* ws, size_t ws_bytes, void* stream);
int dsdf_module_backward(const DsdfNet* net, const float* packed, const float* params, const float* d_sdf,
                         int64_t n, int32_t training, const uint32_t* dropout_key /*[host] the forward's keys; only
                         latent_dropout nets read it (slot 15), may be NULL otherwise*/,
                         float* grads /*arena, overwritten or accumulated*/,
                         int32_t accumulate, float* d_input /*[n, ld_din] or NULL*/, int64_t ld_din,
                         void* ws, size_t ws_bytes, void* stream);
/* Forward-mode tangent (Jacobian-vector product) of Decoder.forward at the point of the LAST dsdf_module_forward on this
 * workspace: jvp_out [n] = d sdf / d input . tangent, tangent [n, ld_t] (L+G columns used).  What
 * torch.autograd.functional.jvp computes through the reference decoder in deep_sdf/mesh.py:420 (d vertices / d latent
 * control points) by double backward; here it is one extra pass through the same GEMMs with the primal pass's ReLU /
 * dropout decisions.  May be called any number of times after one forward (before or after dsdf_module_backward). */
int dsdf_module_jvp(const DsdfNet* net, const float* packed, const float* params, const float* tangent, int64_t ld_t,
                    int64_t n, int32_t training, const uint32_t* dropout_key /*[host] as dsdf_module_backward*/,
                    float* jvp_out, void* ws, size_t ws_bytes, void* stream);

/* ---- training step ---------------------------------------------------------------------------------
 * dsdf_train_forward_backward = train_deep_sdf.py:509-533 for one chunk: max-norm renorm of the looked-up
 * latent rows (in place), gather + concat, decoder forward, clamp, sum-L1 / n_norm, code regulariser,
 * backward.  grads (decoder arena) and dlat [S_tot, L] are overwritten when accumulate == 0, added to
 * otherwise (--batch_split).  loss_out: device float, same accumulate rule.  sdf_out may be NULL. */
int dsdf_train_forward_backward(const DsdfNet* net, const float* packed, const float* params,
                                float* latent_table, int64_t n_scenes, const DsdfBatch* batch,
                                const DsdfLossCfg* cfg, float* grads, float* dlat, float* loss_out,
                                float* sdf_out, int32_t accumulate, void* ws, size_t ws_bytes, void* stream);

/* clip_grad_norm_ over the decoder arena (train_deep_sdf.py:541-543): writes total norm and the clip
 * coefficient min(1, max_norm/(norm+1e-6)) to two device floats. */
int dsdf_grad_norm(const float* grads, int64_t n, float max_norm, float* norm_out, float* coef_out,
                   void* ws, size_t ws_bytes, void* stream);

/* optimizer_all.step() (train_deep_sdf.py:545): fused Adam over the decoder arena (lr_decoder) and the
 * WHOLE latent table (lr_latent; dense update, rows absent from the batch keep moving by momentum),
 * then re-materialises the packed weights. */
int dsdf_adam_step(const DsdfNet* net, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   float* latent_table, const float* dlat, float* lat_exp_avg, float* lat_exp_avg_sq,
                   int64_t n_latent_floats, const DsdfAdamCfg* cfg, float* packed, void* stream);

/* Single-GPU fast path = dsdf_train_forward_backward (accumulate 0) + dsdf_adam_step in one call.  When nothing has to
 * happen between the gradients and the update (no all-reduce, no --batch_split accumulation, no clipping:
 * adam->grad_scale == NULL) the decoder's Adam and the new weight-norm scales are folded into the split-K finalize pass
 * (the gradient arena is then NOT written); otherwise it is exactly the two calls in sequence. */
int dsdf_train_step(const DsdfNet* net, float* packed, float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                    float* latent_table, int64_t n_scenes, float* dlat, float* lat_exp_avg, float* lat_exp_avg_sq,
                    const DsdfBatch* batch, const DsdfLossCfg* cfg, const DsdfAdamCfg* adam, float* loss_out, float* sdf_out,
                    void* ws, size_t ws_bytes, void* stream);

/* latent-only Adam (frozen decoder; config 4): updates only the given latent arena. */
int dsdf_adam_latent_only(float* latent, const float* dlat, float* exp_avg, float* exp_avg_sq, int64_t n,
                          const DsdfAdamCfg* cfg, void* stream);

/* The same update for a loop that is CAPTURED INTO A HIP GRAPH and replayed (deepsdf_amd/reconstruct.py: the 800 iterations of a
 * reconstruction are one captured iteration replayed, no host work per iteration): everything that changes from step to step comes
 * from DEVICE memory.  sched [n_steps][2] = { lr_t / (1 - beta1^t), sqrt(1 - beta2^t) } for t = 1 .. n_steps (computed by the caller
 * in double, as dsdf_adam_latent_only does on the host), indexed by the device counter *step_counter (0-based; entries past the end
 * reuse the last one); the call increments the counter after the update (a second, one-thread launch).  The gradient it applies is
 * dlat + l2_coef * latent (the code regulariser l2reg * mean(z^2) of the reconstruction loss; 0: none). */
int dsdf_adam_latent_sched(float* latent, const float* dlat, float* exp_avg, float* exp_avg_sq, int64_t n, const float* sched,
                           int64_t n_steps, int64_t* step_counter, float beta1, float beta2, float eps, float l2_coef, void* stream);

/* ---- diagnostics: per-kernel-class device time from HIP events recorded on the caller's stream around every
 * launch of that class (bench.py's roofline object).  Off by default; thread-local; read synchronises. */
#define DSDF_PROF_CLASSES 8
enum { DSDF_PROF_GEMM_NT = 0, DSDF_PROF_GEMM_TN = 1, DSDF_PROF_LAST = 2, DSDF_PROF_FUSED_FWD = 3, DSDF_PROF_FUSED_BWD = 4,
       DSDF_PROF_DW_STREAM = 5, DSDF_PROF_OTHER = 6, DSDF_PROF_FUSED_FWD_BWD = 7 /* training: forward + backward in one launch */ };
typedef struct DsdfProfile {
  double ms[DSDF_PROF_CLASSES];     /* summed event-to-event time per class */
  double flops[DSDF_PROF_CLASSES];  /* summed ALGORITHMIC 2*M*N*K of the GEMMs each launch covers (the reference's dense form) */
  int64_t count[DSDF_PROF_CLASSES]; /* launches per class */
  int32_t dropped;                  /* 1 if the event pool overflowed (results incomplete) */
} DsdfProfile;
int dsdf_profile_enable(int32_t on);
int dsdf_profile_read(DsdfProfile* out);

/* ---- per-step subsampling --------------------------------------------------------
 * Replaces deep_sdf/data.py:74-110 unpack_sdf_samples + the DataLoader collate (train_deep_sdf.py:483-501) for
 * samples that are resident in HBM: for each of the B scenes, S = 2*(subsample/2) rows -- subsample/2 positives and
 * negatives drawn WITHOUT replacement (the reference's torch.randperm(len)[:n]), a shortfall of one sign made up by
 * the other, positives first -- gathered into xyz_out [B*S, G] / sdf_out [B*S].
 *   data       [rows, G+1] fp32: xyz then sdf; scene k's positives are rows [pos_start[k], pos_start[k]+n_pos[k]),
 *              its negatives [neg_start[k], ...); the four per-scene arrays are int64 on the device
 *   scene_ids  [B] int64 (device): which scenes, in batch order
 *   key        64-bit draw key (e.g. seed and step); the same key gives the same batch
 * The permutation is the keyed Feistel network specified by oracle/deepsdf_oracle.py sample_perm (bit-exact).
 * Every selected scene needs n_pos + n_neg >= S and each sign at most 2^30 rows (checked by the caller, who owns the
 * sizes; the library cannot read device arrays on the host). */
int dsdf_sample_batch(const float* data, int32_t geom_dim, const int64_t* pos_start, const int64_t* n_pos,
                      const int64_t* neg_start, const int64_t* n_neg, const int64_t* scene_ids, int64_t n_batch_scenes,
                      int64_t subsample, uint64_t key, float* xyz_out, float* sdf_out, void* stream);

/* The same draw inside a graph-captured loop (deepsdf_amd/reconstruct.py): the draw key is key0 + *counter * key_step (mod 2^64),
 * *counter a DEVICE integer read by the kernel -- e.g. the step counter dsdf_adam_latent_sched advances -- so every replay of
 * the captured launch draws a new batch, and the sequence equals that of dsdf_sample_batch called with those keys. */
int dsdf_sample_batch_seq(const float* data, int32_t geom_dim, const int64_t* pos_start, const int64_t* n_pos,
                          const int64_t* neg_start, const int64_t* n_neg, const int64_t* scene_ids, int64_t n_batch_scenes,
                          int64_t subsample, uint64_t key0, uint64_t key_step, const int64_t* counter, float* xyz_out, float* sdf_out,
                          void* stream);

/* ---- building blocks (exported for the parity tests and profiling; not needed by a trainer) --------- */
/* C[M,N] = A[M,K] * B[N,K]^T (+bias) */
int dsdf_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                 int64_t N, int64_t K, const float* bias, void* stream);
/* C[M,N] = A[K,M]^T * B[K,N]  (split-K inside; ws holds the partial slabs) */
int dsdf_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                 int64_t N, int64_t K, void* ws, size_t ws_bytes, void* stream);
/* keep-mask of the dropout hash as 0/1 bytes [rows, cols] (spec: oracle/deepsdf_oracle.py dropout_keep) */
int dsdf_dropout_mask(uint32_t key, float p, int64_t rows, int64_t cols, int64_t row_offset, uint8_t* out,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSDF_H */
