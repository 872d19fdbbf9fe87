/* dr4sr_hip.h — C ABI of the MI355X (gfx950) hot path for DR4SR's target-model training loop.
 *
 * The reference (USTC-StarTeam/DR4SR) has no FFI of its own: the path is pure PyTorch.  The
 * boundary a maintainer would bind is therefore "what the reference asks torch to do" on
 *   model/basemodel.py:177-214  (training_epoch / training_step: neg-sampling, encoder forward,
 *                                tied-embedding scorer, BCE, backward, Adam)
 *   model/sasrec.py:39-75       (SASRecQueryEncoder.forward)
 *   model/loss_func.py:9-38     (BinaryCrossEntropyLoss)
 *   model/basemodel.py:354-365  (topk: full-item scorer)
 * Every entry point below names the reference lines it replaces.  INTEGRATION.md shows the
 * ctypes stub that binds them from the reference's Python.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless said otherwise
 *   - fp32 activations/parameters, int64 ids (as the reference stores them), row-major, contiguous
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (no host sync,
 *     no allocation) so calls may be captured into a hipGraph
 *   - return value: 0 = ok, negative = argument error (DR4SR_E_*; transport: DR4SR_E_RCCL_BASE - ncclResult_t), positive = hipError_t
 *   - nothing is retained across calls except what the caller passes in (plan + workspace)
 */
#ifndef DR4SR_HIP_H
#define DR4SR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR4SR_ABI_VERSION 8

#define DR4SR_E_ARG      (-1)   /* null pointer / bad size                                   */
#define DR4SR_E_SHAPE    (-2)   /* unsupported D / H / F / L combination     */
#define DR4SR_E_WS       (-3)   /* workspace too small                                       */
#define DR4SR_E_RCCL_BASE (-100) /* transport (ABI 8): an RCCL error r is returned as DR4SR_E_RCCL_BASE - r (ncclResult_t, r >= 1) */

/* dropout sites (RNG stream ids); per-layer sites are  kind + 4*layer */
#define DR4SR_SITE_EMB   0      /* sasrec.py:66   dropout(seq_embs + position_embs)          */
#define DR4SR_SITE_ATTN  1      /* torch MHA      dropout on attention probabilities         */
#define DR4SR_SITE_PROJ  2      /* torch TEL      dropout1 after out_proj                    */
#define DR4SR_SITE_ACT   3      /* torch TEL      dropout after the activation               */
#define DR4SR_SITE_FFN   4      /* torch TEL      dropout2 after linear2                     */

/* plan->optimizer (ABI 6): basemodel.py:79-98's `optimizer` choices, each with torch's defaults as the reference constructs them */
#define DR4SR_OPT_ADAM    0     /* torch.optim.Adam(lr, weight_decay); also what an unknown name falls back to (weight_decay 0)   */
#define DR4SR_OPT_SGD     1     /* torch.optim.SGD(lr, weight_decay): no momentum                                                */
#define DR4SR_OPT_ADAGRAD 2     /* torch.optim.Adagrad(lr, weight_decay): adam_v = state_sum, adam_eps = 1e-10                    */
#define DR4SR_OPT_RMSPROP 3     /* torch.optim.RMSprop(lr, weight_decay): adam_v = square_avg, beta2 = alpha 0.99, adam_eps 1e-8  */

#define DR4SR_POOL_NONE   0
#define DR4SR_POOL_ORIGIN 1     /* module/layers.py:41-50  zero rows >= seqlen  -> [B,L,D]   */
#define DR4SR_POOL_LAST   2     /* module/layers.py:69-73  row seqlen-1         -> [B,D]     */
#define DR4SR_POOL_MEAN   3     /* module/functional.py:50-55  mean of rows < seqlen -> [B,D] (CL4SRec views)   */

/* slots of the int32 `state` buffer (device), DR4SR_STATE_WORDS words, owned by the caller */
#define DR4SR_STATE_STEP     0  /* optimizer step counter t (incremented by dr4sr_adam_step) */
#define DR4SR_STATE_T        1  /* number of packed (valid) tokens of the current batch      */
#define DR4SR_STATE_NVALID   2  /* number of loss positions (target != 0)                    */
#define DR4SR_STATE_RNGSTEP  3  /* RNG step (incremented by every fwd_bwd)                   */
#define DR4SR_STATE_WORDS    16

/* Flat parameter layout (fp32), identical for params / grads / adam_m / adam_v:
 *   E[N,D] | P[L,D] | per layer: in_w[3D,D] in_b[3D] out_w[D,D] out_b[D] w1[F,D] b1[F] w2[D,F] b2[D]
 *                                 ln1_w[D] ln1_b[D] ln2_w[D] ln2_b[D]
 * (names: item_embedding.weight, query_encoder.position_emb.weight,
 *  query_encoder.transformer_layer.layers.{i}.{self_attn.in_proj_weight, ...} — SURVEY.md §8a)
 * The grads buffer has DR4SR_GRAD_TAIL extra floats after n_params:
 *   grads[n_params+0] = number of loss positions (as float), grads[n_params+1] = loss SUM,
 *   grads[n_params+2] = poison word (non-zero: a producer of this gradient failed on the device; the optimizer skips the step).
 * Gradients are accumulated UN-normalised (d loss_sum); dr4sr_adam_step divides by
 * grads[n_params+0], so that under data parallelism ONE sum-all-reduce of the whole buffer yields
 * the reference's global-batch normalisation (loss_func.py:18-19, :29-30). */
#define DR4SR_GRAD_TAIL 4

typedef struct dr4sr_sasrec_plan {
    int32_t abi_version;            /* must be DR4SR_ABI_VERSION                                */
    /* ---- model dims (configs/basemodel.yaml, configs/sasrec.yaml) ---- */
    int32_t B, L, D, H, F, n_layer, n_items;
    float   ln_eps;                 /* layer_norm_eps                                            */
    float   p_drop;                 /* dropout_rate (0 => no RNG work at all)                    */
    uint64_t seed;                  /* Philox key                                                */
    /* ---- parameters ---- */
    float*  params;                 /* [n_params]                                                */
    float*  grads;                  /* [n_params + DR4SR_GRAD_TAIL]                              */
    float*  adam_m;                 /* [n_params]                                                */
    float*  adam_v;                 /* [n_params]                                                */
    int64_t n_params;
    /* ---- batch: rows `rows[0..B)` of device-resident dataset tensors (data/dataset.py:79-91).
     *      rows == NULL means rows 0..B-1, i.e. the three pointers ARE the batch tensors. ---- */
    const int64_t* in_item_id;      /* [U,L]  batch['in_item_id']                                */
    const int64_t* item_id;         /* [U,L]  batch['item_id'] (targets), may be NULL for encode */
    const int64_t* seqlen;          /* [U]    batch['seqlen']                                    */
    const int64_t* rows;            /* [B] or NULL                                               */
    int64_t* neg_item;              /* [B,L]  batch['neg_item'] (K = 1)                          */
    int32_t  sample_neg;            /* 1: draw negatives in-kernel (basemodel.py:50-61) and WRITE
                                          them to neg_item; 0: READ neg_item                     */
    /* ---- scratch ---- */
    void*    workspace;             /* >= dr4sr_sasrec_workspace_bytes(plan) bytes               */
    int64_t  workspace_bytes;
    int32_t* state;                 /* [DR4SR_STATE_WORDS] device words, zero-initialised once   */
    /* ---- optimizer (basemodel.py:79-98: torch.optim.Adam) ---- */
    float lr, beta1, beta2, adam_eps, weight_decay;
    /* ---- a1 fused batch selection (optional; replaces a separate dr4sr_select_rows launch): when perm != NULL the first
     *      kernel of dr4sr_sasrec_fwd_bwd/_train_step/_encode(training) FILLS rows[i] = perm[(c*perm_stride + perm_offset + i)
     *      mod n_perm], i < B, with c = *perm_counter, then sets *perm_counter = c + 1.  `rows` must then be writable. ---- */
    const int64_t* perm;            /* [n_perm] one epoch's permutation of dataset rows, or NULL  */
    int64_t  n_perm;
    int64_t  perm_stride;           /* global batch size                                          */
    int64_t  perm_offset;           /* this rank's offset inside the global batch                 */
    int32_t* perm_counter;          /* device int32: batch index within the epoch                 */
    /* ---- optional per-step loss log: dr4sr_adam_step / _train_step write loss_log[slot] = loss_sum / n_valid of the step with
     *      slot = *perm_counter - 1 when perm != NULL (the batch index just consumed), else slot = 0.  NULL = off. ---- */
    float*   loss_log;
    /* ---- regime hint (ABI 4): the host's estimate of the VALID tokens of a batch of B rows (B * mean(min(seqlen, L)) of the
     *      dataset).  The launchers pick their forms from it — token-tile kernels: 16-row tiles + atomics for the table gradient up to
     *      ~7 k packed tokens, 32-row tiles + scatter / owner jobs above; attention: one workgroup per sequence up to ~14 k tokens,
     *      length-class lists above — because the real count lives on the device.  0 = unknown: the capacity B * L decides (boundary
     *      16 384 for both), as before ABI 4.  (Boundaries quoted for d = 64; they scale with 64 / d.)  A wrong hint costs speed, never
     *      correctness. ---- */
    int32_t  expected_tokens;
    int32_t  optimizer;             /* DR4SR_OPT_* (ABI 6; 0 = Adam)                                                              */
} dr4sr_sasrec_plan;

/* -------------------------------------------------------------------------------------------- */
int  dr4sr_abi_version(void);
int  dr4sr_sasrec_plan_sizeof(void);   /* sizeof(dr4sr_sasrec_plan) as compiled: lets a binding verify its struct mirror */
/* Fills offsets[0]=E, [1]=P, [2+12*i+j] = j-th tensor of layer i (order above); returns n_params. */
int64_t dr4sr_sasrec_param_layout(int32_t n_items, int32_t L, int32_t D, int32_t F, int32_t n_layer,
                                  int64_t* offsets /* [2+12*n_layer] or NULL */);
/* Bytes of scratch a plan of this shape needs (plan->workspace may be NULL here), or a negative DR4SR_E_* code: DR4SR_E_SHAPE for an
 * encoder shape without kernels — L > 64; (D, F) outside {(64,128), (64,256), (128,128)}; head_dim other than 32 / 64; a head count != 2
 * whose one-wave-per-head attention would not fit 160 KB of LDS — so an unsupported configuration is refused when the engine is built. */
int64_t dr4sr_sasrec_workspace_bytes(const dr4sr_sasrec_plan* plan);
/* launch forms a step of this plan takes (see expected_tokens): bit 0 = at-scale token-tile kernels (32-row tiles, scatter / owner jobs),
 * bit 1 = the at-scale attention regime of short-sequence plans (length-class lists, or — bit 4 — the wave-per-tile launches that
 * replace them: csrc/attn_wave.hip), bit 2 = the attention runs inside the 16-token tile launches (the latency forms: no attention
 * launches; csrc/attn_tile.h), bit 3 = the opt-in window launches (csrc/attn_tile_sa.hip), bit 5 = bit 4 with the forward folded into the
 * wave-tile forward launches (bits 3 and 5: experiments build only, both measured slower), bit 6 = the deterministic latency form
 * (DR4SR_DETERMINISTIC on a plan of the latency regime: the latency launches of bit 2 with the attention's shared dK | dV rows, the table
 * gradient and the weight-gradient splits summed in a fixed order); < 0: DR4SR_E_* */
int dr4sr_sasrec_at_scale(const dr4sr_sasrec_plan* plan);

/* Deterministic mode (the reference's run-to-run determinism request, utils/utils.py:13-20).  Environment switch DR4SR_DETERMINISTIC=1, read
 * when a plan's workspace is carved (like every switch: cached per call site until the hooks header's reload entry point is called): every reduction of the training
 * steps dr4sr_sasrec_fwd_bwd[_weighted / _phase] / _train_step[s], dr4sr_fmlp_fwd_bwd / _train_step, dr4sr_gru4rec_fwd_bwd / _train_step[s] and
 * of the autograd-path backwards (dr4sr_*_encode_bwd) and of the dense scorer backward (dr4sr_score_bce_bwd / _bpr_bwd) runs in a fixed order — no fp32 atomics — so two runs from one state are bit-identical.
 * The mode needs partial-sum buffers: the *_workspace_bytes entry points answer for the mode that is set when it is called, and a workspace sized without
 * the mode is refused with DR4SR_E_WS once the mode is on.  It is combined with neither DR4SR_WGRAD_F32 (DR4SR_E_SHAPE) nor, for ordered
 * results, DR4SR_DE_ATOMIC. */

/* One reference training step minus the optimizer:  basemodel.py:193-198
 *   (_neg_sampling) -> training_step (sasrec.py:39-75 encoder, basemodel.py:204-214 scorer,
 *   loss_func.py:9-38 BCE) -> loss.backward().
 * Leaves un-normalised gradients + {n_valid, loss_sum} in plan->grads (zeroed first). */
int dr4sr_sasrec_fwd_bwd(const dr4sr_sasrec_plan* plan, void* stream);

/* optimizer.step() (basemodel.py:199) on the flat buffers: dense torch.optim.Adam semantics,
 * g = grads[i] / grads[n_params] (+ weight_decay * p).  Increments state[STEP]. */
int dr4sr_adam_step(const dr4sr_sasrec_plan* plan, void* stream);

/* fwd_bwd + adam in one call (single-GPU fast path). */
int dr4sr_sasrec_train_step(const dr4sr_sasrec_plan* plan, void* stream);

/* n_steps iterations of the loop body of BaseModel.training_epoch (basemodel.py:193-199) in one call: with plan->perm set,
 * consecutive batches of the epoch permutation (perm_counter advances n_steps times), each with fresh dropout masks and
 * negatives.  Same results as n_steps calls of dr4sr_sasrec_train_step; the optimizer launch of every step but the last
 * also prepares the step that follows (batch selection, sequence offsets, zeroed gradients), so the call enqueues one prep
 * kernel instead of n_steps.  After the call plan->grads holds the last step's gradients. */
int dr4sr_sasrec_train_steps(const dr4sr_sasrec_plan* plan, int32_t n_steps, void* stream);

/* The same fusion for a data-parallel loop, where the all-reduce of plan->grads sits between the two halves of a step:
 *   dr4sr_sasrec_fwd_bwd(plan)                                    first step (runs its own prep)
 *   all-reduce ; dr4sr_adam_step_prepare_next(plan)               optimizer + prep of the next batch (selection, offsets, zeroed grads)
 *   dr4sr_sasrec_fwd_bwd_prepared(plan) ; all-reduce ; dr4sr_adam_step_prepare_next(plan) ; ...
 * Finish with dr4sr_adam_step to leave no batch prepared (harmless if not: the next dr4sr_sasrec_fwd_bwd prepares again, but a
 * plan->perm counter has then advanced once more). */
int dr4sr_sasrec_fwd_bwd_prepared(const dr4sr_sasrec_plan* plan, void* stream);
int dr4sr_adam_step_prepare_next(const dr4sr_sasrec_plan* plan, void* stream);

/* ABI 7 — the data-parallel step with the gradient in TWO buckets that become final at different times ("optionally
 * 2 buckets ... overlapped with backward"; the reference has no distributed path, /root/reference/utils/callbacks.py:130 is its TODO).
 *   dr4sr_sasrec_grad_buckets(plan, bounds)  -> number of buckets a step of this plan produces, 1 or 2; bounds[0..2] (floats into
 *       plan->grads): bucket 0 = [bounds[0], bounds[1]), bucket 1 = [bounds[1], bounds[2]).  Two buckets exist where the item-table
 *       gradient is a set of jobs of the last backward launch (the at-scale launch forms, dr4sr_sasrec_at_scale bit 0): bucket 0 = the
 *       item table E and the position table P, [0, offsets[2]) — 92 % of a toys-sized replica's bytes; bucket 1 = the encoder layers and
 *       the {n_valid, loss_sum, poison, -} tail.  One bucket otherwise (latency forms: every producer is one launch), bounds[1] = bounds[2].
 *   dr4sr_sasrec_fwd_bwd_phase(plan, prepared, 1, stream)   everything up to and including the launch that makes bucket 0 final
 *       (prepared = 0: runs its own prep like dr4sr_sasrec_fwd_bwd; != 0: on a batch prepared by dr4sr_adam_step_prepare_next);
 *   dr4sr_sasrec_fwd_bwd_phase(plan, prepared, 2, stream)   the remaining weight-gradient launch (nothing when there is one bucket).
 * Caller:  phase 1 ; all-reduce(bucket 0) on a side stream ; phase 2 ; all-reduce(bucket 1) ; join ; dr4sr_adam_step[_prepare_next].
 * Phase 1 + phase 2 leave exactly what dr4sr_sasrec_fwd_bwd[_prepared] leaves (same kernels, same jobs, cut into two launches). */
int dr4sr_sasrec_grad_buckets(const dr4sr_sasrec_plan* plan, int64_t* bounds /* [3] or NULL */);
int dr4sr_sasrec_fwd_bwd_phase(const dr4sr_sasrec_plan* plan, int32_t prepared, int32_t phase, void* stream);

/* ABI 8 — the data-parallel TRANSPORT: RCCL collectives enqueued on the caller's HIP stream (csrc/comm.hip; library links librccl).
 * SURVEY.md section 8(b) names `allreduce_flat(buf)` (RCCL) among the native entry points and 8(e) `ncclAllReduce(sum, fp32)` over the
 * flat gradient buffer {E | P | encoder layers | n_valid, loss_sum, poison, -}; the reference has no distributed path
 * (/root/reference/utils/callbacks.py:130 is its TODO), so these replace no reference line — they are what a maintainer adding DP to
 * model/basemodel.py:193-199 (between `loss.backward()` and `optimizer.step()`) would call.  One communicator per process and GPU:
 *   rank 0:      dr4sr_comm_unique_id(id)        (HOST buffer of DR4SR_COMM_ID_BYTES) ; carry `id` to every rank by any host means
 *   every rank:  dr4sr_comm_init_rank(id, rank, world, device, &comm)                   (collective; hipSetDevice(device) inside)
 *   per step:    dr4sr_sasrec_fwd_bwd(plan, s) ; dr4sr_allreduce_f32(comm, plan->grads, plan->n_params + 4, s) ; dr4sr_adam_step(plan, s)
 * A collective is an enqueue on `stream`, ordered like a kernel launch: no host thread, no host sync, capturable into a hipGraph with the
 * kernels around it.  Two-bucket step (dr4sr_sasrec_fwd_bwd_phase): dr4sr_allreduce_f32_async puts the collective on the communicator's
 * own side stream BEHIND everything already enqueued on `stream`, which does not wait for it — phase 2's launch runs beside the table
 * bucket's all-reduce (inside a capture: a parallel branch of the graph); dr4sr_comm_join(comm, stream) makes `stream` wait for every
 * collective started that way.  Every rank must enter the same collectives in the same order with the same sizes.
 * Errors: DR4SR_E_ARG, a hipError_t (> 0), or DR4SR_E_RCCL_BASE - ncclResult_t; dr4sr_comm_error_string(rc) names any of them.
 * dr4sr_comm_destroy synchronises the side stream and frees the communicator (collective with RCCL: call on every rank). */
#define DR4SR_COMM_ID_BYTES 128
#define DR4SR_RED_SUM 0
#define DR4SR_RED_MAX 1
#define DR4SR_RED_MIN 2
typedef struct dr4sr_comm dr4sr_comm;
int dr4sr_comm_unique_id(void* id_out /* host, DR4SR_COMM_ID_BYTES */);
int dr4sr_comm_init_rank(const void* id /* host */, int32_t rank, int32_t world, int32_t device, dr4sr_comm** out);
int dr4sr_comm_destroy(dr4sr_comm* comm);
int dr4sr_comm_rank(const dr4sr_comm* comm);
int dr4sr_comm_world(const dr4sr_comm* comm);
int dr4sr_comm_async_error(dr4sr_comm* comm);                 /* 0, or the communicator's asynchronous RCCL error as a return code */
const char* dr4sr_comm_error_string(int rc);
int dr4sr_allreduce_f32(dr4sr_comm* comm, float* buf, int64_t n, void* stream);            /* in place, sum */
int dr4sr_allreduce_f64(dr4sr_comm* comm, double* buf, int64_t n, int32_t op /* DR4SR_RED_* */, void* stream);
int dr4sr_allreduce_f32_async(dr4sr_comm* comm, float* buf, int64_t n, void* stream);      /* on the side stream, after `stream`'s work so far */
int dr4sr_comm_join(dr4sr_comm* comm, void* stream);                                       /* `stream` waits for the async collectives */
int dr4sr_allgather_bytes(dr4sr_comm* comm, const void* send, void* recv /* world * bytes */, int64_t bytes, void* stream);
int dr4sr_broadcast_bytes(dr4sr_comm* comm, void* buf, int64_t bytes, int32_t root, void* stream);

/* SASRecQueryEncoder.forward + SeqPoolingLayer (sasrec.py:39-75, layers.py:41-50/:69-73).
 * training != 0 applies dropout (RNG step = state[RNGSTEP]) and keeps activations in the
 * workspace for dr4sr_sasrec_encode_bwd.  out: [B,L,D] (NONE/ORIGIN; NONE leaves rows >= seqlen
 * ZERO as well — they are never computed) or [B,D] (LAST, MEAN). */
int dr4sr_sasrec_encode(const dr4sr_sasrec_plan* plan, int32_t training, int32_t pooling,
                        float* out, void* stream);
/* autograd of the above (same `training` / `pooling` as the forward call it differentiates, no
 * other encode call in between): d_out has the shape of `out`; parameter gradients are ACCUMULATED
 * into plan->grads (exactly d_out-weighted, no normalisation). */
int dr4sr_sasrec_encode_bwd(const dr4sr_sasrec_plan* plan, int32_t training, int32_t pooling,
                            const float* d_out, void* stream);

/* a1: device-side batch selection replacing DataLoader(shuffle=True) + per-sample __getitem__ +
 * default_collate (data/dataset.py:105-108, :149-164).  rows_out[i] = perm[(c*stride + offset + i)
 * mod n_perm] for i < B where c = *counter (device int32), then *counter = c + 1.  With
 * stride = global batch and offset = rank*B every rank walks its own slice of one permutation. */
int dr4sr_select_rows(const int64_t* perm, int64_t n_perm, int64_t* rows_out, int32_t B,
                      int64_t stride, int64_t offset, int32_t* counter, void* stream);

/* K1: item_encoder(idx) + position_emb(arange(L))  (sasrec.py:43-46, :64) for ALL B*L positions,
 * bit-exact with torch (gather + one IEEE add).  out [B,L,D]. */
int dr4sr_embed_gather_posadd(const float* E, const float* P, const int64_t* idx, float* out,
                              int64_t B, int32_t L, int32_t D, int32_t n_items, void* stream);

/* K4: tied-embedding scorer + BCE on a DENSE query (basemodel.py:204-214, loss_func.py:9-38).
 * query [B,L,D]; target, neg [B,L] int64.  Outputs (any may be NULL): pos/neg score [B,L] (pos is
 * -inf where target==0), loss_pos [B,L] = per-position (-logsigmoid(pos)+softplus(neg)) or 0 at
 * pads (UN-normalised), stats[0] = n_valid, stats[1] = loss sum (floats, accumulated: zero them). */
int dr4sr_score_bce_fwd(const float* query, const float* E, const int64_t* target,
                        const int64_t* neg, float* pos_score, float* neg_score, float* loss_pos,
                        float* stats, int64_t B, int32_t L, int32_t D, void* stream);
/* backward of the above with per-position upstream weight w[B,L] (NULL => 1) times *scale
 * (device float, NULL => 1):  d_query [B,L,D] written, dE [N,D] accumulated (atomics). */
int dr4sr_score_bce_bwd(const float* query, const float* E, const int64_t* target,
                        const int64_t* neg, const float* w, const float* scale, float* d_query,
                        float* dE, int64_t B, int32_t L, int32_t D, void* stream);

/* a8: the same two entry points for BPRLoss (loss_func.py:40-48; selected by loss_fn: 'bpr', basemodel.py:103-104): per position
 * -logsigmoid(pos - neg) (K = 1: softmax(ones) = 1), same masks, same outputs, same normalisation contract (stats / upstream w). */
int dr4sr_score_bpr_fwd(const float* query, const float* E, const int64_t* target,
                        const int64_t* neg, float* pos_score, float* neg_score, float* loss_pos,
                        float* stats, int64_t B, int32_t L, int32_t D, void* stream);
int dr4sr_score_bpr_bwd(const float* query, const float* E, const int64_t* target,
                        const int64_t* neg, const float* w, const float* scale, float* d_query,
                        float* dE, int64_t B, int32_t L, int32_t D, void* stream);
/* The loss modules called on SCORE tensors (model/loss_func.py used outside training_step): pos [n] (-inf marks a padded position),
 * neg [n,K].  kind 0: BinaryCrossEntropyLoss.forward masked branch (:12-31) -logsigmoid(pos) + sum_k softplus(neg_k)/K;
 * kind 1: BPRLoss.forward (:44-49) -sum_k logsigmoid(pos - neg_k)/K.  loss_pos [n] per position UN-normalised (0 at padded
 * positions, NULL ok); stats[0] += #valid, stats[1] += loss sum.  Backward: upstream g[n] (NULL = 1) times *scale (NULL = 1). */
int dr4sr_loss_from_scores_fwd(const float* pos, const float* neg, int64_t n, int32_t K, int32_t kind, float* loss_pos,
                               float* stats, void* stream);
int dr4sr_loss_from_scores_bwd(const float* pos, const float* neg, int64_t n, int32_t K, int32_t kind, const float* g,
                               const float* scale, float* d_pos, float* d_neg, void* stream);

/* K0: _neg_sampling (basemodel.py:50-61): uniform on 1..n_items-1 with replacement, never PAD.
 * out [n] int64.  Stream (seed, step) is the one dr4sr_sasrec_fwd_bwd uses for sample_neg=1. */
int dr4sr_neg_sample(int64_t* out, int64_t n, int32_t n_items, uint64_t seed, uint32_t step,
                     void* stream);
/* the same with the step read from a device word at run time (graph replays) */
int dr4sr_neg_sample_dev(int64_t* out, int64_t n, int32_t n_items, uint64_t seed, const int32_t* step_dev, void* stream);

/* BaseModel.topk (basemodel.py:354-365) for the single-domain case: scores = q @ E[:n_items]^T,
 * column 0 (PAD) and every id in hist[b,:] set to -inf, top-k (k <= 128) by score, ties -> lower id.
 * q [B,D], hist [B,Lh] int64, out_score [B,k] fp32, out_item [B,k] int64. */
/* *bad_count += number of ids outside [0, n_items) among idx[0..n) (device int32, zeroed by the caller).  The gather kernels CLAMP
 * such ids (a launch cannot raise); torch's nn.Embedding — the reference's gather, model/sasrec.py:43 — raises "index out of range in
 * self": a caller that wants that behaviour checks its id tensors with this (dr4sr_amd: once per dataset tensor, and per call of the
 * dense dispatcher op). */
int dr4sr_check_ids(const int64_t* idx, int64_t n, int32_t n_items, int32_t* bad_count, void* stream);

int dr4sr_full_score_topk(const float* q, const float* E, const int64_t* hist, float* out_score,
                          int64_t* out_item, int64_t B, int32_t D, int32_t n_items, int32_t Lh,
                          int32_t k, void* stream);
/* the same through a [B, round_up(n_items, 64)] fp32 score workspace (dr4sr_full_score_topk_workspace_bytes): the scores come from
 * one MFMA GEMM instead of B passes over the table and the top-k from a radix select — identical results, ~15x faster at the
 * reference's eval shape (2048 x 11925, k = 100); no limit on n_items (rows that do not fit LDS are selected from the workspace, whose
 * history columns are then overwritten with -inf). */
int64_t dr4sr_full_score_topk_workspace_bytes(int64_t B, int32_t n_items);
int dr4sr_full_score_topk_ws(const float* q, const float* E, const int64_t* hist, float* out_score, int64_t* out_item, int64_t B,
                             int32_t D, int32_t n_items, int32_t Lh, int32_t k, float* workspace, int64_t workspace_bytes,
                             void* stream);

/* the multi-domain form of basemodel.py:354-365: item_blocked [n_items] bytes, 1 = the item is NOT in domain_item_mapping[eval_domain]
 * (its score becomes -inf exactly like the reference's domain_mask, :358-360); NULL = single domain (only PAD is blocked). */
int dr4sr_full_score_topk_masked_ws(const float* q, const float* E, const int64_t* hist, const uint8_t* item_blocked,
                                    float* out_score, int64_t* out_item, int64_t B, int32_t D, int32_t n_items, int32_t Lh,
                                    int32_t k, float* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FMLP (model/fmlp.py:18-39, module/layers.py:740-807): embedding + position -> LayerNorm -> dropout ->
 * n_layer x { spectral FilterLayer (rfft * complex weight -> irfft, 'ortho') -> dropout -> +x -> LayerNorm ;
 * Intermediate (64 -> 256 GELU -> 64, dropout, +x, LayerNorm) } -> x[:, -1].  Rows are LEFT-padded prefixes with a SCALAR
 * target (dataset/dataset_transform.ipynb), all B*L positions are computed (the filter mixes every position).
 * The reference hard-codes L = 50, D = 64, hidden 256, dropout 0.5 (fmlp.py:11-13, layers.py:743-744,:762); the kernels
 * require D = 64, F = 256, L <= 50... any L <= 50 even.
 * Flat parameter layout: E[N,D] | P[L,D] | ln_w[D] ln_b[D] | per layer: complex_weight[L/2+1, D, 2] | filt_ln_w filt_ln_b |
 *   dense_1.w[F,D] dense_1.b[F] dense_2.w[D,F] dense_2.b[D] | inter_ln_w inter_ln_b      (+ DR4SR_GRAD_TAIL on grads) */
typedef struct dr4sr_fmlp_plan {
    int32_t abi_version;
    int32_t B, L, D, F, n_layer, n_items;
    float   ln_eps, p_drop;
    uint64_t seed;
    float*  params; float* grads; float* adam_m; float* adam_v;
    int64_t n_params;
    const int64_t* in_item_id;      /* [U,L] left-padded prefixes                                  */
    const int64_t* item_id;         /* [U]   scalar targets (may be NULL for encode)               */
    const int64_t* rows;            /* [B] or NULL                                                  */
    int64_t* neg_item;              /* [B]   one negative per row                                   */
    int32_t  sample_neg;
    void*    workspace; int64_t workspace_bytes;
    int32_t* state;                 /* [DR4SR_STATE_WORDS]                                          */
    float lr, beta1, beta2, adam_eps, weight_decay;
    int32_t optimizer;              /* DR4SR_OPT_* (ABI 6)                                          */
    /* ---- fused batch selection + per-step loss log (ABI 6; dr4sr_sasrec_plan's contract): perm != NULL -> the step's first kernel FILLS
     *      rows[i] = perm[(c * perm_stride + perm_offset + i) mod n_perm], c = *perm_counter, and bumps the counter;
     *      dr4sr_fmlp_train_step / _adam_step write loss_log[c] = loss_sum / n_valid ---- */
    const int64_t* perm; int64_t n_perm; int64_t perm_stride; int64_t perm_offset; int32_t* perm_counter;
    float* loss_log;
} dr4sr_fmlp_plan;

int     dr4sr_fmlp_plan_sizeof(void);
/* offsets[0]=E [1]=P [2]=ln_w [3]=ln_b, [4+9*i+j] = j-th tensor of layer i in the order above; returns n_params */
int64_t dr4sr_fmlp_param_layout(int32_t n_items, int32_t L, int32_t D, int32_t F, int32_t n_layer, int64_t* offsets);
/* bytes, or DR4SR_E_SHAPE unless D = 64, F = 256, L even and <= 50 */
int64_t dr4sr_fmlp_workspace_bytes(const dr4sr_fmlp_plan* plan);
/* basemodel.py:193-198 for model = FMLP: negatives, forward, scorer + BCE (1-D targets), backward; un-normalised grads */
int dr4sr_fmlp_fwd_bwd(const dr4sr_fmlp_plan* plan, void* stream);
/* fwd_bwd + dense Adam */
int dr4sr_fmlp_train_step(const dr4sr_fmlp_plan* plan, void* stream);
/* the optimizer half on the plan's buffers (plan->optimizer, loss_log): data parallel = fwd_bwd, all-reduce of plan->grads, this */
int dr4sr_fmlp_adam_step(const dr4sr_fmlp_plan* plan, void* stream);
/* FMLP.forward -> out [B,D] (= encoder output at the last position); training != 0 applies dropout */
int dr4sr_fmlp_encode(const dr4sr_fmlp_plan* plan, int32_t training, float* out, void* stream);
/* autograd of dr4sr_fmlp_encode: d_out [B,D]; parameter gradients ACCUMULATE into plan->grads */
int dr4sr_fmlp_encode_bwd(const dr4sr_fmlp_plan* plan, int32_t training, const float* d_out, void* stream);
/* torch.optim.Adam on arbitrary flat buffers (grads[n] = normaliser, as dr4sr_adam_step); state[STEP] is bumped */
int dr4sr_adam_flat(float* params, const float* grads, float* adam_m, float* adam_v, int64_t n, int32_t* state,
                    float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* the same for any DR4SR_OPT_* (basemodel.py:79-98) */
int dr4sr_optimizer_flat(int32_t optimizer, float* params, const float* grads, float* adam_m, float* adam_v, int64_t n, int32_t* state,
                         float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GRU4Rec (model/gru4rec.py:12-34; module/layers.py:117-136 = torch.nn.GRU(bias=False, batch_first, n_layer) + Linear(H->D)):
 * x = dropout(E[idx]) -> GRU -> Linear -> 'origin'/'last' pooling -> scorer + BCE as SASRec.  D = 64, H in {128, 256},
 * n_layer <= 4, L <= 64.  Only rows < seqlen are computed (post-padded rows; the recurrence is causal).
 * Flat parameter layout: E[N,D] | per layer: weight_ih_l[3H,in] weight_hh_l[3H,H] | out_w[D,H] out_b[D]
 *   (names: item_embedding.weight == query_encoder.0.1.weight, query_encoder.0.3.gru.weight_{ih,hh}_l{l},
 *    query_encoder.1.{weight,bias}; gates ordered r|z|n as torch).  Optimizer: dr4sr_adam_flat (weight_decay 1e-4 in
 *   configs/gru4rec.yaml is torch.optim.Adam's L2 form). */
typedef struct dr4sr_gru4rec_plan {
    int32_t abi_version;
    int32_t B, L, D, H, n_layer, n_items;
    float   p_drop;                 /* dropout on the item embeddings (configs/gru4rec.yaml: 0.2) */
    uint64_t seed;
    float*  params; float* grads; float* adam_m; float* adam_v;
    int64_t n_params;
    const int64_t* in_item_id;      /* [U,L] */
    const int64_t* item_id;         /* [U,L] targets (may be NULL for encode) */
    const int64_t* seqlen;          /* [U]   */
    const int64_t* rows;            /* [B] or NULL */
    int64_t* neg_item;              /* [B,L] */
    int32_t  sample_neg;
    void*    workspace; int64_t workspace_bytes;
    int32_t* state;
    float lr, beta1, beta2, adam_eps, weight_decay;
    int32_t optimizer;              /* DR4SR_OPT_* (ABI 6) */
    /* ---- fused batch selection + per-step loss log (ABI 6; same contract as dr4sr_sasrec_plan's perm / loss_log fields): when
     *      perm != NULL the step's first kernel FILLS rows[i] = perm[(c * perm_stride + perm_offset + i) mod n_perm] with
     *      c = *perm_counter and bumps the counter; dr4sr_gru4rec_train_step / _adam_step write loss_log[c] = loss_sum / n_valid ---- */
    const int64_t* perm; int64_t n_perm; int64_t perm_stride; int64_t perm_offset; int32_t* perm_counter;
    float* loss_log;
} dr4sr_gru4rec_plan;

int     dr4sr_gru4rec_plan_sizeof(void);
/* offsets[0]=E, [1+2l]=weight_ih_l, [2+2l]=weight_hh_l, [1+2n]=out_w, [2+2n]=out_b; returns n_params */
int64_t dr4sr_gru4rec_param_layout(int32_t n_items, int32_t D, int32_t H, int32_t n_layer, int64_t* offsets);
/* The workspace must be ZERO-initialised once by the caller (hipMemset at allocation): for small batches (8*ceil(B/16) <= 192
 * workgroups) the recurrences run cooperatively over 8 CUs per 16 sequences and keep their exchange granules and a launch counter
 * there (csrc/gru_coop.hip); `DR4SR_GRU_NOCOOP=1` forces the single-workgroup recurrence.  The waits of that exchange are bounded:
 * a wait that runs out never hangs the GPU but sets a sticky error flag — the int32 word 2 of the workspace (words 0, 1: launch
 * counter, finish ticket) — and leaves garbage; the caller should read that word whenever it synchronises anyway (per epoch) and
 * treat non-zero as a failed run. */
/* bytes, or DR4SR_E_SHAPE unless D = 64, H in {128, 256}, L <= 64 */
int64_t dr4sr_gru4rec_workspace_bytes(const dr4sr_gru4rec_plan* plan);
/* the optimizer half of dr4sr_gru4rec_train_step on the plan's buffers (plan->optimizer, loss_log): data parallel = fwd_bwd,
 * all-reduce of plan->grads, this */
int dr4sr_gru4rec_adam_step(const dr4sr_gru4rec_plan* plan, void* stream);
/* n consecutive training steps, one prep launch: every optimizer launch but the last prepares the following step (see
 * dr4sr_sasrec_train_steps) */
int dr4sr_gru4rec_train_steps(const dr4sr_gru4rec_plan* plan, int32_t n_steps, void* stream);
/* 1 when a batch of B sequences takes the cooperative multi-CU recurrence on the CURRENT device (its 8*ceil(B/16) workgroups must be
 * co-resident: the budget is 3/4 of the device's compute units, at most 192, and 0 under DR4SR_GRU_NOCOOP), 0 for the
 * single-workgroup recurrence.  A timeout of the cooperative exchange also POISONS the step: the gradient tail word grads[n_params+2]
 * becomes non-zero and dr4sr_adam_flat / dr4sr_gru4rec_train_step then leave parameters, moments and the step counter untouched
 * (under data parallelism the all-reduced tail poisons every replica alike). */
int dr4sr_gru4rec_uses_cooperative(int32_t B, int32_t H);
/* 1 when a two-layer plan runs BOTH layers' recurrences in one launch, the second layer one time step behind the first (layer
 * wavefront, csrc/gru_coop.hip: n_layer == 2, H == 256, batches the 16-slice cooperative form takes; 0 under DR4SR_GRU_NOWAVE) */
int dr4sr_gru4rec_uses_wavefront(int32_t B, int32_t H, int32_t n_layer, int32_t L);
int dr4sr_gru4rec_fwd_bwd(const dr4sr_gru4rec_plan* plan, void* stream);       /* basemodel.py:193-198, un-normalised grads */
int dr4sr_gru4rec_train_step(const dr4sr_gru4rec_plan* plan, void* stream);    /* + dense Adam */
int dr4sr_gru4rec_encode(const dr4sr_gru4rec_plan* plan, int32_t training, int32_t pooling, float* out, void* stream);
int dr4sr_gru4rec_encode_bwd(const dr4sr_gru4rec_plan* plan, int32_t training, int32_t pooling, const float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DR4SR+ MetaModel (model/metamodel.py:19-197, utils/utils.py:134-252).
 *
 * The meta module is nn.Sequential(Linear(D,D), ReLU, Linear(D,2)) (metamodel.py:52-57); phi is its flat parameter vector
 * W1[D,D] | b1[D] | W2[2,D] | b2[2] (dr4sr_meta_param_count floats; D = 64).
 *
 * dr4sr_meta_select_fwd — MetaModel.selection + the two masks of training_step (metamodel.py:169-185):
 *   weight[p] = softmax((meta_module(query[p]) + gumbel[p]) / tau)[0];  1 where user_id[p / L] == 0;  0 where target[p] == 0.
 *   n = B*L positions (L = 1 for scalar-target models).  gumbel [n,2] explicit noise, or NULL: drawn in-kernel from Philox
 *   (seed, step) as -log(-log(u)) like F.gumbel_softmax; step_dev != NULL: the step is *step_dev (a device word such as the plan's
 *   state[RNGSTEP], so that a captured graph draws fresh noise on every replay).  tau = clip(tau, tau_min) is passed by the caller.
 *   gate_in  [n] (NULL = off): a FROZEN ReLU pattern (bit j = unit j active) used instead of (pre > 0);
 *   gate_out [n] (NULL = off): the pattern this call used.
 * dr4sr_meta_select_bwd — backward of the above for upstream d_weight[n] (times *scale if scale != NULL):
 *   d_query [n,D] is ACCUMULATED (+=; NULL = skip), d_phi ACCUMULATED, deterministically (per-block partials in `workspace`,
 *   dr4sr_meta_select_workspace_floats(n) floats, summed in a fixed order). */
int64_t dr4sr_meta_param_count(int32_t D);
int64_t dr4sr_meta_select_workspace_floats(int64_t n);
int dr4sr_meta_select_fwd(const float* query, const float* phi, const float* gumbel, uint64_t seed, uint32_t step,
                          const int32_t* step_dev, float tau,
                          const int64_t* user_id, const int64_t* target, int64_t B, int32_t L, int32_t D,
                          const uint64_t* gate_in, uint64_t* gate_out, float* weight, void* stream);
int dr4sr_meta_select_bwd(const float* query, const float* phi, const float* gumbel, uint64_t seed, uint32_t step,
                          const int32_t* step_dev, float tau,
                          const int64_t* user_id, const int64_t* target, int64_t B, int32_t L, int32_t D,
                          const uint64_t* gate_in, const float* d_weight, const float* scale, float* d_query, float* d_phi,
                          float* workspace, void* stream);

/* MetaModel's weighted loss fused into the SASRec training step (the fast path of dr4sr_amd/model/metamodel.py): the per-token
 * scorer computes weight_t = selection(z_t; phi) itself (same masks, noise keyed by (plan->seed, state[RNGSTEP], b*L+pos) exactly
 * like dr4sr_meta_select_fwd with step_dev = &state[RNGSTEP]) and back-propagates sum_t weight_t loss_t including
 * loss_t * d weight_t / d z_t.  grads tail = {n_valid, sum_t weight_t loss_t}.  d phi is NOT produced (the inner step never uses it;
 * the hyper-gradient takes it from dr4sr_meta_select_bwd).  gate_in / gate_out / weight_out are indexed by PACKED token
 * (workspace order: sequence by sequence, valid positions only), [B*L] words are enough.  D = 64 only. */
typedef struct dr4sr_meta_weighting {
    const float*   phi;             /* flat meta parameters (dr4sr_meta_param_count floats)                      */
    const float*   gumbel;          /* [B,L,2] explicit noise or NULL (Philox)                                   */
    const int64_t* user_id;         /* [U] addressed like the other dataset tensors (through plan->rows), or NULL */
    const uint64_t* gate_in;        /* frozen ReLU pattern per packed token, or NULL                             */
    uint64_t*      gate_out;        /* pattern used, per packed token, or NULL                                   */
    float*         weight_out;      /* weight per packed token, or NULL                                          */
    float          tau;             /* clip(tau, tau_min)                                                        */
} dr4sr_meta_weighting;
int dr4sr_sasrec_fwd_bwd_weighted(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, void* stream);
/* the same on a batch already prepared by dr4sr_adam_step_prepare_next (k-step graphs of the MetaModel inner loop) */
int dr4sr_sasrec_fwd_bwd_weighted_prepared(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, void* stream);

/* Hypergrad.grad (utils/utils.py:145-205) from FIRST-ORDER gradients: with G(W) = dL_train/dW (this library's backward),
 *     H v            = [G(W + e v) - G(W - e v)] / 2e                      (Neumann terms; scaled by hpo_lr, :196-203)
 *     d/dphi (G . p) = [dL_train/dphi(W + e p) - dL_train/dphi(W - e p)] / 2e   (:170-175)
 * evaluated with identical dropout masks / negatives / Gumbel noise and the meta module's ReLU pattern frozen at W (autograd's
 * second derivative of ReLU is 0).  All scalars stay on the device.
 *   step_size: *out_e = rel_step * sqrt(sum_{dir!=0} theta^2 / sum dir^2)
 *   shift    : out = x + sign * (*e) * dir
 *   neumann  : v -= lr * (gp / *np - gm / *nm) / (2 * *e) ;  pacc += v          (gp, gm un-normalised, np/nm their n_valid)
 *   diff     : out = coef * (fp / *np - fm / *nm) / (2 * *e)
 *   diff4    : out = coef * (4 D(e) - D(2e)) / 3 with D(h) = (f(+h)/n - f(-h)/n) / 2h from the probes at +-e (fp1, fm1) and +-2e
 *              (fp2, fm2), nv4 = their four n_valid words {+e, -e, +2e, -2e}: Richardson extrapolation, truncation error O(e^4).
 *              The mixed term is the whole hyper-gradient up to O(hpo_lr); at TRAINED weights (the reference's shipped checkpoint)
 *              its plain central difference sits at 1.0e-3 of the reference's double-backward, the extrapolated one at 1e-5.
 *   scale_by : out = x / *den */
int dr4sr_fd_step_size(const float* theta, const float* dir, int64_t n, float rel_step, float* out_e, void* stream);
/* the same with a caller-owned reduction scratch (dr4sr_fd_step_size_scratch_floats() floats, zeroed once): re-entrant, which the
 * form above — it reduces through a module-level scratch — is not */
int64_t dr4sr_fd_step_size_scratch_floats(void);
int dr4sr_fd_step_size_ws(const float* theta, const float* dir, int64_t n, float rel_step, float* out_e, float* scratch, void* stream);
int dr4sr_fd_shift(float* out, const float* x, const float* dir, const float* e, float sign, int64_t n, void* stream);
int dr4sr_fd_neumann(float* v, float* pacc, const float* gp, const float* gm, const float* np, const float* nm, const float* e,
                     float lr, int64_t n, void* stream);
int dr4sr_fd_diff(float* out, const float* fp, const float* fm, const float* np, const float* nm, const float* e, float coef,
                  int64_t n, void* stream);
int dr4sr_fd_diff4(float* out, const float* fp1, const float* fm1, const float* fp2, const float* fm2, const float* nv4,
                   const float* e, float coef, int64_t n, void* stream);
int dr4sr_scale_by(float* out, const float* x, const float* den, int64_t n, void* stream);

/* MetaOptimizer.step tail (utils/utils.py:240-247) for the reference's default meta optimizer (metamodel.py:68-69):
 * clip_grad_norm_(max_norm; <= 0 disables) then torch.optim.SGD(lr, momentum, weight_decay) on phi[n].
 * step_count (device int32) selects the first-step momentum initialisation and is incremented; out_norm (NULL ok) = |grad|. */
int dr4sr_meta_sgd_step(float* phi, const float* grad, float* momentum_buf, int32_t n, float lr, float momentum,
                        float weight_decay, float max_norm, int32_t* step_count, float* out_norm, void* stream);
/* ABI 7 — the same tail for the reference's other `meta_optimizer` choices (metamodel.py:59-81): optimizer = DR4SR_OPT_ADAM
 * ('adam': torch.optim.Adam(lr), weight_decay 0; any unknown name: Adam(lr, weight_decay = meta_weight_decay)), DR4SR_OPT_ADAGRAD
 * (Adagrad(lr): eps 1e-10, state_v = state_sum), DR4SR_OPT_RMSPROP (RMSprop(lr): beta2 = alpha 0.99, eps 1e-8, state_v = square_avg);
 * state_m / state_v [n] zero-initialised by the caller; step_count = steps taken so far (bias corrections use step_count + 1).
 * `tau` sits in the reference's parameter list but never receives a gradient (it is not in aux_params, metamodel.py:142), so torch
 * skips it: nothing to step.  'sparse_adam' raises in the reference's first step (dense gradients) — the binding raises the same error. */
int dr4sr_meta_opt_step(int32_t optimizer, float* phi, const float* grad, float* state_m, float* state_v, int32_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, float max_norm, int32_t* step_count,
                        float* out_norm, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CL4SRec (model/cl4srec.py, module/data_augmentation.py:20-95,:305-350,:577-619): two augmented views of every sequence are
 * encoded by the SAME SASRec encoder (dr4sr_sasrec_encode with DR4SR_POOL_MEAN, one workspace per view), InfoNCE between the views.
 *
 * dr4sr_cl_augment — Item_Crop (mode 0: contiguous max(1, int(tau n)) items, left-aligned, out_len = that), Item_Mask (mode 1:
 *   int(gamma n) distinct positions -> mask_id), Item_Reorder (mode 2: a contiguous int(beta n) segment shuffled), Item_Random
 *   (mode 3: one of the three, drawn once per call).  seq/out [B,L] int64 (L <= 64), seqlen/out_len [B].  Philox (seed, step):
 *   same distributions as the reference's torch/numpy/random draws, not the same streams.
 * dr4sr_infonce_fwd — InfoNCELoss('inner_product', 'batch_both'): logits[i] = [x_i.x_j^T | x_i.x_i^T, diagonal -inf] / temperature,
 *   cross-entropy with label i.  valid[B] (uint8, NULL = all): rows with 0 are removed from rows and columns (the reference drops
 *   sequences of length 1, data_augmentation.py:613-615).  Outputs lse[B], loss_row[B] (0 at invalid rows), stats[0] += #valid rows,
 *   stats[1] += sum loss_row  (reduce=True loss = stats[1] / stats[0]; reduce=False = loss_row / #valid).
 * dr4sr_infonce_bwd — d(sum loss_row) * (*scale) accumulated into dxi, dxj [B,D]. */
int dr4sr_cl_augment(const int64_t* seq, const int64_t* seqlen, int64_t* out, int64_t* out_len, int32_t B, int32_t L, int32_t mode,
                     double tau, double gamma, double beta, int64_t mask_id, uint64_t seed, uint32_t step, void* stream);
/* the same with step = *step_dev + step_offset read at run time (graph replays of a training step) */
int dr4sr_cl_augment_dev(const int64_t* seq, const int64_t* seqlen, int64_t* out, int64_t* out_len, int32_t B, int32_t L, int32_t mode,
                         double tau, double gamma, double beta, int64_t mask_id, uint64_t seed, const int32_t* step_dev,
                         uint32_t step_offset, void* stream);
/* two views in one launch: what two consecutive dr4sr_cl_augment_dev calls (step_offset, step_offset + 1) produce */
int dr4sr_cl_augment2_dev(const int64_t* seq, const int64_t* seqlen, int64_t* out_i, int64_t* len_i, int64_t* out_j, int64_t* len_j,
                          int32_t B, int32_t L, int32_t mode, double tau, double gamma, double beta, int64_t mask_id, uint64_t seed,
                          const int32_t* step_dev, uint32_t step_offset, void* stream);
/* both views of the batch rows[0..B) of dataset tensors seq [U,L] / seqlen [U] (the batch a captured step selected on the device:
 * dr4sr_sasrec_plan.perm): the draws of dr4sr_cl_augment2_dev on the gathered rows */
int dr4sr_cl_augment2_rows_dev(const int64_t* seq, const int64_t* seqlen, const int64_t* rows, int64_t* out_i, int64_t* len_i,
                               int64_t* out_j, int64_t* len_j, int32_t B, int32_t L, int32_t mode, double tau, double gamma, double beta,
                               int64_t mask_id, uint64_t seed, const int32_t* step_dev, uint32_t step_offset, void* stream);
/* round 4: dr4sr_cl_prepare_rows that also advances the augmentation's device call counter by step_add (module/data_augmentation.py
 * end_step()), and dr4sr_infonce_bwd with the step's scalars computed inside: backward scale = cl_weight * tail[0] / stats[0], and with
 * fold_tail != 0 the contrastive term's share of the reported loss folded into tail[1] (what dr4sr_cl_scalars_dp in front of it gives) —
 * two launches less per CL4SRec step (model/cl4srec.py:_cl_term) */
int dr4sr_cl_prepare_rows_step(const int64_t* seqlen, const int64_t* rows, int32_t B, uint8_t* valid, float* stats, float* zero,
                               int64_t nzero, int32_t* step_dev, int32_t step_add, void* stream);
int dr4sr_infonce_bwd_scaled(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                             const float* lse, float* tail, const float* stats, float cl_weight, int32_t fold_tail,
                             float* dxi, float* dxj, void* stream);
/* glue of a CL4SRec step composed without autograd (model/cl4srec.py:49-73 = BCE + cl_weight * InfoNCE):
 *   dr4sr_cl_prepare: valid[b] = seqlen[b] != 1 (data_augmentation.py:613-615), stats[0..1] = 0, zero[0..nzero) = 0;
 *   dr4sr_cl_scalars: with {n_valid, loss_sum} of the main pass in `tail` and InfoNCE's {rows, loss_sum} in `stats`:
 *     *scale_out = cl_weight * n_valid / rows (the InfoNCE backward scale under an optimizer that divides by n_valid),
 *     *loss_out = loss_sum / n_valid + cl_weight * stats[1] / rows;  either output may be NULL. */
int dr4sr_cl_prepare(const int64_t* seqlen, int32_t B, uint8_t* valid, float* stats, float* zero, int64_t nzero, void* stream);
int dr4sr_cl_prepare_rows(const int64_t* seqlen, const int64_t* rows, int32_t B, uint8_t* valid, float* stats, float* zero, int64_t nzero,
                          void* stream);      /* valid[b] = seqlen[rows[b]] != 1 */
int dr4sr_cl_scalars(const float* tail, const float* stats, float cl_weight, float* scale_out, float* loss_out, void* stream);
/* the same when n_valid is spread over n_parts words nv_parts[r * stride] (data parallel: every rank's count, all-gathered next to its
 * pooled views — InfoNCE's negatives are the GLOBAL batch; a single part = the local tail itself):
 *   *scale_out = cl_weight * sum_r nv_parts[r * stride] / rows;
 *   tail_local[1] += cl_weight * stats[1] / rows * tail_local[0]   (the contrastive term's share of the reported loss, so that the
 *   all-reduced tail[1] / tail[0] = BCE mean + cl_weight * InfoNCE mean).  Either output may be NULL. */
int dr4sr_cl_scalars_dp(const float* nv_parts, int32_t n_parts, int64_t stride, const float* stats, float cl_weight, float* scale_out,
                        float* tail_local, void* stream);
int dr4sr_infonce_fwd(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature, float* lse,
                      float* loss_row, float* stats, void* stream);
int dr4sr_infonce_bwd(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                      const float* lse, const float* scale, float* dxi, float* dxj, void* stream);

/* Test / measurement hooks (dr4sr_dropout_mask, dr4sr_*_launch_kernel) are NOT part of this product surface: they are declared in
 * include/dr4sr_hip_hooks.h, and nothing under dr4sr_amd/ calls them. */

#ifdef __cplusplus
}
#endif
#endif /* DR4SR_HIP_H */
