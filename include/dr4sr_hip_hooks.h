/* dr4sr_hip_hooks.h — TEST / MEASUREMENT hooks of libdr4sr_hip.so.
 *
 * Kept apart from the product ABI (include/dr4sr_hip.h): nothing under dr4sr_amd/ calls these.  Users: tests/ (the oracle runs with
 * the library's own dropout masks) and bench.py / tools/ (ONE kernel of a training step enqueued on the state the last fwd_bwd left
 * in the workspace, so that its launch duration can be bracketed with HIP events on the caller's stream — the `roofline` object).
 */
#ifndef DR4SR_HIP_HOOKS_H
#define DR4SR_HIP_HOOKS_H
#include "dr4sr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Materialise the keep-mask (1.0 / 0.0) the kernels use for (seed, step, site) over n elements
 * (n multiple of 4).  Test hook: lets the oracle run with the library's exact dropout masks. */
int dr4sr_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint32_t step, uint32_t site,
                       void* stream);

/* SASRec step (model/sasrec.py:39-75 under model/basemodel.py:193-199): kernel ids of dr4sr_sasrec_launch_kernel */
#define DR4SR_K_PREP       0
#define DR4SR_K_EMBED_FWD  1
#define DR4SR_K_QKV_FWD    2
#define DR4SR_K_ATTN_FWD   3
#define DR4SR_K_POST_FWD   4
#define DR4SR_K_SCORE      5
#define DR4SR_K_TRANSPOSE  6
#define DR4SR_K_POST_BWD   7
#define DR4SR_K_ATTN_BWD   8
#define DR4SR_K_QKV_BWD    9
#define DR4SR_K_EMBED_BWD  10
#define DR4SR_K_WGRAD      11
#define DR4SR_K_ADAM       12
#define DR4SR_K_ZERO_GRADS 13
/* launches of the fused step (dr4sr_sasrec_train_step): gather + qkv of layer 0; post_fwd + scorer + post_bwd of the last layer;
 * qkv backward of layer 0 + table scatter; weight gradients with the step's extra planes / jobs.  (DR4SR_K_POST_FWD / _BWD with a
 * layer below the last one already are the fused forms: they carry the next layer's qkv projection / its backward.) */
#define DR4SR_K_EMBQKV_FWD     14
#define DR4SR_K_POST_MID       15
#define DR4SR_K_QKV_EMBED_BWD  16
#define DR4SR_K_WGRAD_FUSED    17
int dr4sr_sasrec_launch_kernel(const dr4sr_sasrec_plan* plan, int32_t kernel, int32_t layer, void* stream);

/* The DR4SR_* environment switches (cross-checks and tuning knobs) are read ONCE per process — each site caches its
 * value — and re-read after this call: a test can flip a switch, call dr4sr_reload_env(), and reach the other launch form in the SAME
 * process.  Returns the new generation number.  Not for production use: graphs captured before the call keep their launch forms, and
 * a workspace sized under one setting must not be used under another (DR4SR_BM) — build the engine after the reload. */
int dr4sr_reload_env(void);

/* bit 0: the library was built with -DDR4SR_EXPERIMENTS (`make -C dr4sr_amd/csrc EXPERIMENTS=1` -> libdr4sr_hip_exp.so; load it through
 * DR4SR_LIB_PATH): the experiment / tuning switches of SWITCHES.md's second table are read.  0: the shipped build — they are compile-time
 * constants and only the cross-check switches of the first table exist (tests of experiment switches skip themselves). */
int dr4sr_build_flags(void);

/* the same with the MetaModel weighting (model/metamodel.py:174-194) on the launches that carry it (DR4SR_K_POST_MID); mw may be NULL */
int dr4sr_sasrec_launch_kernel_weighted(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, int32_t kernel, int32_t layer,
                                        void* stream);

/* GRU4Rec step (model/gru4rec.py:12-34, module/layers.py:117-136): the recurrence of one layer (forward: h_t from gi and W_hh;
 * backward: BPTT from the saved gates) and the input GEMM gi = in W_ih^T */
#define DR4SR_GK_REC_FWD  0
#define DR4SR_GK_REC_BWD  1
#define DR4SR_GK_GEMM_IN  2
#define DR4SR_GK_WAVE_FWD 3   /* both layers' recurrences as ONE launch (two-layer plans the wavefront takes; `layer` ignored) */
#define DR4SR_GK_WAVE_BWD 4
int dr4sr_gru4rec_launch_kernel(const dr4sr_gru4rec_plan* plan, int32_t kernel, int32_t layer, void* stream);

/* FMLP step (model/fmlp.py:18-39, module/layers.py:740-807): the filter layer (FFT-free circular convolution with the learned
 * complex weight, LayerNorm, dropout) and the Intermediate block (linear 64 -> 256, GELU, linear 256 -> 64, dropout, LayerNorm) of one
 * layer, forward and backward, and the weight-gradient launch of the step (`layer` ignored) */
#define DR4SR_FK_FILTER_FWD 0
#define DR4SR_FK_FFN_FWD    1
#define DR4SR_FK_FFN_BWD    2
#define DR4SR_FK_FILTER_BWD 3
#define DR4SR_FK_WGRAD      4
int dr4sr_fmlp_launch_kernel(const dr4sr_fmlp_plan* plan, int32_t kernel, int32_t layer, void* stream);

/* Measurement hook of bench.py (the driver's contract is ONE JSON line on stdout): keep a COMPLETE line ready, and if the process is
 * then killed by a fatal signal (SIGABRT from a foreign thread's uncaught exception, SIGSEGV / SIGBUS / SIGFPE / SIGILL, or the SIGTERM
 * a launcher sends to the surviving ranks when another rank died) write it to `fd` and _exit(exit_code) from the signal handler —
 * write(2) and _exit(2) only.  line = NULL disarms and restores the previous handlers.  The line is copied.  Returns 0, or DR4SR_E_ARG. */
int dr4sr_crash_line_set(const char* line, int32_t fd, int32_t exit_code);

#ifdef __cplusplus
}
#endif
#endif /* DR4SR_HIP_HOOKS_H */
