// step.hip — host-side orchestration behind the C ABI: parameter layout, workspace carve-up, the
// fused dense Adam kernel, and the launch sequences of fwd_bwd / encode / encode_bwd.
// Every function only enqueues kernels (and one memset) on the caller's stream: graph-capturable.
//
// Reference control flow being replaced: model/basemodel.py:193-199 (one iteration of training_epoch).
#include "common.h"
#include "kernels.h"

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
void big_lds_impl(const void* kernel, size_t bytes) {
    static std::unordered_map<const void*, size_t> granted;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    size_t& g = granted[kernel];
    if (bytes > g) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        g = bytes;
    }
}

extern "C" int dr4sr_abi_version(void) { return DR4SR_ABI_VERSION; }

// ---- side streams (common.h StepFork).  Created on the first call that asks for them (DR4SR_STREAMS).
static StepFork g_fork = {};
const StepFork& step_fork() {
    if (g_fork.state == 0 && DR4SR_XENV("DR4SR_STREAMS")) {
        // A first call may come from INSIDE a stream capture (the callers warm a step up first, but nothing forces them to): stream /
        // event creation is made legal there by switching this thread to the relaxed capture mode for the duration.  Whatever still
        // fails leaves the state at 0 — side streams off for THIS call only, retried by the next — and nothing half-created behind.
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        bool ok = true;
        int ns = 0, nf = 0, nj = 0;
        for (; ns < StepFork::NSIDE && ok; ++ns) ok = hipStreamCreateWithFlags(&g_fork.side_[ns], hipStreamNonBlocking) == hipSuccess;
        for (; nf < StepFork::NSIDE && ok; ++nf) ok = hipEventCreateWithFlags(&g_fork.fork_ev[nf], hipEventDisableTiming) == hipSuccess;
        for (; nj < StepFork::NSIDE && ok; ++nj) ok = hipEventCreateWithFlags(&g_fork.join_ev[nj], hipEventDisableTiming) == hipSuccess;
        (void)hipThreadExchangeStreamCaptureMode(&mode);                        // back to the caller's mode
        if (!ok) {                                                              // partial creation: destroy what exists, stay retryable
            (void)hipGetLastError();
            for (int i = 0; i < ns; ++i) if (g_fork.side_[i]) { (void)hipStreamDestroy(g_fork.side_[i]); g_fork.side_[i] = nullptr; }
            for (int i = 0; i < nf; ++i) if (g_fork.fork_ev[i]) { (void)hipEventDestroy(g_fork.fork_ev[i]); g_fork.fork_ev[i] = nullptr; }
            for (int i = 0; i < nj; ++i) if (g_fork.join_ev[i]) { (void)hipEventDestroy(g_fork.join_ev[i]); g_fork.join_ev[i] = nullptr; }
            (void)hipGetLastError();
        } else g_fork.state = 1;
    }
    return g_fork;
}
// OPT-IN (DR4SR_STREAMS=1): measured slower on this stack at every size — a fork + join through a captured graph costs ~30 us of cross-queue
// signalling for two branches (tools/probes/graph_branch_probe.py: two 85 us kernels side by side take 115 us, four 141 us), more than the
// 10-25 us launches it overlaps; B = 256: 0.1245 -> 0.1443 ms, B = 8 192: 0.523 -> 0.576 ms, B = 131 072: 5.77 -> 6.22 ms (the early
// weight-gradient launch also takes CUs from the persistent tile kernels).  hipExtAnyOrderLaunch (no barrier bit inside ONE queue) is
// documented as unsupported on gfx9.  Kept as a switch + test: the dependency analysis is right, the platform's price for it is not.
bool StepFork::on() const { return state == 1 && DR4SR_XENV("DR4SR_STREAMS") != nullptr; }
int StepFork::fork(hipStream_t main, int first, int n) const {
    if (!on()) return 0;
    hipError_t e = hipSuccess;
    for (int i = first; i < first + n && e == hipSuccess; ++i) {
        e = hipEventRecord(fork_ev[i], main);
        if (e == hipSuccess) e = hipStreamWaitEvent(side_[i], fork_ev[i], 0);
    }
    return hip_ret(e);
}
int StepFork::join(hipStream_t main, int first, int n) const {
    if (!on()) return 0;
    hipError_t e = hipSuccess;
    for (int i = first; i < first + n && e == hipSuccess; ++i) {
        e = hipEventRecord(join_ev[i], side_[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(main, join_ev[i], 0);
    }
    return hip_ret(e);
}

static std::atomic<int> g_env_generation{0};
int dr4sr_env_generation() { return g_env_generation.load(std::memory_order_acquire); }
std::mutex& dr4sr_env_mutex() { static std::mutex mu; return mu; }
extern "C" int dr4sr_reload_env(void) { return g_env_generation.fetch_add(1, std::memory_order_acq_rel) + 1; }
extern "C" int dr4sr_sasrec_plan_sizeof(void) { return (int)sizeof(dr4sr_sasrec_plan); }
extern "C" int dr4sr_build_flags(void) {
#ifdef DR4SR_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int64_t dr4sr_sasrec_param_layout(int32_t n_items, int32_t L, int32_t D, int32_t F, int32_t n_layer,
                                             int64_t* off) {
    int64_t o = 0;
    auto put = [&](int i, int64_t n) { if (off) off[i] = o; o += n; };
    put(0, (int64_t)n_items * D);
    put(1, (int64_t)L * D);
    for (int l = 0; l < n_layer; ++l) {
        const int b = 2 + 12 * l;
        put(b + P_IN_W, 3LL * D * D); put(b + P_IN_B, 3LL * D);
        put(b + P_OUT_W, (int64_t)D * D); put(b + P_OUT_B, D);
        put(b + P_W1, (int64_t)F * D); put(b + P_B1, F);
        put(b + P_W2, (int64_t)D * F); put(b + P_B2, D);
        put(b + P_LN1_W, D); put(b + P_LN1_B, D); put(b + P_LN2_W, D); put(b + P_LN2_B, D);
    }
    return o;
}

// LDS footprint of the one-wave-per-head VALU attention backward (attn.hip): q|k|v|dctx rows and two score planes per head
static bool valu_attn_fits(const dr4sr_sasrec_plan* p) {
    return sizeof(float) * (4LL * p->L * p->D + 2LL * p->H * p->L * (p->L + 1) + p->L) <= 160 * 1024;
}

// the encoder shapes the kernels are instantiated for (anything else is refused before a workspace is sized or a kernel launched)
static int check_shape(const dr4sr_sasrec_plan* p) {
    if (p->L > 64) return DR4SR_E_SHAPE;
    if (!((p->D == 64 && (p->F == 128 || p->F == 256)) || (p->D == 128 && p->F == 128))) return DR4SR_E_SHAPE;
    if (p->H <= 0 || p->H > 4 || p->D % p->H || (p->D / p->H != 32 && p->D / p->H != 64)) return DR4SR_E_SHAPE;
    // head counts other than 2 run the one-wave-per-head kernels of attn.hip, whose backward keeps q|k|v|dctx and two score
    // planes per head in LDS: reject the shapes that do not fit the 160 KB of a CU instead of failing at the launch
    if (p->H != 2 && !valu_attn_fits(p)) return DR4SR_E_SHAPE;
    return 0;
}

static int check_plan(const dr4sr_sasrec_plan* p) {
    if (!p || p->abi_version != DR4SR_ABI_VERSION) return DR4SR_E_ARG;
    if (p->B <= 0 || p->L <= 0 || p->n_items < 2 || p->n_layer <= 0 || p->n_layer > DR4SR_MAX_LAYERS) return DR4SR_E_ARG;
    const int rc = check_shape(p);
    if (rc) return rc;
    if (!(p->p_drop >= 0.f && p->p_drop < 1.f)) return DR4SR_E_ARG;
    if (!p->params || !p->state || !p->in_item_id || !p->seqlen) return DR4SR_E_ARG;
    return 0;
}

// the deterministic latency form needs the partial blocks of the in-tile attention's shared dK | dV rows: 5 x 2 D floats per token slot and layer
// (a toys plan of the latency regime, 1 280 x 50 slots: 328 MB) — carved for every deterministic-mode workspace of up to 131 072 slots
// (the carving depends on the plan's SHAPE only, not on the cross-check switches attn_tile_capable reads: a workspace sized under one switch
//  setting stays valid under another)
static bool det_kv_shape(const dr4sr_sasrec_plan* p) { return p->H == 2 && p->L <= 64 && (p->D == 64 || p->D == 128) && (int64_t)p->B * p->L <= 131072; }
static bool det_lat_capable(const dr4sr_sasrec_plan* p) { return attn_tile_capable(p) && det_kv_shape(p); }

int carve_workspace(const dr4sr_sasrec_plan* p, Workspace* ws) {
    const int64_t D = p->D, F = p->F, Tmax = (int64_t)p->B * p->L;
    ws->n_params = dr4sr_sasrec_param_layout(p->n_items, p->L, p->D, p->F, p->n_layer, ws->off);
    ws->Tmax = (int)Tmax;
    {
        const bool forced = DR4SR_XENV("DR4SR_LATENCY_TMAX") != nullptr;           // sweeps: the capacity rule with a moved boundary
        const int64_t hint = p->expected_tokens < Tmax ? p->expected_tokens : Tmax;
        const bool known = hint > 0 && !forced;
        // boundaries measured at d = 64; at d = 128 both crossovers sit at half the token count (4 k / 7 k): the work per token doubles
        // round 4: with the attention inside the 16-token tile kernels (attn_tile.h) the latency forms carry toys-shaped batches further
        // (B = 1 536, 8.5 k tokens: 0.227 against 0.240 ms; tie at 11 k) and long-sequence batches less far (all-50 rows, 6 400 tokens:
        // 0.199 against 0.193 ms — every query sees all five key tiles of its window): the boundary follows the expected mean length
        // round 6: with the wave-per-tile attention (attn_wave.hip) the at-scale forms of SHORT-sequence plans overtake the latency forms between
        // 7.0 k and 8.1 k expected tokens (toys rows, same box, latency / at scale: B = 1 280 0.1705 / 0.2043 ms, B = 1 536 0.2218 / 0.2090,
        // B = 2 048 0.2536 / 0.2227, B = 2 560 0.2889 / 0.2343 — profiles/round6_regime_sweep.txt) and the middle regime (at-scale tiles with one
        // attention workgroup per sequence: 0.2419 / 0.2652 / 0.2893 at the last three sizes) is never the fastest: one boundary for both
        const bool wave_capable = p->H == 2 && p->L <= 64 && !DR4SR_ENV("DR4SR_NO_FUSE") && !DR4SR_ENV("DR4SR_ATTN_LISTS") && !DR4SR_ENV("DR4SR_ATTN_NOSPLIT")
                                  && !DR4SR_ENV("DR4SR_ATTN_VALU");
        const bool short_plan = hint <= 16 * (int64_t)p->B;
        const int64_t scale_tokens = !attn_tile_capable(p) ? DR4SR_SCALE_TOKENS
                                     : (short_plan ? (wave_capable ? DR4SR_SCALE_TOKENS_SHORT_WAVE : DR4SR_SCALE_TOKENS_SHORT) : DR4SR_SCALE_TOKENS_LONG);
        ws->scale = known ? hint * D > scale_tokens * 64 : at_scale((int)Tmax);
        // the length-class lists pay off through their short classes (1..8-token VALU class, 16-row tiles); a batch of LONG sequences
        // runs faster as one 8-wave workgroup per sequence at every size (round 3, all-50 batches: B = 2 048 0.929 vs 0.990 ms,
        // B = 8 192 3.36 vs 3.62 ms), so the lists also need an expected mean length of at most 16 tokens
        ws->attn_split = known ? (hint * D > (int64_t)(wave_capable ? (scale_tokens < DR4SR_ATTN_SPLIT_TOKENS ? scale_tokens : DR4SR_ATTN_SPLIT_TOKENS) : DR4SR_ATTN_SPLIT_TOKENS) * 64 && short_plan)
                               : at_scale((int)Tmax);
        // deterministic summation order (DR4SR_DETERMINISTIC=1; Python: train.deterministic): the at-scale forms at every size — their item-table
        // gradient is owner-computed, their attention has no atomics — with the weight-gradient launch's remaining atomics replaced by
        // partial buffers summed in a fixed order (linear.hip k_wgrad_det_reduce)
        ws->det = DR4SR_ENV("DR4SR_DETERMINISTIC") != nullptr && atoi(DR4SR_ENV("DR4SR_DETERMINISTIC")) != 0;
        ws->det_lat = false;
        if (ws->det) {
            // round 6: a plan the rule above leaves in the latency regime keeps its latency launches (16-token tiles, attention
            // inside) — kernels.h Workspace::det_lat; every other plan takes the at-scale forms as before.  DR4SR_DET_SCALE_FORMS (experiments
            // build): the at-scale forms at every size, round 5's mode (A/B)
            ws->det_lat = !ws->scale && !ws->attn_split && det_lat_capable(p) && !DR4SR_XENV("DR4SR_DET_SCALE_FORMS");
            ws->scale = true;
            // round 6: the wave-per-tile attention (attn_wave.hip) writes every dqkv row from exactly one wave — bit-reproducible by
            // construction and two launches per layer at any batch size, where the mode used to fall back to one workgroup per sequence
            // below the lists' threshold (16 + 20 us of the mode's +70 us at B = 256, NOTEBOOK round 5)
            if (p->H == 2 && p->L <= 64 && !DR4SR_ENV("DR4SR_NO_FUSE")) ws->attn_split = true;
        }
        // tests (cached per process until dr4sr_reload_env(), common.h): DR4SR_FORCE_SCALE = 1 / 0 forces every at-scale / latency form, DR4SR_FORCE_ATTN_SPLIT the attention alone
        if (const char* f = DR4SR_ENV("DR4SR_FORCE_SCALE")) {
            ws->scale = ws->attn_split = atoi(f) != 0;
            if (ws->det) ws->det_lat = !ws->scale && det_lat_capable(p) && !DR4SR_XENV("DR4SR_DET_SCALE_FORMS");
        }
        if (const char* f = DR4SR_ENV("DR4SR_FORCE_ATTN_SPLIT")) { ws->attn_split = atoi(f) != 0; if (ws->attn_split) ws->det_lat = false; }
        ws->scale_wg = ws->scale || ws->det;
        if (ws->det_lat) { ws->scale = false; ws->attn_split = false; }
        else if (ws->det) ws->scale = true;
        // round 4, opt-in (DR4SR_ATTN_WINDOW=1): where the lists would run on a short-sequence plan at d = 64, the window attention of
        // attn_tile.h as ONE launch per layer and direction (attn_tile_sa.hip).  Measured slower than the lists (toys B = 8 192: forward
        // 25 against 24 us per layer, backward 68 against 49; B = 32 768: 92 / 239 against 63 / 159): a 16 x 32 window computes ~9x the
        // (query, key) pairs the sequences hold and the launch is bound by instruction issue, not by its latency chain — NOTEBOOK round 4
        ws->attn_tile_sa = DR4SR_XENV("DR4SR_ATTN_WINDOW") && ws->attn_split && p->D == 64 && hint > 0 && hint <= 16 * (int64_t)p->B
                           && attn_tile_capable(p) && !DR4SR_ENV("DR4SR_ATTN_NOSPLIT");
    }
    char* base = (char*)p->workspace;
    int64_t o = 0;
    auto take = [&](int64_t nfloat) -> float* {
        float* r = base ? (float*)(base + o) : nullptr;
        o += ((nfloat * 4 + 255) / 256) * 256;
        return r;
    };
    ws->cu = (int*)take(p->B + 1);
    ws->tile_seq = (int*)take((Tmax + 15) / 16 + 1);
    ws->seq_class = (int*)take(4 + 7LL * p->B);
    ws->len_buf = (int*)take(4 * (int64_t)p->B + 4 * PREP_MAX_BLK);      // two-phase prep: 16 B per sequence + 16 B per workgroup of the optimizer launch
    ws->attn_rd = take(Tmax * p->H);
    for (int i = 0; i <= p->n_layer; ++i) { ws->X[i] = take(Tmax * D); ws->dX[i] = take(Tmax * D); }
    ws->dctx = take(Tmax * D);
    ws->de_rec = (int4*)take(Tmax * 4); ws->idx32 = (int*)take(Tmax); ws->tok = (int2*)take(2 * Tmax);
    ws->de_ent = (int4*)take(Tmax * 12); ws->de_off = (unsigned char*)take((Tmax / 16 + 1) * 257);     // [tiles of >= 16 tokens][G + 4 <= 1028 bytes]
    ws->wT_stride = 4 * D * D + 2 * D * F;
    ws->wT = take(ws->wT_stride * p->n_layer);
    // deterministic mode: partial blocks of the weight-gradient jobs [layer][job][split][stride], LayerNorm [layer][split][4 D], dP [split][L D]
    ws->det_stride = D == 64 ? 64 * 64 + 64 : (int64_t)(D > F ? D : F) * (D > F ? D : F) + (D > F ? D : F);      // d = 64: 64 x 64 block jobs (k_wgrad_bf64); launch_wgrad refuses wider jobs
    ws->det_part = nullptr; ws->det_ln = nullptr; ws->det_dp = nullptr; ws->det_kv = nullptr;
    ws->det_kv_layer = ((Tmax + 15) / 16) * 5 * 16 * 2 * D;
    if (ws->det && det_kv_shape(p)) ws->det_kv = take(ws->det_kv_layer * p->n_layer);      // (whatever THIS plan's regime: the plans of one workspace differ in their hints)
    if (ws->det) {
        //  jobs per layer as launch_wgrad indexes them: the 64 x 64 blocks of d = 64 (4 + 2 F / 64), the six whole GEMMs of d = 128
        // (253 MB per layer were carved for 12 jobs at d = 128, F = 128)
        const int64_t det_jobs = D == 64 ? 4 + 2 * F / 64 : 6;
        ws->det_part = take((int64_t)p->n_layer * det_jobs * DR4SR_DET_MAX_SPLITS * ws->det_stride);
        ws->det_ln = take((int64_t)p->n_layer * DR4SR_DET_MAX_SPLITS * 4 * D);
        ws->det_dp = take((int64_t)DR4SR_DET_MAX_SPLITS * p->L * D);
    }
    ws->wfrag = D == 128 ? take(ws->wT_stride * p->n_layer) : nullptr;      // d = 128 latency forms: fp32 fragment-major images (linear.hip wfrag_image_write)
    // d = 128: split-weight images of the bf16x3 tile GEMMs (common.h WSplit): 2 orientations x (hi | lo) x E bf16 per layer = 2 E floats
    ws->wsplit_E = D == 128 ? ws->wT_stride : 0;
    ws->wsplit = ws->wsplit_E ? reinterpret_cast<unsigned short*>(take(2 * ws->wsplit_E * p->n_layer)) : nullptr;
    if (!base && ws->wsplit_E) ws->wsplit = reinterpret_cast<unsigned short*>(1);      // (size probe: non-NULL so that the launch forms report alike)
    ws->score_part = take(2LL * (p->B > (Tmax + 15) / 16 ? p->B : (Tmax + 15) / 16));   // per sequence, or per token tile (fused last layer)
    ws->ln_part = take((int64_t)p->n_layer * ((Tmax + 15) / 16) * 4 * D);
    for (int l = 0; l < p->n_layer; ++l) {
        LayerWs& w = ws->layer[l];
        w.qkv = take(Tmax * 3 * D); w.ctx = take(Tmax * D); w.attn_st = take(Tmax * p->H * 2);
        w.attn_keep = reinterpret_cast<unsigned*>(take(Tmax * p->H * 2));
        w.u1 = take(Tmax * D); w.y = take(Tmax * D); w.st1 = take(Tmax * 2);
        w.a = take(Tmax * F); w.h = take(Tmax * F); w.u2 = take(Tmax * D); w.st2 = take(Tmax * 2);
        w.df = take(Tmax * D); w.da = take(Tmax * F); w.du1 = take(Tmax * D); w.dout = take(Tmax * D); w.dqkv = take(Tmax * 3 * D);
    }
    ws->bytes = o;
    return 0;
}

extern "C" int64_t dr4sr_sasrec_workspace_bytes(const dr4sr_sasrec_plan* plan) {
    if (!plan) return DR4SR_E_ARG;
    dr4sr_sasrec_plan q = *plan;
    q.workspace = nullptr;
    if (q.B <= 0 || q.L <= 0 || q.n_layer <= 0 || q.n_layer > DR4SR_MAX_LAYERS) return DR4SR_E_ARG;
    if (check_shape(&q)) return DR4SR_E_SHAPE;
    Workspace ws;
    carve_workspace(&q, &ws);
    return ws.bytes;
}

extern "C" int dr4sr_sasrec_at_scale(const dr4sr_sasrec_plan* plan) {
    if (!plan) return DR4SR_E_ARG;
    dr4sr_sasrec_plan q = *plan;
    q.workspace = nullptr;
    if (q.B <= 0 || q.L <= 0 || q.n_layer <= 0 || q.n_layer > DR4SR_MAX_LAYERS) return DR4SR_E_ARG;
    if (check_shape(&q)) return DR4SR_E_SHAPE;
    Workspace ws;
    carve_workspace(&q, &ws);
    // bit 2: attention inside the tile kernels (attn_tile.h); bit 3: the same window attention as launches of its own instead of the lists
    // bit 4 (round 6): where the lists would run, the wave-per-tile attention instead (attn_wave.hip) — one launch per layer and direction
    // bit 5: ... with its forward folded into the wave-tile forward kernels (no attention launch forward)
    // bit 6: the deterministic latency form (Workspace::det_lat): latency tile launches, ordered table / weight-gradient forms in k_wgrad
    return (ws.scale ? 1 : 0) | (ws.attn_split ? 2 : 0) | (attn_in_tile(&q, ws) ? 4 : 0) | (ws.attn_tile_sa ? 8 : 0) | (attn_wave_on(&q, ws) ? 16 : 0)
           | (attn_fold_fwd(&q, ws) ? 32 : 0) | (ws.det_lat ? 64 : 0);
}

static int get_ws(const dr4sr_sasrec_plan* p, Workspace* ws) {
    int rc = check_plan(p);
    if (rc) return rc;
    if (!p->workspace) return DR4SR_E_ARG;
    carve_workspace(p, ws);
    if (ws->bytes > p->workspace_bytes) return DR4SR_E_WS;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Dense Adam over the flat buffers (torch.optim.Adam, single-tensor formula):
//   g = grad/n_valid + wd*p ; m += (g-m)(1-b1) ; v = b2 v + (1-b2) g^2
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// t = state[STEP]+1; the last block to finish bumps state[STEP] (ticket in state[8]).
// grads[n + 2] is a POISON word: non-zero means some producer of this gradient failed on the device (today: an exchange timeout of
// the cooperative GRU recurrence, csrc/gru_coop.hip) — the step is then skipped entirely (parameters, moments and the step counter
// stay as they were).  It lives in the gradient tail so that a data-parallel sum-all-reduce carries it to every replica and all of
// them skip the same step.
// `next` (optional): the launch also prepares the NEXT training step of the same plan — its workgroup 0 runs the prep (batch
// selection, prefix scan, RNG step), every thread zeroes the gradient words it has consumed and the last workgroup to finish
// zeroes the {n_valid, loss} tail — so a k-step graph needs one k_prep, not k (a launch boundary + a single-workgroup
// latency chain per step saved).  enable = 1 (B <= 1024): an extra workgroup, dispatched first, runs the whole prep.  enable = 2
// (larger batches): EVERY workgroup first runs its share of the selection + seqlen gather (prep_select: coalesced over the grid,
// in flight while the Adam loop runs) and the LAST workgroup to finish runs the scan part (prep_body<256, true>) — as a launch of
// its own the single-workgroup prep of 8 192 sequences cost 41 us per step.
// OPT (round 4, basemodel.py:79-98: the reference's `optimizer` choices with torch's defaults): DR4SR_OPT_ADAM as above;
//   DR4SR_OPT_SGD      torch.optim.SGD(lr, weight_decay)        g += wd p ; p -= lr g                         (no momentum; M, V untouched)
//   DR4SR_OPT_ADAGRAD  torch.optim.Adagrad(lr, weight_decay)    g += wd p ; V += g^2 ; p -= lr g / (sqrt(V) + eps)     (eps 1e-10, lr_decay 0)
//   DR4SR_OPT_RMSPROP  torch.optim.RMSprop(lr, weight_decay)    g += wd p ; V = b2 V + (1 - b2) g^2 ; p -= lr g / (sqrt(V) + eps)   (b2 = alpha 0.99)
// The un-normalised gradient, the poison word, the step counter and the fused next-step prep are the same for all four.
struct AdamNext { int enable; int phase2_launch; PrepArgs prep; };
template <int OPT>
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ P, float* __restrict__ G, float* __restrict__ M,
                                              float* __restrict__ V, int64_t n, int* __restrict__ state, float lr, float b1,
                                              float b2, float eps, float wd, float* __restrict__ loss_log,
                                              const int* __restrict__ log_index, const AdamNext next) {
    __shared__ float sh[2];
    __shared__ unsigned long long part[256];
    __shared__ int4 boff[PREP_MAX_BLK];            // two-phase prep: token / class offsets of the workgroups' sequence ranges
    const int t = state[DR4SR_STATE_STEP] + 1;
    const float nvalid = G[n];
    const float gs = nvalid > 0.f ? 1.0f / nvalid : 0.f;
    const bool poisoned = G[n + 2] != 0.f;
    const int nblk = next.enable == 1 ? (int)gridDim.x - 1 : (int)gridDim.x, blk = next.enable == 1 ? (int)blockIdx.x - 1 : (int)blockIdx.x;
    if (next.enable == 2) {                                // two-phase prep, phase 1 (+ this step's loss-log entry, before the batch index moves)
        if (loss_log && blk == 0 && threadIdx.x == 0) loss_log[log_index ? max(*log_index - 1, 0) : 0] = G[n + 1] * gs;
        prep_phase1<256>(next.prep, blk, nblk, part);
    }
    if (blk < 0) {                                         // dispatched first: the next step's prep (and this step's loss-log entry,
        const float lossv = G[n + 1] * gs;                 //  whose index the prep is about to advance)
        if (loss_log && threadIdx.x == 0) loss_log[log_index ? max(*log_index - 1, 0) : 0] = lossv;
        __syncthreads();
        prep_body<256>(next.prep, part);
    } else {
        if (threadIdx.x == 0) {               // double-precision bias corrections, once per block
            const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
            sh[0] = (float)((double)lr / bc1);
            sh[1] = (float)(1.0 / sqrt(bc2));
        }
        __syncthreads();
        const float step_size = sh[0], inv_sqrt_bc2 = sh[1];
        if (!next.enable && loss_log && blk == 0 && threadIdx.x == 0) loss_log[log_index ? max(*log_index - 1, 0) : 0] = G[n + 1] * gs;
        const int64_t n4 = n / 4;
        auto upd = [&](float& pe, float& me, float& ve, float ge) {
            const float g = ge * gs + wd * pe;
            if constexpr (OPT == DR4SR_OPT_SGD) {
                pe = pe - lr * g;
            } else if constexpr (OPT == DR4SR_OPT_ADAGRAD) {
                ve = ve + g * g;
                pe = pe - lr * (g / (sqrtf(ve) + eps));
            } else if constexpr (OPT == DR4SR_OPT_RMSPROP) {
                ve = ve * b2 + (1.0f - b2) * g * g;
                pe = pe - lr * (g / (sqrtf(ve) + eps));
            } else {
                me = me + (g - me) * (1.0f - b1);
                ve = ve * b2 + (1.0f - b2) * g * g;
                const float denom = sqrtf(ve) * inv_sqrt_bc2 + eps;
                pe = pe - step_size * (me / denom);
            }
        };
        constexpr bool useM = OPT == DR4SR_OPT_ADAM, useV = OPT != DR4SR_OPT_SGD;      // moment buffers this optimizer keeps
        // two independent float4 groups per thread per iteration: 8 loads in flight instead of 4 (the loop is latency-, not
        // bandwidth-bound at 3 iterations per thread)
        const int64_t stride = (int64_t)nblk * 256;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n4; i += 2 * stride) {
            const int64_t j = i + stride;
            const bool two = j < n4;
            if (poisoned) {                                // skipped step: only the next step's zeroed gradient is still owed
                if (next.enable) { st4(G + 4 * i, z4); if (two) st4(G + 4 * j, z4); }
                continue;
            }
            float4 p0 = ld4(P + 4 * i), m0 = useM ? ld4(M + 4 * i) : z4, v0 = useV ? ld4(V + 4 * i) : z4;
            const float4 g0 = ld4(G + 4 * i);
            float4 p1 = p0, m1 = m0, v1 = v0, g1 = g0;
            if (two) { p1 = ld4(P + 4 * j); if (useM) m1 = ld4(M + 4 * j); if (useV) v1 = ld4(V + 4 * j); g1 = ld4(G + 4 * j); }
            upd(p0.x, m0.x, v0.x, g0.x); upd(p0.y, m0.y, v0.y, g0.y); upd(p0.z, m0.z, v0.z, g0.z); upd(p0.w, m0.w, v0.w, g0.w);
            st4(P + 4 * i, p0); if (useM) st4(M + 4 * i, m0); if (useV) st4(V + 4 * i, v0);
            if (next.enable) st4(G + 4 * i, z4);
            if (two) {
                upd(p1.x, m1.x, v1.x, g1.x); upd(p1.y, m1.y, v1.y, g1.y); upd(p1.z, m1.z, v1.z, g1.z); upd(p1.w, m1.w, v1.w, g1.w);
                st4(P + 4 * j, p1); if (useM) st4(M + 4 * j, m1); if (useV) st4(V + 4 * j, v1);
                if (next.enable) st4(G + 4 * j, z4);
            }
        }
    }
    __shared__ int s_last;
    // two-phase prep: phase 1's words were published with agent-scope (sc1, write-through) stores and phase 2 reads them with
    // agent-scope loads in the LAST workgroup.  The hand-off is the guide's "sc1 payload -> vmcnt(0) -> flag" form: every wave drains
    // its own stores EXPLICITLY before the barrier in front of the ticket (inline asm: the compiler cannot drop it, and the barrier
    // alone is not required to drain vmcnt) — no L2 write-back fence, which would cost the launch ~6 us.
    if (next.enable == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {                    // no fence needed: the kernel boundary publishes the parameter writes
        const int ticket = atomicAdd(&state[8], 1);
        s_last = ticket == (int)gridDim.x - 1;
        if (s_last) {                          // every workgroup has read the tail and state[STEP] by now
            state[8] = 0;
            if (!poisoned) state[DR4SR_STATE_STEP] = t;
            if (next.enable) { G[n] = 0.f; G[n + 1] = 0.f; G[n + 2] = 0.f; G[n + 3] = 0.f; }
        }
    }
    if (next.enable == 2 && !next.phase2_launch) {   // two-phase prep, phase 2: by the last workgroup, from agent-scope loads of what phase 1 published
        __syncthreads();                       // (run right after phase 1 instead, behind an agent-scope fence and a ticket of its own, the
        if (s_last) prep_phase2<256>(next.prep, (int)gridDim.x, part, boff);     //  launch got longer: 32 -> 43 us at B = 8 192)
    }
}

// phase 2 of the two-phase prep as a launch of its own, spread over the device (round 3): +1 launch boundary, -(one CU's scattered stores)
__global__ __launch_bounds__(256) void k_prep_phase2(const PrepArgs P, const int nblk) {
    __shared__ unsigned long long part[256];
    __shared__ int4 boff[PREP_MAX_BLK];
    prep_phase2<256>(P, nblk, part, boff, (int)blockIdx.x, (int)gridDim.x);
}

int launch_adam_flat(float* P, float* G, float* M, float* V, int64_t n, int* state, float lr, float b1, float b2,
                     float eps, float wd, hipStream_t s, float* loss_log, const int* log_index, const PrepArgs* next, int opt) {
    if (!P || !G || !M || !V || !state || n <= 0 || (n & 3) || opt < DR4SR_OPT_ADAM || opt > DR4SR_OPT_RMSPROP) return DR4SR_E_ARG;
    int64_t blocks = (n / 4 + 255) / 256;
    const int cap = DR4SR_ENV("DR4SR_ADAM_BLOCKS") ? atoi(DR4SR_ENV("DR4SR_ADAM_BLOCKS")) : 256;
    if (blocks > cap) blocks = cap;
    AdamNext nx;
    nx.enable = next ? (next->B > 1024 && next->len_buf && blocks <= PREP_MAX_BLK && (next->B + blocks - 1) / blocks < 65536 ? 2 : 1) : 0;
    nx.phase2_launch = 0;
    if (next) nx.prep = *next; else nx.prep = PrepArgs{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, PermSel{nullptr, 0, 0, 0, nullptr}, nullptr, nullptr, nullptr};
    const bool p2_inline = DR4SR_ENV("DR4SR_PREP2_INLINE") != nullptr;      // cross-check: phase 2 as the tail of the optimizer launch
    nx.phase2_launch = nx.enable == 2 && !p2_inline;
    const dim3 grid((unsigned)blocks + (nx.enable == 1 ? 1 : 0));
    switch (opt) {
        case DR4SR_OPT_SGD: hipLaunchKernelGGL(k_adam<DR4SR_OPT_SGD>, grid, dim3(256), 0, s, P, G, M, V, n, state, lr, b1, b2, eps, wd, loss_log, log_index, nx); break;
        case DR4SR_OPT_ADAGRAD: hipLaunchKernelGGL(k_adam<DR4SR_OPT_ADAGRAD>, grid, dim3(256), 0, s, P, G, M, V, n, state, lr, b1, b2, eps, wd, loss_log, log_index, nx); break;
        case DR4SR_OPT_RMSPROP: hipLaunchKernelGGL(k_adam<DR4SR_OPT_RMSPROP>, grid, dim3(256), 0, s, P, G, M, V, n, state, lr, b1, b2, eps, wd, loss_log, log_index, nx); break;
        default: hipLaunchKernelGGL(k_adam<DR4SR_OPT_ADAM>, grid, dim3(256), 0, s, P, G, M, V, n, state, lr, b1, b2, eps, wd, loss_log, log_index, nx);
    }
    if (nx.phase2_launch) {
        const int B = next->B, per = 8 * 256;
        int g2 = (B + per - 1) / per;
        if (g2 > 256) g2 = 256;
        hipLaunchKernelGGL(k_prep_phase2, dim3(g2), dim3(256), 0, s, nx.prep, (int)blocks);
    }
    return DR4SR_LAUNCH_CHECK();
}
int launch_adam(const dr4sr_sasrec_plan* p, hipStream_t s, const PrepArgs* next) {
    return launch_adam_flat(p->params, p->grads, p->adam_m, p->adam_v, p->n_params, p->state, p->lr, p->beta1, p->beta2,
                            p->adam_eps, p->weight_decay, s, p->loss_log, p->perm ? p->perm_counter : nullptr, next, p->optimizer);
}

extern "C" int dr4sr_adam_step(const dr4sr_sasrec_plan* plan, void* stream) {
    if (!plan || plan->abi_version != DR4SR_ABI_VERSION || !plan->params || !plan->state) return DR4SR_E_ARG;
    return launch_adam(plan, (hipStream_t)stream);
}

// zero the flat gradient (+tail) with our own kernel: a hipMemsetAsync node captured into a hipGraph was observed
// (ROCm 7.2) to fill the last 16 bytes with a stale pattern on later replays, which poisons n_valid.
__global__ __launch_bounds__(256) void k_zero(float* __restrict__ p, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        st4(p + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
}
static int launch_zero_grads(const dr4sr_sasrec_plan* p, int64_t n_params, hipStream_t s) {
    const int64_t n4 = (n_params + DR4SR_GRAD_TAIL) / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_zero, dim3((unsigned)blocks), dim3(256), 0, s, p->grads, n4);
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

// MFMA attention needs exactly 2 heads (one wave pair per head); other head counts use the VALU kernels.
static bool use_mfma_attn(const dr4sr_sasrec_plan* p) {
    const bool off = DR4SR_ENV("DR4SR_ATTN_VALU") != nullptr;
    return p->H == 2 && (!off || !valu_attn_fits(p));       // the cross-check switch applies where the VALU kernels can run
}
static int attn_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int l, int training, hipStream_t s) {
    if (ws.attn_tile_sa) return launch_attn_tile_fwd(p, ws, l, training, s);
    return use_mfma_attn(p) ? launch_attn2_fwd(p, ws, l, training, s) : launch_attn_fwd(p, ws, l, training, s);
}
static int attn_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int l, int training, hipStream_t s) {
    if (ws.attn_tile_sa) return launch_attn_tile_bwd(p, ws, l, training, s);
    return use_mfma_attn(p) ? launch_attn2_bwd(p, ws, l, training, s) : launch_attn_bwd(p, ws, l, training, s);
}

// mid_fused: the last layer's post_fwd / scorer / post_bwd run as ONE launch (launch_post_mid) between the two halves
static int forward_layers(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s, bool mid_fused = false) {
    const bool fuse = DR4SR_ENV("DR4SR_NO_FUSE") == nullptr;     // qkv of layer l>0 is emitted by post_fwd(l-1),
    if (tile_bf3(p, ws)) RC(launch_wsplit(p, ws, s));                // d = 128 at scale: this pass's weights as bf16 hi | lo images
    if (fuse) RC(launch_embqkv_fwd(p, ws, training, s));             // qkv of layer 0 by the embedding gather
    else { RC(launch_embed_fwd(p, ws, training, s)); if (wfrag_img_on(p, ws)) RC(launch_wfrag_write(p, ws, s)); }
    const bool in_tile = attn_in_tile(p, ws);                    // the attention runs at the head of post_fwd / post_mid (attn_tile.h)
    for (int l = 0; l < p->n_layer; ++l) {
        if (!fuse) RC(launch_qkv_fwd(p, ws, l, s));
        if (!in_tile && !attn_fold_fwd(p, ws)) RC(attn_fwd(p, ws, l, training, s));     // (folded: at the head of the wave-tile forward launch, linear_wave.hip wt_attn_ctx)
        if (!(mid_fused && l == p->n_layer - 1)) RC(launch_post_fwd(p, ws, l, training, s));
    }
    return 0;
}

// Two-bucket data-parallel step (include/dr4sr_hip.h: dr4sr_sasrec_fwd_bwd_phase).  Where the table gradient is a set of k_wgrad jobs (at
// scale) the last launch of the backward is cut in two: the FIRST carries the owner / scatter jobs — after it the item and position
// table gradients [0, off[2]) are final, 3.05 of the 3.33 MB of a toys replica — plus the weight-gradient jobs of layers >= split (their
// inputs were ready first, and they keep the CUs busy beside the LDS-bound owners exactly as in the one-launch form); the SECOND carries
// layers < split and the reduce jobs that write the {n_valid, loss_sum} tail.  The caller issues the table bucket's all-reduce between
// the two, so the collective runs BESIDE the second launch.  split = max(1, n_layer / 2); DR4SR_DP_SPLIT_LAYER overrides (n_layer = a
// table-only first launch: the longest cover, but the owners run alone; 0 = one launch, one bucket).
static int dp_split_layer(const dr4sr_sasrec_plan* p) {
    int v = p->n_layer / 2 > 1 ? p->n_layer / 2 : 1;
    if (const char* e = DR4SR_ENV("DR4SR_DP_SPLIT_LAYER")) v = atoi(e);
    return v < 0 ? 0 : (v > p->n_layer ? p->n_layer : v);
}
static bool dp_two_buckets(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    return !DR4SR_ENV("DR4SR_NO_FUSE") && !step_fork().on() && wgrad_table_jobs(p, ws) && dp_split_layer(p) > 0;
}

// phase 0: the whole backward; 1: up to and including the launch that completes the table bucket; 2: what is left (one launch)
static int backward_layers(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, int with_score, hipStream_t s,
                           bool mid_fused = false, bool meta = false, int phase = 0) {
    const bool fused = !DR4SR_ENV("DR4SR_NO_FUSE");
    const bool two = phase != 0 && dp_two_buckets(p, ws);
    if (phase == 2) return two ? launch_wgrad(p, ws, training, with_score, s, fused, meta, 0, dp_split_layer(p), 0) : 0;
    // The weight gradients of layer l >= 1 need dqkv_l (attention backward of layer l) and nothing that the layers below still have to
    // compute: their launch goes to a side stream right behind attn_bwd(l) and runs BESIDE post_bwd(l-1) / attn_bwd(l-1) / the embedding
    // stage — an HBM stream next to latency-bound launches (in a captured step: a parallel branch of the graph).  The launch that
    // holds layer 0 (table-gradient jobs, embedding-stage plane, scorer partials) stays last, behind the join.  DR4SR_WGRAD_ONE_LAUNCH:
    // every layer in the last launch, as before round 4 (cross-check).
    const StepFork& fk = step_fork();
    const bool early = fused && fk.on() && p->n_layer > 1 && !DR4SR_XENV("DR4SR_WGRAD_ONE_LAUNCH");
    bool forked = false;
    for (int l = p->n_layer - 1; l >= 0; --l) {
        if (!(mid_fused && l == p->n_layer - 1)) RC(launch_post_bwd(p, ws, l, training, s));
        if (!attn_in_tile(p, ws)) RC(attn_bwd(p, ws, l, training, s));      // else: the tail of post_bwd / post_mid
        if (!fused) RC(launch_qkv_bwd(p, ws, l, s));      // else folded into post_bwd(l-1) / the embedding scatter
        if (early && l >= 1) {
            RC(fk.fork(s, 2, 1));                           // side 2 is a queue: the layers' launches run one after the other on it
            RC(launch_wgrad(p, ws, training, with_score, fk.side(s, 2), true, meta, l, l + 1));
            forked = true;
        }
    }
    if (!fused) RC(launch_embed_bwd(p, ws, training, s));
    else if (!qeb_in_wgrad(ws)) RC(launch_qkv_embed_bwd(p, ws, training, s));
    if (forked) RC(fk.join(s, 2, 1));
    if (two) return launch_wgrad(p, ws, training, with_score, s, fused, meta, dp_split_layer(p), p->n_layer, 1);
    RC(launch_wgrad(p, ws, training, with_score, s, fused, meta, 0, forked ? 1 : -1));
    return 0;
}

// everything of a training step between the prep and the optimizer
static int fwd_bwd_core(const dr4sr_sasrec_plan* plan, const Workspace& ws, hipStream_t s, int phase = 0) {
    if (!DR4SR_ENV("DR4SR_NO_FUSE")) {
        if (phase == 2) return backward_layers(plan, ws, 1, 2, s, true, false, 2);
        RC(forward_layers(plan, ws, 1, s, true));
        RC(launch_post_mid(plan, ws, 1, s));
        RC(backward_layers(plan, ws, 1, 2, s, true, false, phase));
        return 0;
    }
    if (phase == 2) return 0;
    RC(forward_layers(plan, ws, 1, s));
    RC(launch_score_packed(plan, ws, s));
    RC(backward_layers(plan, ws, 1, 1, s));
    return 0;
}

extern "C" int dr4sr_sasrec_fwd_bwd(const dr4sr_sasrec_plan* plan, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (!plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(launch_prep(plan, ws, 1, 1, s));
    return fwd_bwd_core(plan, ws, s);
}

static int fwd_bwd_weighted(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, void* stream, bool prepared) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (!mw || !mw->phi || !(mw->tau > 0.f) || !plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params)
        return DR4SR_E_ARG;
    if (plan->D != 64) return DR4SR_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (!prepared) RC(launch_prep(plan, ws, 1, 1, s));
    RC(forward_layers(plan, ws, 1, s, true));
    RC(launch_post_mid(plan, ws, 1, s, mw));
    RC(backward_layers(plan, ws, 1, 2, s, true, true));
    return 0;
}
extern "C" int dr4sr_sasrec_fwd_bwd_weighted(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, void* stream) {
    return fwd_bwd_weighted(plan, mw, stream, false);
}
// on a batch prepared by dr4sr_adam_step_prepare_next
extern "C" int dr4sr_sasrec_fwd_bwd_weighted_prepared(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, void* stream) {
    return fwd_bwd_weighted(plan, mw, stream, true);
}

extern "C" int dr4sr_sasrec_train_step(const dr4sr_sasrec_plan* plan, void* stream) {
    RC(dr4sr_sasrec_fwd_bwd(plan, stream));
    return dr4sr_adam_step(plan, stream);
}

// The two halves of a data-parallel step whose prep rides on the PREVIOUS step's optimizer launch (the all-reduce of plan->grads goes
// between them): fwd_bwd on a batch that is already prepared, and Adam + preparation of the next batch.
extern "C" int dr4sr_sasrec_fwd_bwd_prepared(const dr4sr_sasrec_plan* plan, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (!plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    return fwd_bwd_core(plan, ws, (hipStream_t)stream);
}
// ---- the step in two phases around the table bucket's all-reduce (see dp_split_layer above)
extern "C" int dr4sr_sasrec_grad_buckets(const dr4sr_sasrec_plan* plan, int64_t* bounds) {
    if (!plan) return DR4SR_E_ARG;
    dr4sr_sasrec_plan q = *plan;
    q.workspace = nullptr;
    if (q.B <= 0 || q.L <= 0 || q.n_layer <= 0 || q.n_layer > DR4SR_MAX_LAYERS) return DR4SR_E_ARG;
    if (check_shape(&q)) return DR4SR_E_SHAPE;
    Workspace ws;
    carve_workspace(&q, &ws);
    const bool two = dp_two_buckets(&q, ws);
    if (bounds) {
        bounds[0] = 0;
        bounds[1] = two ? ws.off[2] : ws.n_params + DR4SR_GRAD_TAIL;
        bounds[2] = ws.n_params + DR4SR_GRAD_TAIL;
    }
    return two ? 2 : 1;
}
extern "C" int dr4sr_sasrec_fwd_bwd_phase(const dr4sr_sasrec_plan* plan, int32_t prepared, int32_t phase, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (phase < 1 || phase > 2 || !plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (phase == 1 && !prepared) RC(launch_prep(plan, ws, 1, 1, s));
    return fwd_bwd_core(plan, ws, s, phase);
}
extern "C" int dr4sr_adam_step_prepare_next(const dr4sr_sasrec_plan* plan, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (!plan->grads || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    PrepArgs next;
    RC(make_prep_args(plan, ws, 1, &next));
    return launch_adam(plan, (hipStream_t)stream, &next);
}

// n consecutive training steps (consecutive batches of plan->perm when it is set): one k_prep for the first step, every optimizer
// launch but the last also prepares the step that follows it.
extern "C" int dr4sr_sasrec_train_steps(const dr4sr_sasrec_plan* plan, int32_t n_steps, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (n_steps <= 0 || !plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    // B <= 1024: the whole prep as an extra 256-thread workgroup of the optimizer launch (+2.7 % at 256, +1 % at 1024; -1 % at 2048);
    // above: the two-phase form (selection spread over the optimizer launch's workgroups, scan by the last one to finish)
    const bool nofuse_env = DR4SR_ENV("DR4SR_NO_PREP_FUSE") != nullptr;
    const bool nofuse = nofuse_env;
    hipStream_t s = (hipStream_t)stream;
    PrepArgs next;
    RC(make_prep_args(plan, ws, 1, &next));
    RC(launch_prep(plan, ws, 1, 1, s));
    for (int i = 0; i < n_steps; ++i) {
        RC(fwd_bwd_core(plan, ws, s));
        const bool last = i == n_steps - 1;
        if (nofuse && !last) { RC(launch_adam(plan, s)); RC(launch_prep(plan, ws, 1, 1, s)); }
        else RC(launch_adam(plan, s, last ? nullptr : &next));
    }
    return 0;
}

extern "C" int dr4sr_sasrec_encode(const dr4sr_sasrec_plan* plan, int32_t training, int32_t pooling, float* out,
                                   void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (!out || pooling < DR4SR_POOL_NONE || pooling > DR4SR_POOL_MEAN) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(launch_prep(plan, ws, training ? 1 : 0, 0, s));
    RC(forward_layers(plan, ws, training, s));
    return launch_unpack(plan, ws, ws.X[plan->n_layer], out, pooling == DR4SR_POOL_LAST ? 1 : pooling == DR4SR_POOL_MEAN ? 2 : 0, s);
}

extern "C" int dr4sr_sasrec_encode_bwd(const dr4sr_sasrec_plan* plan, int32_t training, int32_t pooling,
                                       const float* d_out, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (!d_out || !plan->grads || pooling < DR4SR_POOL_NONE || pooling > DR4SR_POOL_MEAN) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(launch_pack(plan, ws, d_out, ws.dX[plan->n_layer], pooling == DR4SR_POOL_LAST ? 1 : pooling == DR4SR_POOL_MEAN ? 2 : 0, s));
    return backward_layers(plan, ws, training, 0, s);
}

// ------------------------------------------------------------------------------------------------
// Measurement hook (bench.py / profiles): enqueue ONE kernel of the step so that its launch duration can
// be bracketed with HIP events on the caller's stream.  Uses whatever the last fwd_bwd left in the workspace.
extern "C" int dr4sr_sasrec_launch_kernel(const dr4sr_sasrec_plan* plan, int32_t kernel, int32_t layer, void* stream) {
    return dr4sr_sasrec_launch_kernel_weighted(plan, nullptr, kernel, layer, stream);
}
extern "C" int dr4sr_sasrec_launch_kernel_weighted(const dr4sr_sasrec_plan* plan, const dr4sr_meta_weighting* mw, int32_t kernel,
                                                   int32_t layer, void* stream) {
    Workspace ws;
    RC(get_ws(plan, &ws));
    if (layer < 0 || layer >= plan->n_layer) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (kernel) {
        case DR4SR_K_PREP: return launch_prep(plan, ws, 0, 0, s);
        case DR4SR_K_EMBED_FWD: return launch_embed_fwd(plan, ws, 1, s);
        case DR4SR_K_QKV_FWD: return launch_qkv_fwd(plan, ws, layer, s);
        case DR4SR_K_ATTN_FWD: return attn_fwd(plan, ws, layer, 1, s);
        case DR4SR_K_POST_FWD: return launch_post_fwd(plan, ws, layer, 1, s);
        case DR4SR_K_SCORE: return launch_score_packed(plan, ws, s);
        case DR4SR_K_TRANSPOSE: return launch_transpose_weights(plan, ws, s);
        case DR4SR_K_POST_BWD: return launch_post_bwd(plan, ws, layer, 1, s);
        case DR4SR_K_ATTN_BWD: return attn_bwd(plan, ws, layer, 1, s);
        case DR4SR_K_QKV_BWD: return launch_qkv_bwd(plan, ws, layer, s);
        case DR4SR_K_EMBED_BWD: return launch_embed_bwd(plan, ws, 1, s);
        case DR4SR_K_WGRAD: return launch_wgrad(plan, ws, 1, 1, s);
        case DR4SR_K_ADAM: return launch_adam(plan, s);
        case DR4SR_K_ZERO_GRADS: return launch_zero_grads(plan, ws.n_params, s);
        // the launches of the FUSED step (what dr4sr_sasrec_train_step really enqueues)
        case DR4SR_K_EMBQKV_FWD: return launch_embqkv_fwd(plan, ws, 1, s);
        case DR4SR_K_POST_MID: return launch_post_mid(plan, ws, 1, s, mw);
        case DR4SR_K_QKV_EMBED_BWD: return launch_qkv_embed_bwd(plan, ws, 1, s);
        case DR4SR_K_WGRAD_FUSED: return launch_wgrad(plan, ws, 1, 2, s, true, mw != nullptr);
        default: return DR4SR_E_ARG;
    }
}
