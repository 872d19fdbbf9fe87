// gru.hip — GRU4Rec target model (reference model/gru4rec.py:12-34, module/layers.py:117-136 = torch.nn.GRU(bias=False,
// batch_first, n_layer) + Linear(H -> D)) on gfx950, over ragged (packed) sequences.
//
//   x = dropout(E[idx])                                  packed gather (embed.hip, no position table)
//   per layer:  gi = in W_ih^T                           token-parallel MFMA GEMM (k_gemm, K-chunked through LDS)
//               recurrence over t                        k_gru_fwd: 16 sequences per workgroup, h in LDS, gh = h W_hh^T on
//                                                        v_mfma_f32_16x16x4_f32 with W_hh streamed from L2 every step
//   y = h_top W_out^T + b                                k_gemm
// torch GRU cell (gates r|z|n, no biases):  r = s(gi_r+gh_r)  z = s(gi_z+gh_z)  n = tanh(gi_n + r*gh_n)  h' = (1-z) n + z h.
// Rows >= seqlen never influence loss or gradients ('origin' pooling + causality of the recurrence), so only the valid
// prefix of every sequence is computed.  Backward = BPTT in k_gru_bwd (same tiling, dh W_hh on MFMA with W_hh read
// column-wise), weight gradients as token-parallel split GEMMs with fp32 atomics (k_wgrad64).
//
// W_hh (3H x H fp32 = 786 KB at H=256) fits neither one CU's LDS (160 KB) nor its registers (512 KB); streaming it from L2
// each step bounds a workgroup at ~10 us per step on the fp32 MFMA pipe.
#include "common.h"
#include "kernels.h"
#include "wgrad_bf.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#define GRU_MAX_LAYERS 4

// gru_coop.hip: multi-CU cooperative recurrence for small batches
int64_t gru_coop_words(int B, int H);
int launch_gru_rec_coop(const float* gi, const float* whh, const int* cu, float* r, float* z, float* n, float* ghn, float* hprev,
                        float* hout, const float* dhout, float* dgi, float* dgh, unsigned long long* xch, int* ctl, int B, int H, bool bwd,
                        hipStream_t s);
int64_t gru_xch_words(int B, int H, int L, int n_layer);
int launch_gru_wave(const GruWaveArgs& G, unsigned long long* xch, int* ctl, int B, int H, int L, bool bwd, hipStream_t s);

struct GruLayerWs {
    float* gi; float* r; float* z; float* n; float* ghn; float* hprev; float* hout;
    float* dgi; float* dgh;
};
struct GruWs {
    int64_t off_E, off_wih[GRU_MAX_LAYERS], off_whh[GRU_MAX_LAYERS], off_ow, off_ob, n_params;
    int Tmax;
    int* cu;
    int* tile_seq;                                            // [ceil(Tmax / 16) + 1] sequence slot of token 16 i (k_prep): search hint of the fused glue kernels
    float* X0; float* dX0; float* Y; float* dY; float* dH;     // dH: grad w.r.t. a layer's output rows [T,H]
    float* score_part;
    unsigned long long* xch; int* ctl;                        // cooperative recurrence (gru_coop.hip): granule area, control words
    // deterministic mode (DR4SR_DETERMINISTIC / train.deterministic; round 6): no fp32 atomics in the step.  The separate glue launches run at every
    // size (their y and d x rows live in global memory); the scorer leaves records de_rec [T] {target, negative, dpos, dneg} instead of its
    // atomics, k_gru_det_rows masks d x in place and writes the rows' ids idx32 [T], the item-table gradient is owner-computed in token order
    // (linear.hip launch_table_owner64); the 64 x 64 weight-gradient jobs store one block per token split (det_part [job][split][64 x 64 + 64])
    // and k_wgrad64_det_reduce adds them in split order
    bool det; float* det_part; int* idx32; int4* de_rec;
    GruLayerWs layer[GRU_MAX_LAYERS];
    int64_t bytes;
};

extern "C" int dr4sr_gru4rec_plan_sizeof(void) { return (int)sizeof(dr4sr_gru4rec_plan); }

// offsets: [0]=E, [1+2l]=weight_ih_l, [2+2l]=weight_hh_l, [1+2n]=out_w, [2+2n]=out_b
extern "C" int64_t dr4sr_gru4rec_param_layout(int32_t n_items, int32_t D, int32_t H, int32_t n_layer, int64_t* off) {
    int64_t o = 0;
    auto put = [&](int i, int64_t n) { if (off) off[i] = o; o += n; };
    put(0, (int64_t)n_items * D);
    for (int l = 0; l < n_layer; ++l) {
        put(1 + 2 * l, 3LL * H * (l == 0 ? D : H));
        put(2 + 2 * l, 3LL * H * H);
    }
    put(1 + 2 * n_layer, (int64_t)D * H);
    put(2 + 2 * n_layer, D);
    return o;
}

#define GRU_DET_SPLITS 32                      // deterministic mode: token splits of the weight-gradient launch (gru_backward's cap)
#define GRU_DET_STRIDE (64 * 64 + 64)          // ... floats per stored 64 x 64 block + its bias row
static bool gru_det() { const char* e = DR4SR_ENV("DR4SR_DETERMINISTIC"); return e && atoi(e) != 0; }

static int gru_check(const dr4sr_gru4rec_plan* p) {
    if (!p || p->abi_version != DR4SR_ABI_VERSION) return DR4SR_E_ARG;
    if (p->B <= 0 || p->L <= 0 || p->n_items < 2 || p->n_layer <= 0 || p->n_layer > GRU_MAX_LAYERS) return DR4SR_E_ARG;
    if (p->D != 64 || (p->H != 128 && p->H != 256) || p->L > 64) return DR4SR_E_SHAPE;
    if (!(p->p_drop >= 0.f && p->p_drop < 1.f) || !p->params || !p->state || !p->in_item_id || !p->seqlen) return DR4SR_E_ARG;
    return 0;
}

static void gru_carve(const dr4sr_gru4rec_plan* p, GruWs* ws) {
    const int64_t D = p->D, H = p->H, Tmax = (int64_t)p->B * p->L;
    int64_t off[3 + 2 * GRU_MAX_LAYERS];
    ws->n_params = dr4sr_gru4rec_param_layout(p->n_items, p->D, p->H, p->n_layer, off);
    ws->off_E = off[0];
    for (int l = 0; l < p->n_layer; ++l) { ws->off_wih[l] = off[1 + 2 * l]; ws->off_whh[l] = off[2 + 2 * l]; }
    ws->off_ow = off[1 + 2 * p->n_layer]; ws->off_ob = off[2 + 2 * p->n_layer];
    ws->Tmax = (int)Tmax;
    char* base = (char*)p->workspace;
    int64_t o = 0;
    auto take = [&](int64_t nfloat) -> float* {
        float* r = base ? (float*)(base + o) : nullptr;
        o += ((nfloat * 4 + 255) / 256) * 256;
        return r;
    };
    // control words and granules FIRST: their offsets must not move with B (they hold state that survives across calls)
    ws->ctl = (int*)take(4);
    {   // (an upper bound over every batch size up to B: a partial last batch runs in the workspace sized for the full one)
        const int64_t words = gru_xch_words(p->B, p->H, p->L, p->n_layer);
        ws->xch = words ? (unsigned long long*)take(2 * words) : nullptr;
    }
    ws->cu = (int*)take(p->B + 1);
    ws->tile_seq = (int*)take((Tmax + 15) / 16 + 1);
    ws->X0 = take(Tmax * D); ws->dX0 = take(Tmax * D); ws->Y = take(Tmax * D); ws->dY = take(Tmax * D); ws->dH = take(Tmax * H);
    ws->score_part = take(2LL * (p->B > (Tmax + 15) / 16 ? p->B : (Tmax + 15) / 16));     // per sequence, or per 16-token tile (k_gru_mid)
    for (int l = 0; l < p->n_layer; ++l) {
        GruLayerWs& w = ws->layer[l];
        w.gi = take(Tmax * 3 * H); w.r = take(Tmax * H); w.z = take(Tmax * H); w.n = take(Tmax * H); w.ghn = take(Tmax * H);
        w.hprev = take(Tmax * H); w.hout = take(Tmax * H); w.dgi = take(Tmax * 3 * H); w.dgh = take(Tmax * 3 * H);
    }
    ws->det = gru_det();
    ws->det_part = nullptr; ws->idx32 = nullptr; ws->de_rec = nullptr;
    if (ws->det) {
        int64_t jobs = (D / 64) * (H / 64);                   // as gru_backward builds them
        for (int l = 0; l < p->n_layer; ++l) jobs += (3 * H / 64) * ((l == 0 ? D : H) / 64) + (3 * H / 64) * (H / 64);
        ws->det_part = take(jobs * GRU_DET_SPLITS * GRU_DET_STRIDE);
        ws->idx32 = reinterpret_cast<int*>(take(Tmax));
        ws->de_rec = reinterpret_cast<int4*>(take(4 * Tmax));
    }
    ws->bytes = o;
}

extern "C" int64_t dr4sr_gru4rec_workspace_bytes(const dr4sr_gru4rec_plan* plan) {
    if (!plan || plan->B <= 0 || plan->L <= 0 || plan->n_layer <= 0 || plan->n_layer > GRU_MAX_LAYERS) return DR4SR_E_ARG;
    if (plan->D != 64 || (plan->H != 128 && plan->H != 256) || plan->L > 64) return DR4SR_E_SHAPE;       // as gru_check
    dr4sr_gru4rec_plan q = *plan;
    q.workspace = nullptr;
    GruWs ws;
    gru_carve(&q, &ws);
    return ws.bytes;
}

static int gru_ws(const dr4sr_gru4rec_plan* p, GruWs* ws) {
    int rc = gru_check(p);
    if (rc) return rc;
    if (!p->workspace) return DR4SR_E_ARG;
    gru_carve(p, ws);
    return ws->bytes > p->workspace_bytes ? DR4SR_E_WS : 0;
}

// ------------------------------------------------------------------------------------------------ generic packed-row GEMM
// C[T x N] = A[T x K] * op(W) (+ bias), 64 rows x 64*NTW columns per workgroup, K in chunks of 64 through LDS.
//   COLMODE = false: op(W) = W^T, W is [N][ldw]   (forward linear: x W^T)
//   COLMODE = true : op(W) = W,   W is [K][ldw]   (data gradient: dy W)
template <int NTW, bool COLMODE>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                              const float* __restrict__ bias, float* __restrict__ C, int ldc, int K,
                                              const int* __restrict__ state) {
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * 64;
    if (t0 >= T) return;
    const int n0 = blockIdx.y * 64 * NTW;
    float* As = smem;                                   // [64][68]
    f32x16 acc[NTW];
    acc_zero(acc);
    for (int k0 = 0; k0 < K; k0 += 64) {
        if (k0) lds_barrier();
        load_tile<64>(As, 68, A + k0, lda, t0, T);
        lds_barrier();
        if (COLMODE) mma_64xN_wT<64, NTW>(As, 68, W + (size_t)k0 * ldw + n0, ldw, acc);
        else mma_64xN_ld<64, NTW>(As, 68, W + (size_t)n0 * ldw + k0, ldw, acc);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, g = lane >> 5, rh = w & 1, cg = w >> 1;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int col = n0 + (cg + 2 * i) * 32 + r;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int t = t0 + rh * 32 + (q & 3) + 8 * (q >> 2) + 4 * g;
            if (t < T) C[(size_t)t * ldc + col] = acc[i][q] + bv;
        }
    }
}


// Latency variant of the data-gradient GEMM for small batches: C[T x N] = A[T x K] * W (W [K][ldw] read column-wise), 16-row
// tiles (v_mfma_f32_16x16x4) with the whole K of the A tile in LDS, 64 output columns per workgroup: T/16 x N/64 workgroups
// instead of T/64 x N/128 — the K = 3H = 768 gradients dgi W_ih otherwise run on ~40 workgroups for ~100 us each.
template <int K>
__global__ __launch_bounds__(256) void k_gemm16_col(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                    float* __restrict__ C, int ldc, const int* __restrict__ state) {
    constexpr int LDA = K + 4;
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * 16;
    if (t0 >= T) return;
    const int n0 = blockIdx.y * 64;
    float* As = smem;                                   // [16][LDA]
    load_tile_bm<16, K>(As, LDA, A, lda, t0, T);
    lds_barrier();
    TileAcc<16, 64> acc;
    tile_zero(acc);
    if constexpr (K <= 256) {                           // weight fragments up front where they fit the registers (K / 4 floats per lane)
        WFragC<K, 64> f;
        wfrag_load(f, W + n0, ldw);
        tile_mma_frag<16, K, 64>(As, LDA, f, acc);
    } else tile_mma_xw<16, K, 64>(As, LDA, W + n0, ldw, acc);
    tile_to_global<16, 64>(acc, C + n0, ldc, nullptr, t0, T);
}

// Split-K form of the same product for K = 3H (round 4): in k_gemm16_col a wave owns 16 output columns and walks the whole K as ONE
// dependent chain of K / 4 MFMAs (192 at K = 768: 20.7 us per launch for 0.33 GFLOP).  Here wave w owns the K quarter
// [w K / 4, (w + 1) K / 4) for all 64 columns — four independent accumulator chains of K / 16 MFMAs — and the four partial tiles meet
// in LDS; the sum leaves as one float4 per thread in the (row = tid / 16, columns 4 (tid % 16) ..) layout the row epilogues use.
template <int K>
__device__ __forceinline__ float4 gemm16_col_splitk(const float* __restrict__ As, int lda, const float* __restrict__ W, int ldw, float* __restrict__ red) {
    constexpr int KW = K / 4, KWQ = KW / 4, LDR = 68;
    static_assert(KWQ % 4 == 0, "K must be a multiple of 64");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kb = w * KW + g * KWQ;
#pragma unroll 3
    for (int c = 0; c < KWQ; c += 4) {
        const float4 a = ld4(As + r16 * lda + kb + c);
        const float* wp = W + (size_t)(kb + c) * ldw + r16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float b0 = wp[16 * i], b1 = wp[ldw + 16 * i], b2 = wp[2 * ldw + 16 * i], b3 = wp[3 * ldw + 16 * i];
            acc[i] = mfma16x4(a.x, b0, acc[i]); acc[i] = mfma16x4(a.y, b1, acc[i]);
            acc[i] = mfma16x4(a.z, b2, acc[i]); acc[i] = mfma16x4(a.w, b3, acc[i]);
        }
    }
    float* mine = red + w * 16 * LDR;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[(4 * g + q) * LDR + 16 * i + r16] = acc[i][q];
    lds_barrier();
    const int r = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
    const float4 p0 = ld4(red + r * LDR + c), p1 = ld4(red + (16 + r) * LDR + c), p2 = ld4(red + (32 + r) * LDR + c), p3 = ld4(red + (48 + r) * LDR + c);
    return make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
}
template <int K>
__global__ __launch_bounds__(256) void k_gemm16_col_sk(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                       float* __restrict__ C, int ldc, const int* __restrict__ state) {
    constexpr int LDA = K + 4;
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * 16;
    if (t0 >= T) return;
    const int n0 = blockIdx.y * 64;
    float* As = smem;                                   // [16][LDA]
    float* red = As + 16 * LDA;                         // [4][16][68]
    load_tile_bm<16, K>(As, LDA, A, lda, t0, T);
    lds_barrier();
    const float4 v = gemm16_col_splitk<K>(As, LDA, W + n0, ldw, red);
    const int r = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
    if (t0 + r < T) st4(C + (size_t)(t0 + r) * ldc + n0 + c, v);
}

// ... and of the forward linear C[T x N] = A[T x K] W^T + bias (W [N][ldw]): gi = x W_ih^T (K = D or H, N = 3H) and the output
// projection (K = H, N = D) ran on 21 x N/128 workgroups of 64 rows with K walked in 64-chunks (25 us for 0.5 GFLOP at B = 256).
template <int K>
__global__ __launch_bounds__(256) void k_gemm16_row(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                    const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                    const int* __restrict__ state) {
    constexpr int LDA = K + 4;
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * 16;
    if (t0 >= T) return;
    const int n0 = blockIdx.y * 64;
    float* As = smem;                                   // [16][LDA]
    WFragT<K, 64> f;                                    // the workgroup's 64 x K weight rows: every fragment requested up front, their
    wfrag_load(f, W + (size_t)n0 * ldw, ldw);           // L2 round trips overlap each other and the A tile's (else K / 16 dependent ones)
    load_tile_bm<16, K>(As, LDA, A, lda, t0, T);
    lds_barrier();
    TileAcc<16, 64> acc;
    tile_zero(acc);
    tile_mma_frag<16, K, 64>(As, LDA, f, acc);
    tile_to_global<16, 64>(acc, C + n0, ldc, bias ? bias + n0 : nullptr, t0, T);
}

// ------------------------------------------------------------------------------------------------ fused glue launches (round 4)
// At B = 256 the GRU4Rec step is 15 launches of which 12 are 5-20 us of glue around the three recurrences (profiles/
// round3_timeline_gru4rec.txt: 130 of 493 us).  Three fusions in the latency regime (16-row tiles), each removing launch boundaries and
// a round trip of a [T, D] tensor through global memory; DR4SR_GRU_NOFUSE_GLUE restores the separate launches (cross-check, tested):
//   k_gru_embed_gi : x = drop(E[idx]) gathered straight into the A tile + gi_1 = x W_ih1^T             (was k_embed_fwd + k_gemm16_row<64>)
//   k_gru_mid      : y = h_top W_out^T + b  ->  scorer / BCE / d y (model/basemodel.py:204-214, model/loss_func.py:9-38)  ->  d h_top = d y W_out
//                                                                                                     (was k_gemm16_row<256> + k_score_packed + k_gemm16_col<64>)
//   k_gru_dx_embed : d x = d gi_1 W_ih1 times the embedding dropout mask, scattered into d E            (was k_gemm16_col<768> + k_embed_bwd)
struct GruGlueArgs {
    const float* E; float* dE; const int64_t* idx; const int64_t* target; const int64_t* rows; const int* cu; const int* tile_seq;
    int64_t* neg_item; int sample_neg; float* part; const int* state; uint64_t seed; float p; int training; int n_items, B, L;
    const float* W; const float* bias;                        // the launch's weight (W_ih1 / W_out) and bias (W_out only)
    const float* in; float* out0; float* out1; float* out2;   // kernel-specific tensors, see each kernel
};

// out0 = X0 [T,64] (written by the blockIdx.y == 0 workgroups), out1 = gi_1 [T, 3H]; W = W_ih1 [3H][64].  grid (tiles, 3H / 64)
__global__ __launch_bounds__(256) void k_gru_embed_gi(const GruGlueArgs A, const int N3) {
    constexpr int D = 64, LDA = D + 4;
    const int T = A.state[DR4SR_STATE_T], t0 = blockIdx.x * 16;
    if (t0 >= T) return;
    const int n0 = blockIdx.y * 64;
    float* As = smem;                                   // [16][LDA]
    WFragT<D, 64> f;
    wfrag_load(f, A.W + (size_t)n0 * D, D);
    {
        const int r = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4, t = t0 + r;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            const int b = find_seq_from(A.cu, A.B, t, A.tile_seq[t0 >> 4]), pos = t - A.cu[b];
            const int64_t row = A.rows ? A.rows[b] : b;
            int64_t id = A.idx[row * A.L + pos];
            id = id < 0 ? 0 : (id >= A.n_items ? A.n_items - 1 : id);
            o = ld4(A.E + id * D + c);
            if (A.training && A.p > 0.f) {
                const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
                const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * A.L + pos) * D + c);
                o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
            }
            if (blockIdx.y == 0) st4(A.out0 + (size_t)t * D + c, o);
        }
        st4(As + r * LDA + c, o);
    }
    lds_barrier();
    TileAcc<16, 64> acc;
    tile_zero(acc);
    tile_mma_frag<16, D, 64>(As, LDA, f, acc);
    tile_to_global<16, 64>(acc, A.out1 + n0, N3, nullptr, t0, T);
}

// in = h_top [T,H]; out0 = d y [T,64] (kept for the weight gradient), out1 = d h_top [T,H]; W = W_out [64][H], bias = b_out.  grid (tiles)
template <int H>
__global__ __launch_bounds__(256) void k_gru_mid(const GruGlueArgs A) {
    constexpr int D = 64, LDA = H + 4, LDY = D + 4;
    const int T = A.state[DR4SR_STATE_T], t0 = blockIdx.x * 16, tile = blockIdx.x;
    if (t0 >= T) return;
    float* As = smem;                                   // [16][LDA]  h tile
    float* Ys = As + 16 * LDA;                          // [16][LDY]  y tile, then d y tile
    float* red = Ys + 16 * LDY;                         // [8]
    WFragT<H, 64> fy;
    wfrag_load(fy, A.W, H);
    load_tile_bm<16, H>(As, LDA, A.in, H, t0, T);
    lds_barrier();
    {
        TileAcc<16, 64> acc;
        tile_zero(acc);
        tile_mma_frag<16, H, 64>(As, LDA, fy, acc);
        tile_to_lds<16, 64>(acc, Ys, LDY, A.bias);
    }
    WFragC<D, H> fd;                                    // d h = d y W_out: requested now, the round trip overlaps the scorer
    wfrag_load(fd, A.W, H);
    lds_barrier();
    // ---- scorer: 16 lanes per row (model/basemodel.py:204-214, model/loss_func.py:9-38; same draws / terms as k_score_packed)
    {
        const int r = threadIdx.x >> 4, sub = threadIdx.x & 15, c = sub * 4, t = t0 + r;
        const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], 0.f);
        float lsum = 0.f, cnt = 0.f;
        float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            const int b = find_seq_from(A.cu, A.B, t, A.tile_seq[t0 >> 4]), pos = t - A.cu[b], n = A.cu[b + 1] - A.cu[b];
            const int64_t row = A.rows ? A.rows[b] : b;
            const int64_t tgt = A.target[row * A.L + pos];
            int64_t ng;
            if (A.sample_neg) {
                ng = sample_neg_id(rk, (uint64_t)b * A.L + pos, A.n_items);
                if (sub == 0) A.neg_item[(size_t)b * A.L + pos] = ng;
            } else {
                ng = A.neg_item[(size_t)b * A.L + pos];
            }
            ng = ng < 0 ? 0 : (ng >= A.n_items ? A.n_items - 1 : ng);
            if (tgt > 0 && tgt < A.n_items) {
                const float4 q = ld4(Ys + r * LDY + c), ep = ld4(A.E + tgt * D + c), en = ld4(A.E + ng * D + c);
                const float sp = group16_sum(q.x * ep.x + q.y * ep.y + q.z * ep.z + q.w * ep.w);
                const float sn = group16_sum(q.x * en.x + q.y * en.y + q.z * en.z + q.w * en.w);
                const float dpos = -sigmoid_f(-sp), dneg = sigmoid_f(sn);
                if (sub == 0) { lsum += softplus_f(-sp) + softplus_f(sn); cnt += 1.f; }
                dz = make_float4(dpos * ep.x + dneg * en.x, dpos * ep.y + dneg * en.y, dpos * ep.z + dneg * en.z, dpos * ep.w + dneg * en.w);
                float* gp = A.dE + tgt * D + c;
                float* gn = A.dE + ng * D + c;
                unsafeAtomicAdd(gp, dpos * q.x); unsafeAtomicAdd(gp + 1, dpos * q.y); unsafeAtomicAdd(gp + 2, dpos * q.z); unsafeAtomicAdd(gp + 3, dpos * q.w);
                unsafeAtomicAdd(gn, dneg * q.x); unsafeAtomicAdd(gn + 1, dneg * q.y); unsafeAtomicAdd(gn + 2, dneg * q.z); unsafeAtomicAdd(gn + 3, dneg * q.w);
            }
            st4(A.out0 + (size_t)t * D + c, dz);
            if (pos == n - 1) {                           // positions behind the sequence (zero query): loss terms only, negatives drawn as the unfused scorer draws them
                for (int l = n + sub; l < A.L; l += 16) {
                    const int64_t tl = A.target[row * A.L + l];
                    if (A.sample_neg) A.neg_item[(size_t)b * A.L + l] = sample_neg_id(rk, (uint64_t)b * A.L + l, A.n_items);
                    if (tl > 0 && tl < A.n_items) { lsum += 2.0f * 0.69314718055994530942f; cnt += 1.f; }
                }
            }
        }
        st4(Ys + r * LDY + c, dz);                       // (the lanes of a row have all read their q by now: group16_sum)
        cnt = wave_sum(cnt); lsum = wave_sum(lsum);
        if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = cnt; red[2 * (threadIdx.x >> 6) + 1] = lsum; }
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        A.part[2 * tile] = (red[0] + red[2]) + (red[4] + red[6]);
        A.part[2 * tile + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    TileAcc<16, H> acc2;
    tile_zero(acc2);
    tile_mma_frag<16, D, H>(Ys, LDY, fd, acc2);
    tile_to_global<16, H>(acc2, A.out1, H, nullptr, t0, T);
}

// in = d gi_1 [T, 3H]; W = W_ih1 [3H][64]; scatter into dE.  grid (tiles)
template <int K>
__global__ __launch_bounds__(256) void k_gru_dx_embed(const GruGlueArgs A) {
    constexpr int D = 64, LDA = K + 4;
    const int T = A.state[DR4SR_STATE_T], t0 = blockIdx.x * 16;
    if (t0 >= T) return;
    float* As = smem;                                   // [16][LDA]
    float* red = As + 16 * LDA;                         // [16][68] the product's tile
    load_tile_bm<16, K>(As, LDA, A.in, K, t0, T);
    lds_barrier();
    {                                                   // (the split-K form measured slower HERE — 19.0 against 15.1 us: 83 workgroups, one per
        TileAcc<16, 64> acc;                            //  tile, each streaming the whole 196 KB of W_ih1 either way — and faster for the
        tile_zero(acc);                                 //  N = 256 product between the BPTT launches: 20.7 -> 18.6 us)
        tile_mma_xw<16, K, 64>(As, LDA, A.W, D, acc);
        tile_to_lds<16, 64>(acc, red, 68, nullptr);
    }
    lds_barrier();
    const int r = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4, t = t0 + r;
    float4 g = ld4(red + r * 68 + c);
    if (t >= T) return;
    const int b = find_seq_from(A.cu, A.B, t, A.tile_seq[t0 >> 4]), pos = t - A.cu[b];
    const int64_t row = A.rows ? A.rows[b] : b;
    const int64_t id = A.idx[row * A.L + pos];
    if (id <= 0 || id >= A.n_items) return;             // padding_idx / out of range: no gradient row
    if (A.training && A.p > 0.f) {
        const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
        const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * A.L + pos) * D + c);
        g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
    }
    float* d = A.dE + id * D + c;
    unsafeAtomicAdd(d, g.x); unsafeAtomicAdd(d + 1, g.y); unsafeAtomicAdd(d + 2, g.z); unsafeAtomicAdd(d + 3, g.w);
}

// the fused glue applies to what the 16-row latency GEMMs apply to (small batches), D = 64
static bool gru_glue_fused(const dr4sr_gru4rec_plan* p, int Tmax) {
    return !DR4SR_ENV("DR4SR_GRU_NOFUSE_GLUE") && !DR4SR_XENV("DR4SR_GRU_GEMM64") && !at_scale(Tmax) && p->D == 64 && (p->H == 128 || p->H == 256)
        && !gru_det();                                      // deterministic mode: the separate launches (the owners read y and d x from global memory)
}

static int launch_gemm(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int K, int N,
                       bool colmode, int Tmax, const int* state, hipStream_t s) {
    const size_t lds = sizeof(float) * 64 * 68;
    dim3 blk(256);
    const bool no16 = DR4SR_XENV("DR4SR_GRU_GEMM64") != nullptr;          // cross-check switch: 64-row tiles everywhere
    if (!no16 && !at_scale(Tmax) && N % 64 == 0 && ((colmode && !bias && K == 64) || (!colmode && (K == 64 || K == 128 || K == 256)))) {
        const size_t l16 = sizeof(float) * 16 * (K + 4);
        dim3 grid((Tmax + 15) / 16, N / 64);
        if (colmode) hipLaunchKernelGGL(k_gemm16_col<64>, grid, blk, l16, s, A, lda, W, ldw, C, ldc, state);
        else if (K == 64) hipLaunchKernelGGL(k_gemm16_row<64>, grid, blk, l16, s, A, lda, W, ldw, bias, C, ldc, state);
        else if (K == 128) hipLaunchKernelGGL(k_gemm16_row<128>, grid, blk, l16, s, A, lda, W, ldw, bias, C, ldc, state);
        else hipLaunchKernelGGL(k_gemm16_row<256>, grid, blk, l16, s, A, lda, W, ldw, bias, C, ldc, state);
        return DR4SR_LAUNCH_CHECK();
    }
    if (colmode && !bias && !at_scale(Tmax) && (K == 768 || K == 384) && N % 64 == 0) {      // small batch, K = 3H: latency tiles
        const size_t l16 = sizeof(float) * 16 * (K + 4);
        dim3 grid((Tmax + 15) / 16, N / 64);
        const size_t lsk = l16 + sizeof(float) * 4 * 16 * 68;
        if (DR4SR_XENV("DR4SR_GRU_NO_SPLITK")) {             // cross-check: one K chain per wave (round 2's kernel)
            if (K == 768) { big_lds(k_gemm16_col<768>, l16); hipLaunchKernelGGL(k_gemm16_col<768>, grid, blk, l16, s, A, lda, W, ldw, C, ldc, state); }
            else { big_lds(k_gemm16_col<384>, l16); hipLaunchKernelGGL(k_gemm16_col<384>, grid, blk, l16, s, A, lda, W, ldw, C, ldc, state); }
        } else if (K == 768) { big_lds(k_gemm16_col_sk<768>, lsk); hipLaunchKernelGGL(k_gemm16_col_sk<768>, grid, blk, lsk, s, A, lda, W, ldw, C, ldc, state); }
        else { big_lds(k_gemm16_col_sk<384>, lsk); hipLaunchKernelGGL(k_gemm16_col_sk<384>, grid, blk, lsk, s, A, lda, W, ldw, C, ldc, state); }
        return DR4SR_LAUNCH_CHECK();
    }
    if (N % 128 == 0) {
        dim3 grid((Tmax + 63) / 64, N / 128);
        if (colmode) hipLaunchKernelGGL((k_gemm<2, true>), grid, blk, lds, s, A, lda, W, ldw, bias, C, ldc, K, state);
        else hipLaunchKernelGGL((k_gemm<2, false>), grid, blk, lds, s, A, lda, W, ldw, bias, C, ldc, K, state);
    } else {
        dim3 grid((Tmax + 63) / 64, N / 64);
        if (colmode) hipLaunchKernelGGL((k_gemm<1, true>), grid, blk, lds, s, A, lda, W, ldw, bias, C, ldc, K, state);
        else hipLaunchKernelGGL((k_gemm<1, false>), grid, blk, lds, s, A, lda, W, ldw, bias, C, ldc, K, state);
    }
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ generic weight gradient
// dW[n0..n0+64)[k0..k0+64) += sum_t G[t][gcol+..] X[t][xcol+..]   (one 64x64 output tile per job, token tiles strided over
// gridDim.x workgroups, fp32 atomics at the end);  db[n0..] += column sums of G when db != NULL.
struct Wg64Job { const float* G; int ldg; int gcol; const float* X; int ldx; int xcol; float* dW; int ldw; float* db; };
struct Wg64Mat { const float* G; const float* X; float* dW; float* db; int ldg, NG, ldx, KX, start; };   // dW is [NG][KX]
struct Wg64Args {
    Wg64Mat mat[2 * GRU_MAX_LAYERS + 1]; int nmat; const int* state;
    // one extra job (blockIdx.y == njobs, x == 0): tail[0..1] += the scorer's per-sequence (count, loss) partials, tail[2] += the
    // cooperative recurrence's error word — was a launch of its own (k_sum_score_part, 4.7 us of a 0.49 ms step)
    int njobs; const float* score_part; float* tail; int nscore; const int* err_word;
    int score_tiles;                                          // 1: one (count, loss) pair per 16-token tile (k_gru_mid) instead of per sequence
    float* det;                                               // deterministic mode: [job][split][GRU_DET_STRIDE] stored blocks instead of atomics (NULL: off)
};

// tail[0..1] += sum of the scorer's per-sequence (count, loss) partials
// tail[2] += the cooperative recurrence's sticky error word (the optimizer's poison word, csrc/step.hip k_adam)
__device__ __forceinline__ void sum_score_part(const float* __restrict__ part, float* __restrict__ tail, int B, const int* __restrict__ err_word,
                                               float* red) {                 // red: 512 floats of LDS
    float c = 0.f, l = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) { c += part[2 * b]; l += part[2 * b + 1]; }
    red[threadIdx.x] = c; red[256 + threadIdx.x] = l;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { tail[0] += red[0]; tail[1] += red[256]; if (err_word && *err_word) tail[2] += 1.0f; }
}

// The same launch with every 64 x 64 job on the bf16 matrix cores as a 3-term split (wgrad_bf.h, round 4): the fp32 form runs 1.7 GFLOP
// of v_mfma_f32_32x32x2_f32 at B = 256 — 30 us, a third of the fp32 matrix peak — and is the largest launch of the step after the
// recurrences; max-norm error of the split 5e-6 of the fp32 product (DR4SR_WGRAD_F32 = the fp32 kernel, tested).
__global__ __launch_bounds__(256) void k_wgrad64_bf(const Wg64Args A) {
    if ((int)blockIdx.y == A.njobs) {
        if (blockIdx.x == 0) sum_score_part(A.score_part, A.tail, A.score_tiles && A.nscore ? (A.state[DR4SR_STATE_T] + 15) / 16 : A.nscore, A.err_word, smem);
        return;
    }
    int mi = 0;
    for (int m = 1; m < A.nmat; ++m) if ((int)blockIdx.y >= A.mat[m].start) mi = m;
    const Wg64Mat M = A.mat[mi];
    const int local = blockIdx.y - M.start, nkb = M.KX / 64, n0 = (local / nkb) * 64, k0 = (local % nkb) * 64;
    WgradJob J;
    J.G = M.G; J.ldg = M.ldg; J.gcol = n0; J.X = M.X + k0; J.ldx = M.ldx;
    J.dW = M.dW + (size_t)n0 * M.KX + k0; J.ldw = M.KX; J.db = (M.db && k0 == 0) ? M.db + n0 : nullptr;
    wgrad_body_bf<64, 64>(J, A.state, A.det ? A.det + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * GRU_DET_STRIDE : nullptr);
}
// deterministic mode: dW / db += the stored blocks of the gw token splits, in split order (thread = one element of one job's block)
__global__ __launch_bounds__(256) void k_wgrad64_det_reduce(const Wg64Args A, const int gw) {
    const int job = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= GRU_DET_STRIDE) return;
    int mi = 0;
    for (int m = 1; m < A.nmat; ++m) if (job >= A.mat[m].start) mi = m;
    const Wg64Mat M = A.mat[mi];
    const int local = job - M.start, nkb = M.KX / 64, n0 = (local / nkb) * 64, k0 = (local % nkb) * 64;
    if (e >= 64 * 64 && !(M.db && k0 == 0)) return;
    const int T = A.state[DR4SR_STATE_T], nsplit = min(gw, (T + 63) / 64);
    const float sum = det_sum(A.det + (size_t)job * gw * GRU_DET_STRIDE + e, nsplit, (size_t)GRU_DET_STRIDE);
    if (e < 64 * 64) M.dW[(size_t)(n0 + e / 64) * M.KX + k0 + e % 64] += sum;
    else M.db[n0 + e - 64 * 64] += sum;
}
// deterministic mode: g = d x[t] * the embedding dropout mask, in place, + the row's id (0: no gradient row) for the owners
__global__ __launch_bounds__(256) void k_gru_det_rows(float* __restrict__ dX, int* __restrict__ idx32, const int64_t* __restrict__ idx,
                                                      const int64_t* __restrict__ rows, const int* __restrict__ cu, int B, int L, int n_items,
                                                      const int* __restrict__ state, uint64_t seed, float p, int training) {
    constexpr int D = 64;
    const int T = state[DR4SR_STATE_T], t = blockIdx.x * 16 + (threadIdx.x >> 4), c = (threadIdx.x & 15) * 4;
    if (t >= T) return;
    const int b = find_seq(cu, B, t), pos = t - cu[b];
    const int64_t row = rows ? rows[b] : b;
    const int64_t id = idx[row * L + pos];
    if (training && p > 0.f) {
        const RngKey rk = make_rng(seed, (uint32_t)state[DR4SR_STATE_RNGSTEP], p);
        const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * L + pos) * D + c);
        float4 g = ld4(dX + (size_t)t * D + c);
        g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
        st4(dX + (size_t)t * D + c, g);
    }
    if ((threadIdx.x & 15) == 0) idx32[t] = (id > 0 && id < n_items) ? (int)id : 0;
}

__global__ __launch_bounds__(256) void k_wgrad64(const Wg64Args A) {
    if ((int)blockIdx.y == A.njobs) {
        if (blockIdx.x == 0) sum_score_part(A.score_part, A.tail, A.score_tiles && A.nscore ? (A.state[DR4SR_STATE_T] + 15) / 16 : A.nscore, A.err_word, smem);
        return;
    }
    int mi = 0;
    for (int m = 1; m < A.nmat; ++m) if ((int)blockIdx.y >= A.mat[m].start) mi = m;
    const Wg64Mat M = A.mat[mi];
    Wg64Job J;
    {
        const int local = blockIdx.y - M.start, nkb = M.KX / 64, n0 = (local / nkb) * 64, k0 = (local % nkb) * 64;
        J.G = M.G; J.ldg = M.ldg; J.gcol = n0; J.X = M.X; J.ldx = M.ldx; J.xcol = k0;
        J.dW = M.dW + (size_t)n0 * M.KX + k0; J.ldw = M.KX; J.db = (M.db && k0 == 0) ? M.db + n0 : nullptr;
    }
    const int T = A.state[DR4SR_STATE_T], ntiles = (T + 63) / 64;
    float* Gs = smem;                                   // [64][64]
    float* Xs = smem + 64 * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, g = lane >> 5;
    const int nt = w >> 1, kt = w & 1;                  // wave -> 32x32 tile of the 64x64 block
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float bsum = 0.f;
    float4 gq[4], xq[4];
    auto issue = [&](int tt) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u, row = i >> 4, c = (i & 15) * 4, t = tt * 64 + row;
            gq[u] = t < T ? ld4(J.G + (size_t)t * J.ldg + J.gcol + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            xq[u] = t < T ? ld4(J.X + (size_t)t * J.ldx + J.xcol + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int tt = blockIdx.x;
    if (tt >= ntiles) return;                            // worst-case grid: a workgroup without a token tile has nothing to add
    issue(tt);
    for (; tt < ntiles; tt += gridDim.x) {
        lds_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u, row = i >> 4, c = (i & 15) * 4;
            st4(Gs + row * 64 + c, gq[u]);
            st4(Xs + row * 64 + c, xq[u]);
        }
        lds_barrier();
        if (tt + (int)gridDim.x < ntiles) issue(tt + gridDim.x);
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int t = 2 * s + g;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[t * 64 + nt * 32 + r], Xs[t * 64 + kt * 32 + r], acc, 0, 0, 0);
        }
        if (J.db && threadIdx.x < 64) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
            for (int t = 0; t < 64; t += 2) { s0 += Gs[t * 64 + threadIdx.x]; s1 += Gs[(t + 1) * 64 + threadIdx.x]; }
            bsum += s0 + s1;
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = nt * 32 + (q & 3) + 8 * (q >> 2) + 4 * g;
        unsafeAtomicAdd(J.dW + (size_t)row * J.ldw + kt * 32 + r, acc[q]);
    }
    if (J.db && threadIdx.x < 64) unsafeAtomicAdd(J.db + threadIdx.x, bsum);
}

// ------------------------------------------------------------------------------------------------ recurrence
struct GruRecArgs {
    const float* gi; const float* whh; const int* cu;
    float* r; float* z; float* n; float* ghn; float* hprev; float* hout;          // saved per token [T,H]
    const float* dhout; float* dgi; float* dgh;                                   // backward
    int B;
};

__device__ __forceinline__ f32x4 mfma16g(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// 16 sequences per workgroup, 8 waves; wave w owns hidden units [w*H/8, (w+1)*H/8) of all three gates, so the gate math of a
// unit is register-local (C-layout fragments of the r, z and n tiles coincide: lane = (sequence 4g+q, unit l&15)).
template <int H>
__global__ __launch_bounds__(512) void k_gru_fwd(const GruRecArgs A) {
    constexpr int UT = H / 128;                           // 16-unit tiles per wave
    constexpr int LDH = H + 4, KQ = H / 4;                // lane group g contracts k in [g*KQ, (g+1)*KQ)
    float* hb = smem;                                     // [2][16][LDH]
    int* meta = reinterpret_cast<int*>(hb + 2 * 16 * LDH);   // [16] t0, [16] n
    const int b0 = blockIdx.x * 16;
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    for (int i = threadIdx.x; i < 2 * 16 * LDH; i += 512) hb[i] = 0.f;
    lds_barrier();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    int tq[4], nq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { tq[q] = meta[4 * g + q]; nq[q] = meta[16 + 4 * g + q]; }
    f32x4 hreg[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) hreg[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nmax; ++t) {
        const float* hc = hb + (t & 1) * 16 * LDH + l16 * LDH + g * KQ;       // A operand rows: sequence l16
        float* hn = hb + ((t + 1) & 1) * 16 * LDH;
        // the input-projection terms of this step do not depend on the recurrence: request them before the MFMA loops so that
        // their L2 round trip hides behind ~10 us of matrix work instead of sitting between the loop and the gate math
        float gir[UT][4], giz[UT][4], gin[UT][4];
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool act = t < nq[q];
                const float* gip = A.gi + (size_t)(tq[q] + (act ? t : 0)) * 3 * H + w * (H / 8) + u * 16 + l16;
                gir[u][q] = act ? gip[0] : 0.f; giz[u][q] = act ? gip[H] : 0.f; gin[u][q] = act ? gip[2 * H] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int unit = w * (H / 8) + u * 16 + l16;                     // B operand row / C column of this lane
            f32x4 acc[3];
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) acc[q3] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* wr = A.whh + (size_t)unit * H + g * KQ;
#pragma unroll 8
            for (int c = 0; c < KQ; c += 4) {
                const float4 a = ld4(hc + c);
#pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) {
                    const float4 bv = ld4(wr + (size_t)q3 * H * H + c);
                    acc[q3] = mfma16g(a.x, bv.x, acc[q3]);
                    acc[q3] = mfma16g(a.y, bv.y, acc[q3]);
                    acc[q3] = mfma16g(a.z, bv.z, acc[q3]);
                    acc[q3] = mfma16g(a.w, bv.w, acc[q3]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool act = t < nq[q];
                const size_t tok = (size_t)(tq[q] + t);
                float hnew = hreg[u][q];
                if (act) {
                    const float rr = sigm(gir[u][q] + acc[0][q]);
                    const float zz = sigm(giz[u][q] + acc[1][q]);
                    const float nn = tanh_f(gin[u][q] + rr * acc[2][q]);
                    const float hold = hreg[u][q];
                    hnew = (1.0f - zz) * nn + zz * hold;
                    const size_t o = tok * H + unit;
                    A.r[o] = rr; A.z[o] = zz; A.n[o] = nn; A.ghn[o] = acc[2][q]; A.hprev[o] = hold; A.hout[o] = hnew;
                }
                hreg[u][q] = hnew;
                hn[(4 * g + q) * LDH + unit] = hnew;
            }
        }
        lds_barrier();
    }
}

// BPTT.  Per step (t descending): dh = dhout[t] + carry;  gate derivatives for the wave's own units (registers);
// dgh rows -> LDS;  carry' = dh*z + dgh W_hh  (MFMA, W_hh read column-wise: B[k=j][n=unit] = W_hh[j][unit]).
template <int H>
__global__ __launch_bounds__(512) void k_gru_bwd(const GruRecArgs A) {
    constexpr int UT = H / 128, G3 = 3 * H, LDG = G3 + 4, KQ = G3 / 4;
    float* db = smem;                                     // [16][LDG]  dgh rows of the current step
    int* meta = reinterpret_cast<int*>(db + 16 * LDG);
    const int b0 = blockIdx.x * 16;
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    lds_barrier();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    int tq[4], nq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { tq[q] = meta[4 * g + q]; nq[q] = meta[16 + 4 * g + q]; }
    f32x4 carry[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) carry[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // saved activations of a step are independent of the carry: they are loaded one step AHEAD (registers), so their L2
    // round trip overlaps the previous step's MFMA loop instead of heading every step's serial chain
    float sv[UT][4][6];
    auto load_saved = [&](int t) {
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool act = t >= 0 && t < nq[q];
                const size_t o = (size_t)(tq[q] + (act ? t : 0)) * H + w * (H / 8) + u * 16 + l16;
                sv[u][q][0] = act ? A.dhout[o] : 0.f; sv[u][q][1] = act ? A.r[o] : 0.f; sv[u][q][2] = act ? A.z[o] : 0.f;
                sv[u][q][3] = act ? A.n[o] : 0.f; sv[u][q][4] = act ? A.ghn[o] : 0.f; sv[u][q][5] = act ? A.hprev[o] : 0.f;
            }
    };
    load_saved(nmax - 1);
    for (int t = nmax - 1; t >= 0; --t) {
        f32x4 dhz[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int unit = w * (H / 8) + u * 16 + l16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool act = t < nq[q];
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, keep = 0.f;
                if (act) {
                    const size_t tok = (size_t)(tq[q] + t);
                    const float dh = sv[u][q][0] + carry[u][q];
                    const float rr = sv[u][q][1], zz = sv[u][q][2], nn = sv[u][q][3], gh = sv[u][q][4], hp = sv[u][q][5];
                    dn = dh * (1.0f - zz) * (1.0f - nn * nn);
                    dz = dh * (hp - nn) * zz * (1.0f - zz);
                    dr = dn * gh * rr * (1.0f - rr);
                    dnr = dn * rr;
                    keep = dh * zz;
                    float* gp = A.dgi + tok * G3 + unit;
                    gp[0] = dr; gp[H] = dz; gp[2 * H] = dn;
                    float* hp2 = A.dgh + tok * G3 + unit;
                    hp2[0] = dr; hp2[H] = dz; hp2[2 * H] = dnr;
                }
                dhz[u][q] = keep;
                float* row = db + (4 * g + q) * LDG + unit;
                row[0] = dr; row[H] = dz; row[2 * H] = dnr;
            }
        }
        lds_barrier();
        load_saved(t - 1);
        const float* ar = db + l16 * LDG + g * KQ;                      // A operand: dgh row of sequence l16
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int unit = w * (H / 8) + u * 16 + l16;                // output column (hidden unit) of this lane
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* wc = A.whh + (size_t)(g * KQ) * H + unit;
#pragma unroll 8
            for (int c = 0; c < KQ; c += 4) {
                const float4 a = ld4(ar + c);
                acc = mfma16g(a.x, wc[(size_t)c * H], acc);
                acc = mfma16g(a.y, wc[(size_t)(c + 1) * H], acc);
                acc = mfma16g(a.z, wc[(size_t)(c + 2) * H], acc);
                acc = mfma16g(a.w, wc[(size_t)(c + 3) * H], acc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) carry[u][q] = dhz[u][q] + acc[q];
        }
        lds_barrier();
    }
}

static int launch_gru_rec(const GruRecArgs& A, int H, bool bwd, hipStream_t s, unsigned long long* xch = nullptr, int* ctl = nullptr) {
    {   // small batches: 8 CUs per group of 16 sequences, W_hh slices resident in LDS, per-step exchange inside the launch
        const int rc = launch_gru_rec_coop(A.gi, A.whh, A.cu, A.r, A.z, A.n, A.ghn, A.hprev, A.hout, A.dhout, A.dgi, A.dgh, xch, ctl, A.B, H,
                                           bwd, s);
        if (rc != -100) return rc;
    }
    dim3 grid((A.B + 15) / 16), blk(512);
    if (!bwd) {
        const size_t lds = sizeof(float) * 2 * 16 * (H + 4) + 32 * sizeof(int);
        if (H == 256) hipLaunchKernelGGL(k_gru_fwd<256>, grid, blk, lds, s, A);
        else hipLaunchKernelGGL(k_gru_fwd<128>, grid, blk, lds, s, A);
    } else {
        const size_t lds = sizeof(float) * 16 * (3 * H + 4) + 32 * sizeof(int);
        if (H == 256) { big_lds(k_gru_bwd<256>, lds); hipLaunchKernelGGL(k_gru_bwd<256>, grid, blk, lds, s, A); }
        else hipLaunchKernelGGL(k_gru_bwd<128>, grid, blk, lds, s, A);
    }
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ orchestration
static GruWaveArgs wave_args(const dr4sr_gru4rec_plan* p, const GruWs& ws) {
    GruWaveArgs G{};
    G.gi1 = ws.layer[0].gi; G.wih2 = p->params + ws.off_wih[1]; G.cu = ws.cu; G.dhout = ws.dH;
    for (int l = 0; l < 2; ++l) {
        const GruLayerWs& w = ws.layer[l];
        G.whh[l] = p->params + ws.off_whh[l];
        G.r[l] = w.r; G.z[l] = w.z; G.n[l] = w.n; G.ghn[l] = w.ghn; G.hprev[l] = w.hprev; G.hout[l] = w.hout; G.dgi[l] = w.dgi; G.dgh[l] = w.dgh;
    }
    return G;
}

static GruGlueArgs glue_args(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training) {
    GruGlueArgs A{};
    A.E = p->params + ws.off_E; A.dE = p->grads ? p->grads + ws.off_E : nullptr; A.idx = p->in_item_id; A.target = p->item_id; A.rows = p->rows;
    A.cu = ws.cu; A.tile_seq = ws.tile_seq; A.neg_item = p->neg_item; A.sample_neg = p->sample_neg; A.part = ws.score_part; A.state = p->state;
    A.seed = p->seed; A.p = p->p_drop; A.training = training; A.n_items = p->n_items; A.B = p->B; A.L = p->L;
    return A;
}

// x = drop(E[idx]) and gi_1 = x W_ih1^T: one launch in the latency regime (k_gru_embed_gi), else gather + GEMM
static int gru_embed_gi(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training, hipStream_t s) {
    const int D = p->D, H = p->H;
    if (gru_glue_fused(p, ws.Tmax)) {
        GruGlueArgs A = glue_args(p, ws, training);
        A.W = p->params + ws.off_wih[0]; A.out0 = ws.X0; A.out1 = ws.layer[0].gi;
        hipLaunchKernelGGL(k_gru_embed_gi, dim3((ws.Tmax + 15) / 16, 3 * H / 64), dim3(256), sizeof(float) * 16 * (D + 4), s, A, 3 * H);
        return DR4SR_LAUNCH_CHECK();
    }
    RC(launch_embed_fwd_raw(p->params + ws.off_E, nullptr, p->in_item_id, p->rows, ws.cu, ws.X0, p->B, p->L, D, p->n_items, p->state,
                            p->seed, p->p_drop, training, s));
    return launch_gemm(ws.X0, D, p->params + ws.off_wih[0], D, nullptr, ws.layer[0].gi, 3 * H, D, 3 * H, false, ws.Tmax, p->state, s);
}

// the step's prep: [batch selection,] prefix scan of the lengths, tile -> sequence hints, RNG step, zeroed gradient
static int gru_prep_args(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training, bool select, PrepArgs* out) {
    PermSel sel{nullptr, 0, 0, 0, nullptr};
    if (p->perm && select) {                                // batch selection only in the call that starts a training step
        if (!p->rows || !p->perm_counter || p->n_perm <= 0) return DR4SR_E_ARG;
        sel = PermSel{p->perm, p->n_perm, p->perm_stride, p->perm_offset, p->perm_counter};
    }
    *out = PrepArgs{p->seqlen, p->rows, ws.cu, p->state, p->B, p->L, training ? 1 : 0, sel, ws.tile_seq, nullptr, nullptr};
    return 0;
}
static int gru_prep(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training, int zero_grads, hipStream_t s) {
    PrepArgs P;
    RC(gru_prep_args(p, ws, training, training && zero_grads, &P));
    return launch_prep_raw_hints(p->seqlen, p->rows, ws.cu, p->state, p->B, p->L, P.bump_rng, zero_grads ? p->grads : nullptr,
                                 ws.n_params + DR4SR_GRAD_TAIL, ws.tile_seq, P.sel, s);
}

static int gru_forward(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training, int zero_grads, hipStream_t s, bool skip_out = false,
                       bool prepared = false) {
    const int D = p->D, H = p->H;
    if (!prepared) RC(gru_prep(p, ws, training, zero_grads, s));
    RC(gru_embed_gi(p, ws, training, s));
    const float* in = ws.X0;
    int K = D;
    if (p->n_layer == 2) {                                  // both layers' recurrences in one launch (gru_coop.hip, layer wavefront)
        const int rc = launch_gru_wave(wave_args(p, ws), ws.xch, ws.ctl, p->B, H, p->L, false, s);
        if (rc != -100) {
            RC(rc);
            if (skip_out) return 0;                         // (the fused mid launch forms y itself)
            return launch_gemm(ws.layer[1].hout, H, p->params + ws.off_ow, H, p->params + ws.off_ob, ws.Y, D, H, D, false, ws.Tmax, p->state, s);
        }
    }
    for (int l = 0; l < p->n_layer; ++l) {
        const GruLayerWs& w = ws.layer[l];
        if (l > 0)                                          // (gi_1 is there already: gru_embed_gi)
            RC(launch_gemm(in, K, p->params + ws.off_wih[l], K, nullptr, w.gi, 3 * H, K, 3 * H, false, ws.Tmax, p->state, s));
        GruRecArgs A{};
        A.gi = w.gi; A.whh = p->params + ws.off_whh[l]; A.cu = ws.cu; A.r = w.r; A.z = w.z; A.n = w.n; A.ghn = w.ghn;
        A.hprev = w.hprev; A.hout = w.hout; A.B = p->B;
        RC(launch_gru_rec(A, H, false, s, ws.xch, ws.ctl));
        in = w.hout;
        K = H;
    }
    if (skip_out) return 0;
    return launch_gemm(in, H, p->params + ws.off_ow, H, p->params + ws.off_ob, ws.Y, D, H, D, false, ws.Tmax, p->state, s);
}

// output projection + scorer / BCE + d h_top in one launch (training step, latency regime); returns -100 when it does not apply
static int gru_mid(const dr4sr_gru4rec_plan* p, const GruWs& ws, hipStream_t s) {
    if (!gru_glue_fused(p, ws.Tmax)) return -100;
    const int H = p->H;
    GruGlueArgs A = glue_args(p, ws, 1);
    A.W = p->params + ws.off_ow; A.bias = p->params + ws.off_ob; A.in = ws.layer[p->n_layer - 1].hout; A.out0 = ws.dY; A.out1 = ws.dH;
    const size_t lds = sizeof(float) * (16 * (H + 4) + 16 * 68 + 8);
    dim3 grid((ws.Tmax + 15) / 16);
    if (H == 256) hipLaunchKernelGGL(k_gru_mid<256>, grid, dim3(256), lds, s, A);
    else hipLaunchKernelGGL(k_gru_mid<128>, grid, dim3(256), lds, s, A);
    return DR4SR_LAUNCH_CHECK();
}

// d x = d gi_1 W_ih1 and the embedding-table scatter: one launch in the latency regime (k_gru_dx_embed), else GEMM + scatter
static int gru_dx_embed(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training, hipStream_t s, int with_score = 0) {
    const int D = p->D, H = p->H;
    if (gru_glue_fused(p, ws.Tmax)) {
        GruGlueArgs A = glue_args(p, ws, training);
        A.W = p->params + ws.off_wih[0]; A.in = ws.layer[0].dgi;
        dim3 grid((ws.Tmax + 15) / 16);
        if (H == 256) {
            const size_t lds = sizeof(float) * (16 * (768 + 4) + 16 * 68);
            big_lds(k_gru_dx_embed<768>, lds);
            hipLaunchKernelGGL(k_gru_dx_embed<768>, grid, dim3(256), lds, s, A);
        } else {
            const size_t lds = sizeof(float) * (16 * (384 + 4) + 16 * 68);
            hipLaunchKernelGGL(k_gru_dx_embed<384>, grid, dim3(256), lds, s, A);
        }
        return DR4SR_LAUNCH_CHECK();
    }
    RC(launch_gemm(ws.layer[0].dgi, 3 * H, p->params + ws.off_wih[0], D, nullptr, ws.dX0, D, 3 * H, D, true, ws.Tmax, p->state, s));
    if (ws.det) {                                           // d E: the scorer's records (fused step only) + the masked d x rows, owner-computed in token order
        hipLaunchKernelGGL(k_gru_det_rows, dim3((ws.Tmax + 15) / 16), dim3(256), 0, s, ws.dX0, ws.idx32, p->in_item_id, p->rows, ws.cu, p->B, p->L,
                           p->n_items, p->state, p->seed, p->p_drop, training);
        return launch_table_owner64(p->state, with_score ? ws.de_rec : nullptr, ws.idx32, ws.Y, ws.dX0, p->grads + ws.off_E, p->n_items, s);
    }
    return launch_embed_bwd_raw(ws.dX0, p->in_item_id, p->rows, ws.cu, p->grads + ws.off_E, nullptr, p->B, p->L, D, p->n_items, p->state,
                                p->seed, p->p_drop, training, s);
}

// mid_done: d h_top is in ws.dH already and the scorer's partials are per token tile (gru_mid)
static int gru_backward(const dr4sr_gru4rec_plan* p, const GruWs& ws, int training, int with_score, hipStream_t s, bool mid_done = false) {
    const int D = p->D, H = p->H, nl = p->n_layer;
    // dH_top = dY W_out
    if (!mid_done) RC(launch_gemm(ws.dY, D, p->params + ws.off_ow, H, nullptr, ws.dH, H, D, H, true, ws.Tmax, p->state, s));
    int l_top = nl - 1;
    if (nl == 2) {                                          // both BPTTs in one launch; dh_1 = dgi_2 W_ih2 is formed inside it
        const int rc = launch_gru_wave(wave_args(p, ws), ws.xch, ws.ctl, p->B, H, p->L, true, s);
        if (rc != -100) {
            RC(rc);
            l_top = -1;
        }
    }
    for (int l = l_top; l >= 0; --l) {
        const GruLayerWs& w = ws.layer[l];
        GruRecArgs A{};
        A.whh = p->params + ws.off_whh[l]; A.cu = ws.cu; A.r = w.r; A.z = w.z; A.n = w.n; A.ghn = w.ghn; A.hprev = w.hprev;
        A.dhout = ws.dH; A.dgi = w.dgi; A.dgh = w.dgh; A.B = p->B;
        RC(launch_gru_rec(A, H, true, s, ws.xch, ws.ctl));
        // d(input of this layer) = dgi W_ih
        if (l > 0) RC(launch_gemm(w.dgi, 3 * H, p->params + ws.off_wih[l], H, nullptr, ws.dH, H, 3 * H, H, true, ws.Tmax, p->state, s));
    }
    RC(gru_dx_embed(p, ws, training, s, with_score));
    // weight gradients: one 64x64 output tile per job (jobs decoded in-kernel from per-matrix descriptors)
    Wg64Args WA{};
    int nj = 0;
    auto add = [&](const float* G, int ldg, int NG, const float* X, int ldx, int KX, float* dW, float* dbias) {
        Wg64Mat& M = WA.mat[WA.nmat++];
        M.G = G; M.X = X; M.dW = dW; M.db = dbias; M.ldg = ldg; M.NG = NG; M.ldx = ldx; M.KX = KX; M.start = nj;
        nj += (NG / 64) * (KX / 64);
    };
    float* Gd = p->grads;
    for (int l = 0; l < nl; ++l) {
        const GruLayerWs& w = ws.layer[l];
        add(w.dgi, 3 * H, 3 * H, l == 0 ? ws.X0 : ws.layer[l - 1].hout, l == 0 ? D : H, l == 0 ? D : H, Gd + ws.off_wih[l], nullptr);
        add(w.dgh, 3 * H, 3 * H, w.hprev, H, H, Gd + ws.off_whh[l], nullptr);
    }
    add(ws.dY, D, D, ws.layer[nl - 1].hout, H, H, Gd + ws.off_ow, Gd + ws.off_ob);
    WA.state = p->state;
    const int ntiles = (ws.Tmax + 63) / 64;
    const int gwf = DR4SR_XENV("DR4SR_GRU_WGRAD_GW") ? atoi(DR4SR_XENV("DR4SR_GRU_WGRAD_GW")) : 0;     // tuning knob
    // token-tile splits per 64x64 output tile: every split ends in 4 096 atomics, so fewer, longer splits at small batches (B = 256: 6
    // instead of 12 is worth 0.8 % of the step; 2 is too few workgroups)
    // (bf16x3 jobs: the MFMA phase is 2.7x shorter, the 4 096-atomic tail is not — half as many splits: B = 256 6 -> 3 is worth 0.7 % of the step)
    const bool wg_f32 = DR4SR_ENV("DR4SR_WGRAD_F32") != nullptr;
    const int gdiv = wg_f32 ? 32 : 64, glo = wg_f32 ? 6 : 3;
    int gw = gwf > 0 ? gwf : (ntiles / gdiv > glo ? (ntiles / gdiv > 32 ? 32 : ntiles / gdiv) : glo);
    if (gw > ntiles) gw = ntiles;
    // with_score == 0 (autograd path): no scorer partials to add, the extra job only forwards the recurrence's error word
    WA.njobs = nj; WA.score_part = ws.score_part; WA.tail = p->grads + ws.n_params; WA.nscore = with_score ? p->B : 0; WA.err_word = ws.ctl + 2;
    WA.score_tiles = mid_done ? 1 : 0;
    if (ws.det) {                                           // stored blocks exist for the bf16x3 launch only
        if (wg_f32 || gw > GRU_DET_SPLITS) return DR4SR_E_SHAPE;
        WA.det = ws.det_part;
    }
    if (wg_f32) hipLaunchKernelGGL(k_wgrad64, dim3(gw, nj + 1), dim3(256), sizeof(float) * 2 * 64 * 64, s, WA);
    else hipLaunchKernelGGL(k_wgrad64_bf, dim3(gw, nj + 1), dim3(256), sizeof(float) * 2 * 64 * 64, s, WA);
    if (ws.det) hipLaunchKernelGGL(k_wgrad64_det_reduce, dim3((GRU_DET_STRIDE + 255) / 256, nj), dim3(256), 0, s, WA, gw);
    return DR4SR_LAUNCH_CHECK();
}

// everything of a training step between the prep and the optimizer (prepared: the previous optimizer launch ran this step's prep)
static int gru_fwd_bwd_core(const dr4sr_gru4rec_plan* plan, const GruWs& ws, hipStream_t s, bool prepared);
extern "C" int dr4sr_gru4rec_fwd_bwd(const dr4sr_gru4rec_plan* plan, void* stream) {
    GruWs ws;
    RC(gru_ws(plan, &ws));
    if (!plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    return gru_fwd_bwd_core(plan, ws, (hipStream_t)stream, false);
}
static int gru_fwd_bwd_core(const dr4sr_gru4rec_plan* plan, const GruWs& ws, hipStream_t s, bool prepared) {
    const bool fused_mid = gru_glue_fused(plan, ws.Tmax);
    RC(gru_forward(plan, ws, 1, 1, s, fused_mid, prepared));
    if (fused_mid) {
        RC(gru_mid(plan, ws, s));
        return gru_backward(plan, ws, 1, 1, s, true);
    }
    RC(launch_score_packed_raw(ws.Y, plan->params + ws.off_E, plan->grads + ws.off_E, ws.dY, plan->item_id, plan->rows, ws.cu,
                               plan->neg_item, plan->sample_neg, ws.score_part, plan->state, plan->seed, plan->n_items, plan->B,
                               plan->L, plan->D, s, ws.det ? ws.de_rec : nullptr));
    return gru_backward(plan, ws, 1, 1, s);
}

extern "C" int dr4sr_gru4rec_adam_step(const dr4sr_gru4rec_plan* plan, void* stream) {
    if (!plan || plan->abi_version != DR4SR_ABI_VERSION || !plan->params || !plan->grads || !plan->state) return DR4SR_E_ARG;
    return launch_adam_flat(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->state, plan->lr,
                            plan->beta1, plan->beta2, plan->adam_eps, plan->weight_decay, (hipStream_t)stream, plan->loss_log,
                            plan->perm ? plan->perm_counter : nullptr, nullptr, plan->optimizer);
}

extern "C" int dr4sr_gru4rec_train_step(const dr4sr_gru4rec_plan* plan, void* stream) {
    RC(dr4sr_gru4rec_fwd_bwd(plan, stream));
    return dr4sr_gru4rec_adam_step(plan, stream);
}

// n consecutive training steps (consecutive batches of plan->perm when it is set): ONE prep launch for the first step, every
// optimizer launch but the last also prepares the step that follows it (k_adam's extra workgroup: batch selection, prefix scan, hints,
// RNG step; the gradient words are zeroed as they are consumed) — dr4sr_sasrec_train_steps' contract.  A prep launch per step was
// 7 us of the 0.47 ms step at B = 256.
extern "C" int dr4sr_gru4rec_train_steps(const dr4sr_gru4rec_plan* plan, int32_t n_steps, void* stream) {
    GruWs ws;
    RC(gru_ws(plan, &ws));
    if (n_steps <= 0 || !plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    PrepArgs next;
    RC(gru_prep_args(plan, ws, 1, true, &next));
    const bool fuse = DR4SR_ENV("DR4SR_NO_PREP_FUSE") == nullptr;
    RC(gru_prep(plan, ws, 1, 1, s));
    for (int i = 0; i < n_steps; ++i) {
        RC(gru_fwd_bwd_core(plan, ws, s, i == 0 || fuse));      // (step 0: the launch above; later steps: the previous optimizer launch)
        const bool last = i == n_steps - 1;
        RC(launch_adam_flat(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->state, plan->lr, plan->beta1,
                            plan->beta2, plan->adam_eps, plan->weight_decay, s, plan->loss_log, plan->perm ? plan->perm_counter : nullptr,
                            (last || !fuse) ? nullptr : &next, plan->optimizer));
    }
    return 0;
}

extern "C" int dr4sr_gru4rec_encode(const dr4sr_gru4rec_plan* plan, int32_t training, int32_t pooling, float* out, void* stream) {
    GruWs ws;
    RC(gru_ws(plan, &ws));
    if (!out || pooling < DR4SR_POOL_NONE || pooling > DR4SR_POOL_LAST) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(gru_forward(plan, ws, training, 0, s));
    return launch_unpack_raw(ws.Y, ws.cu, out, plan->B, plan->L, plan->D, pooling == DR4SR_POOL_LAST, s);
}

extern "C" int dr4sr_gru4rec_encode_bwd(const dr4sr_gru4rec_plan* plan, int32_t training, int32_t pooling, const float* d_out,
                                        void* stream) {
    GruWs ws;
    RC(gru_ws(plan, &ws));
    if (!d_out || !plan->grads || pooling < DR4SR_POOL_NONE || pooling > DR4SR_POOL_LAST) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(launch_pack_raw(d_out, ws.cu, ws.dY, plan->B, plan->L, plan->D, pooling == DR4SR_POOL_LAST, s));
    return gru_backward(plan, ws, training, 0, s);
}

// ------------------------------------------------------------------------------------------------
// Measurement hook (bench.py, include/dr4sr_hip_hooks.h): enqueue ONE kernel of the GRU4Rec step on the state the last fwd_bwd left
// in the workspace, so that its launch duration can be bracketed with HIP events on the caller's stream.
extern "C" int dr4sr_gru4rec_launch_kernel(const dr4sr_gru4rec_plan* plan, int32_t kernel, int32_t layer, void* stream) {
    GruWs ws;
    RC(gru_ws(plan, &ws));
    if (layer < 0 || layer >= plan->n_layer) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int D = plan->D, H = plan->H;
    const GruLayerWs& w = ws.layer[layer];
    GruRecArgs A{};
    A.whh = plan->params + ws.off_whh[layer]; A.cu = ws.cu; A.r = w.r; A.z = w.z; A.n = w.n; A.ghn = w.ghn; A.hprev = w.hprev; A.B = plan->B;
    switch (kernel) {
        case DR4SR_GK_REC_FWD: A.gi = w.gi; A.hout = w.hout; return launch_gru_rec(A, H, false, s, ws.xch, ws.ctl);
        case DR4SR_GK_REC_BWD: A.dhout = ws.dH; A.dgi = w.dgi; A.dgh = w.dgh; return launch_gru_rec(A, H, true, s, ws.xch, ws.ctl);
        case DR4SR_GK_GEMM_IN: {
            const float* in = layer == 0 ? ws.X0 : ws.layer[layer - 1].hout;
            const int K = layer == 0 ? D : H;
            return launch_gemm(in, K, plan->params + ws.off_wih[layer], K, nullptr, w.gi, 3 * H, K, 3 * H, false, ws.Tmax, plan->state, s);
        }
        case DR4SR_GK_WAVE_FWD: { const int rc = launch_gru_wave(wave_args(plan, ws), ws.xch, ws.ctl, plan->B, H, plan->L, false, s); return rc == -100 ? DR4SR_E_ARG : rc; }
        case DR4SR_GK_WAVE_BWD: { const int rc = launch_gru_wave(wave_args(plan, ws), ws.xch, ws.ctl, plan->B, H, plan->L, true, s); return rc == -100 ? DR4SR_E_ARG : rc; }
        default: return DR4SR_E_ARG;
    }
}
