// linear.hip — token-tile (64 packed tokens per workgroup) fused GEMM kernels of the SASRec layer,
// fp32 on v_mfma_f32_32x32x2_f32, activations staged through LDS:
//   k_qkv_fwd   : qkv = x W_in^T + b_in
//   k_post_fwd  : ctx -> out_proj -> dropout -> +x -> LayerNorm1 -> linear1 -> GELU -> dropout
//                 -> linear2 -> dropout -> +y -> LayerNorm2            (one launch, y/h stay in LDS)
//   k_post_bwd  : the exact reverse chain (LN2 bwd, linear2/GELU/linear1 data grads, LN1 bwd,
//                 out_proj data grad), LayerNorm affine grads by atomics
//   k_qkv_bwd   : dx = dqkv W_in + du1
//   k_wgrad     : all weight/bias gradients of all layers, split over token tiles, atomics at the end
//
// Reference arithmetic: torch.nn.TransformerEncoderLayer as configured at model/sasrec.py:21-30
// (post-norm, gelu(erf), batch_first), called at model/sasrec.py:65-68.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#define STAMP(i) do { if (A.stamps && blockIdx.x == 0 && threadIdx.x == 0) A.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)

extern __shared__ __attribute__((aligned(16))) float smem[];

// ------------------------------------------------------------------------------------------------
// transposed weight copies for the data-gradient GEMMs (dX = dY W  ==  dY (W^T)^T in "x W^T" form)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ params, float* __restrict__ wT,
                                                   int64_t o_in, int64_t o_out, int64_t o_w1, int64_t o_w2,
                                                   int64_t layer_stride, int64_t wT_stride, int D, int F) {
    const int m = blockIdx.y, layer = blockIdx.z;
    const float* src;
    float* dst = wT + layer * wT_stride;
    int R, C;                                             // src is [R][C], dst is [C][R]
    if (m == 0) { src = params + o_in + layer * layer_stride; R = 3 * D; C = D; }
    else if (m == 1) { src = params + o_out + layer * layer_stride; R = D; C = D; dst += 3 * D * D; }
    else if (m == 2) { src = params + o_w1 + layer * layer_stride; R = F; C = D; dst += 4 * D * D; }
    else { src = params + o_w2 + layer * layer_stride; R = D; C = F; dst += 4 * D * D + D * F; }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < R * C; i += gridDim.x * 256) {
        const int r = i / C, c = i % C;
        dst[c * R + r] = src[i];
    }
}

int launch_transpose_weights(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s) {
    const int64_t layer_stride = p->n_layer > 1 ? ws.off[2 + 12] - ws.off[2] : 0;
    hipLaunchKernelGGL(k_transpose, dim3(16, 4, p->n_layer), dim3(256), 0, s, p->params, ws.wT, poff(ws, 0, P_IN_W),
                       poff(ws, 0, P_OUT_W), poff(ws, 0, P_W1), poff(ws, 0, P_W2), layer_stride, ws.wT_stride, p->D, p->F);
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// bf16 hi | lo images of every layer's four weight matrices, as stored and transposed (common.h WSplit): what the bf16x3 tile GEMMs of
// the d = 128 at-scale step read as their B operand.  One launch per forward pass (the optimizer has just rewritten the weights);
// 98 k elements per layer.  DR4SR_TILE_F32: the fp32 MFMA tile GEMMs (cross-check).
int tile_rows(const Workspace& ws);
bool tile_bf3(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    return ws.wsplit != nullptr && ws.scale && p->D == 128 && tile_rows(ws) == 32 && !DR4SR_ENV("DR4SR_TILE_F32") && !DR4SR_ENV("DR4SR_NO_FUSE");
}
__global__ __launch_bounds__(256) void k_wsplit(const float* __restrict__ params, unsigned short* __restrict__ img, int64_t o_in, int64_t o_out,
                                                int64_t o_w1, int64_t o_w2, int64_t layer_stride, int E, int D, int F) {
    const int layer = blockIdx.y;
    unsigned short* base = img + (size_t)layer * 4 * E;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256)
        wsplit_elem(params, base, e, o_in, o_out, o_w1, o_w2, layer * layer_stride, E, D, F);      // common.h
}
int launch_wsplit(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s) {
    const int64_t layer_stride = p->n_layer > 1 ? ws.off[2 + 12] - ws.off[2] : 0;
    hipLaunchKernelGGL(k_wsplit, dim3(96, p->n_layer), dim3(256), 0, s, p->params, ws.wsplit, poff(ws, 0, P_IN_W), poff(ws, 0, P_OUT_W),
                       poff(ws, 0, P_W1), poff(ws, 0, P_W2), layer_stride, (int)ws.wsplit_E, p->D, p->F);
    return DR4SR_LAUNCH_CHECK();
}
// latency forms at d = 128: the forward GEMMs of k_post_fwd / k_post_mid read their prefetched weight fragments from the fragment-major fp32
// image (common.h wfrag_load_img) that the first launch of the forward pass writes — k_embqkv_fwd<16, 128>, or k_wfrag_write where that kernel
// does not run (DR4SR_NO_FUSE).  Measured on one box, B = 256: k_post_fwd 47.1 -> 36.5 us, k_post_mid 63.7 -> 58.8, the writer + 0.5 us:
// step 0.2369 -> 0.2234 ms (before round 5's other d = 128 work).  At d = 64 the same images gained 1.0 us in k_post_fwd and cost 0.75 us in
// k_embqkv_fwd: not used there.  No run-time switch between the two load forms inside the kernels: at d = 64 such a switch cost the step 11 %.
bool wfrag_img_on(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    return p->D == 128 && p->F == 128 && ws.wfrag != nullptr && tile_rows(ws) == 16;
}
// the layer's split-weight block, or NULL (fp32 MFMA tile GEMMs).  (The embedding-stage kernels take F = 128 for the block geometry: the
// only d = 128 shape, step.hip check_shape.)
int launch_wsplit_raw(const float* params, unsigned short* img, int64_t o_in, int64_t o_out, int64_t o_w1, int64_t o_w2, int64_t layer_stride,
                      int E, int D, int F, int n_layer, hipStream_t s) {
    hipLaunchKernelGGL(k_wsplit, dim3(96, n_layer), dim3(256), 0, s, params, img, o_in, o_out, o_w1, o_w2, layer_stride, E, D, F);
    return DR4SR_LAUNCH_CHECK();
}
static const unsigned short* wsplit_of(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer) {
    return (tile_bf3(p, ws) && layer >= 0 && layer < p->n_layer) ? ws.wsplit + (size_t)layer * 4 * ws.wsplit_E : nullptr;
}

// ------------------------------------------------------------------------------------------------
template <int BM, int D>
__global__ __launch_bounds__(256) void k_qkv_fwd(const float* __restrict__ X, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ QKV,
                                                 const int* __restrict__ state) {
    constexpr int N = 3 * D, LDA = D + 4;
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * BM;
    if (t0 >= T) return;
    float* As = smem;
    load_tile_bm<BM, D>(As, LDA, X, D, t0, T);
    lds_barrier();
    TileAcc<BM, N> acc;
    tile_zero(acc);
    tile_mma_xwT<BM, D, N>(As, LDA, W, D, acc);
    tile_to_global<BM, N>(acc, QKV, N, bias, t0, T);
}

// tile rows per workgroup for the token-tile kernels: 16 while the whole batch is small (latency regime: 4x shorter MFMA
// chains, every CU gets work), 32 at scale (occupancy regime: 4 workgroups per CU interleave their latency chains).
int latency_tmax() {
    const int v = DR4SR_XENV("DR4SR_LATENCY_TMAX") ? atoi(DR4SR_XENV("DR4SR_LATENCY_TMAX")) : 16384;
    return v;
}
int tile_rows(const Workspace& ws) {
    const int forced = DR4SR_ENV("DR4SR_BM") ? atoi(DR4SR_ENV("DR4SR_BM")) : 0;
    if (forced == 16 || forced == 32 || forced == 64) return forced;
    return ws.scale ? 32 : 16;
}
// large batches: the table-gradient scatter of the embedding stage (T x D fp32 atomics, ~55 G/s: 0.47 ms of the dense B=8192 step
// when it sits at the end of k_qkv_embed_bwd) runs as an extra job of k_wgrad, where it overlaps the MFMA-bound weight-gradient jobs
static bool scatter_in_wgrad(const Workspace& ws) {
    const bool off = DR4SR_ENV("DR4SR_SCATTER_INLINE") != nullptr || DR4SR_ENV("DR4SR_NO_FUSE") != nullptr;
    return !off && ws.scale_wg;
}
// large batches: the item-table gradient is NOT accumulated with fp32 atomics (scorer: 2 rows per token, embedding stage: 1) but
// summed row by row by owner workgroups inside k_wgrad (owner_job): deterministic, and ~55 us of a toys-shaped B = 8192 step
// cheaper.  DR4SR_DE_ATOMIC (cached until dr4sr_reload_env()) restores the atomics as a cross-check.
static bool de_owner_mode(const Workspace& ws) { return scatter_in_wgrad(ws) && !DR4SR_ENV("DR4SR_DE_ATOMIC"); }
// Owner geometry of the table gradient: G = 2^logG owners, the smallest power of two (>= 256) whose rows fit k_wgrad's LDS four
// times (one private copy per wave) plus the queues.  Shared by the scorer launch (tile_sort needs G) and the k_wgrad launch.
// at scale with d = 64 the weight-gradient GEMMs run as 64 x 64 blocks (k_wgrad_bf64): 32 KB of operand tiles per workgroup
static bool wgrad_sub64(const dr4sr_sasrec_plan* p) {
    return p->D == 64 && 4 + 2 * (p->F / 64) <= DR4SR_WGRAD_MAX_JOBS && !DR4SR_ENV("DR4SR_WGRAD_F32") && !DR4SR_XENV("DR4SR_WGRAD_WIDE");
}
static size_t wgrad_lds_base(const dr4sr_sasrec_plan* p) {          // (only the at-scale forms size their owners by it)
    const size_t D = p->D, F = p->F;
    size_t lds = sizeof(float) * 64 * (wgrad_sub64(p) && !DR4SR_XENV("DR4SR_WGRAD_LDS48") ? 2 * D : (D + F > 2 * D ? D + F : 2 * D));
    if (sizeof(float) * p->L * D > lds) lds = sizeof(float) * p->L * D;       // the scatter job's position-table accumulator
    return lds;
}
static int owner_logG(const dr4sr_sasrec_plan* p) {
    int logG = DR4SR_XENV("DR4SR_OWNER_LOGG") ? atoi(DR4SR_XENV("DR4SR_OWNER_LOGG")) : 8;      // tuning knob
    const size_t lds = wgrad_lds_base(p);
    while (sizeof(float) * 4 * (size_t)((p->n_items + (1 << logG) - 1) >> logG) * p->D + 4 * 16 * 4 * sizeof(int) > lds && logG < 20) ++logG;
    return logG;
}
// tiles hand their entries over sorted by owner (tile_sort / owner_job_sorted) when the offset tables are byte-sized and small:
// 32- or 64-row tiles, at most 1024 owners.  DR4SR_OWNER_SCAN (cached until dr4sr_reload_env()) keeps the scanning owners as a cross-check.
// the fused last-layer launch takes the wave-tile form (csrc/linear_wave.hip: 16-token tiles) — not for the MetaModel weighting
static bool wt_mid(const dr4sr_sasrec_plan* p, const Workspace& ws, bool meta) { return wave_tiles(p, ws) && wt_bwd_on() && !meta; }
static int mid_tile_rows(const dr4sr_sasrec_plan* p, const Workspace& ws, bool meta) { return wt_mid(p, ws, meta) ? 16 : tile_rows(ws); }
static bool owner_sorted(const dr4sr_sasrec_plan* p, const Workspace& ws, bool meta = false) {
    return de_owner_mode(ws) && (tile_rows(ws) != 16 || wt_mid(p, ws, meta)) && owner_logG(p) <= 10 && !DR4SR_ENV("DR4SR_OWNER_SCAN");
}
#define BM_DISPATCH(bm, CALL) do { if ((bm) == 16) { CALL(16); } else if ((bm) == 32) { CALL(32); } else { CALL(64); } } while (0)

int launch_qkv_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, hipStream_t s) {
    const int D = p->D, bm = tile_rows(ws);
    const size_t lds = sizeof(float) * bm * (D + 4);
    dim3 grid((ws.Tmax + bm - 1) / bm), blk(256);
    const float* W = p->params + poff(ws, layer, P_IN_W);
    const float* b = p->params + poff(ws, layer, P_IN_B);
#define QF(B_) do { if (D == 64) hipLaunchKernelGGL((k_qkv_fwd<B_, 64>), grid, blk, lds, s, ws.X[layer], W, b, ws.layer[layer].qkv, p->state); \
                    else hipLaunchKernelGGL((k_qkv_fwd<B_, 128>), grid, blk, lds, s, ws.X[layer], W, b, ws.layer[layer].qkv, p->state); } while (0)
    BM_DISPATCH(bm, QF);
#undef QF
    return DR4SR_LAUNCH_CHECK();
}


#include "attn_tile.h"
__host__ __device__ constexpr int post_lds_floats(int D, int F, int bm) {
    return bm * ((D + 4) + ((D + 4) + (F + 4) > 3 * D + 4 ? (D + 4) + (F + 4) : 3 * D + 4));     // R0+R2 must hold a [BM][3D+4] tile
}
// attention inside the tile kernels (attn_tile.h): its LDS area sits behind the post kernels' tiles and the scorer's scratch
__host__ __device__ constexpr int att_lds_off(int D, int F) { return post_lds_floats(D, F, 16) + 96; }
static size_t att_lds_bytes(int D) { return sizeof(float) * (D == 64 ? tattn::Lds<64>::floats : tattn::Lds<128>::floats); }
// Latency regime, two heads, L <= 64: no attention launches.  DR4SR_ATTN_SEPARATE: one workgroup per sequence as before (cross-check)
bool attn_tile_capable(const dr4sr_sasrec_plan* p) {
    if (DR4SR_ENV("DR4SR_ATTN_SEPARATE") || DR4SR_ENV("DR4SR_NO_FUSE") || DR4SR_ENV("DR4SR_ATTN_VALU")) return false;
    return p->H == 2 && p->L <= 64 && (p->D == 64 || p->D == 128);
}
bool attn_in_tile(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    return attn_tile_capable(p) && tile_rows(ws) == 16 && !ws.attn_split && !wave_tiles(p, ws);
}
bool attn_wave_on(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    return ws.attn_split && !ws.attn_tile_sa && p->H == 2 && p->L <= 64 && (p->D == 64 || p->D == 128) && !DR4SR_ENV("DR4SR_NO_FUSE")
           && !DR4SR_ENV("DR4SR_ATTN_LISTS") && !DR4SR_ENV("DR4SR_ATTN_NOSPLIT") && !DR4SR_ENV("DR4SR_ATTN_VALU");
}
// OPT-IN, experiments build only (DR4SR_ATTN_FOLD=1): built for the review's "attention + FFN as one block at scale", oracle-tested, measured
// SLOWER than the launch of its own it replaces — toys rows, same box, folded / launch of its own: B = 8 192 0.4665 / 0.4589 ms (k_wt_post_fwd
// 49.1 -> 69.6 us, k_wt_post_mid 102 -> 126 for a 17.6 us launch removed twice), B = 4 096 0.3067 / 0.3035, B = 2 048 0.2357 / 0.2234,
// B = 32 768 1.4745 / 1.4304 (profiles/round6_attn_fold_ab.txt).  The wave-tile kernels run ONE workgroup of 8 - 12 waves per CU (their LDS
// weight image) with about one tile per wave at B = 8 192: the attention's dependent chain (token words -> K | Q rows -> S -> V columns from
// global memory, no LDS left to stage them) is added to every wave's single tile at 3 waves per SIMD, where the launch of its own hides it
// behind 5 - 8 waves per SIMD.  NOTEBOOK round 6.
bool attn_fold_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    return DR4SR_XENV("DR4SR_ATTN_FOLD") && attn_wave_on(p, ws) && wave_tiles(p, ws) && p->D == 64;
}
bool tile_xcd_order(const dr4sr_sasrec_plan* p, const Workspace& ws) { return attn_in_tile(p, ws) && !DR4SR_ENV("DR4SR_TILE_ORDER_PLAIN"); }

// Layer-0 fusion: the token tile is gathered straight from the item/position tables (a3: sasrec.py:42-48,:61-66 —
// x = drop(E[idx] + P[pos]), 16 lanes per token, sequence slot by binary search in cu[]) into LDS, written once to X[0]
// (residual + weight-gradient input) and multiplied by W_in in the same launch.
// the dK | dV rows of a token tile (attn_tile.h adds into them with atomics two launches later)
template <int BM, int D>
__device__ __forceinline__ void zero_kv_rows(float* dqkv, const int t0, const int T) {
    constexpr int C4 = 2 * D / 4;
    for (int i = threadIdx.x; i < BM * C4; i += 256) {
        const int r = i / C4, c = (i % C4) * 4;
        if (t0 + r < T) st4(dqkv + (size_t)(t0 + r) * 3 * D + D + c, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}
// Deterministic latency form (kernels.h Workspace::det_lat): the dK | dV rows of token tile t0 / 16 of a layer, completed by the launch that
// consumes the layer's dqkv.  attn_tile.h's backward stored, per query tile q, the rows it contributes to the five key tiles of its window as
// partial blocks part[q][jt][16][2 D] (key tile q - 4 + jt), except the rows no other tile adds to, which it stored in dqkv itself.  Row tk of
// tile c = (that stored row, if tile-private) + the blocks of q = c .. c + 4 at jt = 4 - (q - c) whose window reaches back to tk (tk >= first
// token of the sequence of token 16 q), added in q order.  The sums replace the K | V columns of the staged tile Aq [16][ldq] and of dqkv (the
// weight-gradient launch reads them).  Every thread of the workgroup; the tile's rows are in LDS and a barrier has passed; ends with no barrier.
template <int D>
__device__ __forceinline__ void det_kv_rows(float* Aq, const int ldq, float* dqkv, const float* __restrict__ part, const int2* __restrict__ tok,
                                            const int t0, const int T) {
    constexpr int C4 = 2 * D / 4, PER = (16 * C4) / 256;
    const int tile = t0 >> 4;
    int first[5];                                         // first token of the sequence of token 16 (tile + q): the reach of query tile tile + q
#pragma unroll
    for (int q = 0; q < 5; ++q) first[q] = t0 + 16 * q < T ? tok[t0 + 16 * q].x : 0x7fffffff;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = threadIdx.x + 256 * u, r = i / C4, c = (i % C4) * 4, tk = t0 + r;
        if (tk >= T) continue;
        const int2 kw = tok[tk];
        const bool own = kw.x + ((kw.y >> 20) & 0x7f) <= t0 + 16;      // the sequence ends inside this tile (tk >= t0): attn_tile.h's plain store
        float4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tk >= first[q] && !(q == 0 && own)) v[q] = ld4(part + (((size_t)(tile + q) * 5 + (4 - q)) * 16 + r) * 2 * D + c);
        }
        float4 a = own ? ld4(Aq + r * ldq + D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 5; ++q) { a.x += v[q].x; a.y += v[q].y; a.z += v[q].z; a.w += v[q].w; }
        st4(Aq + r * ldq + D + c, a);
        st4(dqkv + (size_t)tk * 3 * D + D + c, a);
    }
}
// Fragment-major fp32 images (common.h wfrag_load_img) of every layer's four weight matrices, written by the FIRST launch of a forward pass for
// the launches behind it (latency forms at d = 128; F = 128): item j = one float4 of an image = the four k-consecutive values one lane feeds
// to one MFMA chunk; 24 576 items per layer, about one per thread of the launch.  The source offsets follow the flat parameter layout
// (dr4sr_sasrec_param_layout: E is the first tensor, so A.E is the buffer's base and A.W - A.E layer 0's in_proj offset).
// Only the AS-STORED orientation (the forward GEMMs' WFragT): images of the transposed matrices for the backward's WFragC measured slower.
template <int D>
__device__ __forceinline__ void wfrag_image_write(const EmbQkvArgs& A) {
    constexpr int F = 128, E = 4 * D * D + 2 * D * F, E4 = E / 4;
    constexpr int64_t LSTR = (int64_t)E + 3 * D + D + F + D + 4 * D;           // weights + in_b, out_b, b1, b2 + the two LayerNorms
    const int64_t o_in = A.W - A.E, o_out = o_in + 3 * D * D + 3 * D, o_w1 = o_out + D * D + D, o_w2 = o_w1 + F * D + F;
    float* out = reinterpret_cast<float*>(const_cast<unsigned short*>(A.sp));
    const int total = A.wf_layers * E4;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
        const int layer = j / E4, q = j % E4, e = 4 * q;
        int64_t so; int C, m_off;                            // matrix stored [R][C] at element offset m_off of the image
        if (e < 3 * D * D) { so = o_in; C = D; m_off = 0; }
        else if (e < 4 * D * D) { so = o_out; C = D; m_off = 3 * D * D; }
        else if (e < 4 * D * D + F * D) { so = o_w1; C = D; m_off = 4 * D * D; }
        else { so = o_w2; C = F; m_off = 4 * D * D + F * D; }
        const int K = C, idx = q - m_off / 4, lane = idx & 63, tq = idx >> 6, ci = tq % (K / 16), ct = tq / (K / 16);
        const int n = ct * 16 + (lane & 15), k0 = (lane >> 4) * (K / 4) + 4 * ci;
        st4(out + (size_t)layer * E + e, ld4(A.E + so + layer * LSTR + (size_t)n * C + k0));
    }
}
template <int D>
__global__ __launch_bounds__(256) void k_wfrag_write(const EmbQkvArgs A) { wfrag_image_write<D>(A); }
int launch_wfrag_write(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s) {
    EmbQkvArgs A{};
    A.E = p->params + ws.off[0]; A.W = p->params + poff(ws, 0, P_IN_W); A.wf_layers = p->n_layer;
    A.sp = reinterpret_cast<const unsigned short*>(ws.wfrag);
    hipLaunchKernelGGL(k_wfrag_write<128>, dim3(64), dim3(256), 0, s, A);
    return DR4SR_LAUNCH_CHECK();
}

template <int BM, int D>
__global__ __launch_bounds__(256) void k_embqkv_fwd(const EmbQkvArgs A) {
    constexpr int N = 3 * D, LDA = D + 4, LPT = D / 4, TPB = 256 / LPT;
    if constexpr (BM == 16 && D == 128) { if (A.wf_layers > 0) wfrag_image_write<D>(A); }      // every block, before the early exit of tile-less blocks
    const int T = A.state[DR4SR_STATE_T], t0 = xcd_tile(T, BM, A.xcd) * BM;
    if (t0 >= T) return;
    float* As = smem;
    const int c = (threadIdx.x % LPT) * 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    // latency regime: the in_proj fragments are requested in front of the gather's dependent chain (cu -> idx -> table row)
    constexpr bool PFE = BM == 16 && D == 64;
    WFragT<PFE ? D : 16, PFE ? N : 64> f_in;
    if constexpr (PFE) wfrag_load(f_in, A.W, D);
    const int bh = A.tile_seq[t0 >> 4];
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += TPB) {
        const int r = r0 + threadIdx.x / LPT, t = t0 + r;
        if (r < BM) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) {
                const int b = find_seq_from(A.cu, A.B, t, bh), pos = t - A.cu[b];
                const int64_t row = A.rows ? A.rows[b] : b;
                int64_t id = A.idx[row * A.L + pos];
                if (A.idx32 && c == 0) A.idx32[t] = (id > 0 && id < A.n_items) ? (int)id : 0;      // as the backward's scatter tests it
                if (A.tok && c == 0) A.tok[t] = make_int2(t - pos, b | ((A.cu[b + 1] - A.cu[b]) << 20) | (id == 0 ? 1 << 30 : 0));
                id = id < 0 ? 0 : (id >= A.n_items ? A.n_items - 1 : id);
                o = ld4(A.E + id * D + c);
                const float4 pe = ld4(A.P + (size_t)pos * D + c);
                o = make_float4(o.x + pe.x, o.y + pe.y, o.z + pe.z, o.w + pe.w);
                if (dodrop) {
                    const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * A.L + pos) * D + c);
                    o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
                }
                st4(A.X + (size_t)t * D + c, o);
            }
            st4(As + r * LDA + c, o);
        }
    }
    lds_barrier();
    TileAcc<BM, N> acc;
    tile_zero(acc);
    if constexpr (PFE) tile_mma_frag<BM, D, N>(As, LDA, f_in, acc);
    else tile_gemm<D == 128 && BM == 32, BM, D, N>(As, LDA, A.W, D, false, A.sp, WSplitGeo<D, 128>::E, 0, acc);
    tile_to_global<BM, N>(acc, A.QKV, N, A.bias, t0, T);
    if (A.dqkv_zero) zero_kv_rows<BM, D>(A.dqkv_zero, t0, T);
}

int launch_embqkv_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s) {
    const int D = p->D, bm = tile_rows(ws);
    const size_t lds = sizeof(float) * bm * (D + 4);
    dim3 grid((ws.Tmax + bm - 1) / bm), blk(256);
    EmbQkvArgs A;
    A.E = p->params + ws.off[0]; A.P = p->params + ws.off[1]; A.idx = p->in_item_id; A.rows = p->rows; A.cu = ws.cu; A.tile_seq = ws.tile_seq; A.X = ws.X[0];
    A.W = p->params + poff(ws, 0, P_IN_W); A.bias = p->params + poff(ws, 0, P_IN_B); A.QKV = ws.layer[0].qkv; A.state = p->state;
    A.B = p->B; A.L = p->L; A.n_items = p->n_items; A.training = training; A.seed = p->seed; A.p = p->p_drop;
    A.idx32 = de_owner_mode(ws) ? ws.idx32 : nullptr;
    A.sp = wsplit_of(p, ws, 0); A.wf_layers = 0;
    if (wfrag_img_on(p, ws)) { A.sp = reinterpret_cast<const unsigned short*>(ws.wfrag); A.wf_layers = p->n_layer; }
    const bool in_tile = attn_in_tile(p, ws);
    A.tok = (in_tile || ws.attn_tile_sa || attn_wave_on(p, ws)) ? ws.tok : nullptr; A.dqkv_zero = in_tile ? ws.layer[0].dqkv : nullptr;
    A.xcd = tile_xcd_order(p, ws) ? 1 : 0;
    if (A.xcd) grid.x = xcd_grid((int)grid.x, bm);
    if (wave_tiles(p, ws)) return launch_wt_embqkv_fwd(A, ws.Tmax, s);
#define EQ(B_) do { if (D == 64) hipLaunchKernelGGL((k_embqkv_fwd<B_, 64>), grid, blk, lds, s, A); \
                    else hipLaunchKernelGGL((k_embqkv_fwd<B_, 128>), grid, blk, lds, s, A); } while (0)
    BM_DISPATCH(bm, EQ);
#undef EQ
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------

// dropout + residual + LayerNorm over the 64 rows of a C tile held in LDS, 16 lanes per row, all four row passes of
// a thread issued together (loads first, then four independent reduction chains, then stores) so that the single wave
// per SIMD overlaps global-load / shuffle / Philox latencies across rows instead of serialising them.
//   v = res + drop(C)  -> U (global);  LN(v) -> OUT (global) [+ LDS copy];  (mean, rstd) -> ST
//   yreg (optional): the LayerNorm output rows of THIS thread ([pass][D/64] float4, zero beyond T) — with OUT == NULL nothing of
//   them goes to global memory (the fused last layer hands them to the scorer in registers)
template <int BM, int D, bool RES_IN_LDS, bool COPY_LDS>
__device__ __forceinline__ void ln_rowpass(const float* __restrict__ Cs, int ldc, const float* res, int ldres,
                                           const float* __restrict__ lnw, const float* __restrict__ lnb, float eps,
                                           float* __restrict__ U, float* __restrict__ OUT, float* __restrict__ ST,
                                           float* Ls, int ldl, int t0, int T, bool dodrop, const RngKey& rk,
                                           uint32_t site, float4 (*yreg)[D / 64] = nullptr) {
    constexpr int NV = D / 64, PASSES = BM / 16;
    const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    float4 gam[NV], bet[NV], v[PASSES][NV];
    float mean[PASSES], rstd[PASSES];
#pragma unroll
    for (int j = 0; j < NV; ++j) { gam[j] = ld4(lnw + 4 * l16 + 64 * j); bet[j] = ld4(lnb + 4 * l16 + 64 * j); }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * 16 + rsub, t = t0 + row;
        const bool ok = t < T;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * l16 + 64 * j;
            float4 o = ld4(Cs + row * ldc + c);
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (RES_IN_LDS) r = ld4(res + row * ldres + c);
            else if (ok) r = ld4(res + (size_t)t * ldres + c);
            if (dodrop) { const float4 m = drop4(rk, site, (uint64_t)t * D + c); o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w; }
            v[ps][j] = make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w);
            if (ok) st4(U + (size_t)t * D + c, v[ps][j]);
        }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) ln_stats16<NV>(v[ps], mean[ps], rstd[ps], eps);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * 16 + rsub, t = t0 + row;
        const bool ok = t < T;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * l16 + 64 * j;
            const float m = mean[ps], rs = rstd[ps];
            const float4 yv = make_float4((v[ps][j].x - m) * rs * gam[j].x + bet[j].x, (v[ps][j].y - m) * rs * gam[j].y + bet[j].y,
                                          (v[ps][j].z - m) * rs * gam[j].z + bet[j].z, (v[ps][j].w - m) * rs * gam[j].w + bet[j].w);
            if (ok && OUT) st4(OUT + (size_t)t * D + c, yv);
            if (yreg) yreg[ps][j] = ok ? yv : make_float4(0.f, 0.f, 0.f, 0.f);
            if (COPY_LDS) st4(Ls + row * ldl + c, ok ? yv : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (ok && l16 == 0) { ST[2 * (size_t)t] = mean[ps]; ST[2 * (size_t)t + 1] = rstd[ps]; }
    }
}

template <int BM, int D, int F, bool FFN_ONLY, bool KEEPQ = false>
__device__ __forceinline__ void post_fwd_body(const PostArgs& A, const int t0, const int T, float4 (*zreg)[D / 64] = nullptr,
                                              tattn::Keep* keep_out = nullptr) {
    constexpr int LD = D + 4, LF = F + 4, NVF = F / 64, PASSES = BM / 16;
    float* R0 = smem;                 // [64][LD]  ctx tile, later linear2 output
    float* R1 = R0 + BM * LD;         // [64][LD]  y tile
    float* R2 = R1 + BM * LD;         // [64][LF]  out_proj output (ld LD), then linear1 output / h
    const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const uint32_t sP = A.sP, sA = A.sA, sF = A.sF;
    const bool actdrop = dodrop && sA != 0xffffffffu;

    // latency regime (BM = 16): the weight fragments of the four GEMMs are requested AHEAD of them — not inside their k loops —, so their
    // L2 round trips overlap with the phases in front.  Not all up front any more (round 4): issuing 128 KB of requests took 3.9 us and
    // the attention's window queued behind them; they are spread around the attention's pieces below (SPREAD)
    constexpr bool PF = BM == 16 && !FFN_ONLY;
    WFragT<PF ? D : 16, PF ? D : 64> f_out;
    WFragT<PF ? D : 16, PF ? F : 64> f_w1;
    WFragT<PF ? F : 16, PF ? D : 64> f_w2;
    WFragT<PF ? D : 16, PF ? 3 * D : 64> f_nx;
    STAMP(0);
    if (FFN_ONLY) {                            // FMLP Intermediate block: the input tile IS y
        load_tile_bm<BM, D>(R1, LD, A.x, D, t0, T);
    } else {
        constexpr bool AT = BM == 16 && (D == 64 || D == 128);      // attention of the tile's rows in this launch (attn_tile.h)
        bool at_on = false;
        if constexpr (AT) at_on = A.at.on != 0;
        if (!at_on) load_tile_bm<BM, D>(R0, LD, A.ctx, D, t0, T);
        // Loads return in issue order and a wave stalls ISSUING when the CU's memory queue is full: the 128 KB (d = 64) / 200 KB (d = 128) of
        // weight fragments per workgroup took 3.9 us to request in front of the attention (stamps, round 4).  So: the window first, then
        // out_proj's fragments only; the rest is requested where the attention's arithmetic hides it (DR4SR_WFRAG_UPFRONT: all up front).
        tattn::FwdPre<AT ? D : 64, KEEPQ> att_pre;
#ifdef DR4SR_WFRAG_UPFRONT
        constexpr bool SPREAD = false;
#else
        constexpr bool SPREAD = AT && PF;
#endif
        if constexpr (SPREAD) {
            if (at_on) {
                tattn::fwd_issue<D, KEEPQ>(A, t0, T, att_pre);
                __builtin_amdgcn_sched_barrier(0);
                { if constexpr (D == 128) wfrag_load_img(f_out, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::OUT); else wfrag_load(f_out, A.out_w, D); }
                __builtin_amdgcn_sched_barrier(0);
                tattn::fwd_stage<D, KEEPQ>(A, t0, T, smem + att_lds_off(D, F), att_pre);
                __builtin_amdgcn_sched_barrier(0);
                { if constexpr (D == 128) wfrag_load_img(f_w1, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::W1); else wfrag_load(f_w1, A.w1, D); }
                { if constexpr (D == 128) wfrag_load_img(f_w2, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::W2); else wfrag_load(f_w2, A.w2, F); }
                __builtin_amdgcn_sched_barrier(0);
                const tattn::Keep k = tattn::fwd_compute<D, KEEPQ>(A, t0, T, R0, LD, smem + att_lds_off(D, F), att_pre);
                if (keep_out) *keep_out = k;
                __builtin_amdgcn_sched_barrier(0);
                if (A.nx_qkv) { if constexpr (D == 128) wfrag_load_img(f_nx, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::E); else wfrag_load(f_nx, A.nx_in_w, D); }
            } else {
                { if constexpr (D == 128) wfrag_load_img(f_out, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::OUT); else wfrag_load(f_out, A.out_w, D); }
                { if constexpr (D == 128) wfrag_load_img(f_w1, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::W1); else wfrag_load(f_w1, A.w1, D); }
                { if constexpr (D == 128) wfrag_load_img(f_w2, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::W2); else wfrag_load(f_w2, A.w2, F); }
                if (A.nx_qkv) { if constexpr (D == 128) wfrag_load_img(f_nx, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::E); else wfrag_load(f_nx, A.nx_in_w, D); }
            }
        } else {
            constexpr bool EARLY = D == 128;               // (window before the fragments: no register spill at d = 128)
            if constexpr (AT && EARLY) { if (at_on) { tattn::fwd_issue<D, KEEPQ>(A, t0, T, att_pre); __builtin_amdgcn_sched_barrier(0); } }
            if constexpr (PF) {
                { if constexpr (D == 128) wfrag_load_img(f_out, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::OUT); else wfrag_load(f_out, A.out_w, D); }
                { if constexpr (D == 128) wfrag_load_img(f_w1, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::W1); else wfrag_load(f_w1, A.w1, D); }
                { if constexpr (D == 128) wfrag_load_img(f_w2, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::W2); else wfrag_load(f_w2, A.w2, F); }
                if (A.nx_qkv) { if constexpr (D == 128) wfrag_load_img(f_nx, reinterpret_cast<const float*>(A.sp) + WSplitGeo<D, F>::E); else wfrag_load(f_nx, A.nx_in_w, D); }
            }
            if constexpr (AT) {
                if (at_on) {
                    if constexpr (!EARLY) tattn::fwd_issue<D, KEEPQ>(A, t0, T, att_pre);
                    tattn::fwd_stage<D, KEEPQ>(A, t0, T, smem + att_lds_off(D, F), att_pre);
                    const tattn::Keep k = tattn::fwd_compute<D, KEEPQ>(A, t0, T, R0, LD, smem + att_lds_off(D, F), att_pre);
                    if (keep_out) *keep_out = k;
                }
            }
        }
        lds_barrier(); STAMP(1);
        {
            TileAcc<BM, D> acc;
            tile_zero(acc);
            if constexpr (PF) tile_mma_frag<BM, D, D>(R0, LD, f_out, acc);
            else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, D, D>(R0, LD, A.out_w, D, false, A.sp, WSplitGeo<D, F>::E, WSplitGeo<D, F>::OUT, acc);
            tile_to_lds<BM, D>(acc, R2, LD, A.out_b);
        }
        lds_barrier(); STAMP(2);
        // ---- dropout1 + residual + LayerNorm1
        ln_rowpass<BM, D, false, true>(R2, LD, A.x, D, A.ln1_w, A.ln1_b, A.eps, A.u1, A.y, A.st1, R1, LD, t0, T, dodrop, rk, sP);
    }
    lds_barrier(); STAMP(3);
    // ---- linear1 + GELU + dropout
    {
        TileAcc<BM, F> acc;
        tile_zero(acc);
        if constexpr (PF) tile_mma_frag<BM, D, F>(R1, LD, f_w1, acc);
        else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, D, F>(R1, LD, A.w1, D, false, A.sp, WSplitGeo<D, F>::E, WSplitGeo<D, F>::W1, acc);
        tile_to_lds<BM, F>(acc, R2, LF, A.b1);
    }
    lds_barrier(); STAMP(4);
    {
        float4 av[PASSES][NVF];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
            for (int j = 0; j < NVF; ++j) av[ps][j] = ld4(R2 + (ps * 16 + rsub) * LF + 4 * l16 + 64 * j);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = ps * 16 + rsub, t = t0 + row;
            const bool ok = t < T;
#pragma unroll
            for (int j = 0; j < NVF; ++j) {
                const int c = 4 * l16 + 64 * j;
                const float4 a4 = av[ps][j];
                float4 h = make_float4(gelu_erf(a4.x), gelu_erf(a4.y), gelu_erf(a4.z), gelu_erf(a4.w));
                if (actdrop) { const float4 m = drop4(rk, sA, (uint64_t)t * F + c); h.x *= m.x; h.y *= m.y; h.z *= m.z; h.w *= m.w; }
                if (ok) { st4(A.a + (size_t)t * F + c, a4); st4(A.h + (size_t)t * F + c, h); }
                st4(R2 + row * LF + c, h);
            }
        }
    }
    lds_barrier(); STAMP(5);
    // ---- linear2 + dropout2 + residual + LayerNorm2
    {
        TileAcc<BM, D> acc;
        tile_zero(acc);
        if constexpr (PF) tile_mma_frag<BM, F, D>(R2, LF, f_w2, acc);
        else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, F, D>(R2, LF, A.w2, F, false, A.sp, WSplitGeo<D, F>::E, WSplitGeo<D, F>::W2, acc);
        tile_to_lds<BM, D>(acc, R0, LD, A.b2);
    }
    lds_barrier(); STAMP(6);
    if (!FFN_ONLY && A.nx_qkv) {               // layer-boundary fusion: keep z in LDS and emit the next layer's in_proj
        ln_rowpass<BM, D, true, true>(R0, LD, R1, LD, A.ln2_w, A.ln2_b, A.eps, A.u2, A.z, A.st2, R1, LD, t0, T, dodrop, rk, sF);
        lds_barrier();
        TileAcc<BM, 3 * D> acc;
        tile_zero(acc);
        if constexpr (PF) tile_mma_frag<BM, D, 3 * D>(R1, LD, f_nx, acc);
        else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, D, 3 * D>(R1, LD, A.nx_in_w, D, false, A.sp ? A.sp + 4 * WSplitGeo<D, F>::E : nullptr, WSplitGeo<D, F>::E, 0, acc);
        tile_to_global<BM, 3 * D>(acc, A.nx_qkv, 3 * D, A.nx_in_b, t0, T);
        if (A.nx_dqkv_zero) zero_kv_rows<BM, D>(A.nx_dqkv_zero, t0, T);
    } else {
        ln_rowpass<BM, D, true, false>(R0, LD, R1, LD, A.ln2_w, A.ln2_b, A.eps, A.u2, A.z, A.st2, nullptr, 0, t0, T, dodrop, rk, sF, zreg);
    }
    STAMP(15);
}

template <int BM, int D, int F, bool FFN_ONLY>
__global__ __launch_bounds__(256) void k_post_fwd(const PostArgs A) {
    const int T = A.state[DR4SR_STATE_T], t0 = xcd_tile(T, BM, A.xcd) * BM;
    if (t0 >= T) return;
    post_fwd_body<BM, D, F, FFN_ONLY>(A, t0, T);
}

// fold the 16 row-groups of the workgroup (4 per wave x 4 waves) into ONE partial row per token tile:
// dst[0..D) = sum dgam, dst[D..2D) = sum dbet.  `scr` is a [4 waves][2*D] LDS scratch.  Deterministic, no atomics;
// the tile partials are summed by k_wgrad's reduce job.
template <int NV>
__device__ __forceinline__ void flush_affine(const float4 (&dgam)[NV], const float4 (&dbet)[NV], float* scr, float* dst) {
    constexpr int D = 64 * NV;
    const int l16 = threadIdx.x & 15, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto fold = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * l16 + 64 * j;
        const float4 a = make_float4(fold(dgam[j].x), fold(dgam[j].y), fold(dgam[j].z), fold(dgam[j].w));
        const float4 b = make_float4(fold(dbet[j].x), fold(dbet[j].y), fold(dbet[j].z), fold(dbet[j].w));
        if (lane < 16) { st4(scr + w * 2 * D + c, a); st4(scr + w * 2 * D + D + c, b); }
    }
    lds_barrier();
    for (int i = threadIdx.x; i < 2 * D; i += 256)
        dst[i] = (scr[i] + scr[2 * D + i]) + (scr[4 * D + i] + scr[6 * D + i]);
    lds_barrier();
}

// LayerNorm backward over the 64 rows of a tile with the four row passes of a thread issued together (see ln_rowpass).
//   g = Gg[t] (SRC 0) | La[row] + Lb[row] (SRC 1) | La[row] + Gg[t] (SRC 2);   du = LN'(g; u, mean, rstd, gamma)
//   du -> DUg (global, optional) and DUl (LDS, optional);   du * dropout(site) -> DMg (global) and DMl (LDS)
template <int BM, int D, int SRC>
__device__ __forceinline__ void ln_bwd_rowpass(const float* __restrict__ Gg, const float* La,
                                               const float* Lb, int ldl, const float* __restrict__ Ug,
                                               const float* __restrict__ ST, const float* __restrict__ lnw,
                                               float* __restrict__ DUg, float* DUl, float* __restrict__ DMg,
                                               float* DMl, float4 (&dgam)[D / 64], float4 (&dbet)[D / 64],
                                               int t0, int T, bool dodrop, const RngKey& rk, uint32_t site,
                                               const float4 (*greg)[D / 64] = nullptr) {
    constexpr int NV = D / 64, PASSES = BM / 16;
    const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    float4 gam[NV], g[PASSES][NV], u[PASSES][NV];
    float mean[PASSES], rstd[PASSES];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gam[j] = ld4(lnw + 4 * l16 + 64 * j);
        dgam[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        dbet[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * 16 + rsub, t = t0 + row;
        const bool ok = t < T;
        mean[ps] = ok ? ST[2 * (size_t)t] : 0.f;
        rstd[ps] = ok ? ST[2 * (size_t)t + 1] : 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * l16 + 64 * j;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            u[ps][j] = ok ? ld4(Ug + (size_t)t * D + c) : z4;
            if (SRC == 0) g[ps][j] = ok ? ld4(Gg + (size_t)t * D + c) : z4;
            else if (SRC == 3) g[ps][j] = ok ? greg[ps][j] : z4;     // this thread's rows, straight from the scorer (registers)
            else {
                const float4 p0 = ld4(La + row * ldl + c);
                float4 p1 = z4;
                if (SRC == 1) p1 = ld4(Lb + row * ldl + c);
                else if (ok) p1 = ld4(Gg + (size_t)t * D + c);
                g[ps][j] = ok ? make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w) : z4;
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) ln_bwd_row<NV>(g[ps], u[ps], mean[ps], rstd[ps], gam, dgam, dbet);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * 16 + rsub, t = t0 + row;
        const bool ok = t < T;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * l16 + 64 * j;
            const float4 du = g[ps][j];
            if (DUg && ok) st4(DUg + (size_t)t * D + c, du);
            if (DUl) st4(DUl + row * ldl + c, du);
            float4 dm = du;
            if (dodrop) { const float4 m = drop4(rk, site, (uint64_t)t * D + c); dm.x *= m.x; dm.y *= m.y; dm.z *= m.z; dm.w *= m.w; }
            if (ok) st4(DMg + (size_t)t * D + c, dm);
            st4(DMl + row * ldl + c, dm);
        }
    }
}

// DET: the deterministic latency form (kernels.h Workspace::det_lat) — a separate instantiation: as a run-time branch it cost the default step
// 0.8 % (d = 64) / 2 % (d = 128) at B = 256
template <int BM, int D, int F, bool FFN_ONLY, bool DET = false>
__device__ __forceinline__ void post_bwd_body(const PostArgs& A, const int t0, const int T, const int tile,
                                              const float4 (*dzreg)[D / 64] = nullptr, const tattn::Keep* att_staged = nullptr) {
    constexpr int LD = D + 4, LF = F + 4, NV = D / 64, NVF = F / 64, PASSES = BM / 16;
    float* R1 = smem;                          // R1 first: R0 and R2 are contiguous and together hold a [64][3D+4] dqkv tile
    float* R0 = R1 + BM * LD;
    float* R2 = R0 + BM * LD;
    const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const uint32_t sP = A.sP, sA = A.sA, sF = A.sF;
    const bool actdrop = dodrop && sA != 0xffffffffu;
    float4 dgam[NV], dbet[NV];
    // attention backward of the tile's rows at the end of this launch (attn_tile.h): its operands are requested first
    // (att_staged: k_post_mid — the forward half of the launch left window, Q rows, statistics and keep decisions behind)
    // latency regime, d = 64 (k_post_bwd: 90 VGPRs without them): the data-gradient GEMMs' weight fragments are requested one phase AHEAD of
    // the GEMM that consumes them — the L2 round trip of each of the four phases overlaps the phase in front of it (as the forward's do)
#ifdef DR4SR_BWD_NO_PREFETCH
    constexpr bool PFB = false;
#else
    constexpr bool PFB = BM == 16 && !FFN_ONLY && (D == 64 || D == 128);
#endif
    // d = 64: requested BEFORE the GEMM in front (two sets in flight); d = 128 (64 VGPRs per set, none to spare): right BEHIND it, one set
    // in flight while the row pass between two GEMMs runs — and the [3D x D] set of the first phase stays inside its k loop
    constexpr bool PF64 = PFB && D == 64, PF128 = PFB && D == 128;
    WFragC<PF64 ? 3 * D : 16, PF64 ? D : 64> fr_up;
    WFragC<PFB ? D : 16, PFB ? F : 64> fr_w2;
    WFragC<PFB ? F : 16, PFB ? D : 64> fr_w1;
    WFragC<PFB ? D : 16, PFB ? D : 64> fr_out;
    constexpr bool AT = BM == 16 && !FFN_ONLY && (D == 64 || D == 128);
    bool at_on = false, at_stage = false;
    tattn::Keep keep{0xffffffffu, 0xffffffffu};
    tattn::Stage<AT ? D : 64, true, true> att_st;
    constexpr int UPQ = AT ? (16 * 3 * D / 4) / 256 : 1;
    float4 updq[UPQ];
    bool up_pre = false;
    if constexpr (AT) {
        at_on = A.at.on != 0;
        at_stage = at_on && !att_staged;
        if (att_staged) keep = *att_staged;
        // the layer below accumulates its dK | dV in the NEXT launch: zeroed here too (the forward did it once; a second backward pass
        // on the same forward — dr4sr_sasrec_encode_bwd twice, retain_graph — must not add onto the first)
        if (at_on && A.dn_dqkv_zero) zero_kv_rows<BM, D>(A.dn_dqkv_zero, t0, T);
        if (at_stage && A.up_dqkv) {
            // loads return in issue order and the Philox calls below are ~1 us of VALU: the FIRST phase's own tile (layer + 1's dqkv rows)
            // is requested in front of the attention's window, which is consumed last
            up_pre = true;
#pragma unroll
            for (int q = 0; q < UPQ; ++q) {
                const int i = threadIdx.x + 256 * q, row = i / (3 * D / 4), c = (i % (3 * D / 4)) * 4;
                updq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t0 + row < T) updq[q] = ld4(A.up_dqkv + (size_t)(t0 + row) * 3 * D + c);
            }
            if constexpr (PF64) wfrag_load(fr_up, A.up_in_w, D);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (at_stage) {
            const int2 mq = tattn::own_word(A.at, t0, T);
            att_st.issue(A.at, t0, T, (A.at.on & 4) ? tattn::NEAR0 : 0, tattn::WR);
            __builtin_amdgcn_sched_barrier(0);
            keep = tattn::own_keep(A, mq, t0, T);          // Philox calls while the window is in flight
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMP(16);
    if constexpr (PFB) { if (!A.up_dqkv) wfrag_load(fr_w2, A.w2, F); else if (PF64 && !up_pre) wfrag_load(fr_up, A.up_in_w, D); }
    auto att_commit = [&]() { if constexpr (AT) { if (at_stage) att_st.commit(tattn::Lds<D>(smem + att_lds_off(D, F)), (A.at.on & 4) ? tattn::NEAR0 : 0, tattn::WR); } };

    // ---- LayerNorm2 backward: du2 -> R1 (residual branch); df = du2*mask -> global + R0
    if (!FFN_ONLY && A.up_dqkv) {
        // layer-boundary fusion: dz = dqkv(layer+1) W_in(layer+1) + du1(layer+1), computed here instead of a separate launch
        constexpr int LQ = 3 * D + 4;
        float* Aq = R0;                            // [64][LQ] spans R0 + R2 (see post_lds)
        if (AT && up_pre) {
#pragma unroll
            for (int q = 0; q < UPQ; ++q) {
                const int i = threadIdx.x + 256 * q, row = i / (3 * D / 4), c = (i % (3 * D / 4)) * 4;
                st4(Aq + row * LQ + c, updq[q]);
            }
        } else load_tile_bm<BM, 3 * D>(Aq, LQ, A.up_dqkv, 3 * D, t0, T);
        att_commit();                              // behind this phase's own loads: one round trip for both
        lds_barrier();
        if constexpr (AT && DET) {                 // layer + 1's shared dK | dV rows arrive as partial blocks
            det_kv_rows<D>(Aq, LQ, const_cast<float*>(A.up_dqkv), A.up_kv_part, A.at.tok, t0, T);
            lds_barrier();
        }
        TileAcc<BM, D> acc;
        tile_zero(acc);
        if constexpr (PF64) { wfrag_load(fr_w2, A.w2, F); tile_mma_frag<BM, 3 * D, D>(Aq, LQ, fr_up, acc); }
        else { tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, 3 * D, D>(Aq, LQ, A.up_in_w, D, true, A.sp ? A.sp + 4 * WSplitGeo<D, F>::E : nullptr, WSplitGeo<D, F>::E, 0, acc); if constexpr (PF128) wfrag_load(fr_w2, A.w2, F); }
        tile_to_lds<BM, D>(acc, R1, LD, nullptr);
        lds_barrier();
        ln_bwd_rowpass<BM, D, 2>(A.up_du1, R1, nullptr, LD, A.u2, A.st2, A.ln2_w, nullptr, R1, A.df, R0, dgam, dbet, t0, T, dodrop, rk, sF);
    } else if (dzreg) {
        att_commit();
        ln_bwd_rowpass<BM, D, 3>(nullptr, nullptr, nullptr, LD, A.u2, A.st2, A.ln2_w, nullptr, R1, A.df, R0, dgam, dbet, t0, T, dodrop, rk, sF, dzreg);
    } else {
        att_commit();
        ln_bwd_rowpass<BM, D, 0>(A.dz, nullptr, nullptr, LD, A.u2, A.st2, A.ln2_w, nullptr, R1, A.df, R0, dgam, dbet, t0, T, dodrop, rk, sF);
    }
    flush_affine<NV>(dgam, dbet, R2, A.ln_part + (size_t)tile * 4 * D);
    // ---- dh = df W2   (x W^T form with W2^T [F][D])
    {
        TileAcc<BM, F> acc;
        tile_zero(acc);
        if constexpr (PF64) { wfrag_load(fr_w1, A.w1, D); tile_mma_frag<BM, D, F>(R0, LD, fr_w2, acc); }
        else if constexpr (PF128) { tile_mma_frag<BM, D, F>(R0, LD, fr_w2, acc); wfrag_load(fr_w1, A.w1, D); }
        else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, D, F>(R0, LD, A.w2, F, true, A.sp, WSplitGeo<D, F>::E, WSplitGeo<D, F>::W2, acc);
        tile_to_lds<BM, F>(acc, R2, LF, nullptr);
    }
    lds_barrier();
    // ---- da = dh * mask_act * gelu'(a)
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * 16 + rsub, t = t0 + row;
#pragma unroll
        for (int j = 0; j < NVF; ++j) {
            const int c = 4 * l16 + 64 * j;
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) {
                d = ld4(R2 + row * LF + c);
                const float4 av = ld4(A.a + (size_t)t * F + c);
                if (actdrop) { const float4 m = drop4(rk, sA, (uint64_t)t * F + c); d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w; }
                d.x *= gelu_erf_grad(av.x); d.y *= gelu_erf_grad(av.y); d.z *= gelu_erf_grad(av.z); d.w *= gelu_erf_grad(av.w);
                st4(A.da + (size_t)t * F + c, d);
            }
            st4(R2 + row * LF + c, d);
        }
    }
    lds_barrier();
    // ---- dy = da W1 + du2 ;  LayerNorm1 backward -> du1 ;  do = du1*mask -> R1
    {
        TileAcc<BM, D> acc;
        tile_zero(acc);
        if constexpr (PF64) { wfrag_load(fr_out, A.out_w, D); tile_mma_frag<BM, F, D>(R2, LF, fr_w1, acc); }
        else if constexpr (PF128) { tile_mma_frag<BM, F, D>(R2, LF, fr_w1, acc); wfrag_load(fr_out, A.out_w, D); }
        else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, F, D>(R2, LF, A.w1, D, true, A.sp, WSplitGeo<D, F>::E, WSplitGeo<D, F>::W1, acc);
        tile_to_lds<BM, D>(acc, R0, LD, nullptr);
    }
    lds_barrier();
    if (FFN_ONLY) {                            // FMLP: d(input) = da W1 + du2, nothing upstream inside this kernel
        constexpr int C4f = D / 4;
        for (int i = threadIdx.x; i < BM * C4f; i += 256) {
            const int row = i / C4f, c = (i % C4f) * 4;
            if (t0 + row < T) {
                const float4 p0 = ld4(R0 + row * LD + c), p1 = ld4(R1 + row * LD + c);
                st4(A.du1 + (size_t)(t0 + row) * D + c, make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w));
            }
        }
        return;
    }
    ln_bwd_rowpass<BM, D, 1>(nullptr, R0, R1, LD, A.u1, A.st1, A.ln1_w, A.du1, nullptr, A.dout, R1, dgam, dbet, t0, T, dodrop, rk, sP);
    flush_affine<NV>(dgam, dbet, R2, A.ln_part + (size_t)tile * 4 * D + 2 * D);
    // ---- dctx = do W_out
    {
        TileAcc<BM, D> acc;
        tile_zero(acc);
        if constexpr (PFB) tile_mma_frag<BM, D, D>(R1, LD, fr_out, acc);
        else tile_gemm<BM == 32 && (D == 128 || FFN_ONLY), BM, D, D>(R1, LD, A.out_w, D, true, A.sp, WSplitGeo<D, F>::E, WSplitGeo<D, F>::OUT, acc);
        tile_to_lds<BM, D>(acc, R0, LD, nullptr);
    }
    lds_barrier();
    constexpr int C4 = D / 4;
    for (int i = threadIdx.x; i < BM * C4; i += 256) {
        const int row = i / C4, c = (i % C4) * 4;
        const bool ok = t0 + row < T;
        const float4 v = ld4(R0 + row * LD + c);
        if (ok && !at_on) st4(A.dctx + (size_t)(t0 + row) * D + c, v);        // (in-tile attention: dctx never leaves the launch)
        if (A.rd) {                                    // softmax-backward row term of the attention: sum_j P dP = <dctx, ctx> per head
            const int lph = C4 / A.n_head;             // lanes per head (contiguous, power of two)
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) o = ld4(A.ctx + (size_t)(t0 + row) * D + c);
            float d = (v.x * o.x + v.y * o.y) + (v.z * o.z + v.w * o.w);
            for (int off = lph >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
            if (ok && (i % C4) % lph == 0) A.rd[(size_t)(t0 + row) * A.n_head + (i % C4) / lph] = d;
            if constexpr (AT) { if (at_on && (i % C4) % lph == 0) tattn::Lds<D>(smem + att_lds_off(D, F)).rd[row * 2 + (i % C4) / lph] = d; }
        }
    }
    if constexpr (AT) {
        if (at_on) {
            STAMP(17);
            lds_barrier();
            if (at_stage) tattn::far_rows_if_needed<D>(A.at, tattn::Lds<D>(smem + att_lds_off(D, F)), t0, T);
            STAMP(18);
            tattn::bwd<D, DET>(A, t0, T, R0, LD, smem + att_lds_off(D, F), keep);
        }
    }
    STAMP(26);
}

template <int BM, int D, int F, bool FFN_ONLY, bool DET = false>
__global__ __launch_bounds__(256) void k_post_bwd(const PostArgs A) {
    const int T = A.state[DR4SR_STATE_T], bx = xcd_tile(T, BM, A.xcd), t0 = bx * BM;
    if (t0 >= T) return;
    post_bwd_body<BM, D, F, FFN_ONLY, DET>(A, t0, T, bx);
}

// ------------------------------------------------------------------------------------------------
// Last-layer fusion: the scorer is per TOKEN (basemodel.py:204-214), so the last post_fwd, the scorer + BCE (fwd and bwd)
// and the last post_bwd of a token tile need nothing from any other workgroup — one launch instead of three.
// Scorer per token (LPT lanes per token, like the embedding kernels): negative drawn in-kernel or read, pos/neg dot
// products by lane-group shuffles, un-normalised dz + table-gradient atomics; the tile's (count, loss) partial goes to
// part[2*tile] (summed by k_wgrad's reduce job).  Valid targets at positions >= seqlen (query row is zero there) are
// counted by the sequence's last token, as the per-sequence scorer does.
template <int LPT>
__device__ __forceinline__ float lane_group_sum(float v) {
#pragma unroll
    for (int o = LPT / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// MetaModel.selection for ONE token inside the scorer (D = 64: 16 lanes per token, lane l holds z[4l..4l+3] and owns hidden units
// 4l..4l+3 of the 64->64->2 MLP; W1 is read from L1/L2 — phi is 17 KB).  Returns weight_t and, in dzw, loss_t * d weight_t / d z
// for this lane's 4 features.  Same noise / mask conventions as csrc/meta.hip (position index p = b*L + pos).
__device__ __forceinline__ float meta_weight_token(const ScoreTileArgs& S, const float4& q, float lt, int b, int pos, int64_t row, int t,
                                                   int sub, uint32_t step, float4& dzw) {
    constexpr int MD = 64;
    const float* W1 = S.phi;
    const float* b1 = S.phi + MD * MD;
    const float* W2 = b1 + MD;                               // [2][64]
    const float* b2 = W2 + 2 * MD;
    float pre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pre[k] = b1[4 * sub + k];
#pragma unroll
    for (int s = 0; s < 16; ++s) {                           // z[4s..4s+3] from lane s of the group
        const float z0 = __shfl(q.x, s, 16), z1 = __shfl(q.y, s, 16), z2 = __shfl(q.z, s, 16), z3 = __shfl(q.w, s, 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 wv = ld4(W1 + (size_t)(4 * sub + k) * MD + 4 * s);
            pre[k] = fmaf(wv.x, z0, fmaf(wv.y, z1, fmaf(wv.z, z2, fmaf(wv.w, z3, pre[k]))));
        }
    }
    unsigned long long gate;
    if (S.gate_in) gate = S.gate_in[t];
    else {
        unsigned long long mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mine |= (unsigned long long)(pre[k] > 0.f) << (4 * sub + k);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mine |= __shfl_xor(mine, o, 16);
        gate = mine;
    }
    float h[4], s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = 4 * sub + k;
        h[k] = ((gate >> j) & 1ull) ? pre[k] : 0.f;
        s0 = fmaf(W2[j], h[k], s0);
        s1 = fmaf(W2[MD + j], h[k], s1);
    }
    s0 = lane_group_sum<16>(s0) + b2[0];
    s1 = lane_group_sum<16>(s1) + b2[1];
    float g0, g1;
    const int64_t p = (int64_t)b * S.L + pos;
    if (S.gumbel) { g0 = S.gumbel[2 * p]; g1 = S.gumbel[2 * p + 1]; }
    else {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)p, (uint32_t)(p >> 32), 0x6D657461u, step),
                                      make_uint2((uint32_t)S.meta_seed, (uint32_t)(S.meta_seed >> 32)));
        const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
        g0 = -logf(-logf(u0)); g1 = -logf(-logf(u1));
    }
    const float y = 1.0f / (1.0f + expf(-((s0 + g0) - (s1 + g1)) * S.inv_tau));
    const bool forced = S.user_id && S.user_id[row] == 0;        // metamodel.py:180-183: pattern rows -> weight 1
    const float wt = forced ? 1.0f : y;
    if (sub == 0) {
        if (S.w_out) S.w_out[t] = wt;
        if (S.gate_out) S.gate_out[t] = gate;
    }
    dzw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!forced) {
        const float dzl = lt * y * (1.0f - y) * S.inv_tau;
        float dh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = 4 * sub + k; dh[k] = ((gate >> j) & 1ull) ? (W2[j] - W2[MD + j]) * dzl : 0.f; }
#pragma unroll
        for (int s = 0; s < 16; ++s) {                       // d z[4 sub..] += sum_j W1[j][4 sub..] dh_j, units j = 4s..4s+3 from lane s
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dj = __shfl(dh[k], s, 16);
                const float4 wv = ld4(W1 + (size_t)(4 * s + k) * MD + 4 * sub);
                dzw.x = fmaf(wv.x, dj, dzw.x); dzw.y = fmaf(wv.y, dj, dzw.y); dzw.z = fmaf(wv.z, dj, dzw.z); dzw.w = fmaf(wv.w, dj, dzw.w);
            }
        }
    }
    return wt;
}

// META: with the MetaModel selection weight (a separate instantiation — its registers cost the plain kernel an occupancy step)
template <int BM, int D, bool META>
__device__ __forceinline__ void score_tile(const PostArgs& A, const ScoreTileArgs& S, const int t0, const int T, const int tile, float* lrec_base) {
    constexpr int LPT = D / 4, TPB = 256 / LPT;
    const int c = (threadIdx.x % LPT) * 4, sub = threadIdx.x % LPT;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], 0.f);
    float lsum = 0.f, cnt = 0.f;
    float* dZ = const_cast<float*>(A.dz);
    const int bh = S.tile_seq[t0 >> 4];
    int4* lrec = reinterpret_cast<int4*>(lrec_base);
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += TPB) {
        const int r = r0 + threadIdx.x / LPT, t = t0 + r;
        if (r < BM && t < T) {
            const int b = find_seq_from(S.cu, S.B, t, bh), pos = t - S.cu[b], n = S.cu[b + 1] - S.cu[b];
            const int64_t row = S.rows ? S.rows[b] : b;
            int64_t tgt = S.target[row * S.L + pos], ng;
            if (S.sample_neg) {
                ng = sample_neg_id(rk, (uint64_t)b * S.L + pos, S.n_items);
                if (sub == 0) S.neg_item[(size_t)b * S.L + pos] = ng;
            } else {
                ng = S.neg_item[(size_t)b * S.L + pos];
            }
            ng = ng < 0 ? 0 : (ng >= S.n_items ? S.n_items - 1 : ng);
            float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (S.ent && sub == 1) reinterpret_cast<int*>(lrec + BM)[r] = S.idx32[t];
            if (tgt > 0 && tgt < S.n_items) {
                const float4 q = ld4(A.z + (size_t)t * D + c), ep = ld4(S.E + tgt * D + c), en = ld4(S.E + ng * D + c);
                const float sp = lane_group_sum<LPT>(q.x * ep.x + q.y * ep.y + q.z * ep.z + q.w * ep.w);
                const float sn = lane_group_sum<LPT>(q.x * en.x + q.y * en.y + q.z * en.z + q.w * en.w);
                const float lt = softplus_f(-sp) + softplus_f(sn);
                float dpos = -sigmoid_f(-sp), dneg = sigmoid_f(sn), wt = 1.0f;
                float4 dzw = make_float4(0.f, 0.f, 0.f, 0.f);          // loss_t * d weight_t / d z_t
                if constexpr (D == 64 && META) {
                    if (S.phi) wt = meta_weight_token(S, q, lt, b, pos, row, t, sub, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], dzw);
                }
                if (sub == 0) { lsum += wt * lt; cnt += 1.f; }
                dpos *= wt; dneg *= wt;
                dz = make_float4(dpos * ep.x + dneg * en.x + dzw.x, dpos * ep.y + dneg * en.y + dzw.y, dpos * ep.z + dneg * en.z + dzw.z,
                                 dpos * ep.w + dneg * en.w + dzw.w);
                float* gp = S.dE + tgt * D + c;
                float* gn = S.dE + ng * D + c;
                if (!S.rec) {
                    unsafeAtomicAdd(gp, dpos * q.x); unsafeAtomicAdd(gp + 1, dpos * q.y); unsafeAtomicAdd(gp + 2, dpos * q.z); unsafeAtomicAdd(gp + 3, dpos * q.w);
                    unsafeAtomicAdd(gn, dneg * q.x); unsafeAtomicAdd(gn + 1, dneg * q.y); unsafeAtomicAdd(gn + 2, dneg * q.z); unsafeAtomicAdd(gn + 3, dneg * q.w);
                } else if (sub == 0) {
                    const int4 rec = make_int4((int)tgt, (int)ng, __float_as_int(dpos), __float_as_int(dneg));
                    if (S.ent) lrec[r] = rec; else S.rec[t] = rec;
                }
            } else if (S.rec && sub == 0) { if (S.ent) lrec[r] = make_int4(0, 0, 0, 0); else S.rec[t] = make_int4(0, 0, 0, 0); }
            st4(dZ + (size_t)t * D + c, dz);
            if (pos == n - 1) {                               // tail positions of this sequence (zero query): loss terms only
                for (int l = n + sub; l < S.L; l += LPT) {
                    const int64_t tl = S.target[row * S.L + l];
                    if (S.sample_neg) S.neg_item[(size_t)b * S.L + l] = sample_neg_id(rk, (uint64_t)b * S.L + l, S.n_items);
                    if (tl > 0 && tl < S.n_items) { lsum += 2.0f * 0.69314718055994530942f; cnt += 1.f; }
                }
            }
        }
    }
    // workgroup reduction of (cnt, lsum) -> one partial per tile
    float* red = smem;
    cnt = wave_sum(cnt); lsum = wave_sum(lsum);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = cnt; red[2 * (threadIdx.x >> 6) + 1] = lsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        S.part[2 * tile] = (red[0] + red[2]) + (red[4] + red[6]);
        S.part[2 * tile + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

// Owner-computes table gradient, producer side.  A token tile contributes up to 3 BM rows to dE — (target, d pos score, z_t),
// (negative, d neg score, z_t), (input id, 1, dx0_t) — and each belongs to the owner workgroup `id & (G - 1)` of k_wgrad's owner job.
// Instead of every owner scanning every token of the batch (G x T record reads: the instruction-issue cost of that scan was
// ~20 % of k_wgrad at B = 8192), the tile leaves its entries SORTED BY OWNER plus a [G + 1] table of start offsets (bytes: a
// tile has at most 192 entries), so an owner reads two bytes per tile and then exactly its own entries: O(T) in total.  The
// order inside a bucket is the entry index (token-major), a pure function of the batch: the gradient stays bit-reproducible.
//   entry = {packed token, id >> logG (the owner's local row), coefficient bits, source (0: query rows z, 1: dx0 rows)}
// Runs between the two halves of k_post_mid: the tile regions of LDS are free, `scratch` = their base.
// LDS of tile_sort, private to it (behind the scorer's reduction scratch): [BM] int4 records | [BM] int input ids | [G + 4] int histogram / offsets | [4] int
__host__ __device__ constexpr int tile_sort_lds_bytes(int bm, int G) { return bm * 20 + (G + 4 + 4) * 4; }
template <int BM>
__device__ __forceinline__ void tile_sort_init(const ScoreTileArgs& S, float* area) {     // at kernel start: nothing else touches the area
    int* hist = reinterpret_cast<int*>(area) + 5 * BM;
    for (int g = threadIdx.x; g < (1 << S.logG) + 4; g += 256) hist[g] = 0;
}
template <int BM>
__device__ __forceinline__ void tile_sort(const ScoreTileArgs& S, const int tile, const int t0, const int T, float* area) {
    constexpr int E = 3 * BM, EPL = (E + 63) / 64;        // wave 0 takes every entry: EPL per lane, in entry order
    static_assert(E <= 255, "byte offsets");
    const int G = 1 << S.logG, per = G >> 8;              // G in {256, 512, 1024}: `per` consecutive buckets per thread in the scan
    const int4* lrec = reinterpret_cast<const int4*>(area);
    const int* lidx = reinterpret_cast<const int*>(area) + 4 * BM;
    int* hist = reinterpret_cast<int*>(area) + 5 * BM;    // [G + 4], zero since tile_sort_init
    int* wsum = hist + G + 4;                             // [4]
    const bool w0 = threadIdx.x < 64;
    lds_barrier();                                        // the scorer's records and input ids are in LDS (LDS-only hand-offs: no vmcnt drain)
    // rank inside the bucket = value returned by the LDS counter: the entries of one instruction are served in lane order and the
    // instructions of the one wave in program order, so the order is the entry index (a function of the batch; checked by the
    // bit-reproducibility tests)
    int4 ent[EPL]; int k[EPL], rank[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int e = threadIdx.x + 64 * j, r = e / 3, kind = e % 3, t = t0 + r;
        k[j] = -1; rank[j] = 0; ent[j] = make_int4(0, 0, 0, 0);
        if (w0 && e < E && t < T) {
            int id = kind == 2 ? lidx[r] : 0, cf = __float_as_int(1.0f);
            if (kind < 2) {
                const int4 rec = lrec[r];
                id = rec.x > 0 ? (kind ? rec.y : rec.x) : 0; cf = kind ? rec.w : rec.z;
            }
            if (id > 0) { k[j] = id & (G - 1); ent[j] = make_int4(t, id >> S.logG, cf, kind == 2); rank[j] = atomicAdd(&hist[k[j]], 1); }
        }
    }
    lds_barrier();
    // exclusive scan of hist[0..G) -> start offsets (in place), hist[G] = the tile's entry count
    int loc[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { loc[j] = j < per ? hist[threadIdx.x * per + j] : 0; sum += loc[j]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += u; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    lds_barrier();
    int ex = incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) ex += wsum[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < per) { hist[threadIdx.x * per + j] = ex; ex += loc[j]; }
    if (threadIdx.x == 255) hist[G] = ex;
    lds_barrier();
    unsigned* offw = reinterpret_cast<unsigned*>(S.off + (size_t)tile * (G + 4));
    for (int q = threadIdx.x; q < (G + 4) / 4; q += 256)
        offw[q] = (unsigned)hist[4 * q] | ((unsigned)hist[4 * q + 1] << 8) | ((unsigned)hist[4 * q + 2] << 16) | ((unsigned)hist[4 * q + 3] << 24);
#pragma unroll
    for (int j = 0; j < EPL; ++j)
        if (k[j] >= 0) S.ent[(size_t)tile * E + hist[k[j]] + rank[j]] = ent[j];
    // the area stays private to the next tile_sort of this workgroup (one tile per workgroup: none)
}

// D = 64: the row passes of the post kernels and the scorer use the SAME thread -> (row, 4 columns) map (16 lanes per token), so the
// query row goes from LayerNorm2 to the scorer and dz from the scorer to LayerNorm2's backward IN REGISTERS: no global round trip
// of z / dz, no vmcnt-draining workgroup barriers around the scorer (each was ~1-2 us of the latency-bound B = 256 launch).  The
// index chain of the scorer (tile hint -> cu -> rows -> target / negative -> two table rows) does not depend on any activation: it is
// requested BEFORE the forward half (ScorePre) and has landed long before the scorer needs it.
template <int BM>
struct ScorePre {
    static constexpr int PASSES = BM / 16;
    int b[PASSES], pos[PASSES], n[PASSES], idin[PASSES]; int64_t row[PASSES], tgt[PASSES], ng[PASSES]; float4 ep[PASSES], en[PASSES]; bool ok[PASSES];
};
template <int BM>
__device__ __forceinline__ void score_prefetch(const PostArgs& A, const ScoreTileArgs& S, const int t0, const int T, ScorePre<BM>& P) {
    constexpr int D = 64;
    const int c = (threadIdx.x & 15) * 4, sub = threadIdx.x & 15;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], 0.f);
    const int bh = S.tile_seq[t0 >> 4];
#pragma unroll
    for (int ps = 0; ps < ScorePre<BM>::PASSES; ++ps) {
        const int t = t0 + ps * 16 + (threadIdx.x >> 4);
        P.ok[ps] = t < T;
        P.b[ps] = 0; P.pos[ps] = 0; P.n[ps] = 0; P.row[ps] = 0; P.tgt[ps] = 0; P.ng[ps] = 0;
        P.idin[ps] = (S.ent && P.ok[ps]) ? S.idx32[t] : 0;
        P.ep[ps] = make_float4(0.f, 0.f, 0.f, 0.f); P.en[ps] = P.ep[ps];
        if (P.ok[ps]) {
            const int b = find_seq_from(S.cu, S.B, t, bh), c0 = S.cu[b];
            P.b[ps] = b; P.pos[ps] = t - c0; P.n[ps] = S.cu[b + 1] - c0;
            P.row[ps] = S.rows ? S.rows[b] : b;
            P.tgt[ps] = S.target[P.row[ps] * S.L + P.pos[ps]];
            int64_t ng;
            if (S.sample_neg) {
                ng = sample_neg_id(rk, (uint64_t)b * S.L + P.pos[ps], S.n_items);
                if (sub == 0) S.neg_item[(size_t)b * S.L + P.pos[ps]] = ng;
            } else {
                ng = S.neg_item[(size_t)b * S.L + P.pos[ps]];
            }
            P.ng[ps] = ng < 0 ? 0 : (ng >= S.n_items ? S.n_items - 1 : ng);
            if (P.tgt[ps] > 0 && P.tgt[ps] < S.n_items) { P.ep[ps] = ld4(S.E + P.tgt[ps] * D + c); P.en[ps] = ld4(S.E + P.ng[ps] * D + c); }
        }
    }
}
// the scorer on register rows: q = zreg[pass][0] in, dz out; same arithmetic as score_tile.
// META (MetaModel weighting, metamodel.py:169-194): the 64 -> 64 -> 2 selection MLP of the tile's tokens runs as two small MFMA GEMMs on
// the workgroup's LDS tile instead of per-lane dot loops (each token's 16 lanes read all of W1 twice from L1: +11.5 us on the
// 24 us launch at B = 256): pre = Z W1^T + b1 [BM x 64], then the per-token gate / softmax weight / d hidden, then dzw = dh W1.
// `tile` points at two [BM][68] LDS tiles (the post kernels' regions are free between the two halves).
template <int BM, bool META>
__device__ __forceinline__ void score_tile_regs(const PostArgs& A, const ScoreTileArgs& S, const int t0, const int T, const int tile,
                                                const ScorePre<BM>& P, const float4 (*zreg)[1], float4 (*dzreg)[1], float* red, float* lds_tile) {
    int4* lrec = reinterpret_cast<int4*>(red + 8);        // [BM] the tile's records for tile_sort (S.ent), instead of S.rec in global memory
    constexpr int D = 64, LPT = 16, LD = D + 4, PASSES = ScorePre<BM>::PASSES;
    const int c = (threadIdx.x & 15) * 4, sub = threadIdx.x & 15;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], 0.f);
    float lsum = 0.f, cnt = 0.f;
    float lt_[PASSES], dpos_[PASSES], dneg_[PASSES], wt_[PASSES];
    bool on_[PASSES];
    float4 dzw_[PASSES];
    // ---- pass 1: scores and loss terms (needs nothing but the query row)
    auto terms = [&](const int ps, float& lt, float& dpos, float& dneg) {
        const float4 q = zreg[ps][0], ep = P.ep[ps], en = P.en[ps];
        const float sp = lane_group_sum<LPT>(q.x * ep.x + q.y * ep.y + q.z * ep.z + q.w * ep.w);
        const float sn = lane_group_sum<LPT>(q.x * en.x + q.y * en.y + q.z * en.z + q.w * en.w);
        lt = softplus_f(-sp) + softplus_f(sn);
        dpos = -sigmoid_f(-sp); dneg = sigmoid_f(sn);
    };
    auto pass1 = [&](const int ps) {
        on_[ps] = false; lt_[ps] = 0.f; dpos_[ps] = 0.f; dneg_[ps] = 0.f; wt_[ps] = 1.0f; dzw_[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.ok[ps] && P.tgt[ps] > 0 && P.tgt[ps] < S.n_items) { terms(ps, lt_[ps], dpos_[ps], dneg_[ps]); on_[ps] = true; }
    };
    if constexpr (META) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) pass1(ps);
    }
    if constexpr (META) {
        if (S.phi) {
            constexpr int MD = 64;
            const float* W1 = S.phi;
            const float* b1 = S.phi + MD * MD;
            const float* W2 = b1 + MD;                           // [2][64]
            const float* b2 = W2 + 2 * MD;
            float* Zs = lds_tile;                                // [BM][LD] query rows, later d hidden
            float* Ps = lds_tile + BM * LD;                      // [BM][LD] pre-activations, later dzw
            const uint32_t step = (uint32_t)A.state[DR4SR_STATE_RNGSTEP];
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) st4(Zs + (ps * 16 + (threadIdx.x >> 4)) * LD + c, zreg[ps][0]);      // rows beyond T are zero
            lds_barrier();
            {
                TileAcc<BM, MD> acc;
                tile_zero(acc);
                tile_mma_xwT<BM, MD, MD>(Zs, LD, W1, MD, acc);
                tile_to_lds<BM, MD>(acc, Ps, LD, b1);
            }
            lds_barrier();
            float4 dh4[PASSES];
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                dh4[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on_[ps]) {
                    const int row = ps * 16 + (threadIdx.x >> 4), t = t0 + row, b = P.b[ps], pos = P.pos[ps];
                    const float4 pre4 = ld4(Ps + row * LD + c);
                    const float pre[4] = {pre4.x, pre4.y, pre4.z, pre4.w};
                    unsigned long long gate;
                    if (S.gate_in) gate = S.gate_in[t];
                    else {
                        unsigned long long mine = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) mine |= (unsigned long long)(pre[k] > 0.f) << (4 * sub + k);
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) mine |= __shfl_xor(mine, o, 16);
                        gate = mine;
                    }
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = 4 * sub + k;
                        const float h = ((gate >> j) & 1ull) ? pre[k] : 0.f;
                        s0 = fmaf(W2[j], h, s0);
                        s1 = fmaf(W2[MD + j], h, s1);
                    }
                    s0 = lane_group_sum<16>(s0) + b2[0];
                    s1 = lane_group_sum<16>(s1) + b2[1];
                    float g0, g1;
                    const int64_t pidx = (int64_t)b * S.L + pos;
                    if (S.gumbel) { g0 = S.gumbel[2 * pidx]; g1 = S.gumbel[2 * pidx + 1]; }
                    else {
                        const uint4 r = philox4x32_10(make_uint4((uint32_t)pidx, (uint32_t)(pidx >> 32), 0x6D657461u, step),
                                                      make_uint2((uint32_t)S.meta_seed, (uint32_t)(S.meta_seed >> 32)));
                        const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
                        g0 = -logf(-logf(u0)); g1 = -logf(-logf(u1));
                    }
                    const float y = 1.0f / (1.0f + expf(-((s0 + g0) - (s1 + g1)) * S.inv_tau));
                    const bool forced = S.user_id && S.user_id[P.row[ps]] == 0;        // metamodel.py:180-183: pattern rows -> weight 1
                    wt_[ps] = forced ? 1.0f : y;
                    if (sub == 0) {
                        if (S.w_out) S.w_out[t] = wt_[ps];
                        if (S.gate_out) S.gate_out[t] = gate;
                    }
                    if (!forced) {
                        const float dzl = lt_[ps] * y * (1.0f - y) * S.inv_tau;
                        float dh[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const int j = 4 * sub + k; dh[k] = ((gate >> j) & 1ull) ? (W2[j] - W2[MD + j]) * dzl : 0.f; }
                        dh4[ps] = make_float4(dh[0], dh[1], dh[2], dh[3]);
                    }
                }
            }
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) st4(Zs + (ps * 16 + (threadIdx.x >> 4)) * LD + c, dh4[ps]);       // Zs: the pre GEMM has consumed it
            lds_barrier();
            {
                TileAcc<BM, MD> acc;                             // dzw[t][i] = sum_j dh[t][j] W1[j][i]   (x W form, W1 [out = j][in = i])
                tile_zero(acc);
                tile_mma_xw<BM, MD, MD>(Zs, LD, W1, MD, acc);
                tile_to_lds<BM, MD>(acc, Ps, LD, nullptr);
            }
            lds_barrier();
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps)
                if (on_[ps]) dzw_[ps] = ld4(Ps + (ps * 16 + (threadIdx.x >> 4)) * LD + c);
        }
    }
    // ---- pass 2: dz, table gradient, loss partials.  Without the selection MLP the loss terms are computed here, inside the same
    // branch (one pass, the control flow and register budget of the plain instantiations are those of the form without META)
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.ok[ps]) {
            const int t = t0 + ps * 16 + (threadIdx.x >> 4), b = P.b[ps], pos = P.pos[ps], n = P.n[ps];
            const int64_t row = P.row[ps], tgt = P.tgt[ps], ng = P.ng[ps];
            int4 rec = make_int4(0, 0, 0, 0);
            bool on;
            if constexpr (META) on = on_[ps]; else on = tgt > 0 && tgt < S.n_items;
            if (on) {
                const float4 q = zreg[ps][0], ep = P.ep[ps], en = P.en[ps];
                float lt, dpos, dneg, wt = 1.0f;
                float4 dzw = make_float4(0.f, 0.f, 0.f, 0.f);          // loss_t * d weight_t / d z_t
                if constexpr (META) { lt = lt_[ps]; dpos = dpos_[ps]; dneg = dneg_[ps]; wt = wt_[ps]; dzw = dzw_[ps]; }
                else terms(ps, lt, dpos, dneg);
                if (sub == 0) { lsum += wt * lt; cnt += 1.f; }
                dpos *= wt; dneg *= wt;
                dz = make_float4(dpos * ep.x + dneg * en.x + dzw.x, dpos * ep.y + dneg * en.y + dzw.y, dpos * ep.z + dneg * en.z + dzw.z,
                                 dpos * ep.w + dneg * en.w + dzw.w);
                float* gp = S.dE + tgt * D + c;
                float* gn = S.dE + ng * D + c;
                if (!S.rec) {
                    unsafeAtomicAdd(gp, dpos * q.x); unsafeAtomicAdd(gp + 1, dpos * q.y); unsafeAtomicAdd(gp + 2, dpos * q.z); unsafeAtomicAdd(gp + 3, dpos * q.w);
                    unsafeAtomicAdd(gn, dneg * q.x); unsafeAtomicAdd(gn + 1, dneg * q.y); unsafeAtomicAdd(gn + 2, dneg * q.z); unsafeAtomicAdd(gn + 3, dneg * q.w);
                } else if (sub == 0) {                    // owner-computes mode: one 16-byte record per token, the table rows are summed by k_wgrad's owner job
                    rec = make_int4((int)tgt, (int)ng, __float_as_int(dpos), __float_as_int(dneg));
                }
            }
            if (S.rec && sub == 0) { if (S.ent) lrec[ps * 16 + (threadIdx.x >> 4)] = rec; else S.rec[t] = rec; }
            if (S.ent && sub == 1) reinterpret_cast<int*>(lrec + BM)[ps * 16 + (threadIdx.x >> 4)] = P.idin[ps];
            if (pos == n - 1) {                               // tail positions of this sequence (zero query): loss terms only
                for (int l = n + sub; l < S.L; l += LPT) {
                    const int64_t tl = S.target[row * S.L + l];
                    if (S.sample_neg) S.neg_item[(size_t)b * S.L + l] = sample_neg_id(rk, (uint64_t)b * S.L + l, S.n_items);
                    if (tl > 0 && tl < S.n_items) { lsum += 2.0f * 0.69314718055994530942f; cnt += 1.f; }
                }
            }
        }
        dzreg[ps][0] = dz;
    }
    // workgroup reduction of (cnt, lsum) -> one partial per tile; `red` is scratch no tile region overlaps
    cnt = wave_sum(cnt); lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = cnt; red[2 * (threadIdx.x >> 6) + 1] = lsum; }
    lds_barrier();
    if (threadIdx.x == 0) {
        S.part[2 * tile] = (red[0] + red[2]) + (red[4] + red[6]);
        S.part[2 * tile + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

template <int BM, int D, int F, bool META, bool DET = false>
__device__ __forceinline__ void post_mid_body(const PostArgs& A, const ScoreTileArgs& S) {
    const int T = A.state[DR4SR_STATE_T], bx = xcd_tile(T, BM, A.xcd), t0 = bx * BM;
    if (t0 >= T) return;
    if constexpr (BM != 16) { if (S.ent) tile_sort_init<BM>(S, smem + post_lds_floats(D, F, BM) + 8); }
    if constexpr (D == 64) {
        ScorePre<BM> P;
        if constexpr (BM == 16) score_prefetch<BM>(A, S, t0, T, P);             // latency regime: ahead of the forward half
        float4 zreg[BM / 16][1], dzreg[BM / 16][1];
        PostArgs Af = A;
        if (!S.rec) Af.z = nullptr;                          // the query rows leave the kernel only when the owner job will gather them
        tattn::Keep keep{0xffffffffu, 0xffffffffu};
        post_fwd_body<BM, D, F, false, true>(Af, t0, T, zreg, &keep);
        if constexpr (BM != 16) score_prefetch<BM>(A, S, t0, T, P);             // occupancy regime: its registers would cost a workgroup per CU
        if constexpr (META) lds_barrier();                   // every wave is past the forward half's last LDS reads: the tiles are free
        score_tile_regs<BM, META>(A, S, t0, T, bx, P, zreg, dzreg, smem + post_lds_floats(D, F, BM), smem);
        if constexpr (BM != 16) { if (S.ent) tile_sort<BM>(S, bx, t0, T, smem + post_lds_floats(D, F, BM) + 8); }
        post_bwd_body<BM, D, F, false, DET>(A, t0, T, bx, dzreg, &keep);
    } else {
        tattn::Keep keep{0xffffffffu, 0xffffffffu};
        post_fwd_body<BM, D, F, false, true>(A, t0, T, nullptr, &keep);
        __syncthreads();                               // z rows of this tile are visible to the whole workgroup
        score_tile<BM, D, META>(A, S, t0, T, bx, smem + post_lds_floats(D, F, BM) + 8);
        if constexpr (BM != 16) { if (S.ent) tile_sort<BM>(S, bx, t0, T, smem + post_lds_floats(D, F, BM) + 8); }
        __syncthreads();                               // dz rows written, LDS scratch free again
        post_bwd_body<BM, D, F, false, DET>(A, t0, T, bx, nullptr, &keep);
    }
}
template <int BM, int D, int F, bool META, bool DET = false>
__global__ __launch_bounds__(256) void k_post_mid(const PostArgs A, const ScoreTileArgs S) { post_mid_body<BM, D, F, META, DET>(A, S); }
// The 32-row (at-scale) form with its registers capped at 128: the fused kernel carries the query / dz rows and the scorer's prefetch
// across its two halves (160 VGPRs: 3 waves per SIMD, where k_post_fwd / k_post_bwd run 4); capped it allocates 119 without a
// vector spill and the fourth workgroup per CU (LDS: 4 x 36 KB) is worth 3 % (toys) to 6 % (dense) of the launch at B = 8192.
// (d = 64, plain scorer only: the MetaModel weighting and d = 128 spill under the cap.)
template <int D, int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_post_mid32(const PostArgs A, const ScoreTileArgs S) {
    post_mid_body<32, D, F, false>(A, S);
}

PostArgs make_post_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training) {
    PostArgs A;
    const LayerWs& lw = ws.layer[layer];
    const float* P = p->params;
    A.ctx = lw.ctx; A.x = ws.X[layer];
    A.out_w = P + poff(ws, layer, P_OUT_W); A.out_b = P + poff(ws, layer, P_OUT_B);
    A.ln1_w = P + poff(ws, layer, P_LN1_W); A.ln1_b = P + poff(ws, layer, P_LN1_B);
    A.w1 = P + poff(ws, layer, P_W1); A.b1 = P + poff(ws, layer, P_B1);
    A.w2 = P + poff(ws, layer, P_W2); A.b2 = P + poff(ws, layer, P_B2);
    A.ln2_w = P + poff(ws, layer, P_LN2_W); A.ln2_b = P + poff(ws, layer, P_LN2_B);
    A.u1 = lw.u1; A.y = lw.y; A.st1 = lw.st1; A.a = lw.a; A.h = lw.h; A.u2 = lw.u2; A.st2 = lw.st2; A.z = ws.X[layer + 1];
    A.dz = ws.dX[layer + 1];
    const float* wT = ws.wT + layer * ws.wT_stride;
    const int D = p->D, F = p->F;
    A.out_wT = wT + 3 * D * D; A.w1T = wT + 4 * D * D; A.w2T = wT + 4 * D * D + D * F;
    A.df = lw.df; A.da = lw.da; A.du1 = lw.du1; A.dout = lw.dout; A.dctx = ws.dctx;
    A.rd = p->H == 2 ? ws.attn_rd : nullptr; A.n_head = p->H;
    A.ln_part = ws.ln_part + (size_t)layer * ((ws.Tmax + 15) / 16) * 4 * p->D;      // stride sized for the smallest tile
    A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.eps = p->ln_eps; A.layer = layer; A.training = training;
    A.sP = DR4SR_SITE_PROJ + 4 * layer; A.sA = DR4SR_SITE_ACT + 4 * layer; A.sF = DR4SR_SITE_FFN + 4 * layer;
    A.nx_in_w = A.nx_in_b = nullptr; A.nx_qkv = nullptr; A.up_dqkv = A.up_in_w = A.up_du1 = nullptr;
    if (layer + 1 < p->n_layer && !DR4SR_ENV("DR4SR_NO_FUSE")) {
        A.nx_in_w = P + poff(ws, layer + 1, P_IN_W); A.nx_in_b = P + poff(ws, layer + 1, P_IN_B); A.nx_qkv = ws.layer[layer + 1].qkv;
        A.up_dqkv = ws.layer[layer + 1].dqkv; A.up_in_w = P + poff(ws, layer + 1, P_IN_W); A.up_du1 = ws.layer[layer + 1].du1;
    }
    A.stamps = DR4SR_XENV("DR4SR_STAMPS") ? reinterpret_cast<unsigned long long*>(ws.dctx) : nullptr;   // debug only: dctx is free during fwd
    A.at.on = attn_in_tile(p, ws) ? 1 : 0;
    A.xcd = tile_xcd_order(p, ws) ? 1 : 0;
    // short-sequence plans at d = 128 stage the near half of the window first (attn_tile.h far_rows_if_needed): toys B = 256 0.2115 -> 0.2074 ms;
    // at d = 64 the half window is not worth the second round trip of one tile in ten (0.1049 -> 0.1053).  DR4SR_ATTN_TILE_FULL / _NEAR force
    const bool near_ok = p->expected_tokens > 0 && p->expected_tokens <= 16 * (int64_t)p->B;
    if (A.at.on && !DR4SR_XENV("DR4SR_ATTN_TILE_FULL") && ((near_ok && p->D == 128) || DR4SR_XENV("DR4SR_ATTN_TILE_NEAR"))) A.at.on |= 4;
    if (A.at.on && DR4SR_ENV("DR4SR_ATTN_TILE_ATOMICS")) A.at.on |= 2;       // cross-check: every dK | dV row through atomics (no plain stores)
    A.at.qkv = lw.qkv; A.at.dqkv = lw.dqkv; A.at.ctx = lw.ctx; A.at.stat = lw.attn_st; A.at.tok = ws.tok; A.at.L = p->L; A.at.keep = lw.attn_keep;
    A.wt_attn = attn_fold_fwd(p, ws) ? 1 : 0;
    A.sp = wsplit_of(p, ws, layer);
    if (wfrag_img_on(p, ws)) A.sp = reinterpret_cast<const unsigned short*>(ws.wfrag + (size_t)layer * ws.wT_stride);
    A.at.kv_part = (A.at.on && ws.det_lat && ws.det_kv) ? ws.det_kv + (size_t)layer * ws.det_kv_layer : nullptr;
    A.up_kv_part = (A.at.kv_part && A.up_dqkv) ? ws.det_kv + (size_t)(layer + 1) * ws.det_kv_layer : nullptr;
    A.nx_dqkv_zero = (A.at.on && A.nx_qkv) ? ws.layer[layer + 1].dqkv : nullptr;
    A.dn_dqkv_zero = (A.at.on && layer > 0) ? ws.layer[layer - 1].dqkv : nullptr;
    return A;
}

static size_t post_lds(int D, int F, int bm = 64) { return sizeof(float) * post_lds_floats(D, F, bm); }

template <int BM>
static int post_launch_bm(const dr4sr_sasrec_plan* p, const Workspace& ws, const PostArgs& A, bool bwd, hipStream_t s) {
    dim3 grid((ws.Tmax + BM - 1) / BM), blk(256);
    if (A.xcd) grid.x = xcd_grid((int)grid.x, BM);
    const size_t lds = (BM == 16 && A.at.on) ? sizeof(float) * att_lds_off(p->D, p->F) + att_lds_bytes(p->D) : post_lds(p->D, p->F, BM);
#define PL(D_, F_) do { if (bwd && BM == 16 && A.at.kv_part) { big_lds((k_post_bwd<BM == 16 ? 16 : 64, D_, F_, false, BM == 16>), lds); \
                                                              hipLaunchKernelGGL((k_post_bwd<BM == 16 ? 16 : 64, D_, F_, false, BM == 16>), grid, blk, lds, s, A); } \
                        else if (bwd) { big_lds(k_post_bwd<BM, D_, F_, false>, lds); hipLaunchKernelGGL((k_post_bwd<BM, D_, F_, false>), grid, blk, lds, s, A); } \
                        else { big_lds(k_post_fwd<BM, D_, F_, false>, lds); hipLaunchKernelGGL((k_post_fwd<BM, D_, F_, false>), grid, blk, lds, s, A); } } while (0)
    if (p->D == 64 && p->F == 128) PL(64, 128);
    else if (p->D == 128 && p->F == 128) PL(128, 128);
    else if (p->D == 64 && p->F == 256) PL(64, 256);
    else return DR4SR_E_SHAPE;
#undef PL
    return DR4SR_LAUNCH_CHECK();
}
template <int BM>
static int post_mid_bm(const dr4sr_sasrec_plan* p, const Workspace& ws, const PostArgs& A, const ScoreTileArgs& S, hipStream_t s) {
    dim3 grid((ws.Tmax + BM - 1) / BM), blk(256);
    if (A.xcd) grid.x = xcd_grid((int)grid.x, BM);
    const size_t lds = (BM == 16 && A.at.on) ? sizeof(float) * att_lds_off(p->D, p->F) + att_lds_bytes(p->D)
                       : post_lds(p->D, p->F, BM) + 8 * sizeof(float) + (S.ent ? tile_sort_lds_bytes(BM, 1 << S.logG) : 0);   // + the scorer's (count, loss) reduction scratch + tile_sort's records
#define PM(D_, F_) do { if (BM == 16 && A.at.kv_part) { big_lds((k_post_mid<BM == 16 ? 16 : 64, D_, F_, false, BM == 16>), lds); \
                                                       hipLaunchKernelGGL((k_post_mid<BM == 16 ? 16 : 64, D_, F_, false, BM == 16>), grid, blk, lds, s, A, S); } \
                        else if constexpr (BM == 32 && D_ == 64) { big_lds(k_post_mid32<D_, F_>, lds); hipLaunchKernelGGL((k_post_mid32<D_, F_>), grid, blk, lds, s, A, S); } \
                        else { big_lds(k_post_mid<BM, D_, F_, false>, lds); hipLaunchKernelGGL((k_post_mid<BM, D_, F_, false>), grid, blk, lds, s, A, S); } } while (0)
    if (S.phi) {                                       // MetaModel weighting: D = 64 only (checked by the entry point)
        if (p->D != 64 || p->F != 128) return DR4SR_E_SHAPE;
        if (BM == 16 && A.at.kv_part) { big_lds((k_post_mid<BM == 16 ? 16 : 64, 64, 128, true, BM == 16>), lds);
                                        hipLaunchKernelGGL((k_post_mid<BM == 16 ? 16 : 64, 64, 128, true, BM == 16>), grid, blk, lds, s, A, S); }
        else { big_lds(k_post_mid<BM, 64, 128, true>, lds); hipLaunchKernelGGL((k_post_mid<BM, 64, 128, true>), grid, blk, lds, s, A, S); }
        return DR4SR_LAUNCH_CHECK();
    }
    if (p->D == 64 && p->F == 128) PM(64, 128);
    else if (p->D == 128 && p->F == 128) PM(128, 128);
    else if (p->D == 64 && p->F == 256) PM(64, 256);
    else return DR4SR_E_SHAPE;
#undef PM
    return DR4SR_LAUNCH_CHECK();
}
// post_fwd + scorer/BCE fwd+bwd + post_bwd of the LAST layer in one launch (training step only)
int launch_post_mid(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s, const dr4sr_meta_weighting* mw) {
    const int layer = p->n_layer - 1;
    const PostArgs A = make_post_args(p, ws, layer, training);
    ScoreTileArgs S;
    S.E = p->params + ws.off[0]; S.dE = p->grads + ws.off[0]; S.target = p->item_id; S.rows = p->rows; S.cu = ws.cu; S.tile_seq = ws.tile_seq;
    S.neg_item = p->neg_item; S.part = ws.score_part; S.sample_neg = p->sample_neg; S.n_items = p->n_items; S.B = p->B; S.L = p->L;
    S.rec = de_owner_mode(ws) ? ws.de_rec : nullptr;
    S.ent = nullptr; S.off = nullptr; S.idx32 = ws.idx32; S.logG = 0;
    if (owner_sorted(p, ws, mw != nullptr)) { S.ent = ws.de_ent; S.off = ws.de_off; S.logG = owner_logG(p); }
    S.phi = nullptr; S.gumbel = nullptr; S.user_id = nullptr; S.gate_in = nullptr; S.gate_out = nullptr; S.w_out = nullptr; S.inv_tau = 1.f;
    S.meta_seed = p->seed;
    if (mw) {
        S.phi = mw->phi; S.gumbel = mw->gumbel; S.user_id = mw->user_id; S.gate_in = (const unsigned long long*)mw->gate_in;
        S.gate_out = (unsigned long long*)mw->gate_out; S.w_out = mw->weight_out; S.inv_tau = 1.0f / mw->tau;
    }
    if (wt_mid(p, ws, mw != nullptr)) {                 // DR4SR_WT_RECOMPUTE_A: the backward half recomputes the linear1 pre-activations
        PostArgs Aw = A;
        if (DR4SR_XENV("DR4SR_WT_RECOMPUTE_A")) Aw.a = nullptr;
        return launch_wt_post_mid(Aw, S, ws.Tmax, s);
    }
    const int bm = tile_rows(ws);
    return bm == 16 ? post_mid_bm<16>(p, ws, A, S, s) : bm == 32 ? post_mid_bm<32>(p, ws, A, S, s) : post_mid_bm<64>(p, ws, A, S, s);
}
int launch_post_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    PostArgs A = make_post_args(p, ws, layer, training);
    if (wave_tiles(p, ws) && !A.stamps) {
        // DR4SR_WT_RECOMPUTE_A (round 4, measured slower, opt-in: linear_wave.hip wt_bwd_tile): the forward does not store the linear1
        // pre-activations for a wave-tile backward, which recomputes them from y
        if (wt_bwd_on() && layer + 1 < p->n_layer && DR4SR_XENV("DR4SR_WT_RECOMPUTE_A")) A.a = nullptr;
        return launch_wt_post_fwd(A, ws.Tmax, s);
    }
    const int bm = tile_rows(ws);
    return bm == 16 ? post_launch_bm<16>(p, ws, A, false, s) : bm == 32 ? post_launch_bm<32>(p, ws, A, false, s) : post_launch_bm<64>(p, ws, A, false, s);
}
int launch_post_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    PostArgs A = make_post_args(p, ws, layer, training);
    if (wave_tiles(p, ws) && wt_bwd_on() && layer + 1 < p->n_layer) {
        if (DR4SR_XENV("DR4SR_WT_RECOMPUTE_A")) A.a = nullptr;          // recompute a = y W1^T + b1 (the forward did not store it)
        return launch_wt_post_bwd(A, ws.Tmax, s);
    }
    const int bm = tile_rows(ws);
    return bm == 16 ? post_launch_bm<16>(p, ws, A, true, s) : bm == 32 ? post_launch_bm<32>(p, ws, A, true, s) : post_launch_bm<64>(p, ws, A, true, s);
}

// FMLP Intermediate block (module/layers.py:761-779): linear1 -> GELU -> linear2 -> dropout -> +x -> LayerNorm, D=64, F=256
// token rows per workgroup of the FMLP Intermediate kernels (ln_part rows follow the same tiling)
int ffn_tile_rows(int Tmax) { return at_scale(Tmax) ? 64 : 32; }
int launch_ffn_fwd(const PostArgs& A, int Tmax, hipStream_t s) {
    const int bm = ffn_tile_rows(Tmax);
    dim3 grid((Tmax + bm - 1) / bm), blk(256);
    const size_t lds = post_lds(64, 256, bm);
    if (bm == 32) { big_lds(k_post_fwd<32, 64, 256, true>, lds); hipLaunchKernelGGL((k_post_fwd<32, 64, 256, true>), grid, blk, lds, s, A); }
    else { big_lds(k_post_fwd<64, 64, 256, true>, lds); hipLaunchKernelGGL((k_post_fwd<64, 64, 256, true>), grid, blk, lds, s, A); }
    return DR4SR_LAUNCH_CHECK();
}
int launch_ffn_bwd(const PostArgs& A, int Tmax, hipStream_t s) {
    const int bm = ffn_tile_rows(Tmax);
    dim3 grid((Tmax + bm - 1) / bm), blk(256);
    const size_t lds = post_lds(64, 256, bm);
    if (bm == 32) { big_lds(k_post_bwd<32, 64, 256, true>, lds); hipLaunchKernelGGL((k_post_bwd<32, 64, 256, true>), grid, blk, lds, s, A); }
    else { big_lds(k_post_bwd<64, 64, 256, true>, lds); hipLaunchKernelGGL((k_post_bwd<64, 64, 256, true>), grid, blk, lds, s, A); }
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// dx = dqkv W_in + du1   (x W^T form with W_in^T [D][3D])
template <int BM, int D>
__global__ __launch_bounds__(256) void k_qkv_bwd(const float* __restrict__ dQKV, const float* __restrict__ WT,
                                                 const float* __restrict__ dU1, float* __restrict__ dXo,
                                                 const int* __restrict__ state) {
    constexpr int K = 3 * D, LDA = K + 4, LDC = D + 4;
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * BM;
    if (t0 >= T) return;
    float* As = smem;
    float* Cs = smem + BM * LDA;
    load_tile_bm<BM, K>(As, LDA, dQKV, K, t0, T);
    lds_barrier();
    TileAcc<BM, D> acc;
    tile_zero(acc);
    tile_mma_xw<BM, K, D>(As, LDA, WT, D, acc);
    tile_to_lds<BM, D>(acc, Cs, LDC, nullptr);
    lds_barrier();
    constexpr int C4 = D / 4;
    for (int i = threadIdx.x; i < BM * C4; i += 256) {
        const int row = i / C4, c = (i % C4) * 4;
        if (t0 + row < T) {
            const float4 v = ld4(Cs + row * LDC + c), r = ld4(dU1 + (size_t)(t0 + row) * D + c);
            st4(dXo + (size_t)(t0 + row) * D + c, make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w));
        }
    }
}

int launch_qkv_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, hipStream_t s) {
    const int D = p->D, bm = tile_rows(ws);
    const size_t lds = sizeof(float) * bm * ((3 * D + 4) + (D + 4));
    dim3 grid((ws.Tmax + bm - 1) / bm), blk(256);
    const float* WT = p->params + poff(ws, layer, P_IN_W);      // forward weight used as-is (column-mode B operand)
    const LayerWs& lw = ws.layer[layer];
#define QB(B_) do { if (D == 64) { big_lds(k_qkv_bwd<B_, 64>, lds); hipLaunchKernelGGL((k_qkv_bwd<B_, 64>), grid, blk, lds, s, lw.dqkv, WT, lw.du1, ws.dX[layer], p->state); } \
                    else { big_lds(k_qkv_bwd<B_, 128>, lds); hipLaunchKernelGGL((k_qkv_bwd<B_, 128>), grid, blk, lds, s, lw.dqkv, WT, lw.du1, ws.dX[layer], p->state); } } while (0)
    BM_DISPATCH(bm, QB);
#undef QB
    return DR4SR_LAUNCH_CHECK();
}


// Layer-0 fusion of the backward tail: dx0 = dqkv W_in + du1 stays in LDS and is scattered straight into the tables
// (a3 backward: g = dx0 * mask_emb; dE[idx] += g except padding_idx 0; dP[pos] += g), 16 lanes per token; dP is first
// accumulated in LDS and flushed with one atomic per touched element per workgroup.
template <int BM, int D, bool DET = false>
__device__ __forceinline__ void qkv_embed_bwd_body(const QkvEmbBwdArgs& A, const int t0, const int T) {
    constexpr int K = 3 * D, LDA = K + 4, LDC = D + 4, LPT = D / 4, TPB = 256 / LPT;
    float* As = smem;
    float* Cs = As + BM * LDA;
    float* accP = Cs + BM * LDC;                        // [L][D]
    int* ids = reinterpret_cast<int*>(accP + A.L * D);  // [BM] item id of each row of the tile (0: contributes nothing to dE)
    // latency regime, d = 64: the [3D x D] in_proj fragments are requested with the tile (48 VGPRs) instead of inside the k loop
    constexpr bool PFQ = BM == 16 && D == 64;
    WFragC<PFQ ? K : 16, 64> f_in;
    if constexpr (PFQ) wfrag_load(f_in, A.W, D);
    for (int i = threadIdx.x; i < A.L * D; i += 256) accP[i] = 0.f;
    load_tile_bm<BM, K>(As, LDA, A.dQKV, K, t0, T);
    lds_barrier();
    if constexpr (BM == 16 && DET) {                        // deterministic latency form: layer 0's shared dK | dV rows arrive as partial blocks
        det_kv_rows<D>(As, LDA, const_cast<float*>(A.dQKV), A.kv_part, A.tok, t0, T);
        lds_barrier();
    }
    TileAcc<BM, D> acc;
    tile_zero(acc);
    if constexpr (PFQ) tile_mma_frag<BM, K, D>(As, LDA, f_in, acc);
    else tile_gemm<D == 128 && BM == 32, BM, K, D>(As, LDA, A.W, D, true, A.sp, WSplitGeo<D, 128>::E, 0, acc);
    tile_to_lds<BM, D>(acc, Cs, LDC, nullptr);
    lds_barrier();
    const int c = (threadIdx.x % LPT) * 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const int bh = A.tile_seq[t0 >> 4];
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += TPB) {
        const int r = r0 + threadIdx.x / LPT, t = t0 + r;
        if (r < BM && t < T) {
            const int b = find_seq_from(A.cu, A.B, t, bh), pos = t - A.cu[b];
            const int64_t row = A.rows ? A.rows[b] : b;
            const float4 v = ld4(Cs + r * LDC + c), u = ld4(A.dU1 + (size_t)t * D + c);
            float4 g = make_float4(v.x + u.x, v.y + u.y, v.z + u.z, v.w + u.w);
            if (dodrop) {
                const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * A.L + pos) * D + c);
                g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
            }
            if (A.gout) { st4(A.gout + (size_t)t * D + c, g); continue; }
            float* a = accP + pos * D + c;
            atomicAdd(a, g.x); atomicAdd(a + 1, g.y); atomicAdd(a + 2, g.z); atomicAdd(a + 3, g.w);
            const int64_t id = A.idx[row * A.L + pos];
            st4(Cs + r * LDC + c, g);                        // this thread's own cells: g replaces the GEMM output it has just read
            if (c == 0) ids[r] = (id > 0 && id < A.n_items) ? (int)id : 0;
        } else if (r < BM && c == 0 && !A.gout) ids[r] = 0;
    }
    if (A.gout) return;
    __syncthreads();
    // dE[id] += g: rows of the tile that share an item id are summed FIRST and leave as one row of atomics (by the first of them).
    // A hot row — CL4SRec's mask token is 70 % of a masked view's tokens — otherwise turns the scatter into one chain of
    // same-address atomics per column: +85 us on a 14 us launch, measured (tools/maskprobe.py).
#pragma unroll
    for (int r0 = 0; r0 < BM; r0 += TPB) {
        const int r = r0 + threadIdx.x / LPT;
        const int id = r < BM ? ids[r] : 0;
        unsigned long long same = 0;                        // bit q: row q of the tile carries this row's id
        if constexpr (BM <= LPT) {                          // lane j of a row's lane group looks at row j: one ballot
            const int j = threadIdx.x % LPT;
            const unsigned long long bal = __ballot(j < BM && id > 0 && ids[j < BM ? j : 0] == id);
            same = (bal >> ((threadIdx.x & 63) / LPT * LPT)) & ((1ull << BM) - 1ull);
        } else if (id > 0) {
            for (int q = 0; q < BM; ++q) same |= (unsigned long long)(ids[q] == id) << q;
        }
        if (id > 0 && (same & ((1ull << r) - 1ull)) == 0) {  // no earlier row with this id: this row carries the sum
            float4 g = ld4(Cs + r * LDC + c);
            for (same = r + 1 < 64 ? same >> (r + 1) : 0; same; same &= same - 1) {
                const int q = r + __ffsll((long long)same);
                const float4 o = ld4(Cs + q * LDC + c); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
            }
            float* d = A.dE + (size_t)id * D + c;
            unsafeAtomicAdd(d, g.x); unsafeAtomicAdd(d + 1, g.y); unsafeAtomicAdd(d + 2, g.z); unsafeAtomicAdd(d + 3, g.w);
        }
    }
    for (int i = threadIdx.x; i < A.L * D; i += 256) {
        const float v = accP[i];
        if (v != 0.f) unsafeAtomicAdd(A.dP + i, v);
    }
}
template <int BM, int D, bool DET = false>
__global__ __launch_bounds__(256) void k_qkv_embed_bwd(const QkvEmbBwdArgs A) {
    const int T = A.state[DR4SR_STATE_T], t0 = blockIdx.x * BM;
    if (t0 >= T) return;
    qkv_embed_bwd_body<BM, D, DET>(A, t0, T);
}

static QkvEmbBwdArgs make_qeb_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int training) {
    const LayerWs& lw = ws.layer[0];
    QkvEmbBwdArgs A;
    A.dQKV = lw.dqkv; A.W = p->params + poff(ws, 0, P_IN_W); A.dU1 = lw.du1; A.idx = p->in_item_id; A.rows = p->rows; A.cu = ws.cu; A.tile_seq = ws.tile_seq;
    A.dE = p->grads + ws.off[0]; A.dP = p->grads + ws.off[1]; A.state = p->state; A.B = p->B; A.L = p->L; A.n_items = p->n_items;
    A.training = training; A.seed = p->seed; A.p = p->p_drop;
    A.gout = scatter_in_wgrad(ws) ? ws.dX[0] : nullptr;
    A.kv_part = (ws.det_lat && ws.det_kv && attn_in_tile(p, ws)) ? ws.det_kv : nullptr; A.tok = ws.tok;
    A.sp = wsplit_of(p, ws, 0);
    return A;
}
// latency regime: k_qkv_embed_bwd and k_wgrad are independent (the weight gradients read dqkv / X, not dx0), so the embedding tiles
// run as the first plane of the k_wgrad launch: one launch boundary less per step and the two overlap
bool qeb_in_wgrad(const Workspace& ws) {
    const bool off = DR4SR_ENV("DR4SR_QEB_SEPARATE") != nullptr || DR4SR_ENV("DR4SR_NO_FUSE") != nullptr;
    return !off && tile_rows(ws) == 16 && !ws.det_lat;       // (deterministic latency form: the embedding tiles complete layer 0's dqkv rows BEFORE k_wgrad reads them)
}

int launch_qkv_embed_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s) {
    const int D = p->D, bm = tile_rows(ws);
    const size_t lds = sizeof(float) * (bm * ((3 * D + 4) + (D + 4)) + (size_t)p->L * D + bm);
    dim3 grid((ws.Tmax + bm - 1) / bm), blk(256);
    const QkvEmbBwdArgs A = make_qeb_args(p, ws, training);
    if (wave_tiles(p, ws) && wt_bwd_on() && A.gout) return launch_wt_qkv_embed_bwd(A, ws.Tmax, s);
#define QE(B_) do { if (B_ == 16 && A.kv_part) { if (D == 64) { big_lds((k_qkv_embed_bwd<16, 64, true>), lds); hipLaunchKernelGGL((k_qkv_embed_bwd<16, 64, true>), grid, blk, lds, s, A); } \
                                                 else { big_lds((k_qkv_embed_bwd<16, 128, true>), lds); hipLaunchKernelGGL((k_qkv_embed_bwd<16, 128, true>), grid, blk, lds, s, A); } } \
                    else if (D == 64) { big_lds(k_qkv_embed_bwd<B_, 64>, lds); hipLaunchKernelGGL((k_qkv_embed_bwd<B_, 64>), grid, blk, lds, s, A); } \
                    else { big_lds(k_qkv_embed_bwd<B_, 128>, lds); hipLaunchKernelGGL((k_qkv_embed_bwd<B_, 128>), grid, blk, lds, s, A); } } while (0)
    BM_DISPATCH(bm, QE);
#undef QE
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Weight gradients: dW[n][k] = sum_t G[t][n] X[t][k], db[n] = sum_t G[t][n].
// MFMA 32x32x2 with the TOKEN axis as the contraction: lane (r, g) feeds A[n0+r][t=2s+g] = Gs[t][n0+r]
// and B[t][k0+r] = Xs[t][k0+r] (conflict-free ds_read_b32).  Each workgroup owns the full [NG x KX]
// output (tiles spread over its 4 waves) for a strided subset of token tiles and adds it to the
// flat gradient with 128-B-coalesced fp32 atomics at the end.

template <int NG, int KX>
__device__ __forceinline__ void wgrad_body(const WgradJob& J, const WgradArgs& A, float* __restrict__ part = nullptr) {      // part: wgrad_bf.h
    constexpr int NT = NG / 32, KT = KX / 32, TPW = (NT * KT) / 4;
    static_assert((NT * KT) % 4 == 0, "tile count must split over 4 waves");
    const int T = A.state[DR4SR_STATE_T];
    const int ntiles = (T + 63) / 64;
    float* Gs = smem;                 // [64][NG]
    float* Xs = smem + 64 * NG;       // [64][KX]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, g = lane >> 5;
    f32x16 acc[TPW];
    acc_zero(acc);
    float bsum[(NG + 255) / 256];
#pragma unroll
    for (int i = 0; i < (NG + 255) / 256; ++i) bsum[i] = 0.f;

    // software pipeline (async-STAGE split): the global loads of the NEXT token tile are issued into registers
    // before the MFMA phase of the current one and committed to LDS after it (plain stores: the operands are the saved activations —
    // the recompute modes this commit once had cost ~200 ISA lines per vector even when skipped).
    constexpr int GQ = NG / 16, XQ = KX / 16;      // float4 per thread per tile
    float4 gq[GQ], xq[XQ];
    auto issue = [&](int tt) {                       // rows past T: the last row again (unconditional loads issue back to back), zeroed in commit
        const int t0 = tt * 64;
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / (NG / 4), c = (i % (NG / 4)) * 4, t = min(t0 + row, T - 1);
            gq[u] = ld4(J.G + (size_t)t * J.ldg + J.gcol + c);
        }
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / (KX / 4), c = (i % (KX / 4)) * 4, t = min(t0 + row, T - 1);
            xq[u] = ld4(J.X + (size_t)t * J.ldx + c);
        }
    };
    auto commit = [&](int tt) {
        const int t0 = tt * 64;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / (NG / 4), c = (i % (NG / 4)) * 4;
            st4(Gs + row * NG + c, t0 + row < T ? gq[u] : z4);
        }
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int i = threadIdx.x + 256 * u, row = i / (KX / 4), c = (i % (KX / 4)) * 4;
            st4(Xs + row * KX + c, t0 + row < T ? xq[u] : z4);
        }
    };
    // token tiles of this workgroup: blockIdx.x, + gridDim.x, ... — or, behind token-tile kernels that ran in the XCD-aware order (A.xcd =
    // their tile rows, kernels.h xcd_tile), the 64-token tiles XCD (blockIdx.x & 7) produced: the operands are in this XCD's L2
    const bool xo = A.xcd > 0 && (gridDim.x & 7) == 0;
    const int p64 = xo ? xcd_per((T + A.xcd - 1) / A.xcd, A.xcd) * A.xcd / 64 : 0, xbase = ((int)blockIdx.x & 7) * p64;
    const int kstep = xo ? (int)gridDim.x >> 3 : (int)gridDim.x;
    auto tile_of = [&](int k) { return !xo ? k : (k < p64 ? xbase + k : ntiles); };
    int k = xo ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    int tt = tile_of(k);
    if (tt >= ntiles) return;                            // the grid covers the worst case B*L tokens: no tile, nothing to add
    issue(tt);
    for (; tt < ntiles; k += kstep, tt = tile_of(k)) {
        lds_barrier();                                   // previous MFMA phase has finished reading LDS
        commit(tt);
        lds_barrier();
        { const int tn = tile_of(k + kstep); if (tn < ntiles) issue(tn); }
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
            const int t = 2 * s + g;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int q = w + 4 * i, nt = q / KT, kt = q % KT;
                const float a = Gs[t * NG + nt * 32 + r];
                const float b = Xs[t * KX + kt * 32 + r];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < (NG + 255) / 256; ++i) {
            const int n = threadIdx.x + 256 * i;
            if (n < NG) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
                for (int t = 0; t < 64; t += 4) {
                    s0 += Gs[t * NG + n]; s1 += Gs[(t + 1) * NG + n]; s2 += Gs[(t + 2) * NG + n]; s3 += Gs[(t + 3) * NG + n];
                }
                bsum[i] += (s0 + s1) + (s2 + s3);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int q = w + 4 * i, nt = q / KT, kt = q % KT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
            if (part) part[(size_t)row * KX + kt * 32 + r] = acc[i][e];
            else unsafeAtomicAdd(J.dW + (size_t)row * (J.ldw ? J.ldw : KX) + kt * 32 + r, acc[i][e]);
        }
    }
#pragma unroll
    for (int i = 0; i < (NG + 255) / 256; ++i) {
        const int n = threadIdx.x + 256 * i;
        if (n < NG && J.db) { if (part) part[(size_t)NG * KX + n] = bsum[i]; else unsafeAtomicAdd(J.db + n, bsum[i]); }
    }
}

#include "wgrad_bf.h"        // wgrad_body_bf: the same job as a bf16x3 split (shared with gru.hip)

// sum per-token-tile LayerNorm partials [ntiles][4][D] (rows: ln2_w, ln2_b, ln1_w, ln1_b) into the flat gradient
// (layout ln1_w | ln1_b | ln2_w | ln2_b), tiles strided over gridDim.x blocks; block (0, layer 0) also folds the
// scorer's per-sequence (count, loss) partials into the gradient tail.
__device__ __forceinline__ void reduce_jobs(const WgradArgs& A, const int layer) {
    const int T = A.state[DR4SR_STATE_T], ntiles = (T + A.ln_rows[layer] - 1) / A.ln_rows[layer], D = A.D;
    const float* part = A.ln_part + (size_t)layer * A.ln_layer_stride;
    float* g = A.grads + A.o_ln1_w + (size_t)layer * A.layer_stride;
    for (int c = threadIdx.x; c < 4 * D; c += 256) {
        float s = 0.f;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) s += part[(size_t)t * 4 * D + c];
        const int dstc = c < 2 * D ? c + 2 * D : c - 2 * D;          // partial rows are (ln2, ln1); gradient is (ln1, ln2)
        if (A.det_ln) A.det_ln[((size_t)layer * gridDim.x + blockIdx.x) * 4 * D + dstc] = s;      // deterministic mode: summed over the blocks in order by k_wgrad_det_reduce
        else if (gridDim.x == 1) g[dstc] += s; else unsafeAtomicAdd(g + dstc, s);
    }
    if (blockIdx.x == 0 && layer == 0 && A.score_part) {
        __shared__ float red[512];
        float c = 0.f, l = 0.f;
        const int nsc = A.score_tiles ? (T + A.ln_tile_rows - 1) / A.ln_tile_rows : A.B;      // per-tile partials (fused last layer) or per-sequence
        for (int b = threadIdx.x; b < nsc; b += 256) { c += A.score_part[2 * b]; l += A.score_part[2 * b + 1]; }
        red[threadIdx.x] = c; red[256 + threadIdx.x] = l;
        lds_barrier();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
            lds_barrier();
        }
        if (threadIdx.x == 0) { A.tail[0] += red[0]; A.tail[1] += red[256]; }
    }
}

// a3 backward for large batches: dE[idx[t]] += g[t] (not PAD), dP[pos[t]] += g[t] with g = masked dx0 rows left by
// k_qkv_embed_bwd.  16 lanes per token; dP accumulates in LDS and is flushed once per workgroup.
// deterministic position-table gradient: block k owns the sequences [k C, (k + 1) C); thread (position group pg, column c) adds, sequence by
// sequence in slot order, the masked dx0 rows of the positions pos = pg (mod 256 / D) into register accumulators, and stores them as this
// block's [L][D] partial (k_wgrad_det_reduce sums the blocks in order).  No search, no atomics; loads coalesced over c.
template <int D>
__device__ __forceinline__ void scatter_job_det(const WgradArgs& A) {
    constexpr int PG = 256 / D, NK = (64 + PG - 1) / PG;
    const int c = threadIdx.x % D, pg = threadIdx.x / D, C = (A.B + (int)gridDim.x - 1) / (int)gridDim.x;
    const int b0 = min(A.B, (int)blockIdx.x * C), b1 = min(A.B, b0 + C);
    float acc[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[k] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int pos = pg + PG * k;
            if (pos < n) acc[k] += A.sc_g[(size_t)(t0 + pos) * D + c];
        }
    }
    float* out = A.det_dp + (size_t)blockIdx.x * A.sc_L * D;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int pos = pg + PG * k;
        if (pos < A.sc_L) out[pos * D + c] = acc[k];
    }
}
template <int D>
__device__ __forceinline__ void scatter_job(const WgradArgs& A) {
    constexpr int LPT = D / 4, TPB = 256 / LPT;
    if (A.det_dp) { scatter_job_det<D>(A); return; }
    const int T = A.state[DR4SR_STATE_T], ntiles = (T + 63) / 64;
    if ((int)blockIdx.x >= ntiles) return;
    float* accP = smem;                                   // [L][D]
    for (int i = threadIdx.x; i < A.sc_L * D; i += 256) accP[i] = 0.f;
    __syncthreads();
    const int c = (threadIdx.x % LPT) * 4;
    for (int tt = blockIdx.x; tt < ntiles; tt += gridDim.x) {
#pragma unroll
        for (int r0 = 0; r0 < 64; r0 += TPB) {
            const int t = tt * 64 + r0 + threadIdx.x / LPT;
            if (t < T) {
                const int b = find_seq_from(A.cu, A.B, t, A.sc_tile_seq[t >> 4]), pos = t - A.cu[b];
                const int64_t row = A.sc_rows ? A.sc_rows[b] : b;
                const int64_t id = A.sc_idx[row * A.sc_L + pos];
                const float4 g = ld4(A.sc_g + (size_t)t * D + c);
                float* a = accP + pos * D + c;
                atomicAdd(a, g.x); atomicAdd(a + 1, g.y); atomicAdd(a + 2, g.z); atomicAdd(a + 3, g.w);
                if (id > 0 && id < A.sc_n_items && !A.ow_on) {
                    float* d = A.sc_dE + id * D + c;
                    unsafeAtomicAdd(d, g.x); unsafeAtomicAdd(d + 1, g.y); unsafeAtomicAdd(d + 2, g.z); unsafeAtomicAdd(d + 3, g.w);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A.sc_L * D; i += 256) {
        const float v = accP[i];
        if (v != 0.f) unsafeAtomicAdd(A.sc_dP + i, v);
    }
}

// Owner-computes item-table gradient (large batches).  dE[r] = sum_{t: target_t = r} dpos_t z_t + sum_{t: neg_t = r} dneg_t z_t
//                                                               + sum_{t: idx_t = r} dx0_t
// Owner workgroup o holds the rows {r : r mod G == o} (G = 2^logG; round-robin so that the Zipf-popular low ids spread over the
// owners) as fp32 accumulators in LDS, ONE PRIVATE COPY PER WAVE.  Wave w scans a contiguous quarter of the packed tokens — the
// scorer's 16-byte records {target, negative, dpos, dneg} and the input ids — in order, queues its matches sixteen at a time (so the
// sixteen 256-byte row gathers are in flight together) and adds them to its own copy; at the end the four copies are summed in a
// fixed order and the owner writes its rows.  No atomics anywhere: the result is a pure function of the batch (bit-reproducible),
// and the 8.5 M conflicting fp32 atomics of a toys-shaped B = 8192 step (memory-side, ~55 G/s) are gone.
template <int D>
__device__ __forceinline__ void owner_job(const WgradArgs& A, const int owner) {
    constexpr int NW = 4, VPL = D / 64, QN = 16;
    const int G = 1 << A.ow_logG;
    if (owner >= G) return;
    const int T = A.state[DR4SR_STATE_T], lane = threadIdx.x & 63, w = threadIdx.x >> 6, rpo = A.ow_rpo;
    float* acc = smem + (size_t)w * rpo * D;
    int* qw = reinterpret_cast<int*>(smem + (size_t)NW * rpo * D) + w * QN * 4;        // per-wave queue: {token, local row, coef bits, source}
    for (int i = lane; i < rpo * D; i += 64) acc[i] = 0.f;
    const int Tq = ((T + NW * 64 - 1) / (NW * 64)) * 64, ta = w * Tq, tb = min(T, ta + Tq);
    int qn = 0;
    auto flush = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[QN][VPL]; int lr[QN]; float cf[QN];
#pragma unroll
        for (int k = 0; k < QN; ++k) {
            lr[k] = 0; cf[k] = 0.f;
#pragma unroll
            for (int u = 0; u < VPL; ++u) v[k][u] = 0.f;
            if (k < qn) {
                const int tok = qw[4 * k]; lr[k] = qw[4 * k + 1]; cf[k] = __int_as_float(qw[4 * k + 2]);
                const float* src = (qw[4 * k + 3] ? A.sc_g : A.ow_z) + (size_t)tok * D + lane;
#pragma unroll
                for (int u = 0; u < VPL; ++u) v[k][u] = src[64 * u];
            }
        }
#pragma unroll
        for (int k = 0; k < QN; ++k)
            if (k < qn) {
#pragma unroll
                for (int u = 0; u < VPL; ++u) acc[lr[k] * D + lane + 64 * u] = fmaf(cf[k], v[k][u], acc[lr[k] * D + lane + 64 * u]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        qn = 0;
    };
    auto push = [&](int tok, int id, int cfbits, int which) {
        if (lane == 0) { qw[4 * qn] = tok; qw[4 * qn + 1] = id >> A.ow_logG; qw[4 * qn + 2] = cfbits; qw[4 * qn + 3] = which; }
        if (++qn == QN) flush();
    };
    constexpr int CH = 8;                                   // 64-token slices requested together: one memory round trip per 512 tokens (16: slower)
    if (A.ow_rec) {                                         // stream 1: the scorer's records (fused training step only)
        for (int base = ta; base < tb; base += 64 * CH) {
            int4 r[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                r[k] = make_int4(0, 0, 0, 0);
                if (base + 64 * k + lane < tb) r[k] = A.ow_rec[base + 64 * k + lane];
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                unsigned long long m = __ballot(r[k].x > 0 && (r[k].x & (G - 1)) == owner);
                while (m) {
                    const int L = __ffsll((long long)m) - 1; m &= m - 1;
                    push(base + 64 * k + L, __builtin_amdgcn_readlane(r[k].x, L), __builtin_amdgcn_readlane(r[k].z, L), 0);
                }
                m = __ballot(r[k].x > 0 && r[k].y > 0 && (r[k].y & (G - 1)) == owner);
                while (m) {
                    const int L = __ffsll((long long)m) - 1; m &= m - 1;
                    push(base + 64 * k + L, __builtin_amdgcn_readlane(r[k].y, L), __builtin_amdgcn_readlane(r[k].w, L), 0);
                }
            }
        }
    }
    {                                                       // stream 2: the embedding stage (masked dx0 rows left by k_qkv_embed_bwd)
        for (int base = ta; base < tb; base += 64 * CH) {
            int r[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                r[k] = 0;
                if (base + 64 * k + lane < tb) r[k] = A.ow_idx32[base + 64 * k + lane];
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                unsigned long long m = __ballot(r[k] > 0 && (r[k] & (G - 1)) == owner);
                while (m) {
                    const int L = __ffsll((long long)m) - 1; m &= m - 1;
                    push(base + 64 * k + L, __builtin_amdgcn_readlane(r[k], L), __float_as_int(1.0f), 1);
                }
            }
        }
    }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < rpo * D; i += 256) {
        const float sum = (smem[i] + smem[(size_t)rpo * D + i]) + (smem[(size_t)2 * rpo * D + i] + smem[(size_t)3 * rpo * D + i]);
        const int id = ((i / D) << A.ow_logG) | owner;
        if (sum != 0.f && id < A.sc_n_items) A.sc_dE[(size_t)id * D + (i % D)] += sum;
    }
}

// The same owner job fed by k_post_mid's tile_sort: wave w walks a quarter of the token TILES reading, per tile, the two bytes that
// bracket this owner's bucket (64 tiles per load instruction, CH instructions in flight), then exactly its own entries.  An entry
// is requested by lane `qn` the moment it is queued, so its round trip overlaps the rest of the scan; the flush hands the queued
// entries to every lane through LDS and gathers the sixteen source rows together, as above.  Work per owner: T / BM offset pairs +
// 3 T / G entries instead of T records.
template <int D>
__device__ __forceinline__ void owner_job_sorted(const WgradArgs& A, const int owner) {
    constexpr int NW = 4, VPL = D / 64, QN = 16;
    const int G = 1 << A.ow_logG;
    if (owner >= G) return;
    const int T = A.state[DR4SR_STATE_T], lane = threadIdx.x & 63, w = threadIdx.x >> 6, rpo = A.ow_rpo;
    const int bm = A.ln_tile_rows, E = 3 * bm, ntiles = (T + bm - 1) / bm, OS = G + 4;
    float* acc = smem + (size_t)w * rpo * D;
    int4* qw = reinterpret_cast<int4*>(smem + (size_t)NW * rpo * D) + w * QN;          // per-wave queue of entries
    for (int i = lane; i < rpo * D; i += 64) acc[i] = 0.f;
    const int Tq = ((ntiles + NW * 64 - 1) / (NW * 64)) * 64, ia = w * Tq, ib = min(ntiles, ia + Tq);
    int qn = 0;
    int4 pend = make_int4(0, 0, 0, 0);
    auto flush = [&]() {
        if (lane < qn) qw[lane] = pend;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[QN][VPL]; int lr[QN]; float cf[QN];
#pragma unroll
        for (int k = 0; k < QN; ++k) {
            lr[k] = 0; cf[k] = 0.f;
#pragma unroll
            for (int u = 0; u < VPL; ++u) v[k][u] = 0.f;
            if (k < qn) {
                const int4 e = qw[k];
                lr[k] = e.y; cf[k] = __int_as_float(e.z);
                const float* src = (e.w ? A.sc_g : A.ow_z) + (size_t)e.x * D + lane;
#pragma unroll
                for (int u = 0; u < VPL; ++u) v[k][u] = src[64 * u];
            }
        }
#pragma unroll
        for (int k = 0; k < QN; ++k)
            if (k < qn) {
#pragma unroll
                for (int u = 0; u < VPL; ++u) acc[lr[k] * D + lane + 64 * u] = fmaf(cf[k], v[k][u], acc[lr[k] * D + lane + 64 * u]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        qn = 0;
    };
    constexpr int CH = 4;
    for (int base = ia; base < ib; base += 64 * CH) {
        int lo[CH], hi[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int i = base + 64 * k + lane;
            lo[k] = 0; hi[k] = 0;
            if (i < ib) { const unsigned char* o = A.ow_off + (size_t)i * OS + owner; lo[k] = o[0]; hi[k] = o[1]; }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            unsigned long long m = __ballot(hi[k] > lo[k]);
            while (m) {
                const int L = __ffsll((long long)m) - 1; m &= m - 1;
                const int tile = base + 64 * k + L, e0 = __builtin_amdgcn_readlane(lo[k], L), e1 = __builtin_amdgcn_readlane(hi[k], L);
                for (int e = e0; e < e1; ++e) {
                    if (lane == qn) pend = A.ow_ent[(size_t)tile * E + e];
                    if (++qn == QN) flush();
                }
            }
        }
    }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < rpo * D; i += 256) {
        const float sum = (smem[i] + smem[(size_t)rpo * D + i]) + (smem[(size_t)2 * rpo * D + i] + smem[(size_t)3 * rpo * D + i]);
        const int id = ((i / D) << A.ow_logG) | owner;
        if (sum != 0.f && id < A.sc_n_items) A.sc_dE[(size_t)id * D + (i % D)] += sum;
    }
}

// blockIdx.y = job within layer (0..5: dWq dWk dWv dWo dW1 dW2, 6: reductions; with the embedding scatter: y = 0 is the scatter
// [layer 0 only] and the others shift by one), blockIdx.z = layer
// SUB (at scale, d = 64): every GEMM job is a 64 x 64 block of a weight gradient (the 128-wide jobs of linear1 / linear2 are cut in two),
// so the kernel holds ONE GEMM instantiation of 120 VGPRs instead of the 188 of the <128, 64> / <64, 128> bodies, which capped ALL jobs at
// 2 waves per SIMD: 61 % of a wave's life was s_waitcnt on the operand loads (SQ_WAIT_ANY, profiles/round4_sq_pmc_B8192_toys.txt) with two
// workgroups per CU to hide them.  The cut re-reads y and df once more (+512 B per token and layer: algorithmic bytes unchanged).
// BLK (fp32, batches below the at-scale forms, d = 128): the six whole jobs are cut into 64 x 64 blocks ON THE DEVICE (blockIdx.y ->
// job, row block, column block): a [128 x 128] job is 128 MFMAs per wave and 16 384 atomics per workgroup on 24 token tiles —
// four blocks spread that over four workgroups (measured in DESIGN.md section 5.00, round 4).
template <int D, int F, bool BF, bool SUB = false, bool BLK = false>
__device__ __forceinline__ void wgrad_kernel_body(const WgradArgs& A, const QkvEmbBwdArgs& Q) {
    // latency regime (A.qeb_plane): plane z = 0 of the grid is k_qkv_embed_bwd's work (16-row tiles, block-strided), layers shift by one
    const int layer = A.layer0 + (A.qeb_plane ? (int)blockIdx.z - 1 : (int)blockIdx.z);
    if (!SUB && layer < A.layer0) {
        const int T = A.state[DR4SR_STATE_T], nblk = gridDim.x * gridDim.y;
        for (int tile = blockIdx.y * gridDim.x + blockIdx.x; tile * 16 < T; tile += nblk) {
            qkv_embed_bwd_body<16, D>(Q, tile * 16, T);
            __syncthreads();
        }
        return;
    }
    // the scatter blocks come FIRST in dispatch order (y = 0): the other jobs are persistent loops, so blocks dispatched after
    // the first resident wave would only start when those finish — no overlap
    // (owner planes first, then the dP / atomic scatter plane)
    // (table jobs: plane z = 0 of the launch that carries them — layer 0's unless the data-parallel step moved them, launch_wgrad `table`)
    // (the 64 x 64 fp32 block form runs in the latency regime only, where the table gradient is NOT a set of jobs: no owner / scatter code there)
    int j = (int)blockIdx.y;
    if constexpr (!BLK) {
        const bool tz = (int)blockIdx.z == A.qeb_plane;
        if (A.ow_on && (int)blockIdx.y < A.ow_planes) {
            if (tz) { if (A.ow_ent) owner_job_sorted<D>(A, blockIdx.y * gridDim.x + blockIdx.x); else owner_job<D>(A, blockIdx.y * gridDim.x + blockIdx.x); }
            return;
        }
        j = (int)blockIdx.y - (A.ow_on ? A.ow_planes : 0) - (A.sc_g ? 1 : 0);
        if (j < 0) { if (tz) scatter_job<D>(A); return; }
    }
    if constexpr (BLK) {
        constexpr int RD = D / 64, RF = F / 64, NDD = RD * RD, NB = 4 * NDD + 2 * RD * RF;
        if (j == NB) { reduce_jobs(A, layer); return; }
        int jj, rb, cb;
        if (j < 4 * NDD) { jj = j / NDD; rb = (j % NDD) / RD; cb = (j % NDD) % RD; }
        else if (j < 4 * NDD + RF * RD) { jj = 4; rb = (j - 4 * NDD) / RD; cb = (j - 4 * NDD) % RD; }
        else { jj = 5; rb = (j - 4 * NDD - RF * RD) / RF; cb = (j - 4 * NDD - RF * RD) % RF; }
        WgradJob Jb = A.job[layer * A.jobs_per_layer + jj];
        const int ldw = Jb.ldw ? Jb.ldw : (jj == 5 ? F : D);
        Jb.gcol += 64 * rb; Jb.X += 64 * cb; Jb.dW += (size_t)64 * rb * ldw + 64 * cb; Jb.ldw = ldw;
        Jb.db = (Jb.db && cb == 0) ? Jb.db + 64 * rb : nullptr;
        wgrad_body<64, 64>(Jb, A);
        return;
    }
    if (j == A.jobs_per_layer) { reduce_jobs(A, layer); return; }
    const WgradJob& J = A.job[layer * A.jobs_per_layer + j];
    // deterministic mode: this (job, token split)'s partial block instead of atomics into the gradient
    float* part = A.det ? A.det + ((size_t)(layer * A.jobs_per_layer + j) * gridDim.x + blockIdx.x) * A.det_stride : nullptr;
    if constexpr (SUB) {
        wgrad_body_bf<64, 64>(J, A.state, part);       // (DEEP measured: 176 VGPRs -> 2 waves per SIMD again, 88.6 -> 102 us toys, 770 -> 780 us dense)
    } else if constexpr (BF) {
        if (j < 4) wgrad_body_bf<D, D>(J, A.state, part);
        else if (j == 4) wgrad_body_bf<F, D>(J, A.state, part);
        else wgrad_body_bf<D, F>(J, A.state, part);
    } else {
        if (j < 4) wgrad_body<D, D>(J, A, part);
        else if (j == 4) wgrad_body<F, D>(J, A, part);
        else wgrad_body<D, F>(J, A, part);
    }
}
template <int D, int F>
__global__ __launch_bounds__(256) void k_wgrad(const WgradArgs A, const QkvEmbBwdArgs Q) { wgrad_kernel_body<D, F, false>(A, Q); }
// the at-scale form with the weight-gradient GEMMs on the bf16 matrix cores (a separate kernel: its registers must not weigh on the fp32 one)
template <int D, int F>
__global__ __launch_bounds__(256) void k_wgrad_bf(const WgradArgs A, const QkvEmbBwdArgs Q) { wgrad_kernel_body<D, F, true>(A, Q); }
template <int D, int F>
__global__ __launch_bounds__(256) void k_wgrad_bf64(const WgradArgs A, const QkvEmbBwdArgs Q) { wgrad_kernel_body<D, F, true, true>(A, Q); }
template <int D, int F>
__global__ __launch_bounds__(256) void k_wgrad_blk(const WgradArgs A, const QkvEmbBwdArgs Q) { wgrad_kernel_body<D, F, false, false, true>(A, Q); }


// ---- deterministic mode: the partial results k_wgrad's jobs stored (WgradArgs::det*) summed in a FIXED order into the flat gradient.
// blockIdx.z = layer of the launch, blockIdx.y = GEMM job | jobs_per_layer: the LayerNorm partials | jobs_per_layer + 1 (plane of the launch's
// first layer only): the position-table partials.  One thread per gradient element, the token splits / blocks walked in index order:
// the result is a pure function of the batch.  gw = the k_wgrad launch's grid width, gemm / table: which job kinds that launch carried.
struct DetDims { short ng[DR4SR_WGRAD_MAX_JOBS], kx[DR4SR_WGRAD_MAX_JOBS]; };
// ln_cols: LayerNorm affine sums per layer in det_ln's [4 D] rows (SASRec: 4 D = ln1 w | b | ln2 w | b; FMLP: 2 D = Intermediate LayerNorm w | b)
__global__ __launch_bounds__(256) void k_wgrad_det_reduce(const WgradArgs A, const DetDims dm, const int gw, const int gemm, const int table,
                                                          const int ln_cols) {
    const int layer = A.layer0 + (int)blockIdx.z, j = blockIdx.y, T = A.state[DR4SR_STATE_T];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (j < A.jobs_per_layer) {
        if (!gemm) return;
        const WgradJob& J = A.job[layer * A.jobs_per_layer + j];
        const int NG = dm.ng[j], KX = dm.kx[j], nsplit = min(gw, (T + 63) / 64);
        if (e >= NG * KX + NG) return;
        const float* p = A.det + (size_t)(layer * A.jobs_per_layer + j) * gw * A.det_stride + e;
        if (e >= NG * KX && !J.db) return;
        const float s = det_sum(p, nsplit, (size_t)A.det_stride);
        if (e < NG * KX) J.dW[(size_t)(e / KX) * (J.ldw ? J.ldw : KX) + e % KX] += s;
        else J.db[e - NG * KX] += s;
    } else if (j == A.jobs_per_layer) {
        if (!gemm || e >= ln_cols) return;
        const float* p = A.det_ln + (size_t)layer * gw * 4 * A.D + e;
        const float s = det_sum(p, gw, (size_t)4 * A.D);
        A.grads[A.o_ln1_w + (size_t)layer * A.layer_stride + e] += s;
    } else {
        if (!table || blockIdx.z != 0 || !A.det_dp || e >= A.sc_L * A.D) return;
        A.sc_dP[e] += det_sum(A.det_dp + e, gw, (size_t)A.sc_L * A.D);
    }
}

// ---- FMLP: weight gradients of the Intermediate blocks (dense_1, dense_2) + LayerNorm / scorer partial reductions
__device__ __forceinline__ void reduce_jobs_fmlp(const WgradArgs& A) {
    const int T = A.state[DR4SR_STATE_T], ntiles = (T + A.ln_tile_rows - 1) / A.ln_tile_rows, D = A.D, layer = blockIdx.z;
    const float* part = A.ln_part + (size_t)layer * A.ln_layer_stride;          // [ntiles][4][D], rows 0,1 = (d ln_w, d ln_b)
    float* g = A.grads + A.o_ln1_w + (size_t)layer * A.layer_stride;           // intermediate.LayerNorm.weight | .bias
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        float s = 0.f;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) s += part[(size_t)t * 4 * D + c];
        if (A.det_ln) A.det_ln[((size_t)layer * gridDim.x + blockIdx.x) * 4 * D + c] = s;       // deterministic mode: summed over the blocks in order by k_wgrad_det_reduce
        else unsafeAtomicAdd(g + c, s);
    }
    if (blockIdx.x == 0 && layer == 0 && A.score_part) {
        __shared__ float red[512];
        float c = 0.f, l = 0.f;
        for (int b = threadIdx.x; b < A.B; b += 256) { c += A.score_part[2 * b]; l += A.score_part[2 * b + 1]; }
        red[threadIdx.x] = c; red[256 + threadIdx.x] = l;
        lds_barrier();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
            lds_barrier();
        }
        if (threadIdx.x == 0) { A.tail[0] += red[0]; A.tail[1] += red[256]; }
    }
}
// d(complex_weight)[k][d][2] += fold(dm) (fmlp.hip: k_fmlp_coef is the forward): the first blocks of each layer's reduce row, one
// (k, d) pair per thread — a 6.5 us launch of its own before.  This launch is the only writer of those gradient rows.
__device__ __forceinline__ void fmlp_coef_bwd_job(const WgradArgs& A) {
    __shared__ float ct[64], sn[64];
    const int layer = blockIdx.z, L = A.fc_L, K = L / 2 + 1, D = A.D;
    if ((int)blockIdx.x * 256 >= K * D) return;
    if ((int)threadIdx.x < L) sincospif(2.0f * threadIdx.x / (float)L, &sn[threadIdx.x], &ct[threadIdx.x]);
    lds_barrier();
    float* g = A.grads + A.fc_o_cw + layer * A.layer_stride;
    const float* dml = A.fc_dm + (size_t)layer * L * D;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < K * D; i += gridDim.x * 256) {      // (tiny batches run fewer blocks than pairs / 256)
        const int k = i / D, d = i % D;
        float gr = 0.f, gi = 0.f;
        for (int r = 0; r < L; ++r) {
            const int j = (k * r) % L;
            const float v = dml[r * D + d];
            gr += ct[j] * v;
            gi -= sn[j] * v;
        }
        const float c = ((k == 0 || 2 * k == L) ? 1.f : 2.f) / (float)L;
        g[(k * D + d) * 2] += c * gr;
        g[(k * D + d) * 2 + 1] += c * gi;
    }
}
__global__ __launch_bounds__(256) void k_fmlp_wgrad(const WgradArgs A) {
    const int j = blockIdx.y;
    if (j == 2) { reduce_jobs_fmlp(A); if (A.fc_dm) fmlp_coef_bwd_job(A); return; }
    const WgradJob& J = A.job[blockIdx.z * 6 + 4 + j];
    if (j == 0) wgrad_body<256, 64>(J, A);
    else wgrad_body<64, 256>(J, A);
}
// the two FFN weight-gradient jobs on the bf16 matrix cores as a 3-term split (round 4; wgrad_bf.h): the fp32 form was 37 us of FMLP's
// 0.215 ms step at B = 256 (12 800 positions).  DR4SR_WGRAD_F32: the fp32 kernel (cross-check)
__global__ __launch_bounds__(256) void k_fmlp_wgrad_bf(const WgradArgs A) {
    const int j = blockIdx.y;
    if (j == 2) { reduce_jobs_fmlp(A); if (A.fc_dm) fmlp_coef_bwd_job(A); return; }
    const WgradJob& J = A.job[blockIdx.z * 6 + 4 + j];
    if (j == 0) wgrad_body_bf<256, 64>(J, A.state);
    else wgrad_body_bf<64, 256>(J, A.state);
}
// ... and as 64 x 64 blocks (the k_wgrad_bf64 idea: 32 KB of LDS instead of 80, four workgroups per CU): blockIdx.y 0..3 = the row
// blocks of dW1 [256 x 64], 4..7 = the column blocks of dW2 [64 x 256], 8 = the reduce jobs
__global__ __launch_bounds__(256) void k_fmlp_wgrad_bf64(const WgradArgs A) {
    const int j = blockIdx.y;
    if (j == 8) { reduce_jobs_fmlp(A); if (A.fc_dm) fmlp_coef_bwd_job(A); return; }
    if (A.det) {                                            // deterministic mode (fmlp.hip fmlp_backward): the eight blocks are explicit jobs, results stored per split
        wgrad_body_bf<64, 64>(A.job[blockIdx.z * 8 + j], A.state, A.det + ((size_t)(blockIdx.z * 8 + j) * gridDim.x + blockIdx.x) * A.det_stride);
        return;
    }
    WgradJob J = A.job[blockIdx.z * 6 + 4 + (j >= 4 ? 1 : 0)];
    if (j < 4) { J.gcol += 64 * j; J.dW += (size_t)64 * j * 64; J.ldw = 64; if (J.db) J.db += 64 * j; }
    else { const int kb = j - 4; J.X += 64 * kb; J.dW += 64 * kb; J.ldw = 256; if (kb) J.db = nullptr; }
    wgrad_body_bf<64, 64>(J, A.state);
}
// the owner-computed item-table gradient (owner_job above) as a launch of its own: FMLP's deterministic mode.  rec == NULL: no scorer stream
__global__ __launch_bounds__(256) void k_table_owner64(const WgradArgs A) { owner_job<64>(A, blockIdx.x); }
int launch_table_owner64(const int* state, const int4* rec, const int* idx32, const float* z, const float* g, float* dE, int n_items, hipStream_t s) {
    WgradArgs A{};
    int logG = 8;
    auto lds_of = [&](int lg) { return sizeof(float) * 4 * (size_t)((n_items + (1 << lg) - 1) >> lg) * 64 + 4 * 16 * 4 * sizeof(int); };
    while (lds_of(logG) > 96 * 1024 && logG < 20) ++logG;
    A.state = state; A.ow_on = 1; A.ow_rec = rec; A.ow_idx32 = idx32; A.ow_z = z; A.sc_g = g; A.sc_dE = dE; A.sc_n_items = n_items;
    A.ow_logG = logG; A.ow_rpo = (n_items + (1 << logG) - 1) >> logG;
    const size_t lds = lds_of(logG);
    big_lds(k_table_owner64, lds);
    hipLaunchKernelGGL(k_table_owner64, dim3(1 << logG), dim3(256), lds, s, A);
    return DR4SR_LAUNCH_CHECK();
}
int launch_fmlp_wgrad(const WgradArgs& A, int Tmax, int n_layer, hipStream_t s) {
    const int ntiles = (Tmax + 63) / 64;
    const int gwf = DR4SR_XENV("DR4SR_FMLP_WGRAD_GW") ? atoi(DR4SR_XENV("DR4SR_FMLP_WGRAD_GW")) : 0;     // tuning knob
    int gw_t = gwf > 0 ? gwf : (ntiles / 16 > 64 ? (ntiles / 16 > 160 ? 160 : ntiles / 16) : 64);   // 64: every CU holds one heavy workgroup at B = 256 (48: 37.6 us, 64: 35.4)
    // (64 x 64 bf16x3 blocks, the default: 16 block jobs per layer pair instead of 4 whole ones -> fewer token splits: B = 256 measured
    //  16 / 24 / 32 / 48 / 64 splits = 0.1985 / 0.1926 / 0.1950 / 0.1985 / 0.2015 ms per step)
    const bool blocks64 = !DR4SR_ENV("DR4SR_WGRAD_F32") && !DR4SR_XENV("DR4SR_FMLP_WGRAD_WIDE");
    if (blocks64 && gwf <= 0) gw_t = ntiles / 8 > 24 ? (ntiles / 8 > 160 ? 160 : ntiles / 8) : 24;
    int gw = ntiles < gw_t ? ntiles : gw_t;
    const size_t lds = sizeof(float) * 64 * (64 + 256);
    if (A.det && (!blocks64 || gw > 160 || A.jobs_per_layer != 8)) return DR4SR_E_SHAPE;      // partial blocks exist for the 64 x 64 block form only
    if (DR4SR_ENV("DR4SR_WGRAD_F32")) {
        big_lds(k_fmlp_wgrad, lds);
        hipLaunchKernelGGL(k_fmlp_wgrad, dim3(gw, 3, n_layer), dim3(256), lds, s, A);
    } else if (DR4SR_XENV("DR4SR_FMLP_WGRAD_WIDE")) {          // the two whole jobs per layer (cross-check of the 64 x 64 blocks)
        big_lds(k_fmlp_wgrad_bf, lds);
        hipLaunchKernelGGL(k_fmlp_wgrad_bf, dim3(gw, 3, n_layer), dim3(256), lds, s, A);
    } else {
        hipLaunchKernelGGL(k_fmlp_wgrad_bf64, dim3(gw, 9, n_layer), dim3(256), sizeof(float) * 2 * 64 * 64, s, A);
    }
    if (A.det) {                                            // the stored blocks and LayerNorm sums, added in split order
        DetDims dm;
        for (int jj = 0; jj < 8; ++jj) { dm.ng[jj] = 64; dm.kx[jj] = 64; }
        hipLaunchKernelGGL(k_wgrad_det_reduce, dim3((64 * 64 + 64 + 255) / 256, 8 + 1, n_layer), dim3(256), 0, s, A, dm, gw, 1, 0, 2 * A.D);
    }
    return DR4SR_LAUNCH_CHECK();
}

static bool blk_env_on() { const char* e = DR4SR_XENV("DR4SR_WGRAD_BLK"); return e && atoi(e) != 0; }      // (the fp32 64 x 64 block form has no partial outputs)
bool wgrad_table_jobs(const dr4sr_sasrec_plan* p, const Workspace& ws) { (void)p; return scatter_in_wgrad(ws); }

int launch_wgrad(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, int with_score, hipStream_t s, bool qeb, bool meta,
                 int l_lo, int l_hi, int table) {
    WgradArgs A;
    if (l_hi < 0) l_hi = p->n_layer;
    const bool tbl = table < 0 ? l_lo == 0 : table != 0;    // this launch carries the table jobs
    const bool gemm = l_lo < l_hi;                          // ... and / or weight-gradient jobs of layers [l_lo, l_hi)
    if (l_lo < 0 || l_lo > l_hi || l_hi > p->n_layer || (!gemm && !(tbl && table > 0))) return DR4SR_E_ARG;
    if (table >= 0 && (!scatter_in_wgrad(ws) || with_score == 1 || (qeb && qeb_in_wgrad(ws)))) return DR4SR_E_ARG;   // only where the table gradient IS a set of jobs
    const bool has0 = tbl;                                  // the launch that carries the table / embedding-stage jobs
    A.layer0 = l_lo;
    const int D = p->D, F = p->F;
    float* G = p->grads;
    const bool wg_f32 = DR4SR_ENV("DR4SR_WGRAD_F32") != nullptr;
    A.bf16x3 = (ws.scale_wg && !wg_f32) ? 1 : 0;              // at scale: weight gradients on the bf16 matrix cores (3-term split)
    A.xcd = (tile_xcd_order(p, ws) && !DR4SR_ENV("DR4SR_WGRAD_ORDER_PLAIN")) ? tile_rows(ws) : 0;       // follow the token-tile kernels' XCD-aware order
    // at scale with d = 64: 64 x 64 blocks (k_wgrad_bf64, see wgrad_kernel_body); DR4SR_WGRAD_WIDE: the six whole jobs (cross-check)
    const bool sub64 = A.bf16x3 && wgrad_sub64(p);
    const int NJ = sub64 ? 4 + 2 * (F / 64) : 6;
    A.jobs_per_layer = NJ;
    for (int l = 0; l < p->n_layer; ++l) {
        const LayerWs& lw = ws.layer[l];
        WgradJob* jb = A.job + l * NJ;
        for (int part = 0; part < 3; ++part) {          // in_proj rows [part*D, (part+1)*D)
            WgradJob& J = jb[part];
            J.G = lw.dqkv; J.ldg = 3 * D; J.gcol = part * D;
            J.X = ws.X[l]; J.ldx = D; J.ldw = 0;
            J.dW = G + poff(ws, l, P_IN_W) + (int64_t)part * D * D; J.db = G + poff(ws, l, P_IN_B) + part * D;
        }
        { WgradJob& J = jb[3];                            // out_proj: G = dout (= du1 * mask_proj), X = ctx
          J.G = lw.dout; J.ldg = D; J.gcol = 0;
          J.X = lw.ctx; J.ldx = D; J.ldw = 0;
          J.dW = G + poff(ws, l, P_OUT_W); J.db = G + poff(ws, l, P_OUT_B); }
        if (!sub64) {
            { WgradJob& J = jb[4];                        // linear1: G = da, X = y
              J.G = lw.da; J.ldg = F; J.gcol = 0;
              J.X = lw.y; J.ldx = D; J.ldw = 0;
              J.dW = G + poff(ws, l, P_W1); J.db = G + poff(ws, l, P_B1); }
            { WgradJob& J = jb[5];                        // linear2: G = df (= du2 * mask_ffn), X = h = drop(gelu(a))
              J.G = lw.df; J.ldg = D; J.gcol = 0;
              J.X = lw.h; J.ldx = F; J.ldw = 0;
              J.dW = G + poff(ws, l, P_W2); J.db = G + poff(ws, l, P_B2); }
        } else {
            for (int b = 0; b < F / 64; ++b) {
                WgradJob& J1 = jb[4 + b];                 // linear1 rows [64 b, 64 b + 64): G = da[:, 64 b ..], X = y
                J1.G = lw.da; J1.ldg = F; J1.gcol = 64 * b;
                J1.X = lw.y; J1.ldx = D; J1.ldw = 0;
                J1.dW = G + poff(ws, l, P_W1) + (int64_t)64 * b * D; J1.db = G + poff(ws, l, P_B1) + 64 * b;
                WgradJob& J2 = jb[4 + F / 64 + b];        // linear2 columns [64 b, 64 b + 64): G = df, X = h[:, 64 b ..]
                J2.G = lw.df; J2.ldg = D; J2.gcol = 0;
                J2.X = lw.h + 64 * b; J2.ldx = F; J2.ldw = F;
                J2.dW = G + poff(ws, l, P_W2) + 64 * b; J2.db = b == 0 ? G + poff(ws, l, P_B2) : nullptr;
            }
        }
    }
    A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.training = training;
    const int ntiles = (ws.Tmax + 63) / 64;
    A.ln_part = ws.ln_part; A.ln_layer_stride = (int64_t)((ws.Tmax + 15) / 16) * 4 * D; A.grads = G; A.ln_tile_rows = with_score == 2 ? mid_tile_rows(p, ws, meta) : tile_rows(ws);
    // per-layer partial rows: the wave-tile backward (layers below the last, fused step) leaves one row per 16 tokens
    for (int l = 0; l < p->n_layer; ++l)
        A.ln_rows[l] = (l + 1 < p->n_layer && wave_tiles(p, ws) && wt_bwd_on()) ? 16 : A.ln_tile_rows;
    A.o_ln1_w = poff(ws, 0, P_LN1_W); A.layer_stride = p->n_layer > 1 ? ws.off[2 + 12] - ws.off[2] : 0;
    A.score_part = with_score ? ws.score_part : nullptr; A.tail = G + ws.n_params; A.B = p->B; A.D = D;
    A.score_tiles = with_score == 2;
    const int gw_max = DR4SR_XENV("DR4SR_WGRAD_GW") ? atoi(DR4SR_XENV("DR4SR_WGRAD_GW")) : 48;   // tuning knob
    // token splits per job at scale: 160 up to ~2 500 expected 64-token tiles, then tiles / 16 up to 320 (measured: toys B = 8 192,
    // 1 170 tiles: 128 / 160 / 224 splits -> 120.9 / 117.3 / 121.4 us; dense B = 8 192, 6 400 tiles: 160 / 224 / 320 / 448 -> 905 / 868 /
    // 840 / 854 us; toys B = 32 768: 160 -> 256 splits +1.4 % step).  Without a hint the capacity counts as before (cap 160).
    const int gw_cap_env = DR4SR_XENV("DR4SR_WGRAD_GW_CAP") ? atoi(DR4SR_XENV("DR4SR_WGRAD_GW_CAP")) : 0;
    const int hint_tiles = p->expected_tokens > 0 ? (int)((p->expected_tokens < ws.Tmax ? p->expected_tokens : ws.Tmax) / 64) : 0;
    // round 4 (64 x 64 block jobs, four workgroups per CU): dense B = 8 192 128 / 192 / 256 / 320 splits -> 713 / 681 / 680 / 720 us, toys B = 8 192
    // 96 / 128 / 160 / 192 / 256 -> 92 / 83 / 83 / 89 / 105 us: the upper cap comes down from 320 to 224
    const int gw_hi = sub64 ? 224 : 320;
    // round 5 (64 x 64 block jobs, B = 3 072 / 4 096 / 6 144 toys-shaped rows = 262 / 350 / 525 expected tiles): 96 splits 0.3208 / 0.3467 ms
    // against 0.3350 / 0.3593 at 160 (each workgroup sums >= 3 tiles before its 4 096 atomics); 112 at 525 tiles; 128..160 equal at 700
    const int gw_small = (sub64 && hint_tiles > 0 && hint_tiles < 450) ? 96 : (sub64 && hint_tiles > 0 && hint_tiles < 620) ? 112 : 160;
    const int gw_cap = gw_cap_env > 0 ? gw_cap_env : (hint_tiles / 16 > 160 ? (hint_tiles / 16 > gw_hi ? gw_hi : hint_tiles / 16) : gw_small);
    // floor: 48 splits (36 real tiles at the toys B = 256 batch: one each), 64 once the batch is expected to hold >= 100 tiles
    // (dense B = 256, 200 tiles: k_wgrad 49.4 -> 44.2 us; 80 splits: 44.8)
    const bool gw_env = DR4SR_XENV("DR4SR_WGRAD_GW") != nullptr;
    const int gw_floor = !gw_env && hint_tiles >= 100 && gw_max < 64 ? 64 : gw_max;
    int gw_t = ntiles / 16 > gw_floor ? (ntiles / 16 > gw_cap ? gw_cap : ntiles / 16) : gw_floor;   // >= 16 token tiles per workgroup at scale
    int gw = ntiles < gw_t ? ntiles : gw_t;
    A.sc_g = nullptr;
    const bool scatter = has0 && scatter_in_wgrad(ws) && with_score != 1;       // paired with launch_qkv_embed_bwd (not the unfused debug path)
    if (scatter) {
        A.sc_g = ws.dX[0]; A.sc_idx = p->in_item_id; A.sc_rows = p->rows; A.sc_tile_seq = ws.tile_seq; A.cu = ws.cu;
        A.sc_dE = G + ws.off[0]; A.sc_dP = G + ws.off[1]; A.sc_L = p->L; A.sc_n_items = p->n_items;
    }
    A.qeb_plane = (has0 && qeb && qeb_in_wgrad(ws)) ? 1 : 0;
    const QkvEmbBwdArgs Q = make_qeb_args(p, ws, training);
    size_t lds = sub64 ? wgrad_lds_base(p) : sizeof(float) * 64 * (D + F > 2 * D ? D + F : 2 * D);
    if (scatter && sizeof(float) * p->L * D > lds) lds = sizeof(float) * p->L * D;
    // deterministic mode (Workspace::det): partial buffers instead of atomics + the ordered reduce launch below
    A.det = nullptr; A.det_ln = nullptr; A.det_dp = nullptr; A.det_stride = 0;
    if (ws.det && ws.det_part && ws.scale_wg && !blk_env_on()) {
        //  the ordered position-table partials belong to the OWNER form of the table gradient: with DR4SR_DE_ATOMIC the scatter
        // job keeps its atomics for dE and dP alike (complete, not ordered) instead of building dP partials only and dropping dE
        A.det = ws.det_part; A.det_stride = (int)ws.det_stride; A.det_ln = ws.det_ln; A.det_dp = (scatter && de_owner_mode(ws)) ? ws.det_dp : nullptr;
        A.xcd = 0;          //  k_wgrad_det_reduce sums partial blocks x in [0, min(gw, tiles)): block x must own tile x first — the plain order
    }
    A.ow_ent = nullptr; A.ow_off = nullptr;
    A.ow_on = 0; A.ow_rec = nullptr; A.ow_idx32 = nullptr; A.ow_z = nullptr; A.ow_logG = 0; A.ow_planes = 0; A.ow_rpo = 0;
    if (scatter && de_owner_mode(ws)) {
        // owners = the smallest power of two (>= 256) whose rows fit the launch's LDS four times (one private copy per wave) + queues
        const int logG = owner_logG(p);
        A.ow_on = 1; A.ow_logG = logG; A.ow_rpo = (p->n_items + (1 << logG) - 1) >> logG; A.ow_planes = ((1 << logG) + gw - 1) / gw;
        A.ow_rec = with_score == 2 ? ws.de_rec : nullptr; A.ow_idx32 = ws.idx32; A.ow_z = ws.X[p->n_layer];
        if (with_score == 2 && owner_sorted(p, ws, meta)) { A.ow_ent = ws.de_ent; A.ow_off = ws.de_off; }
    }
    // fp32 jobs as 64 x 64 blocks cut on the device (k_wgrad_blk) below the at-scale forms: toys B = 256 step 0.1234 -> 0.1221 ms at d = 64
    // (three alternating pairs), 0.2284 -> 0.2152 ms at d = 128 (the [128 x 128] jobs were 42 us of it).  DR4SR_WGRAD_BLK=0: whole jobs
    const char* blk_env = DR4SR_XENV("DR4SR_WGRAD_BLK");
    const bool blk64 = !A.bf16x3 && !scatter && (blk_env ? atoi(blk_env) != 0 : !ws.scale);      // (never with table jobs in the launch: k_wgrad_blk has none)
    const int NY = blk64 ? 4 * (D / 64) * (D / 64) + 2 * (D / 64) * (F / 64) : NJ;
    if (blk64 && !scatter) lds = sizeof(float) * 64 * 128;
    if (A.det) {                                            // partial blocks must hold this launch's widest job (checked BEFORE anything is written)
        if (blk64 || gw > DR4SR_DET_MAX_SPLITS) return DR4SR_E_SHAPE;
        for (int jj = 0; jj < NJ; ++jj) {
            const int ng = sub64 ? 64 : (jj == 4 ? F : D), kx = sub64 ? 64 : (jj == 5 ? F : D);
            if (ng * kx + ng > A.det_stride) return DR4SR_E_SHAPE;       // (e.g. DR4SR_WGRAD_WIDE at d = 64: whole jobs do not fit the 64 x 64 partial blocks)
        }
    }
    // (a table-only launch has no GEMM / reduce rows: y counts the owner planes and the scatter plane, one z plane)
    dim3 grid(gw, (gemm ? NY + 1 : 0) + (scatter ? 1 : 0) + A.ow_planes, (gemm ? l_hi - l_lo : 1) + A.qeb_plane), blk(256);
    const size_t lds_q = sizeof(float) * (16 * ((3 * D + 4) + (D + 4)) + (size_t)p->L * D + 16);
    if (A.qeb_plane && lds_q > lds) lds = lds_q;
#define WG(D_, F_) do { if (blk64) { big_lds(k_wgrad_blk<D_, F_>, lds); hipLaunchKernelGGL((k_wgrad_blk<D_, F_>), grid, blk, lds, s, A, Q); } \
                        else if (sub64) { big_lds(k_wgrad_bf64<D_, F_>, lds); hipLaunchKernelGGL((k_wgrad_bf64<D_, F_>), grid, blk, lds, s, A, Q); } \
                        else if (A.bf16x3) { big_lds(k_wgrad_bf<D_, F_>, lds); hipLaunchKernelGGL((k_wgrad_bf<D_, F_>), grid, blk, lds, s, A, Q); } \
                        else { big_lds(k_wgrad<D_, F_>, lds); hipLaunchKernelGGL((k_wgrad<D_, F_>), grid, blk, lds, s, A, Q); } } while (0)
    if (D == 64 && F == 128) WG(64, 128);
    else if (D == 128 && F == 128) WG(128, 128);
    else if (D == 64 && F == 256) WG(64, 256);
    else return DR4SR_E_SHAPE;
#undef WG
    if (A.det) {
        DetDims dm;
        for (int jj = 0; jj < NJ; ++jj) {
            if (sub64) { dm.ng[jj] = 64; dm.kx[jj] = 64; }
            else { dm.ng[jj] = (short)(jj == 4 ? F : D); dm.kx[jj] = (short)(jj == 5 ? F : D); }
        }
        int maxe = 4 * D > p->L * D ? 4 * D : p->L * D;
        for (int jj = 0; jj < NJ; ++jj) {
            if (dm.ng[jj] * dm.kx[jj] + dm.ng[jj] > maxe) maxe = dm.ng[jj] * dm.kx[jj] + dm.ng[jj];
        }
        hipLaunchKernelGGL(k_wgrad_det_reduce, dim3((maxe + 255) / 256, NJ + 2, gemm ? l_hi - l_lo : 1), dim3(256), 0, s, A, dm, gw, gemm ? 1 : 0,
                           scatter ? 1 : 0, 4 * D);
    }
    return DR4SR_LAUNCH_CHECK();
}
