// kernels.h — host-side workspace carve-up and launcher prototypes (internal to libdr4sr_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dr4sr_hip.h"
#include "../../include/dr4sr_hip_hooks.h"
#include "prep_body.h"

#define DR4SR_MAX_LAYERS 8

// per-layer saved activations / gradients, all packed [Tmax, *] fp32 (Tmax = B*L)
struct LayerWs {
    float* qkv;    // [T,3D]  in_proj output (q|k|v)
    float* ctx;    // [T,D]   attention output (heads merged), before out_proj
    float* attn_st; // [T,H,2] softmax row max and 1/row sum per (token, head), saved by the MFMA attention forward
    unsigned* attn_keep; // [T,H,2] dropout keep bits of (token, head)'s key positions 0..31 | 32..63, saved by the wave-per-tile forward (attn_wave.hip)
    float* u1;     // [T,D]   x + drop(out_proj(ctx))                     (LayerNorm1 input)
    float* y;      // [T,D]   LayerNorm1 output
    float* st1;    // [T,2]   (mean, rstd) of u1
    float* a;      // [T,F]   linear1 pre-activation
    float* h;      // [T,F]   drop(gelu(a))  (kept so the weight-gradient pass is a plain GEMM)
    float* u2;     // [T,D]   y + drop(linear2(drop(gelu(a))))            (LayerNorm2 input)
    float* st2;    // [T,2]
    // gradients kept for the weight-gradient pass
    float* df;     // [T,D]   d(linear2 output) = du2 * dropout2 mask
    float* da;     // [T,F]
    float* du1;    // [T,D]   d(LayerNorm1 input) (residual branch of qkv_bwd)
    float* dout;   // [T,D]   d(out_proj output) = du1 * dropout1 mask
    float* dqkv;   // [T,3D]
};

struct Workspace {
    int64_t off[2 + 12 * DR4SR_MAX_LAYERS];   // flat parameter offsets (dr4sr_sasrec_param_layout)
    int64_t n_params;
    int Tmax;
    int* cu;                                   // [B+1] packed row offset of each sequence slot
    bool scale, attn_split;                    // at-scale token-tile forms / length-class attention lists for this plan (see at_scale below)
    bool attn_tile_sa;                         // ... and, for short-sequence plans at d = 64, the window attention of attn_tile.h as launches of its own INSTEAD of the lists (attn_tile_sa.hip)
    int* len_buf;                              // scratch of the two-phase prep (prep_body.h: 16 B per sequence + 16 B per optimizer workgroup)
    int* seq_class;                            // [4 + 7B] n_short, n_long, n_tiny, - | tiny_desc[B] int4 {t0, n, slot, row} | short_list[B] (9..16) | long_list[B] (> 16) | tiny_list[B] (1..8)  (k_prep)
    float* attn_rd;                            // [Tmax][H] <dctx, ctx> per (token, head): softmax-backward row term, from k_post_bwd
    int* tile_seq;                             // [ceil(Tmax/16)] sequence slot of token 16*i (k_prep), search hint of the token-tile kernels
    float* X[DR4SR_MAX_LAYERS + 1];            // X[0] = embedding stage output, X[i+1] = output of layer i
    float* dX[DR4SR_MAX_LAYERS + 1];           // gradients w.r.t. X[i]
    float* dctx;                               // [T,D] scratch
    int4* de_rec;                              // [Tmax] scorer records {target id or 0, negative id, d pos score, d neg score} of the owner-computes table gradient
    int* idx32;                                // [Tmax] input item id per packed token (0 = contributes nothing), written by k_embqkv_fwd
    int2* tok;                                 // [Tmax] {first token of the token's sequence, sequence slot | sequence length << 20 | PAD << 30}, written by k_embqkv_fwd (attn_tile.h)
    int4* de_ent;                              // [3 Tmax] table-gradient entries of each token tile sorted by owner (linear.hip tile_sort)
    unsigned char* de_off;                     // [Tmax / 32 + 1][1028] start offsets of the owners' buckets inside each tile's entries
    bool det;                                  // deterministic summation order (DR4SR_DETERMINISTIC): the at-scale forms whatever the size + ordered partial sums in k_wgrad
    // round 6: `scale` picks the token-tile / attention forms, `scale_wg` the table-gradient and weight-gradient forms (owner / scatter jobs and bf16x3
    // blocks inside k_wgrad).  They differ in ONE case — det_lat, the deterministic mode of a plan in the latency regime: the six latency launches
    // keep their 16-token tiles with the attention inside (scale = false), the attention's shared dK | dV rows go through partial blocks (det_kv
    // [layer][tile][5][16][2 D], TileAttnArgs::kv_part), the embedding stage runs as a launch of its own in front of k_wgrad, and k_wgrad takes its
    // at-scale, ordered forms (scale_wg = true)
    bool scale_wg, det_lat; float* det_kv; int64_t det_kv_layer;
    float* det_part; float* det_ln; float* det_dp; int64_t det_stride;     // ... their partial buffers (NULL: mode off)
    float* wfrag;                              // d = 128, latency forms: fragment-major fp32 image of every layer's weights, E floats per layer (common.h wfrag_load_img)
    unsigned short* wsplit;                    // d = 128 at scale: bf16 hi | lo images of every layer's weights, both orientations (common.h WSplit; k_wsplit)
    int64_t wsplit_E;                          // elements per part and layer (4 D^2 + 2 D F); a layer's block is 4 of them; 0: off
    float* wT;                                 // transposed weights, per layer: in_wT[D,3D] out_wT[D,D] w1T[D,F] w2T[F,D]
    int64_t wT_stride;                         // floats per layer in wT
    float* score_part;                         // [B][2]  per-sequence (count, loss sum) of the scorer
    float* ln_part;                            // [n_layer][ntiles][4][D]  per-token-tile LayerNorm affine grad partials
                                               //   rows: d ln2_w, d ln2_b, d ln1_w, d ln1_b
    LayerWs layer[DR4SR_MAX_LAYERS];
    int64_t bytes;
};

// index of tensor j of layer i in Workspace::off
enum { P_IN_W = 0, P_IN_B, P_OUT_W, P_OUT_B, P_W1, P_B1, P_W2, P_B2, P_LN1_W, P_LN1_B, P_LN2_W, P_LN2_B };
static inline int64_t poff(const Workspace& ws, int layer, int j) { return ws.off[2 + 12 * layer + j]; }
// Launch forms of the SASRec step, set by carve_workspace from the plan (Workspace::scale, Workspace::attn_split).
//   token-tile kernels — latency forms: 16-row tiles, atomics for the table gradient, embedding-stage backward inside k_wgrad (every CU
//     gets work, 4x shorter MFMA chains); at scale (`scale`): 32-row tiles, scatter / owner jobs inside k_wgrad.
//   attention — one workgroup per sequence, or (`attn_split`) the length-class lists walked by persistent launches.
// Measured crossovers (tools/regime_sweep.sh + per-kernel times, toys-shaped and dense batches alike): the tile kernels' at-scale forms
// win from ~5.5 k (toys-shaped) to ~8 k (dense) VALID tokens per step — boundary 7 k —, the attention lists from ~14 k.  Only the host knows the dataset's lengths before the launch, so
// the choice follows the plan's expected_tokens hint and falls back to the capacity B * L (boundary latency_tmax() = 16 384 for both,
// the pre-hint rule; DR4SR_LATENCY_TMAX moves it) when there is none.  GRU4Rec / FMLP keep the capacity rule (at_scale(Tmax)).
int latency_tmax();
static inline bool at_scale(int Tmax) { return Tmax > latency_tmax(); }
constexpr int DR4SR_SCALE_TOKENS = 7168, DR4SR_ATTN_SPLIT_TOKENS = 14336;      // at d = 64; scaled by 64 / d
constexpr int DR4SR_SCALE_TOKENS_SHORT = 10240, DR4SR_SCALE_TOKENS_LONG = 6144;    // ... with the attention inside the tile kernels: expected mean length <= 16 / above
constexpr int DR4SR_SCALE_TOKENS_SHORT_WAVE = 7680;     // round 6: short-sequence plans whose at-scale attention is the wave-per-tile form (attn_wave.hip): tiles AND attention switch here
bool attn_tile_capable(const dr4sr_sasrec_plan* p);
// XCD-aware block -> tile order of the 256-thread token-tile kernels when the attention runs inside them (round 4).  A tile's window
// holds the rows of the tiles in FRONT of it, written by the previous launch: with tile = blockIdx.x those ran on another XCD (block b
// runs on XCD b % 8 — observed, MI355X_MICROARCH.md "Workgroup dispatch": for speed only, never for correctness) and the window comes
// through the fabric; with XCD x owning the CONTIGUOUS tiles [x * per, (x + 1) * per) it is in this XCD's L2 except at the 7 seams.
// Every tile kernel of a step uses the same order, so a tile's own rows stay on one XCD from launch to launch as before.
bool tile_xcd_order(const dr4sr_sasrec_plan* p, const Workspace& ws);       // DR4SR_TILE_ORDER_PLAIN: tile = blockIdx.x
// tiles of BM rows per XCD: ceil(tiles / 8), rounded up to whole 64-token tiles (the weight-gradient launch reads 64-token tiles and
// follows the same order: a workgroup on XCD x sums the tiles XCD x produced — wgrad_body)
__host__ __device__ inline int xcd_per(const int tiles, const int BM) {
    const int q = BM < 64 ? 64 / BM : 1, per = (tiles + 7) >> 3;
    return (per + q - 1) / q * q;
}
static inline int xcd_grid(int tiles, int BM) { return 8 * xcd_per(tiles, BM); }
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_tile(const int T, const int BM, const int on) {
    if (!on) return (int)blockIdx.x;
    const int nt = (T + BM - 1) / BM, per = xcd_per(nt, BM), x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    return (j < per && x * per + j < nt) ? x * per + j : nt;         // nt: no such tile (the caller's t0 >= T test exits)
}
#endif        // shape and switches allow attn_tile.h (the launch forms decide the rest)

// argument blocks shared by the tile kernels of linear.hip (SASRec layer) and their FMLP re-use
// attention inside the 16-token tile kernels of the latency regime (attn_tile.h): this layer's qkv / dqkv / ctx / statistics and the
// per-token words {first token of the sequence, sequence slot | sequence length << 20 | PAD << 30} the embedding stage wrote
struct TileAttnArgs {
    const float* qkv; float* dqkv; float* ctx; float* stat; const int2* tok; int L;
    unsigned* keep;                            // [T][H][2] dropout keep bits saved by the wave-per-tile forward (attn_wave.hip / wt_attn_ctx) for its backward
    int on;                                    // bit 0: on; bit 1 (DR4SR_ATTN_TILE_ATOMICS): no plain stores for tile-private dK | dV rows; bit 2: near rows first (short-sequence plans)
    // deterministic latency form (Workspace::det_lat): the dK | dV rows a query tile adds to OTHER tiles' tokens (and to shared rows of its own) are
    // stored as this layer's partial blocks [query tile][key tile of the window: 5][16 rows][2 D] instead of added with atomics; the launch that
    // consumes the layer's dqkv sums them in query-tile order (linear.hip det_kv_rows).  NULL: atomics
    float* kv_part;
};
struct alignas(16) PostArgs {                   // (16: the argument block behind it in a kernel's kernarg segment keeps its alignment — s_load grouping)
    // forward inputs / saved activations
    const float* ctx; const float* x;
    const float* out_w; const float* out_b; const float* ln1_w; const float* ln1_b;
    const float* w1; const float* b1; const float* w2; const float* b2; const float* ln2_w; const float* ln2_b;
    float* u1; float* y; float* st1; float* a; float* h; float* u2; float* st2; float* z;
    // backward
    const float* dz; const float* w2T; const float* w1T; const float* out_wT;
    float* df; float* da; float* du1; float* dout; float* dctx;
    float* rd; int n_head;                     // optional: rd[t][h] = <dctx[t], ctx[t]> over head h's columns (MFMA attention backward)
    float* ln_part;                            // this layer's [ntiles][4][D] LayerNorm affine partials
    const int* state; uint64_t seed; float p; float eps; int layer; int training;
    uint32_t sP, sA, sF;                       // dropout sites: after out_proj / after activation (0xffffffff = none) / after linear2
    unsigned long long* stamps;                // debug: per-phase s_memtime of block 0 (NULL normally)
    // layer-boundary fusions (NULL = not fused)
    const float* nx_in_w; const float* nx_in_b; float* nx_qkv;     // fwd: also emit qkv of layer+1 = z W_in^T + b
    const float* up_dqkv; const float* up_in_w; const float* up_du1;   // bwd: dz = up_dqkv W_in(layer+1) + up_du1 instead of reading A.dz
    const float* up_kv_part;                   // deterministic latency form: layer + 1's dK | dV partial blocks (TileAttnArgs::kv_part), summed into up_dqkv's tile first
    TileAttnArgs at;                           // at.on: the attention of this layer runs inside the tile kernels (no attention launches)
    int xcd;                                   // 1: XCD-aware block -> tile order (xcd_tile below); the grid is a multiple of 8
    float* nx_dqkv_zero;                       // ... and the launch that emits layer+1's qkv zeroes the K | V rows of its dqkv (atomics target)
    float* dn_dqkv_zero;                       // backward: the K | V rows of layer-1's dqkv, zeroed by the launch in front of their accumulation
    int wt_attn;                               // wave-tile forward kernels (linear_wave.hip), round 6: 1 = the layer's attention forward runs at the head of every tile
                                               // (wt_attn_ctx: from at.qkv / at.tok; ctx, statistics and keep bits are still stored for the backward) — no attention launch
    const unsigned short* sp;                  // bf16x3 tile GEMMs (d = 128 at scale): this layer's split-weight block, layer + 1's right behind (common.h); NULL: fp32.
                                               // Latency forms at d = 128 (16-row tiles): this layer's fragment-major fp32 image instead (as float*; layer + 1's E floats behind)
};

// layer-0 fusions (linear.hip k_embqkv_fwd / k_qkv_embed_bwd and their wave-tile forms)
struct EmbQkvArgs {
    const float* E; const float* P; const int64_t* idx; const int64_t* rows; const int* cu; const int* tile_seq; float* X;
    const float* W; const float* bias; float* QKV; const int* state;
    int B, L, n_items, training; uint64_t seed; float p;
    int* idx32;                                // optional: the item id whose table row receives this token's gradient (0 = none)
    int2* tok; float* dqkv_zero;               // attention in the tile kernels (attn_tile.h): per-token words out, layer 0's dK | dV rows zeroed
    int xcd;                                   // 1: XCD-aware block -> tile order (xcd_tile below); the grid is a multiple of 8
    int wf_layers;                             // latency forms at d = 128: > 0 = this launch also writes the fragment-major fp32 weight images of that many layers to `sp` (wfrag_image_write)
    const unsigned short* sp;                  // at scale: layer 0's split-weight block (bf16x3 tile GEMMs), NULL: fp32 | latency forms, wf_layers > 0: the fp32 image buffer (as float*)
};
struct QkvEmbBwdArgs {
    const float* dQKV; const float* W; const float* dU1; const int64_t* idx; const int64_t* rows; const int* cu; const int* tile_seq;
    float* dE; float* dP; const int* state; int B, L, n_items, training; uint64_t seed; float p;
    float* gout;                 // large batches: masked dx0 rows are stored here and scattered by a job of k_wgrad (overlaps its MFMA work)
    const float* kv_part; const int2* tok;     // deterministic latency form: layer 0's dK | dV partial blocks + the tokens' words (NULL: dQKV is complete)
    const unsigned short* sp;    // layer 0's split-weight block (bf16x3 tile GEMMs), NULL: fp32
};

// scorer half of the fused last-layer launch (k_post_mid / k_wt_post_mid)
struct ScoreTileArgs {
    const float* E; float* dE; const int64_t* target; const int64_t* rows; const int* cu; const int* tile_seq; int64_t* neg_item; float* part;
    int sample_neg, n_items, B, L;
    int4* rec;                                 // owner-computes table gradient: per-token records instead of atomics into dE (NULL: atomics)
    // ... and the tile's table-gradient entries sorted by owner (tile_sort): ent [tiles][3 BM] int4, off [tiles][G + 4] bytes; NULL: the
    // owners scan rec / idx32 themselves
    int4* ent; unsigned char* off; const int* idx32; int logG;
    // MetaModel (DR4SR+) weighted loss, fused: weight_t = selection(z_t; phi) with the masks of metamodel.py:180-185; the loss
    // becomes sum_t weight_t loss_t and dz gains loss_t * d weight_t / d z_t.  phi == NULL: plain BCE.  (d phi is NOT produced
    // here: the inner step never uses it and the hyper-gradient takes it from the deterministic dr4sr_meta_select_bwd.)
    const float* phi; const float* gumbel; const int64_t* user_id; const unsigned long long* gate_in; unsigned long long* gate_out;
    float* w_out; float inv_tau; uint64_t meta_seed;
};

struct WgradJob {
    const float* G; int ldg; int gcol;        // G rows start at column gcol
    const float* X; int ldx;
    float* dW; float* db;                     // db == NULL: no bias sum from this job (a column block of X: the row block's other job adds it)
    int ldw;                                  // row pitch of dW (0: the job's own KX)
};
#define DR4SR_DET_MAX_SPLITS 320              // deterministic mode: token splits the partial buffers are carved for (the launch's cap)
#define DR4SR_WGRAD_MAX_JOBS 12               // per layer: 6 GEMMs, or their 64 x 64 blocks (d = 64: 4 + 2 F / 64)
struct WgradArgs {
    WgradJob job[DR4SR_WGRAD_MAX_JOBS * DR4SR_MAX_LAYERS];
    int jobs_per_layer;                        // stride of job[] (6 unless the launch runs 64 x 64 blocks)
    const int* state; uint64_t seed; float p; int training;
    // reduce jobs (blockIdx.y == 6): LayerNorm affine partials of every layer, scorer partials
    const float* ln_part; int64_t ln_layer_stride; float* grads; int64_t o_ln1_w; int64_t layer_stride;   // ln1_w,ln1_b,ln2_w,ln2_b contiguous
    const float* score_part; float* tail; int B; int D;
    int score_tiles;                           // 1: score_part holds one (count, loss) pair per token tile instead of per sequence
    int ln_tile_rows;                          // token rows per tile of the LAST layer's post kernel (LayerNorm partials, owner-sorted entries)
    int ln_rows[DR4SR_MAX_LAYERS];             // token rows per LayerNorm-partial row, per layer (wave-tile kernels: 16)
    int qeb_plane;                             // 1: grid plane z = 0 runs the embedding-stage backward tiles, layers are z - 1
    int layer0;                                // first layer of this launch (grid z counts from it)
    int bf16x3;                                // 1: weight-gradient GEMMs as a 3-term bf16 split (wgrad_body_bf)
    int xcd;                                   // > 0: the token-tile kernels ran in the XCD-aware order with tiles of this many rows (xcd_tile): wgrad_body follows it
    const float* fc_dm; int64_t fc_o_cw; int fc_L;     // FMLP: filter-coefficient backward as part of the reduce blocks (fc_dm == NULL: none)
    // embedding scatter job (blockIdx.y == 7, large batches; sc_g == NULL: none)
    const float* sc_g; const int64_t* sc_idx; const int64_t* sc_rows; const int* sc_tile_seq; const int* cu;
    float* sc_dE; float* sc_dP; int sc_L; int sc_n_items;
    // owner-computes table gradient (large batches; ow_rec == NULL: the scorer / scatter job use fp32 atomics instead): blockIdx.y <
    // ow_planes are the owner workgroups, owner o = y * gridDim.x + x accumulates the rows {id : id mod 2^ow_logG == o}
    const int4* ow_ent; const unsigned char* ow_off;     // per-tile entries sorted by owner + byte offset tables (k_post_mid's tile_sort); NULL: scan
    int ow_on; const int4* ow_rec; const int* ow_idx32; const float* ow_z; int ow_logG, ow_planes, ow_rpo;     // ow_rec == NULL: no scorer stream (autograd path)
    // deterministic mode (DR4SR_DETERMINISTIC / train.deterministic; det == NULL: off): every job STORES its partial result — GEMM jobs
    // [job][token split][det_stride], the LayerNorm reduce blocks [layer][block][4 D], the position-table scatter [block][L D] — and
    // k_wgrad_det_reduce sums them in a fixed order; the item table is owner-computed as always at scale
    float* det; int det_stride; float* det_ln; float* det_dp;
};

// linear_wave.hip: wave-autonomous 16-token tiles with LDS-resident weights (at scale, d = 64 / FFN 128)
bool wave_tiles(const dr4sr_sasrec_plan* p, const Workspace& ws);
bool wt_bwd_on();                               // the backward wave-tile kernels too (DR4SR_WT_FWD_ONLY: forward only)
int launch_wt_post_fwd(const PostArgs& A, int Tmax, hipStream_t s);
int launch_wt_post_bwd(const PostArgs& A, int Tmax, hipStream_t s);     // writes ln_part rows per 16-token tile

int launch_wt_post_mid(const PostArgs& A, const ScoreTileArgs& S, int Tmax, hipStream_t s);
int launch_wt_embqkv_fwd(const EmbQkvArgs& A, int Tmax, hipStream_t s);
int launch_wt_qkv_embed_bwd(const QkvEmbBwdArgs& A, int Tmax, hipStream_t s);          // gout form only (the scatter is a k_wgrad job)   // score_part / entries per 16-token tile

int ffn_tile_rows(int Tmax);
int launch_ffn_fwd(const PostArgs& A, int Tmax, hipStream_t s);
int launch_ffn_bwd(const PostArgs& A, int Tmax, hipStream_t s);
int launch_fmlp_wgrad(const WgradArgs& A, int Tmax, int n_layer, hipStream_t s);
// deterministic item-table gradient as a launch of its own (FMLP): dE[r] += sum over the tokens, in token order, of rec's scorer terms (coefficient x z
// row; rec == NULL: none) and of the rows g whose id idx32[t] == r — linear.hip owner_job<64>, no atomics
int launch_table_owner64(const int* state, const int4* rec, const int* idx32, const float* z, const float* g, float* dE, int n_items, hipStream_t s);
int launch_adam_flat(float* P, float* G, float* M, float* V, int64_t n, int* state, float lr, float b1, float b2, float eps, float wd, hipStream_t s,
                     float* loss_log = nullptr, const int* log_index = nullptr, const PrepArgs* next = nullptr, int opt = DR4SR_OPT_ADAM);

int carve_workspace(const dr4sr_sasrec_plan* p, Workspace* ws);   // fills ws from p->workspace (or sizes only if NULL)

int launch_prep(const dr4sr_sasrec_plan* p, const Workspace& ws, int bump_rng, int zero_grads, hipStream_t s);
int make_prep_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int bump_rng, PrepArgs* out);
int launch_embed_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s);
int launch_embed_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s);
int launch_unpack(const dr4sr_sasrec_plan* p, const Workspace& ws, const float* X, float* out, int mode, hipStream_t s);
int launch_pack(const dr4sr_sasrec_plan* p, const Workspace& ws, const float* dout, float* dX, int mode, hipStream_t s);

int launch_transpose_weights(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s);
bool tile_bf3(const dr4sr_sasrec_plan* p, const Workspace& ws);              // the 256-thread tile kernels' GEMMs run as a bf16x3 split (d = 128 at scale)
int launch_wsplit(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s);
// the same images for any parameter buffer (FMLP's Intermediate blocks: o_in < 0 = no in_proj / out_proj); E = 4 D^2 + 2 D F elements per part
int launch_wfrag_write(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s);      // d = 128 latency forms without k_embqkv_fwd (DR4SR_NO_FUSE): the fp32 fragment images
bool wfrag_img_on(const dr4sr_sasrec_plan* p, const Workspace& ws);
int launch_wsplit_raw(const float* params, unsigned short* img, int64_t o_in, int64_t o_out, int64_t o_w1, int64_t o_w2, int64_t layer_stride,
                      int E, int D, int F, int n_layer, hipStream_t s);       // ... from the images this launch writes (once per forward pass)
int launch_embqkv_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s);
int launch_qkv_embed_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s);
int launch_post_mid(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s, const dr4sr_meta_weighting* mw = nullptr);
int launch_qkv_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, hipStream_t s);
int launch_post_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);
int launch_post_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);
int launch_qkv_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, hipStream_t s);
// layers [l_lo, l_hi) only (l_hi < 0: all): the upper layers' weight gradients need nothing the lower layers' backward still has to
// compute, so the fused step launches them early on a side stream (step.hip backward_layers); the table-gradient jobs (owner / scatter
// planes), the embedding-stage plane and the scorer partials belong to the launch that holds layer 0
int launch_wgrad(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, int with_score, hipStream_t s, bool qeb = false,
                 bool meta = false, int l_lo = 0, int l_hi = -1, int table = -1);     // meta: the fused last-layer launch carried the MetaModel weighting (its tile size differs)
// table: -1 = the launch that holds layer 0 carries the item / position table jobs (owner planes, scatter plane); 1 = this launch carries
// them whatever its layer range (which may be EMPTY: l_lo == l_hi), 0 = it does not — the two-bucket data-parallel step (step.hip)
bool wgrad_table_jobs(const dr4sr_sasrec_plan* p, const Workspace& ws);      // the table gradient is a set of k_wgrad jobs (at scale, fused step)
bool attn_in_tile(const dr4sr_sasrec_plan* p, const Workspace& ws);    // latency regime: attention inside k_post_fwd / k_post_mid / k_post_bwd (attn_tile.h)
// at scale, where the length-class lists would run (Workspace::attn_split), two heads, fused step: ONE launch per layer and direction, a wave
// per (16-token tile of the packed stream, head[, phase]) from the embedding stage's per-token words (attn_wave.hip, round 6); the lists
// stay as the cross-check (DR4SR_ATTN_LISTS) and for the un-fused step, which writes no token words
bool attn_wave_on(const dr4sr_sasrec_plan* p, const Workspace& ws);
// ... and its FORWARD folded into the wave-tile forward kernels (d = 64 at scale: k_wt_post_fwd / k_wt_post_mid compute the tile's ctx rows
// themselves: one attention launch per layer — the backward — instead of two).  Experiments build only, DR4SR_ATTN_FOLD=1: measured slower
bool attn_fold_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws);
// at scale, short sequences, d = 64 (Workspace::attn_tile_sa): one window-attention launch per layer and direction instead of the
// two / three length-class list launches (attn_tile_sa.hip; the same tattn::fwd / tattn::bwd bodies as the in-tile form)
PostArgs make_post_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training);
int launch_attn_tile_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);
int launch_attn_tile_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);
bool qeb_in_wgrad(const Workspace& ws);         // latency regime: k_qkv_embed_bwd's tiles run as the first plane of the k_wgrad launch

int launch_attn_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);
int launch_attn_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);

// two-layer GRU wavefront (gru_coop.hip): per-layer tensors of both layers, [0] = first layer
struct GruWaveArgs {
    const float* gi1; const float* whh[2]; const float* wih2; const int* cu;
    float* r[2]; float* z[2]; float* n[2]; float* ghn[2]; float* hprev[2]; float* hout[2];
    const float* dhout; float* dgi[2]; float* dgh[2];
};
// raw (plan-independent) launchers shared with the GRU4Rec path
int launch_prep_raw(const int64_t* seqlen, const int64_t* rows, int* cu, int* state, int B, int L, int bump_rng, float* zero,
                    int64_t zero_floats, hipStream_t s);
int launch_prep_raw_hints(const int64_t* seqlen, const int64_t* rows, int* cu, int* state, int B, int L, int bump_rng, float* zero,
                          int64_t zero_floats, int* tile_seq, const PermSel& sel, hipStream_t s);
int launch_embed_fwd_raw(const float* E, const float* P, const int64_t* idx, const int64_t* rows, const int* cu, float* X, int B,
                         int L, int D, int n_items, const int* state, uint64_t seed, float p, int training, hipStream_t s);
int launch_embed_bwd_raw(const float* dX, const int64_t* idx, const int64_t* rows, const int* cu, float* dE, float* dP, int B, int L,
                         int D, int n_items, const int* state, uint64_t seed, float p, int training, hipStream_t s);
int launch_unpack_raw(const float* X, const int* cu, float* out, int B, int L, int D, int last, hipStream_t s);
int launch_pack_raw(const float* dout, const int* cu, float* dX, int B, int L, int D, int last, hipStream_t s);
int launch_score_packed_raw(const float* Z, const float* E, float* dE, float* dZ, const int64_t* target, const int64_t* rows,
                            const int* cu, int64_t* neg_item, int sample_neg, float* part, const int* state, uint64_t seed,
                            int n_items, int B, int L, int D, hipStream_t s, int4* rec = nullptr);      // rec: records instead of dE atomics (gru.hip, deterministic mode)

int launch_attn2_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);   // MFMA, H == 2
int launch_attn2_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s);

int launch_score_packed(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s);
int launch_adam(const dr4sr_sasrec_plan* p, hipStream_t s, const PrepArgs* next = nullptr);
