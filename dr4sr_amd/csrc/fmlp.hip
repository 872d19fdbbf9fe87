// fmlp.hip — FMLP target model (reference model/fmlp.py:18-39, module/layers.py:740-807) on gfx950.
//
// The spectral FilterLayer  irfft(rfft(x, dim=seq, 'ortho') * W, 'ortho')  is, per feature d, a circular convolution
// along the sequence with the real kernel
//     m[r,d] = (1/L) sum_k c_k (Wre[k,d] cos(2 pi k r / L) - Wim[k,d] sin(2 pi k r / L)),   c_0 = c_{L/2} = 1, else 2
// (irfft ignores the imaginary part of the DC and Nyquist bins).  k_fmlp_coef builds m from the complex weight once per
// step; k_fmlp_filter_fwd/bwd apply it (and its transpose) with the whole [L x 64] sequence tile resident in LDS, fused
// with dropout + residual + LayerNorm; the weight gradient is accumulated as dm and folded back to d(complex weight) by
// fmlp_coef_bwd_job (linear.hip, inside k_fmlp_wgrad).  The Intermediate block re-uses the MFMA tile kernels of linear.hip (FFN_ONLY instantiation, F = 256).
// All B*L positions are computed: FMLP rows are left-padded prefixes and the filter mixes every position.
#include "common.h"
#include "kernels.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

#define FM_D 64
#define FM_F 256
#define FM_DMBLK 256                           // workgroups of k_fmlp_filter_bwd (= rows of dm_part per layer)
#define FM_DET_SPLITS 160                      // deterministic mode: token splits of the weight-gradient launch (launch_fmlp_wgrad's cap)
#define FM_DET_STRIDE (64 * 64 + 64)           // ... floats per stored 64 x 64 block + its bias row
#define FS_EMB 0u
#define FS_FILT(l) (1u + 2u * (l))
#define FS_FFN(l) (2u + 2u * (l))

struct FmlpLayerWs {
    float* uf; float* stf; float* xf;        // filter: LN input, (mean,rstd), LN output (= Intermediate input)
    float* a; float* h; float* u2; float* st2;
    float* df; float* da; float* dxf;        // grads: d(dense_2 out)*mask, d(pre-activation), d(xf)
};
struct FmlpWs {
    int64_t off[4 + 9 * DR4SR_MAX_LAYERS];
    int64_t n_params;
    int Tn;
    float* e0; float* st0;
    float* X[DR4SR_MAX_LAYERS + 1]; float* dX[DR4SR_MAX_LAYERS + 1];
    float* m; float* dm;                      // [n_layer][L][D]
    float* dm_part;                           // [n_layer + 1][FM_DMBLK][L][D] per-block partials of dm, last slab: of dP (summed by k_fmlp_dm_reduce)
    float* ln_wg;                             // [n_layer + 1][FM_DMBLK][2][D] per-block partials of the filter LayerNorms' (last slab: the embedding LayerNorm's) d weight | d bias
    float* score_part; float* ln_part;        // [B][2]; [n_layer][ntiles][4][D]
    unsigned short* wsplit;                   // bf16 hi | lo images of linear1 / linear2, both orientations, fragment-major (common.h WSplitGeo): 4 E per layer
    // deterministic mode (DR4SR_DETERMINISTIC / train.deterministic; round 6): no fp32 atomics anywhere in the step.  The item-table gradient is
    // owner-computed (linear.hip owner_job) from the scorer's records de_rec [Tn] {target, negative, dpos, dneg} (zero off the last position) and
    // the ids idx32 [Tn] of the masked LayerNorm-backward rows k_fmlp_embed_bwd leaves in det_g [Tn][D]; the dense_1 / dense_2 weight-gradient blocks and
    // the Intermediate LayerNorm sums are stored per token split (det_part [layer][8][split][64 x 64 + 64], det_ln [layer][split][4 D]) and
    // summed in split order by k_wgrad_det_reduce
    bool det; float* det_part; float* det_ln; int* idx32; int4* de_rec; float* det_g;
    FmlpLayerWs layer[DR4SR_MAX_LAYERS];
    int64_t bytes;
};
enum { FP_CW = 0, FP_FLN_W, FP_FLN_B, FP_W1, FP_B1, FP_W2, FP_B2, FP_ILN_W, FP_ILN_B };
static inline int64_t foff(const FmlpWs& ws, int l, int j) { return ws.off[4 + 9 * l + j]; }

extern "C" int dr4sr_fmlp_plan_sizeof(void) { return (int)sizeof(dr4sr_fmlp_plan); }

extern "C" int64_t dr4sr_fmlp_param_layout(int32_t n_items, int32_t L, int32_t D, int32_t F, int32_t n_layer, int64_t* off) {
    int64_t o = 0;
    auto put = [&](int i, int64_t n) { if (off) off[i] = o; o += n; };
    put(0, (int64_t)n_items * D); put(1, (int64_t)L * D); put(2, D); put(3, D);
    for (int l = 0; l < n_layer; ++l) {
        const int b = 4 + 9 * l;
        put(b + FP_CW, (int64_t)(L / 2 + 1) * D * 2); put(b + FP_FLN_W, D); put(b + FP_FLN_B, D);
        put(b + FP_W1, (int64_t)F * D); put(b + FP_B1, F); put(b + FP_W2, (int64_t)D * F); put(b + FP_B2, D);
        put(b + FP_ILN_W, D); put(b + FP_ILN_B, D);
    }
    return o;
}

static int fmlp_check(const dr4sr_fmlp_plan* p) {
    if (!p || p->abi_version != DR4SR_ABI_VERSION) return DR4SR_E_ARG;
    if (p->B <= 0 || p->n_items < 2 || p->n_layer <= 0 || p->n_layer > DR4SR_MAX_LAYERS) return DR4SR_E_ARG;
    if (p->D != FM_D || p->F != FM_F || p->L <= 0 || p->L > 50 || (p->L & 1)) return DR4SR_E_SHAPE;
    if (!(p->p_drop >= 0.f && p->p_drop < 1.f) || !p->params || !p->state || !p->in_item_id) return DR4SR_E_ARG;
    return 0;
}

static void fmlp_carve(const dr4sr_fmlp_plan* p, FmlpWs* ws) {
    const int64_t D = p->D, F = p->F, Tn = (int64_t)p->B * p->L;
    ws->n_params = dr4sr_fmlp_param_layout(p->n_items, p->L, p->D, p->F, p->n_layer, ws->off);
    ws->Tn = (int)Tn;
    char* base = (char*)p->workspace;
    int64_t o = 0;
    auto take = [&](int64_t nfloat) -> float* {
        float* r = base ? (float*)(base + o) : nullptr;
        o += ((nfloat * 4 + 255) / 256) * 256;
        return r;
    };
    ws->e0 = take(Tn * D); ws->st0 = take(Tn * 2);
    for (int i = 0; i <= p->n_layer; ++i) { ws->X[i] = take(Tn * D); ws->dX[i] = take(Tn * D); }
    ws->m = take((int64_t)p->n_layer * p->L * D); ws->dm = take((int64_t)p->n_layer * p->L * D);
    ws->dm_part = take((int64_t)(p->n_layer + 1) * FM_DMBLK * p->L * D);
    ws->ln_wg = take((int64_t)(p->n_layer + 1) * FM_DMBLK * 2 * D);
    ws->score_part = take(2LL * p->B);
    ws->ln_part = take((int64_t)p->n_layer * ((Tn + 31) / 32) * 4 * D);        // sized for the smallest FFN tile
    ws->wsplit = reinterpret_cast<unsigned short*>(take((int64_t)p->n_layer * 2 * (4 * D * D + 2 * D * F)));
    for (int l = 0; l < p->n_layer; ++l) {
        FmlpLayerWs& w = ws->layer[l];
        w.uf = take(Tn * D); w.stf = take(Tn * 2); w.xf = take(Tn * D);
        w.a = take(Tn * F); w.h = take(Tn * F); w.u2 = take(Tn * D); w.st2 = take(Tn * 2);
        w.df = take(Tn * D); w.da = take(Tn * F); w.dxf = take(Tn * D);
    }
    ws->det = DR4SR_ENV("DR4SR_DETERMINISTIC") != nullptr && atoi(DR4SR_ENV("DR4SR_DETERMINISTIC")) != 0;
    ws->det_part = nullptr; ws->det_ln = nullptr; ws->idx32 = nullptr; ws->de_rec = nullptr; ws->det_g = nullptr;
    if (ws->det) {
        ws->det_part = take((int64_t)p->n_layer * 8 * FM_DET_SPLITS * FM_DET_STRIDE);
        ws->det_ln = take((int64_t)p->n_layer * FM_DET_SPLITS * 4 * D);
        ws->idx32 = reinterpret_cast<int*>(take(Tn));
        ws->de_rec = reinterpret_cast<int4*>(take(4 * Tn));
        ws->det_g = take(Tn * D);
    }
    ws->bytes = o;
}

extern "C" int64_t dr4sr_fmlp_workspace_bytes(const dr4sr_fmlp_plan* plan) {
    if (!plan || plan->B <= 0 || plan->L <= 0 || plan->n_layer <= 0 || plan->n_layer > DR4SR_MAX_LAYERS) return DR4SR_E_ARG;
    if (plan->D != FM_D || plan->F != FM_F || plan->L > 50 || (plan->L & 1)) return DR4SR_E_SHAPE;       // as fmlp_check
    dr4sr_fmlp_plan q = *plan;
    q.workspace = nullptr;
    FmlpWs ws;
    fmlp_carve(&q, &ws);
    return ws.bytes;
}

static int fmlp_ws(const dr4sr_fmlp_plan* p, FmlpWs* ws) {
    int rc = fmlp_check(p);
    if (rc) return rc;
    if (!p->workspace) return DR4SR_E_ARG;
    fmlp_carve(p, ws);
    return ws->bytes > p->workspace_bytes ? DR4SR_E_WS : 0;
}

// ------------------------------------------------------------------------------------------------ prep + coefficients
// block 0: state[T] = B*L, RNG bump; blocks 1 .. zb: zero the flat gradient (+tail); the last n_layer * FM_COEF_BLK blocks:
// m[l][r][d] from complex_weight[k][d][2] (one output per thread) and dm = 0 — a launch of its own (4.9 us) before.
#define FM_COEF_BLK 4                          // x 1024 threads >= L * 64 outputs per layer (L <= 50)
struct FPrepArgs {
    int* state; int Tn, bump; float* zero; int64_t n4; int zb;
    const float* params; int64_t o_cw0, layer_stride; float* m; float* dm; int L;
    PermSel sel; int64_t* rows; int B;                      // round 4: the batch selection of the fused step (sel.perm == NULL: none)
    // round 5: FM_SPLIT_BLK more blocks per layer, behind the n_coef_blocks coefficient blocks, write the bf16 hi | lo fragment-major images of
    // linear1 / linear2 (common.h wsplit_elem; wsplit == NULL: none)
    unsigned short* wsplit; int64_t o_w1, o_w2; int n_coef_blocks;
};
constexpr int FM_SPLIT_BLK = 8;
__global__ __launch_bounds__(1024) void k_fmlp_prep(const FPrepArgs A) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) { A.state[DR4SR_STATE_T] = A.Tn; if (A.bump) A.state[DR4SR_STATE_RNGSTEP] += 1; }
        if (A.sel.perm) {                                   // rows[i] = perm[(c stride + offset + i) mod n]; c++  (was a launch of its own)
            const int64_t c = *A.sel.counter;
            for (int i = threadIdx.x; i < A.B; i += 1024) A.rows[i] = A.sel.perm[(c * A.sel.stride + A.sel.offset + i) % A.sel.n];
            __syncthreads();
            if (threadIdx.x == 0) *A.sel.counter = (int)(c + 1);
        }
        return;
    }
    if ((int)blockIdx.x <= A.zb) {
        for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < A.n4; i += (int64_t)A.zb * 1024)
            st4(A.zero + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    __shared__ float ct[64], sn[64];
    const int c = blockIdx.x - 1 - A.zb, L = A.L, K = L / 2 + 1;
    if (c >= A.n_coef_blocks) {                              // split-weight blocks
        constexpr int E = 4 * FM_D * FM_D + 2 * FM_D * FM_F;
        const int q = c - A.n_coef_blocks, layer_s = q / FM_SPLIT_BLK;
        unsigned short* base = A.wsplit + (size_t)layer_s * 4 * E;
        for (int e = 4 * FM_D * FM_D + (q % FM_SPLIT_BLK) * 1024 + threadIdx.x; e < E; e += FM_SPLIT_BLK * 1024)
            wsplit_elem(A.params, base, e, -1, -1, A.o_w1, A.o_w2, layer_s * A.layer_stride, E, FM_D, FM_F);
        return;
    }
    const int layer = c / FM_COEF_BLK;
    if ((int)threadIdx.x < L) sincospif(2.0f * threadIdx.x / (float)L, &sn[threadIdx.x], &ct[threadIdx.x]);
    __syncthreads();
    const float* cw = A.params + A.o_cw0 + layer * A.layer_stride;
    const int i = (c % FM_COEF_BLK) * 1024 + threadIdx.x;
    if (i < L * FM_D) {
        float acc = 0.f;
        const int r = i / FM_D, d = i % FM_D;
        for (int k = 0; k < K; ++k) {
            const int j = (k * r) % L;
            const float cf = (k == 0 || 2 * k == L) ? 1.f : 2.f;
            acc += cf * (cw[(k * FM_D + d) * 2] * ct[j] - cw[(k * FM_D + d) * 2 + 1] * sn[j]);
        }
        A.m[(size_t)layer * L * FM_D + i] = acc / (float)L;
        A.dm[(size_t)layer * L * FM_D + i] = 0.f;
    }
}
// dm[layer][i] = sum over the nblk per-workgroup partials written by k_fmlp_filter_bwd (fixed order -> deterministic).
// History of this launch: one thread walking all 256 partials of a column (26 workgroups): 60 us of dependent-latency loads; 64
// columns per workgroup, partials split over 4 waves, 8 four-byte loads in flight per thread: 19.5 us; now 6.1 us (below).
// blockIdx.y == n_layer: the slab of the position-table gradient partials left by k_fmlp_embed_bwd -> dP (this launch is its only writer)
// Thread = (4 columns, one of 16 partial sub-rows): 16-byte loads, 4 in flight per thread (16 KB per workgroup: the launch is bound
// by the bytes it keeps in flight, 9.8 MB through ~2 us round trips), then the 16 sub-rows meet in LDS in a fixed order.
// blockIdx.x >= nx (two more blocks per slab): the LayerNorm weight | bias partials of the same workgroups (ln_wg) -> grads; 256
// workgroups adding into the same 128 words were 6 of k_fmlp_filter_bwd's 18.4 us.
struct FDmRedArgs {
    const float* part; float* dm; float* dP; const float* ln_wg; float* grads;
    int64_t o_fln_w, o_fln_b, layer_stride, o_eln_w, o_eln_b;
    int nblk, n, n_layer, nx;
};
__global__ __launch_bounds__(256) void k_fmlp_dm_reduce(const FDmRedArgs A) {
    __shared__ float4 red[16][16];
    const int layer = blockIdx.y, c4 = threadIdx.x & 15, sub = threadIdx.x >> 4, nblk = A.nblk;
    const bool ln = (int)blockIdx.x >= A.nx;
    const int n = ln ? 2 * FM_D : A.n, i = (ln ? (int)blockIdx.x - A.nx : (int)blockIdx.x) * 64 + 4 * c4;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {                                           // n is a multiple of 64 (L * 64 columns)
        const float* p = (ln ? A.ln_wg : A.part) + (size_t)layer * FM_DMBLK * n + i;
        for (int b = sub; b < nblk; b += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (b + 16 * u < nblk) {
                    const float4 v = ld4(p + (size_t)(b + 16 * u) * n);
                    acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
                }
        }
    }
    red[sub][c4] = make_float4((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                               (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w));
    __syncthreads();
    if (sub == 0 && i < n) {
        float4 s4 = red[0][c4];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][c4]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
        if (!ln) st4((layer == A.n_layer ? A.dP : A.dm + (size_t)layer * n) + i, s4);
        else {                                             // this launch is the only writer of these rows of the step's gradient
            const bool wgt = i < FM_D;
            float* g = A.grads + (layer == A.n_layer ? (wgt ? A.o_eln_w : A.o_eln_b) : (wgt ? A.o_fln_w : A.o_fln_b) + layer * A.layer_stride) + (wgt ? i : i - FM_D);
            const float4 o = ld4(g);
            st4(g, make_float4(o.x + s4.x, o.y + s4.y, o.z + s4.z, o.w + s4.w));
        }
    }
}
// ------------------------------------------------------------------------------------------------ embedding + LayerNorm
struct FEmbArgs {
    const float* E; const float* P; const float* lnw; const float* lnb;
    const int64_t* idx; const int64_t* rows;
    float* e0; float* st0; float* x0;
    const float* dx0; float* dE; float* dP; float* ln_wg;
    int B, L, n_items; float eps; const int* state; uint64_t seed; float p; int training;
    float* g_out; int* idx32;                  // deterministic mode (idx32 != NULL): the rows go to g_out + their ids, no atomics
};

__global__ __launch_bounds__(256) void k_fmlp_embed_fwd(const FEmbArgs A) {
    const int l16 = threadIdx.x & 15, c = 4 * l16;
    const int64_t Tn = (int64_t)A.B * A.L;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const float4 gam = ld4(A.lnw + c), bet = ld4(A.lnb + c);
    for (int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); t < Tn; t += (int64_t)gridDim.x * 16) {
        const int b = (int)(t / A.L), l = (int)(t % A.L);
        const int64_t row = A.rows ? A.rows[b] : b;
        int64_t id = A.idx[row * A.L + l];
        id = id < 0 ? 0 : (id >= A.n_items ? A.n_items - 1 : id);
        const float4 e = ld4(A.E + id * FM_D + c), pe = ld4(A.P + (size_t)l * FM_D + c);
        float4 v[1] = {make_float4(e.x + pe.x, e.y + pe.y, e.z + pe.z, e.w + pe.w)};
        st4(A.e0 + t * FM_D + c, v[0]);
        float mean, rstd;
        ln_stats16<1>(v, mean, rstd, A.eps);
        float4 y = make_float4((v[0].x - mean) * rstd * gam.x + bet.x, (v[0].y - mean) * rstd * gam.y + bet.y,
                               (v[0].z - mean) * rstd * gam.z + bet.z, (v[0].w - mean) * rstd * gam.w + bet.w);
        if (dodrop) { const float4 mk = drop4(rk, FS_EMB, (uint64_t)t * FM_D + c); y.x *= mk.x; y.y *= mk.y; y.z *= mk.z; y.w *= mk.w; }
        st4(A.x0 + t * FM_D + c, y);
        if (l16 == 0) { A.st0[2 * t] = mean; A.st0[2 * t + 1] = rstd; }
    }
}

// backward: g = dx0*mask -> LN' -> de;  dE[idx] += de (idx != 0), dP[pos] += de, LN affine grads.
// Block-strided over sequences; dP and the affine grads are accumulated in registers and added once per block.
__global__ __launch_bounds__(256) void k_fmlp_embed_bwd(const FEmbArgs A) {
    __shared__ float scr[4 * 2 * FM_D];
    const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4, c = 4 * l16;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const float4 gamv = ld4(A.lnw + c);
    float4 gam[1] = {gamv}, dgam[1] = {make_float4(0.f, 0.f, 0.f, 0.f)}, dbet[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
    float4 accP[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accP[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
        const int64_t row = A.rows ? A.rows[b] : b;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int l = ps * 16 + rsub;
            if (l < A.L) {                               // uniform per 16-lane group
                const int64_t t = (int64_t)b * A.L + l;
                float4 g[1] = {ld4(A.dx0 + t * FM_D + c)};
                if (dodrop) { const float4 mk = drop4(rk, FS_EMB, (uint64_t)t * FM_D + c); g[0].x *= mk.x; g[0].y *= mk.y; g[0].z *= mk.z; g[0].w *= mk.w; }
                const float4 u[1] = {ld4(A.e0 + t * FM_D + c)};
                ln_bwd_row<1>(g, u, A.st0[2 * t], A.st0[2 * t + 1], gam, dgam, dbet);
                accP[ps].x += g[0].x; accP[ps].y += g[0].y; accP[ps].z += g[0].z; accP[ps].w += g[0].w;
                const int64_t id = A.idx[row * A.L + l];
                if (A.idx32) {                           // the owners of linear.hip launch_table_owner64 add the rows in token order
                    st4(A.g_out + t * FM_D + c, g[0]);
                    if (l16 == 0) A.idx32[t] = (id > 0 && id < A.n_items) ? (int)id : 0;
                } else if (id > 0 && id < A.n_items) {
                    float* d = A.dE + id * FM_D + c;
                    unsafeAtomicAdd(d, g[0].x); unsafeAtomicAdd(d + 1, g[0].y); unsafeAtomicAdd(d + 2, g[0].z); unsafeAtomicAdd(d + 3, g[0].w);
                }
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {                   // deterministic: one dP partial [L][64] per workgroup, summed by k_fmlp_dm_reduce
        const int l = ps * 16 + rsub;                  // (atomics from 64 workgroups into the 3 200 dP words were 2/3 of this launch)
        if (l < A.L) st4(A.dP + ((size_t)blockIdx.x * A.L + l) * FM_D + c, accP[ps]);
    }
    // fold the 16 row groups -> one [2*D] row in LDS, then one atomic per column per block
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        auto fold = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
        const float4 a = make_float4(fold(dgam[0].x), fold(dgam[0].y), fold(dgam[0].z), fold(dgam[0].w));
        const float4 bb = make_float4(fold(dbet[0].x), fold(dbet[0].y), fold(dbet[0].z), fold(dbet[0].w));
        if (lane < 16) { st4(scr + w * 2 * FM_D + c, a); st4(scr + w * 2 * FM_D + FM_D + c, bb); }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * FM_D; i += 256) {
            const float v = (scr[i] + scr[2 * FM_D + i]) + (scr[4 * FM_D + i] + scr[6 * FM_D + i]);
            A.ln_wg[(size_t)blockIdx.x * 2 * FM_D + i] = v;            // summed by k_fmlp_dm_reduce
        }
    }
}

// ------------------------------------------------------------------------------------------------ filter layer
struct FFiltArgs {
    const float* m; float* dm;                 // this layer's [L][D]
    const float* x; const float* lnw; const float* lnb;
    float* uf; float* stf; float* xf;          // forward outputs
    const float* dxf; float* dx; float* ln_wg;               // backward
    int B, L; float eps; const int* state; uint64_t seed; float p; int training; uint32_t site;
};

// One workgroup per sequence.  Thread (dq = tid & 15, rg = tid >> 4) owns the feature quad d = 4dq..4dq+3 of rows l = 4 rg + i:
// every LDS access of the L x L circular-convolution loops is a conflict-free ds_read_b128 feeding 4 FMAs (one b32 load per
// FMA was LDS-issue bound: 36 / 111 us), and a row's LayerNorm statistics reduce over the 16 lanes of a row group.
__device__ __forceinline__ void fma4(float4& a, const float4& m, const float4& x) {
    a.x = fmaf(m.x, x.x, a.x); a.y = fmaf(m.y, x.y, a.y); a.z = fmaf(m.z, x.z, a.z); a.w = fmaf(m.w, x.w, a.w);
}

// Four CONSECUTIVE rows per thread and a sliding window: acc[i] += C(s) * V(s + off_i), off_i = i (or 3 - i), s = 0 .. S-1.  Row i at
// step s and row i -/+ 1 at step s + 1 read the same V, so a 4-slot circular register window needs ONE new value per step: 2 LDS
// reads (coefficient + new value) per 16 FMAs instead of 5 with rows 16 apart — these kernels run one wave per SIMD and were bound by
// LDS bandwidth / latency (k_fmlp_filter_fwd 14.4 us, _bwd 28.8 us at B = 256).  V and C must be safe (in bounds, any value) for s up to S + 10.
template <bool REV, class VF, class CF>
__device__ __forceinline__ void conv4(float4 (&acc)[4], const int S, VF V, CF C) {
    float4 w[4], nv[4], nc[4];
    w[0] = V(0); w[1] = V(1); w[2] = V(2);
#pragma unroll
    for (int j = 0; j < 4; ++j) { nv[j] = V(3 + j); nc[j] = C(j); }
    const int S4 = S & ~3;
    for (int s0 = 0; s0 < S4; s0 += 4) {
        float4 cv[4], cc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { cv[j] = nv[j]; cc[j] = nc[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { nv[j] = V(s0 + 7 + j); nc[j] = C(s0 + 4 + j); }   // the next 4 steps' reads fly over these 64 FMAs:
#pragma unroll                                                                          // one wave per SIMD hides no LDS latency itself
        for (int j = 0; j < 4; ++j) {
            w[(j + 3) & 3] = cv[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) fma4(acc[i], cc[j], w[(j + (REV ? 3 - i : i)) & 3]);
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)                    // the S & 3 tail steps (their operands are already in nv / nc)
        if (S4 + j < S) {
            w[(j + 3) & 3] = nv[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) fma4(acc[i], nc[j], w[(j + (REV ? 3 - i : i)) & 3]);
        }
}

__global__ __launch_bounds__(256) void k_fmlp_filter_fwd(const FFiltArgs A) {
    const int L = A.L, b = blockIdx.x, dq = threadIdx.x & 15, rg = threadIdx.x >> 4, c = 4 * dq;
    float* ML = smem;                          // [L][64]
    float* X2 = ML + L * FM_D;                 // [2L][64]   x twice: x[(l-r) mod L] = X2[l - r + L]
    const float* xb = A.x + (size_t)b * L * FM_D;
    float4 mreg[4], xreg[4];                   // L <= 64: at most 4 quads per thread; every global read is issued before the first wait
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = min((int)threadIdx.x + 256 * k, L * FM_D / 4 - 1);
        mreg[k] = ld4(A.m + 4 * i); xreg[k] = ld4(xb + 4 * i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < L * FM_D / 4) { st4(ML + 4 * i, mreg[k]); st4(X2 + 4 * i, xreg[k]); st4(X2 + L * FM_D + 4 * i, xreg[k]); }
    }
    __syncthreads();
    float4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int l0 = 4 * rg;                     // this thread's rows l0 .. l0 + 3: y[l0 + i] = sum_r m[r] x[(l0 + i - r) mod L]
    conv4<true>(acc, L, [&](int k) { return ld4(X2 + min(max(l0 + 3 + L - k, 0), 2 * L - 1) * FM_D + c); },
                [&](int r) { return ld4(ML + min(r, 3 * L - 1) * FM_D + c); });
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const float4 gam = ld4(A.lnw + c), bet = ld4(A.lnb + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int l = l0 + i;
        if (l < L) {                               // uniform over the 16 lanes of a row group
            const size_t t = (size_t)b * L + l;
            float4 y = acc[i];
            if (dodrop) { const float4 m = drop4(rk, A.site, (uint64_t)t * FM_D + c); y.x *= m.x; y.y *= m.y; y.z *= m.z; y.w *= m.w; }
            const float4 x0 = ld4(X2 + l * FM_D + c);
            const float4 u = make_float4(y.x + x0.x, y.y + x0.y, y.z + x0.z, y.w + x0.w);
            const float mean = group16_sum((u.x + u.y) + (u.z + u.w)) * (1.0f / FM_D);
            const float4 dv = make_float4(u.x - mean, u.y - mean, u.z - mean, u.w - mean);
            const float rstd = 1.0f / sqrtf(group16_sum((dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w)) * (1.0f / FM_D) + A.eps);
            st4(A.uf + t * FM_D + c, u);
            st4(A.xf + t * FM_D + c, make_float4(dv.x * rstd * gam.x + bet.x, dv.y * rstd * gam.y + bet.y, dv.z * rstd * gam.z + bet.z,
                                                 dv.w * rstd * gam.w + bet.w));
            if (dq == 0) { A.stf[2 * t] = mean; A.stf[2 * t + 1] = rstd; }
        }
    }
}

// backward; block-strided over sequences so that dm and the LN affine grads cost one atomic per element per block
__global__ __launch_bounds__(256) void k_fmlp_filter_bwd(const FFiltArgs A) {
    const int L = A.L, dq = threadIdx.x & 15, rg = threadIdx.x >> 4, c = 4 * dq;
    float* ML = smem;                          // [L][64]
    float* X2 = ML + L * FM_D;                 // [2L][64]
    float* DY2 = X2 + 2 * L * FM_D;            // [2L][64]  dy twice: dy[(l'+r) mod L] = DY2[l' + r]
    float* red = DY2 + 2 * L * FM_D;           // [16][2][64]
    float4 mreg[4];                            // stored to LDS once the first sequence's reads are in flight too
#pragma unroll
    for (int k = 0; k < 4; ++k) mreg[k] = ld4(A.m + 4 * min((int)threadIdx.x + 256 * k, L * FM_D / 4 - 1));
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const float4 gam = ld4(A.lnw + c);
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam, dmacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dmacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
        __syncthreads();                           // previous iteration finished with X2 / DY2
        const float* xb = A.x + (size_t)b * L * FM_D;
        const int l0 = 4 * rg;                     // this thread's rows (dx) / filter taps (dm): l0 .. l0 + 3
        float4 xreg[4], dzv[4], uuv[4], du[4];
        float2 stv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xreg[k] = ld4(xb + 4 * min((int)threadIdx.x + 256 * k, L * FM_D / 4 - 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {              // unconditional (clamped) so that all of them are in flight together
            const size_t t = (size_t)b * L + min(l0 + i, L - 1);
            dzv[i] = ld4(A.dxf + t * FM_D + c); uuv[i] = ld4(A.uf + t * FM_D + c);
            stv[i] = *reinterpret_cast<const float2*>(A.stf + 2 * t);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < L * FM_D / 4) {
                if (b == (int)blockIdx.x) st4(ML + 4 * i, mreg[k]);
                st4(X2 + 4 * i, xreg[k]); st4(X2 + L * FM_D + 4 * i, xreg[k]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = l0 + i;
            du[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l < L) {
                const size_t t = (size_t)b * L + l;
                const float4 dz = dzv[i], uu = uuv[i];
                const float mean = stv[i].x, rstd = stv[i].y;
                const float4 xh = make_float4((uu.x - mean) * rstd, (uu.y - mean) * rstd, (uu.z - mean) * rstd, (uu.w - mean) * rstd);
                const float4 gg = make_float4(dz.x * gam.x, dz.y * gam.y, dz.z * gam.z, dz.w * gam.w);
                const float s1 = group16_sum((gg.x + gg.y) + (gg.z + gg.w)) * (1.0f / FM_D);
                const float s2 = group16_sum((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * (1.0f / FM_D);
                dgam.x += dz.x * xh.x; dgam.y += dz.y * xh.y; dgam.z += dz.z * xh.z; dgam.w += dz.w * xh.w;
                dbet.x += dz.x; dbet.y += dz.y; dbet.z += dz.z; dbet.w += dz.w;
                du[i] = make_float4(rstd * (gg.x - s1 - xh.x * s2), rstd * (gg.y - s1 - xh.y * s2), rstd * (gg.z - s1 - xh.z * s2),
                                    rstd * (gg.w - s1 - xh.w * s2));
                float4 dy = du[i];
                if (dodrop) { const float4 m = drop4(rk, A.site, (uint64_t)t * FM_D + c); dy.x *= m.x; dy.y *= m.y; dy.z *= m.z; dy.w *= m.w; }
                st4(DY2 + l * FM_D + c, dy);
                st4(DY2 + (l + L) * FM_D + c, dy);
            }
        }
        __syncthreads();
        // dx[l'] = du[l'] + sum_r m[r] dy[(l'+r) mod L]
        float4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = du[i];
        conv4<false>(acc, L, [&](int k) { return ld4(DY2 + min(l0 + k, 2 * L - 1) * FM_D + c); },
                     [&](int r) { return ld4(ML + min(r, 3 * L - 1) * FM_D + c); });
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = l0 + i;
            if (l < L) st4(A.dx + ((size_t)b * L + l) * FM_D + c, acc[i]);
        }
        // dm[r] += sum_l dy[l] x[(l-r) mod L]   (this thread owns the taps r = l0 + i)
        conv4<true>(dmacc, L, [&](int k) { return ld4(X2 + min(max(k + L - l0 - 3, 0), 2 * L - 1) * FM_D + c); },
                    [&](int l) { return ld4(DY2 + min(l, 2 * L - 1) * FM_D + c); });
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                  // deterministic: one partial [L][64] per workgroup, summed by k_fmlp_dm_reduce
        const int r = 4 * rg + i;
        if (r < L) st4(A.dm + ((size_t)blockIdx.x * L + r) * FM_D + c, dmacc[i]);
    }
    __syncthreads();
    st4(red + (rg * 2) * FM_D + c, dgam);
    st4(red + (rg * 2 + 1) * FM_D + c, dbet);
    __syncthreads();
    if (threadIdx.x < 2 * FM_D) {
        const int which = threadIdx.x >> 6, dd = threadIdx.x & 63;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[(k * 2 + which) * FM_D + dd];
        A.ln_wg[(size_t)blockIdx.x * 2 * FM_D + threadIdx.x] = v;          // [d weight | d bias], summed by k_fmlp_dm_reduce
    }
}

// ------------------------------------------------------------------------------------------------ scorer (1-D targets)
// one wave per row b: q = z[b, L-1, :];  loss / gradient as model/basemodel.py:204-214 + loss_func.py:9-38 with pos [B], neg [B,1]
#define DR4SR_SITE_NEG_F 0x4e454722u
__global__ __launch_bounds__(256) void k_fmlp_score(const float* __restrict__ Z, const float* __restrict__ E, float* __restrict__ dE,
                                                    float* __restrict__ dZ, const int64_t* __restrict__ target,
                                                    const int64_t* __restrict__ rows, int64_t* __restrict__ neg_item, int sample_neg,
                                                    float* __restrict__ part, const int* __restrict__ state, uint64_t seed,
                                                    int n_items, int B, int L, int4* __restrict__ rec) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int64_t row = rows ? rows[b] : b;
    const int64_t tgt = target[row];
    int64_t ng;
    if (sample_neg) {
        const RngKey rk = make_rng(seed, (uint32_t)state[DR4SR_STATE_RNGSTEP], 0.f);
        const uint4 r = rng_call(rk, DR4SR_SITE_NEG_F, (uint64_t)b >> 2);
        const uint32_t cc = b & 3, wv = cc == 0 ? r.x : cc == 1 ? r.y : cc == 2 ? r.z : r.w;
        ng = 1 + (int64_t)__umulhi(wv, (uint32_t)(n_items - 1));
        if (lane == 0) neg_item[b] = ng;
    } else {
        ng = neg_item[b];
    }
    ng = ng < 0 ? 0 : (ng >= n_items ? n_items - 1 : ng);
    float* dzb = dZ + (size_t)b * L * FM_D;
    for (int i = lane; i < (L - 1) * FM_D / 4; i += 64) st4(dzb + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
    if (rec && lane < L - 1) rec[(size_t)b * L + lane] = make_int4(0, 0, 0, 0);      // deterministic mode: records instead of atomics (L <= 50)
    const size_t tl = ((size_t)b * L + L - 1) * FM_D;
    float cnt = 0.f, ls = 0.f;
    if (tgt > 0 && tgt < n_items) {
        const float q = Z[tl + lane], ep = E[tgt * FM_D + lane], en = E[ng * FM_D + lane];
        const float sp = wave_sum(q * ep), sn = wave_sum(q * en);
        ls = softplus_f(-sp) + softplus_f(sn);
        cnt = 1.f;
        const float dpos = -sigmoid_f(-sp), dneg = sigmoid_f(sn);
        dZ[tl + lane] = dpos * ep + dneg * en;
        if (rec) {
            if (lane == 0) rec[(size_t)b * L + L - 1] = make_int4((int)tgt, (int)ng, __float_as_int(dpos), __float_as_int(dneg));
        } else {
            unsafeAtomicAdd(dE + tgt * FM_D + lane, dpos * q);
            unsafeAtomicAdd(dE + ng * FM_D + lane, dneg * q);
        }
    } else {
        dZ[tl + lane] = 0.f;
        if (rec && lane == 0) rec[(size_t)b * L + L - 1] = make_int4(0, 0, 0, 0);
    }
    if (lane == 0) { part[2 * b] = cnt; part[2 * b + 1] = ls; }
}

// out[b,:] = X[b, L-1, :]  /  dX[b, l, :] = (l == L-1) ? d_out[b,:] : 0
__global__ void k_fmlp_last(const float* __restrict__ X, float* __restrict__ out, int B, int L) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * FM_D; i += gridDim.x * blockDim.x)
        out[i] = X[((size_t)(i / FM_D) * L + L - 1) * FM_D + (i % FM_D)];
}
__global__ void k_fmlp_last_bwd(const float* __restrict__ dout, float* __restrict__ dX, int B, int L) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)B * L * FM_D; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / FM_D;
        const int l = (int)(t % L);
        dX[i] = l == L - 1 ? dout[(t / L) * FM_D + (i % FM_D)] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ orchestration
static bool fmlp_bf3(const FmlpWs& ws) { return ws.wsplit != nullptr && ffn_tile_rows(ws.Tn) == 32 && !DR4SR_ENV("DR4SR_TILE_F32"); }
#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

// `only` >= 0 (measurement hook dr4sr_fmlp_launch_kernel): just that launch (DR4SR_FK_*) of layer `only_layer`, on the state the last step left
static int fmlp_forward(const dr4sr_fmlp_plan* p, const FmlpWs& ws, int training, int zero_grads, hipStream_t s, int only = -1, int only_layer = 0) {
    const int L = p->L, nl = p->n_layer;
    const bool all = only < 0;
    const int64_t n4 = zero_grads ? (ws.n_params + DR4SR_GRAD_TAIL) / 4 : 0;
    int zb = (int)((n4 + 1023) / 1024); if (zb > 255) zb = 255;
    const int64_t lstride = nl > 1 ? ws.off[4 + 9] - ws.off[4] : 0;
    PermSel sel{nullptr, 0, 0, 0, nullptr};
    if (p->perm && training && zero_grads) {                // batch selection only in the call that starts a training step
        if (!p->rows || !p->perm_counter || p->n_perm <= 0) return DR4SR_E_ARG;
        sel = PermSel{p->perm, p->n_perm, p->perm_stride, p->perm_offset, p->perm_counter};
    }
    // 32-row Intermediate kernels (below the at-scale tile size): GEMMs as a bf16x3 split from fragment-major bf16 hi | lo weight images
    // (round 5: the d = 128 SASRec tile kernels' form, csrc/common.h tile_mma_xwT_bf3; DR4SR_TILE_F32: fp32 MFMA from the parameters); the
    // images are written by extra blocks of the prep launch (as a launch of its own, k_wsplit: +5 us per step)
    const bool bf3 = fmlp_bf3(ws);
    constexpr int FE = 4 * FM_D * FM_D + 2 * FM_D * FM_F;
    const FPrepArgs PA{p->state, ws.Tn, training ? 1 : 0, zero_grads ? p->grads : nullptr, n4, zb,
                       p->params, foff(ws, 0, FP_CW), lstride, ws.m, ws.dm, L, sel, const_cast<int64_t*>(p->rows), p->B,
                       bf3 ? ws.wsplit : nullptr, foff(ws, 0, FP_W1), foff(ws, 0, FP_W2), nl * FM_COEF_BLK};
    if (all) hipLaunchKernelGGL(k_fmlp_prep, dim3(1 + zb + nl * FM_COEF_BLK + (bf3 ? nl * FM_SPLIT_BLK : 0)), dim3(1024), 0, s, PA);
    FEmbArgs E{};
    E.E = p->params + ws.off[0]; E.P = p->params + ws.off[1]; E.lnw = p->params + ws.off[2]; E.lnb = p->params + ws.off[3];
    E.idx = p->in_item_id; E.rows = p->rows; E.e0 = ws.e0; E.st0 = ws.st0; E.x0 = ws.X[0];
    E.B = p->B; E.L = L; E.n_items = p->n_items; E.eps = p->ln_eps; E.state = p->state; E.seed = p->seed; E.p = p->p_drop; E.training = training;
    int eb = (ws.Tn + 15) / 16; if (eb > 2048) eb = 2048;
    if (all) hipLaunchKernelGGL(k_fmlp_embed_fwd, dim3(eb), dim3(256), 0, s, E);
    for (int l = 0; l < nl; ++l) {
        const FmlpLayerWs& w = ws.layer[l];
        FFiltArgs Fa{};
        Fa.m = ws.m + (size_t)l * L * FM_D; Fa.x = ws.X[l];
        Fa.lnw = p->params + foff(ws, l, FP_FLN_W); Fa.lnb = p->params + foff(ws, l, FP_FLN_B);
        Fa.uf = w.uf; Fa.stf = w.stf; Fa.xf = w.xf; Fa.B = p->B; Fa.L = L; Fa.eps = p->ln_eps; Fa.state = p->state; Fa.seed = p->seed;
        Fa.p = p->p_drop; Fa.training = training; Fa.site = FS_FILT(l);
        const size_t lds = sizeof(float) * 3 * L * FM_D;
        if (all || (only == DR4SR_FK_FILTER_FWD && l == only_layer)) hipLaunchKernelGGL(k_fmlp_filter_fwd, dim3(p->B), dim3(256), lds, s, Fa);
        PostArgs A{};
        A.x = w.xf; A.w1 = p->params + foff(ws, l, FP_W1); A.b1 = p->params + foff(ws, l, FP_B1);
        A.w2 = p->params + foff(ws, l, FP_W2); A.b2 = p->params + foff(ws, l, FP_B2);
        A.ln2_w = p->params + foff(ws, l, FP_ILN_W); A.ln2_b = p->params + foff(ws, l, FP_ILN_B);
        A.a = w.a; A.h = w.h; A.u2 = w.u2; A.st2 = w.st2; A.z = ws.X[l + 1];
        A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.eps = p->ln_eps; A.layer = l; A.training = training;
        A.sP = 0xffffffffu; A.sA = 0xffffffffu; A.sF = FS_FFN(l); A.stamps = nullptr; A.rd = nullptr; A.n_head = 1;
        A.sp = bf3 ? ws.wsplit + (size_t)l * 4 * FE : nullptr;
        if (all || (only == DR4SR_FK_FFN_FWD && l == only_layer)) RC(launch_ffn_fwd(A, ws.Tn, s));
    }
    return DR4SR_LAUNCH_CHECK();
}

static int fmlp_backward(const dr4sr_fmlp_plan* p, const FmlpWs& ws, int training, int with_score, hipStream_t s, int only = -1, int only_layer = 0) {
    const int L = p->L, nl = p->n_layer;
    const bool all = only < 0;
    const int bm = ffn_tile_rows(ws.Tn), ntiles = (ws.Tn + bm - 1) / bm;
    for (int l = nl - 1; l >= 0; --l) {
        const FmlpLayerWs& w = ws.layer[l];
        PostArgs A{};
        A.dz = ws.dX[l + 1]; A.u2 = w.u2; A.st2 = w.st2; A.ln2_w = p->params + foff(ws, l, FP_ILN_W);
        A.a = w.a; A.w1 = p->params + foff(ws, l, FP_W1); A.w2 = p->params + foff(ws, l, FP_W2);
        A.df = w.df; A.da = w.da; A.du1 = w.dxf;
        A.ln_part = ws.ln_part + (size_t)l * ntiles * 4 * FM_D;
        A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.eps = p->ln_eps; A.layer = l; A.training = training;
        A.sP = 0xffffffffu; A.sA = 0xffffffffu; A.sF = FS_FFN(l); A.stamps = nullptr; A.rd = nullptr; A.n_head = 1;
        A.sp = fmlp_bf3(ws) ? ws.wsplit + (size_t)l * 4 * (4 * FM_D * FM_D + 2 * FM_D * FM_F) : nullptr;      // (written by this step's forward)
        if (all || (only == DR4SR_FK_FFN_BWD && l == only_layer)) RC(launch_ffn_bwd(A, ws.Tn, s));
        FFiltArgs Fa{};
        Fa.m = ws.m + (size_t)l * L * FM_D; Fa.dm = ws.dm_part + (size_t)l * FM_DMBLK * L * FM_D; Fa.x = ws.X[l];
        Fa.lnw = p->params + foff(ws, l, FP_FLN_W); Fa.uf = w.uf; Fa.stf = w.stf;
        Fa.dxf = w.dxf; Fa.dx = ws.dX[l]; Fa.ln_wg = ws.ln_wg + (size_t)l * FM_DMBLK * 2 * FM_D;
        Fa.B = p->B; Fa.L = L; Fa.eps = p->ln_eps; Fa.state = p->state; Fa.seed = p->seed; Fa.p = p->p_drop; Fa.training = training;
        Fa.site = FS_FILT(l);
        const size_t lds = sizeof(float) * (5 * L * FM_D + 32 * FM_D);
        big_lds(k_fmlp_filter_bwd, lds);
        const int gb = p->B < FM_DMBLK ? p->B : FM_DMBLK;
        if (all || (only == DR4SR_FK_FILTER_BWD && l == only_layer)) hipLaunchKernelGGL(k_fmlp_filter_bwd, dim3(gb), dim3(256), lds, s, Fa);
    }
    FEmbArgs E{};
    E.lnw = p->params + ws.off[2]; E.idx = p->in_item_id; E.rows = p->rows; E.e0 = ws.e0; E.st0 = ws.st0;
    E.dx0 = ws.dX[0]; E.dE = p->grads + ws.off[0]; E.dP = ws.dm_part + (size_t)nl * FM_DMBLK * L * FM_D; E.ln_wg = ws.ln_wg + (size_t)nl * FM_DMBLK * 2 * FM_D;
    E.B = p->B; E.L = L; E.n_items = p->n_items; E.eps = p->ln_eps; E.state = p->state; E.seed = p->seed; E.p = p->p_drop; E.training = training;
    if (ws.det) { E.g_out = ws.det_g; E.idx32 = ws.idx32; }
    if (all) hipLaunchKernelGGL(k_fmlp_embed_bwd, dim3(p->B < FM_DMBLK ? p->B : FM_DMBLK), dim3(256), 0, s, E);    // one dP partial per workgroup
    if (all && ws.det)                                      // dE: scorer records (fused step only) + embedding rows, owner-computed in token order
        RC(launch_table_owner64(p->state, with_score ? ws.de_rec : nullptr, ws.idx32, ws.X[nl], ws.det_g, p->grads + ws.off[0], p->n_items, s));
    const int64_t lstride = nl > 1 ? ws.off[4 + 9] - ws.off[4] : 0;
    FDmRedArgs R{};
    R.part = ws.dm_part; R.dm = ws.dm; R.dP = p->grads + ws.off[1]; R.ln_wg = ws.ln_wg; R.grads = p->grads;
    R.o_fln_w = foff(ws, 0, FP_FLN_W); R.o_fln_b = foff(ws, 0, FP_FLN_B); R.layer_stride = lstride; R.o_eln_w = ws.off[2]; R.o_eln_b = ws.off[3];
    R.nblk = p->B < FM_DMBLK ? p->B : FM_DMBLK; R.n = L * FM_D; R.n_layer = nl; R.nx = (L * FM_D + 63) / 64;
    if (all) hipLaunchKernelGGL(k_fmlp_dm_reduce, dim3(R.nx + 2, nl + 1), dim3(256), 0, s, R);
    WgradArgs W{};
    for (int l = 0; l < nl; ++l) {
        const FmlpLayerWs& w = ws.layer[l];
        WgradJob J1, J2;
        J1.G = w.da; J1.ldg = FM_F; J1.gcol = 0; J1.X = w.xf; J1.ldx = FM_D; J1.ldw = 0;
        J1.dW = p->grads + foff(ws, l, FP_W1); J1.db = p->grads + foff(ws, l, FP_B1);
        J2.G = w.df; J2.ldg = FM_D; J2.gcol = 0; J2.X = w.h; J2.ldx = FM_F; J2.ldw = 0;
        J2.dW = p->grads + foff(ws, l, FP_W2); J2.db = p->grads + foff(ws, l, FP_B2);
        if (!ws.det) { W.job[l * 6 + 4] = J1; W.job[l * 6 + 5] = J2; continue; }
        for (int j = 0; j < 8; ++j) {                       // deterministic mode: the eight 64 x 64 blocks as explicit jobs (k_wgrad_det_reduce reads them)
            WgradJob J = j < 4 ? J1 : J2;
            if (j < 4) { J.gcol += 64 * j; J.dW += (size_t)64 * j * 64; J.ldw = 64; J.db += 64 * j; }
            else { const int kb = j - 4; J.X += 64 * kb; J.dW += 64 * kb; J.ldw = 256; if (kb) J.db = nullptr; }
            W.job[l * 8 + j] = J;
        }
    }
    W.jobs_per_layer = ws.det ? 8 : 6;
    if (ws.det) { W.det = ws.det_part; W.det_stride = FM_DET_STRIDE; W.det_ln = ws.det_ln; }
    W.state = p->state; W.seed = p->seed; W.p = p->p_drop; W.training = training;
    W.ln_part = ws.ln_part; W.ln_layer_stride = (int64_t)ntiles * 4 * FM_D; W.grads = p->grads; W.ln_tile_rows = bm;
    W.o_ln1_w = foff(ws, 0, FP_ILN_W); W.layer_stride = lstride;
    W.score_part = with_score ? ws.score_part : nullptr; W.tail = p->grads + ws.n_params; W.B = p->B; W.D = FM_D;
    W.fc_dm = ws.dm; W.fc_o_cw = foff(ws, 0, FP_CW); W.fc_L = L;      // d(complex_weight) = fold(dm) rides in the reduce blocks
    if (all || only == DR4SR_FK_WGRAD) RC(launch_fmlp_wgrad(W, ws.Tn, nl, s));
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_fmlp_fwd_bwd(const dr4sr_fmlp_plan* plan, void* stream) {
    FmlpWs ws;
    RC(fmlp_ws(plan, &ws));
    if (!plan->grads || !plan->item_id || !plan->neg_item || plan->n_params != ws.n_params) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(fmlp_forward(plan, ws, 1, 1, s));
    hipLaunchKernelGGL(k_fmlp_score, dim3((plan->B + 3) / 4), dim3(256), 0, s, ws.X[plan->n_layer], plan->params + ws.off[0],
                       plan->grads + ws.off[0], ws.dX[plan->n_layer], plan->item_id, plan->rows, plan->neg_item, plan->sample_neg,
                       ws.score_part, plan->state, plan->seed, plan->n_items, plan->B, plan->L, ws.det ? ws.de_rec : nullptr);
    return fmlp_backward(plan, ws, 1, 1, s);
}

// measurement hook (include/dr4sr_hip_hooks.h): ONE launch of the step on the workspace the last dr4sr_fmlp_fwd_bwd left
extern "C" int dr4sr_fmlp_launch_kernel(const dr4sr_fmlp_plan* plan, int32_t kernel, int32_t layer, void* stream) {
    FmlpWs ws;
    RC(fmlp_ws(plan, &ws));
    if (!plan->grads || layer < 0 || layer >= plan->n_layer) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (kernel) {
        case DR4SR_FK_FILTER_FWD: case DR4SR_FK_FFN_FWD: return fmlp_forward(plan, ws, 1, 0, s, kernel, layer);
        case DR4SR_FK_FFN_BWD: case DR4SR_FK_FILTER_BWD: case DR4SR_FK_WGRAD: return fmlp_backward(plan, ws, 1, 1, s, kernel, layer);
        default: return DR4SR_E_ARG;
    }
}

extern "C" int dr4sr_adam_flat(float* params, const float* grads, float* adam_m, float* adam_v, int64_t n, int32_t* state,
                               float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    return launch_adam_flat(params, const_cast<float*>(grads), adam_m, adam_v, n, state, lr, beta1, beta2, eps, weight_decay, (hipStream_t)stream);   // (written only when a next-step prep is fused)
}

extern "C" int dr4sr_optimizer_flat(int32_t optimizer, float* params, const float* grads, float* adam_m, float* adam_v, int64_t n, int32_t* state,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    return launch_adam_flat(params, const_cast<float*>(grads), adam_m, adam_v, n, state, lr, beta1, beta2, eps, weight_decay, (hipStream_t)stream,
                            nullptr, nullptr, nullptr, optimizer);
}

extern "C" int dr4sr_fmlp_adam_step(const dr4sr_fmlp_plan* plan, void* stream) {
    if (!plan || plan->abi_version != DR4SR_ABI_VERSION || !plan->params || !plan->grads || !plan->state) return DR4SR_E_ARG;
    return launch_adam_flat(plan->params, plan->grads, plan->adam_m, plan->adam_v, plan->n_params, plan->state, plan->lr,
                            plan->beta1, plan->beta2, plan->adam_eps, plan->weight_decay, (hipStream_t)stream, plan->loss_log,
                            plan->perm ? plan->perm_counter : nullptr, nullptr, plan->optimizer);
}

extern "C" int dr4sr_fmlp_train_step(const dr4sr_fmlp_plan* plan, void* stream) {
    RC(dr4sr_fmlp_fwd_bwd(plan, stream));
    return dr4sr_fmlp_adam_step(plan, stream);
}

extern "C" int dr4sr_fmlp_encode(const dr4sr_fmlp_plan* plan, int32_t training, float* out, void* stream) {
    FmlpWs ws;
    RC(fmlp_ws(plan, &ws));
    if (!out) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    RC(fmlp_forward(plan, ws, training, 0, s));
    hipLaunchKernelGGL(k_fmlp_last, dim3((plan->B * FM_D + 255) / 256), dim3(256), 0, s, ws.X[plan->n_layer], out, plan->B, plan->L);
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_fmlp_encode_bwd(const dr4sr_fmlp_plan* plan, int32_t training, const float* d_out, void* stream) {
    FmlpWs ws;
    RC(fmlp_ws(plan, &ws));
    if (!d_out || !plan->grads) return DR4SR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    int gb = (int)(((int64_t)ws.Tn * FM_D + 255) / 256); if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(k_fmlp_last_bwd, dim3(gb), dim3(256), 0, s, d_out, ws.dX[plan->n_layer], plan->B, plan->L);
    return fmlp_backward(plan, ws, training, 0, s);
}
