// attn.hip — causal multi-head self-attention over ragged (packed) sequences, forward and backward.
// One workgroup per sequence slot, one wave per head, one lane per query row (L <= 64).
// K/V (and Q/dCtx in the backward) of the sequence live in LDS and are read as wave-wide broadcasts;
// the softmax row lives in LDS at a bank-conflict-free stride (L+1); probabilities are recomputed in
// the backward (nothing but q|k|v is saved).  Dropout on the probabilities is regenerated from Philox
// at element ((b*H + h)*64 + i)*64 + j.
//
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer with
// attn_mask = triu(ones(L,L),1) (model/sasrec.py:58) and key_padding_mask = (idx == 0)
// (model/sasrec.py:48); scale 1/sqrt(head_dim); dropout on softmax output; need_weights=False.
//
// v1 keeps the QK^T / PV products on the VALU (8 % of the step's flops); the MFMA version is the
// next optimisation target.
#include "common.h"
#include "kernels.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

struct AttnArgs {
    const float* qkv; float* ctx;             // forward
    const float* dctx; float* dqkv;           // backward
    const int64_t* idx; const int64_t* rows; const int* cu;
    const int* state; uint64_t seed; float p; int layer; int training; int L; int D; int H;
};

template <int DH>
__device__ __forceinline__ float dot_bcast(const float (&q)[DH], const float* __restrict__ k) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const float4 kv = ld4(k + c);
        s0 += q[c] * kv.x; s1 += q[c + 1] * kv.y; s2 += q[c + 2] * kv.z; s3 += q[c + 3] * kv.w;
    }
    return (s0 + s1) + (s2 + s3);
}

template <int DH>
__global__ __launch_bounds__(256) void k_attn_fwd(const AttnArgs A) {
    const int b = blockIdx.x;
    const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
    if (n <= 0) return;
    const int D = A.D, L = A.L, H = A.H, LS = L + 1;
    float* Ks = smem;                         // [n][D]
    float* Vs = Ks + L * D;                   // [n][D]
    float* S = Vs + L * D;                    // [H][L][LS]
    int* kpad = reinterpret_cast<int*>(S + H * L * LS);   // [L]
    const int64_t row = A.rows ? A.rows[b] : b;
    for (int i = threadIdx.x; i < n * (D / 4); i += blockDim.x) {
        const int r = i / (D / 4), c = (i % (D / 4)) * 4;
        st4(Ks + r * D + c, ld4(A.qkv + (size_t)(t0 + r) * 3 * D + D + c));
        st4(Vs + r * D + c, ld4(A.qkv + (size_t)(t0 + r) * 3 * D + 2 * D + c));
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) kpad[j] = A.idx[row * L + j] == 0;
    __syncthreads();
    const int h = threadIdx.x >> 6, i = threadIdx.x & 63;
    if (i >= n) return;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
    float q[DH];
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const float4 v = ld4(A.qkv + (size_t)(t0 + i) * 3 * D + h * DH + c);
        q[c] = v.x; q[c + 1] = v.y; q[c + 2] = v.z; q[c + 3] = v.w;
    }
    float* Srow = S + (h * L + i) * LS;
    const float scale = 1.0f / sqrtf((float)DH);
    float m = -INFINITY;
    for (int j = 0; j <= i; ++j) {
        float s = -INFINITY;
        if (!kpad[j]) s = dot_bcast<DH>(q, Ks + j * D + h * DH) * scale;
        Srow[j] = s;
        m = fmaxf(m, s);
    }
    float sum = 0.f;
    for (int j = 0; j <= i; ++j) {
        const float e = expf(Srow[j] - m);
        Srow[j] = e;
        sum += e;
    }
    const float inv = 1.0f / sum;
    float acc[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = 0.f;
    const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
    for (int j4 = 0; j4 <= i; j4 += 4) {
        float mk[4] = {1.f, 1.f, 1.f, 1.f};
        if (dodrop) { const float4 m4 = drop4(rk, site, ebase + j4); mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j4 + u;
            if (j <= i) {
                const float pj = Srow[j] * inv * mk[u];
                const float* v = Vs + j * D + h * DH;
#pragma unroll
                for (int c = 0; c < DH; c += 4) {
                    const float4 vv = ld4(v + c);
                    acc[c] += pj * vv.x; acc[c + 1] += pj * vv.y; acc[c + 2] += pj * vv.z; acc[c + 3] += pj * vv.w;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < DH; c += 4)
        st4(A.ctx + (size_t)(t0 + i) * D + h * DH + c, make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]));
}

template <int DH>
__global__ __launch_bounds__(256) void k_attn_bwd(const AttnArgs A) {
    const int b = blockIdx.x;
    const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
    if (n <= 0) return;
    const int D = A.D, L = A.L, H = A.H, LS = L + 1;
    float* Ks = smem;
    float* Vs = Ks + L * D;
    float* Qs = Vs + L * D;
    float* Cs = Qs + L * D;                    // dctx rows
    float* S = Cs + L * D;                     // [H][L][LS]  probabilities, then dropped probabilities
    float* DS = S + H * L * LS;                // [H][L][LS]  d(prob), then d(score)*scale
    int* kpad = reinterpret_cast<int*>(DS + H * L * LS);
    const int64_t row = A.rows ? A.rows[b] : b;
    for (int i = threadIdx.x; i < n * (D / 4); i += blockDim.x) {
        const int r = i / (D / 4), c = (i % (D / 4)) * 4;
        const float* src = A.qkv + (size_t)(t0 + r) * 3 * D + c;
        st4(Qs + r * D + c, ld4(src));
        st4(Ks + r * D + c, ld4(src + D));
        st4(Vs + r * D + c, ld4(src + 2 * D));
        st4(Cs + r * D + c, ld4(A.dctx + (size_t)(t0 + r) * D + c));
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) kpad[j] = A.idx[row * L + j] == 0;
    __syncthreads();
    const int h = threadIdx.x >> 6, i = threadIdx.x & 63;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
    const float scale = 1.0f / sqrtf((float)DH);
    if (i < n) {
        // ---- phase 1: row i of head h -> dq, and the P~ / dS rows for phase 2
        float q[DH], dc[DH];
#pragma unroll
        for (int c = 0; c < DH; c += 4) {
            const float4 v = ld4(A.qkv + (size_t)(t0 + i) * 3 * D + h * DH + c);
            q[c] = v.x; q[c + 1] = v.y; q[c + 2] = v.z; q[c + 3] = v.w;
            const float4 g = ld4(A.dctx + (size_t)(t0 + i) * D + h * DH + c);
            dc[c] = g.x; dc[c + 1] = g.y; dc[c + 2] = g.z; dc[c + 3] = g.w;
        }
        float* Srow = S + (h * L + i) * LS;
        float* Drow = DS + (h * L + i) * LS;
        float m = -INFINITY;
        for (int j = 0; j <= i; ++j) {
            float s = -INFINITY;
            if (!kpad[j]) s = dot_bcast<DH>(q, Ks + j * D + h * DH) * scale;
            Srow[j] = s;
            m = fmaxf(m, s);
        }
        float sum = 0.f;
        for (int j = 0; j <= i; ++j) {
            const float e = expf(Srow[j] - m);
            Srow[j] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
        float rowdot = 0.f;
        for (int j4 = 0; j4 <= i; j4 += 4) {
            float mk[4] = {1.f, 1.f, 1.f, 1.f};
            if (dodrop) { const float4 m4 = drop4(rk, site, ebase + j4); mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j4 + u;
                if (j <= i) {
                    const float pj = Srow[j] * inv;
                    const float dp = dot_bcast<DH>(dc, Vs + j * D + h * DH) * mk[u];
                    rowdot += pj * dp;
                    Srow[j] = pj;
                    Drow[j] = dp;
                }
            }
        }
        float dq[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) dq[c] = 0.f;
        for (int j4 = 0; j4 <= i; j4 += 4) {
            float mk[4] = {1.f, 1.f, 1.f, 1.f};
            if (dodrop) { const float4 m4 = drop4(rk, site, ebase + j4); mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j4 + u;
                if (j <= i) {
                    const float pj = Srow[j];
                    const float ds = pj * (Drow[j] - rowdot) * scale;
                    Drow[j] = ds;
                    Srow[j] = pj * mk[u];
                    const float* k = Ks + j * D + h * DH;
#pragma unroll
                    for (int c = 0; c < DH; c += 4) {
                        const float4 kv = ld4(k + c);
                        dq[c] += ds * kv.x; dq[c + 1] += ds * kv.y; dq[c + 2] += ds * kv.z; dq[c + 3] += ds * kv.w;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < DH; c += 4)
            st4(A.dqkv + (size_t)(t0 + i) * 3 * D + h * DH + c, make_float4(dq[c], dq[c + 1], dq[c + 2], dq[c + 3]));
    }
    __syncthreads();
    if (i < n) {
        // ---- phase 2: key/value column j = i of head h
        const int j = i;
        float dk[DH], dv[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int r = j; r < n; ++r) {
            const float ds = DS[(h * L + r) * LS + j];
            const float pt = S[(h * L + r) * LS + j];
            const float* qq = Qs + r * D + h * DH;
            const float* cc = Cs + r * D + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const float4 qv = ld4(qq + c), cv = ld4(cc + c);
                dk[c] += ds * qv.x; dk[c + 1] += ds * qv.y; dk[c + 2] += ds * qv.z; dk[c + 3] += ds * qv.w;
                dv[c] += pt * cv.x; dv[c + 1] += pt * cv.y; dv[c + 2] += pt * cv.z; dv[c + 3] += pt * cv.w;
            }
        }
#pragma unroll
        for (int c = 0; c < DH; c += 4) {
            st4(A.dqkv + (size_t)(t0 + j) * 3 * D + D + h * DH + c, make_float4(dk[c], dk[c + 1], dk[c + 2], dk[c + 3]));
            st4(A.dqkv + (size_t)(t0 + j) * 3 * D + 2 * D + h * DH + c, make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]));
        }
    }
}

static AttnArgs make_attn_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training) {
    AttnArgs A;
    const LayerWs& lw = ws.layer[layer];
    A.qkv = lw.qkv; A.ctx = lw.ctx; A.dctx = ws.dctx; A.dqkv = lw.dqkv;
    A.idx = p->in_item_id; A.rows = p->rows; A.cu = ws.cu;
    A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.layer = layer; A.training = training;
    A.L = p->L; A.D = p->D; A.H = p->H;
    return A;
}

int launch_attn_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    const AttnArgs A = make_attn_args(p, ws, layer, training);
    const int dh = p->D / p->H;
    const size_t lds = sizeof(float) * (2 * p->L * p->D + p->H * p->L * (p->L + 1) + p->L);
    dim3 grid(p->B), blk(64 * p->H);
    if (dh == 32) { big_lds(k_attn_fwd<32>, lds); hipLaunchKernelGGL(k_attn_fwd<32>, grid, blk, lds, s, A); }
    else if (dh == 64) { big_lds(k_attn_fwd<64>, lds); hipLaunchKernelGGL(k_attn_fwd<64>, grid, blk, lds, s, A); }
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}

int launch_attn_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    const AttnArgs A = make_attn_args(p, ws, layer, training);
    const int dh = p->D / p->H;
    const size_t lds = sizeof(float) * (4 * p->L * p->D + 2 * p->H * p->L * (p->L + 1) + p->L);
    dim3 grid(p->B), blk(64 * p->H);
    if (dh == 32) { big_lds(k_attn_bwd<32>, lds); hipLaunchKernelGGL(k_attn_bwd<32>, grid, blk, lds, s, A); }
    else if (dh == 64) { big_lds(k_attn_bwd<64>, lds); hipLaunchKernelGGL(k_attn_bwd<64>, grid, blk, lds, s, A); }
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}
