// attn_tiny_body.h (included by attn_mfma.hip) — causal 2-head self-attention for sequences of 1..8 tokens, on the VALU: the third length class of the split launches.
//
// Why: two thirds of the Amazon-toys sequences have at most 4 tokens and 85 % at most 8, yet the MFMA kernels (attn_mfma.hip) spend a
// full 16x16 tile chain — 16 MFMAs per (sequence, head) forward, 56 backward, plus LDS staging and two barriers — on each of them
// whatever its length (rocprofv3 counters, round 1: 9x more MFMA flops issued than the algorithm has; MfmaUtil 1.9 %).  A sequence
// this short is not matrix-shaped work.  Here 8 lanes serve one (sequence, head): lane i owns query row i (forward; backward phase
// A) and key row i (backward phase B), keeps its row in registers and reads the other rows of ITS sequence with loads whose
// address is the same for the 8 lanes of the group (one request per group).  A wave carries 8 (sequence, head) pairs = 4 sequences,
// a 128-thread workgroup 8 sequences.  The workgroup first stages the K | V rows of its sequences in LDS with coalesced loads (one
// contiguous 2D-float run per token), so that the row loops run at LDS latency with no dependent global round trips inside them
// (a first version read the rows straight from L2 inside divergent branches: 14 / 27 us per launch, latency chains);
// the backward adds one 8x8 exchange tile per (sequence, head): dS and P~ are computed once by the query-row owner and consumed by
// the key-row owner.  No MFMA.
//
// Arithmetic = torch.nn.MultiheadAttention as configured at model/sasrec.py:21-34 (attn_mask triu(1) :58, key_padding_mask idx == 0
// :48, scale 1/sqrt(head_dim), dropout on the probabilities); saved statistics, dropout element indexing
// ((b*H+h)*64 + i)*64 + j and the <dctx, ctx> row term are those of attn_mfma.hip, so the two classes are interchangeable per
// sequence (tests: DR4SR_ATTN_NOTINY runs the tiny list through the 16-row MFMA kernels instead).
#pragma once
#include "common.h"
#include "kernels.h"
#include "attn_args.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

namespace tiny {

constexpr int NMAX = DR4SR_TINY_MAX;            // 8 rows = 8 lanes per (sequence, head)
static_assert(NMAX == 8, "lane layout: 8 lanes per (sequence, head)");
constexpr int SPB = 4;                          // sequences per workgroup: ONE wave.  The backward's LDS (22 KB) then allows 7 workgroups per
constexpr int NT = SPB * 2 * NMAX;              // CU and the whole toys list (B = 8192: ~1 750 workgroups) is resident in one round
constexpr int XS = 2 * NMAX * NMAX + 8;         // floats per (sequence, head) exchange tile (+8: groups land on different banks)

// LDS map: K rows [SPB*8][LD] | V rows [SPB*8][LD] | (backward) exchange tiles [SPB*2][XS] | per-sequence words
template <int DH>
struct TinyLds {
    static constexpr int D = 2 * DH, LD = D + 4, ROWS = SPB * NMAX;
    float *Ks, *Vs, *X; int* meta;               // meta: [SPB] t0, [SPB] n, [SPB] b, [SPB] dataset row
    __device__ __forceinline__ explicit TinyLds(bool bwd) {
        Ks = smem; Vs = Ks + ROWS * LD; X = Vs + ROWS * LD;
        meta = reinterpret_cast<int*>(X + (bwd ? SPB * 2 * XS : 0));
    }
    static size_t bytes(bool bwd) { return sizeof(float) * (2 * ROWS * LD + (bwd ? SPB * 2 * XS : 0)) + sizeof(int) * 4 * SPB; }
};

template <int DH>
__device__ __forceinline__ void load_row(float (&f)[DH], const float* __restrict__ p) {
#pragma unroll
    for (int c = 0; c < DH; c += 4) { const float4 v = ld4(p + c); f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w; }
}
template <int DH>
__device__ __forceinline__ float dot_row(const float (&a)[DH], const float* __restrict__ b) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const float4 v = ld4(b + c);
        s0 = fmaf(a[c], v.x, s0); s1 = fmaf(a[c + 1], v.y, s1); s2 = fmaf(a[c + 2], v.z, s2); s3 = fmaf(a[c + 3], v.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
}
template <int DH>
__device__ __forceinline__ void axpy_row(float (&acc)[DH], float a, const float* __restrict__ x) {
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const float4 v = ld4(x + c);
        acc[c] = fmaf(a, v.x, acc[c]); acc[c + 1] = fmaf(a, v.y, acc[c + 1]); acc[c + 2] = fmaf(a, v.z, acc[c + 2]); acc[c + 3] = fmaf(a, v.w, acc[c + 3]);
    }
}

// Workgroup prologue: the (t0, n, b) words of the block's 8 sequences, then their K | V rows (contiguous in a qkv row: one coalesced
// 2D-float run per token) into LDS, rows >= n zero-filled.  Returns the largest n of the block (the row loops stop there).
template <int DH>
__device__ __forceinline__ int tiny_stage(const AttnArgs2& A, const TinyLds<DH>& S, int cnt, unsigned& padbits, const int bid) {
    constexpr int D = 2 * DH, LD = D + 4, ROWS = SPB * NMAX, F4 = 2 * D / 4;      // float4 per staged row (K | V)
    if (threadIdx.x < SPB) {                    // ONE 16-byte load per sequence (k_prep's descriptor) instead of list -> cu -> rows
        const int k = bid * SPB + threadIdx.x;
        int4 d = make_int4(0, 0, 0, 0);
        if (k < cnt) d = reinterpret_cast<const int4*>(A.desc)[k];
        S.meta[threadIdx.x] = d.x; S.meta[SPB + threadIdx.x] = d.y; S.meta[2 * SPB + threadIdx.x] = d.z; S.meta[3 * SPB + threadIdx.x] = d.w;
    }
    lds_barrier();
    // raw item id of this thread's key position (model/sasrec.py:48 key_padding_mask): requested first, consumed after the row loads
    // below have been issued, so the two round trips overlap
    int64_t myid = 1;
    {
        const int sq = threadIdx.x >> 4, i = threadIdx.x & 7;
        if (i < S.meta[SPB + sq]) myid = A.idx[(int64_t)S.meta[3 * SPB + sq] * A.L + i];
    }
    int nmax = 0;
#pragma unroll
    for (int q = 0; q < SPB; ++q) nmax = max(nmax, S.meta[SPB + q]);
    float4 v[ROWS * F4 / NT];
#pragma unroll
    for (int q = 0; q < ROWS * F4 / NT; ++q) {
        const int f = threadIdx.x + q * NT, r = f / F4, c4 = f % F4, sq = r / NMAX, i = r % NMAX;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < S.meta[SPB + sq]) v[q] = ld4(A.qkv + (size_t)(S.meta[sq] + i) * 3 * D + D + 4 * c4);
    }
    {
        const unsigned long long bal = __ballot(myid == 0);
        padbits = (unsigned)(bal >> (threadIdx.x & 56)) & 0xffu;
    }
#pragma unroll
    for (int q = 0; q < ROWS * F4 / NT; ++q) {
        const int f = threadIdx.x + q * NT, r = f / F4, c4 = f % F4;
        float* dst = (4 * c4 < D ? S.Ks + r * LD + 4 * c4 : S.Vs + r * LD + 4 * c4 - D);
        st4(dst, v[q]);
    }
    return nmax;
}

// ------------------------------------------------------------------------------------------------ forward
// (bodies take the workgroup index: the same code runs as its own launch and as the tail blocks of k_attn_small_*, attn_mfma.hip)
template <int DH>
__device__ __forceinline__ void fwd_body(const AttnArgs2& A, const int bid) {
    constexpr int D = 2 * DH, H = 2, LD = D + 4;
    const int cnt = *A.list_count;
    if (bid * SPB >= cnt) return;
    const TinyLds<DH> S(false);
    const int i = threadIdx.x & 7, h = (threadIdx.x >> 3) & 1, sq = threadIdx.x >> 4;
    unsigned padmask;
    const int nmax = tiny_stage<DH>(A, S, cnt, padmask, bid);
    const int t0 = S.meta[sq], n = S.meta[SPB + sq], b = S.meta[2 * SPB + sq];
    const bool act = i < n;
    float q[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] = 0.f;
    if (act) load_row<DH>(q, A.qkv + (size_t)(t0 + i) * 3 * D + h * DH);      // own row, used once: straight from global
    float mk[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) mk[j] = 1.f;
    if (A.training && A.p > 0.f) {
        const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
        const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
        const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
        float4 m0, m1;                                                         // keys 0..7 of query row i: ONE Philox call
        drop8(rk, site, ebase, m0, m1);
        mk[0] = m0.x; mk[1] = m0.y; mk[2] = m0.z; mk[3] = m0.w; mk[4] = m1.x; mk[5] = m1.y; mk[6] = m1.z; mk[7] = m1.w;
    }
    lds_barrier();
    const float scale = 1.0f / sqrtf((float)DH);
    const float* Kr = S.Ks + (sq * NMAX) * LD + h * DH;
    const float* Vr = S.Vs + (sq * NMAX) * LD + h * DH;
    float s[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        s[j] = -INFINITY;
        if (j < nmax) {                                                        // block-uniform
            const float d = dot_row<DH>(q, Kr + j * LD) * scale;
            if (act && j <= i && !((padmask >> j) & 1u)) s[j] = d;
        }
    }
    float m = s[0];
#pragma unroll
    for (int j = 1; j < NMAX; ++j) m = fmaxf(m, s[j]);
    if (!act) m = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) { s[j] = __expf(s[j] - m); sum += s[j]; }
    const float inv = act ? 1.0f / sum : 0.f;
    if (act) { float* st = A.stat + ((size_t)(t0 + i) * H + h) * 2; st[0] = m; st[1] = inv; }
    float o[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = 0.f;
#pragma unroll
    for (int j = 0; j < NMAX; ++j)
        if (j < nmax) axpy_row<DH>(o, s[j] * inv * mk[j], Vr + j * LD);        // rows >= n are zero in LDS, s[j] = 0 beyond the causal bound
    if (act) {
        float* dst = A.ctx + (size_t)(t0 + i) * D + h * DH;
#pragma unroll
        for (int c = 0; c < DH; c += 4) st4(dst + c, make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]));
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Phase A (lane = query row i): P from the saved statistics, dP~ = V dctx_i, dS = P (dP~ mask - <dctx_i, ctx_i>) scale, dQ_i += dS K_j;
// dS and P~ = P mask go to the (sequence, head)'s 8x8 LDS tile, transposed.  The lanes' own dctx and Q rows replace V and K in
// LDS and phase B (lane = key row j) forms dK_j = sum_i dS[i][j] Q_i, dV_j = sum_i P~[i][j] dctx_i — nothing is recomputed, no
// second round of Philox.
template <int DH>
__device__ __forceinline__ void bwd_body(const AttnArgs2& A, const int bid) {
    constexpr int D = 2 * DH, H = 2, LD = D + 4;
    const int cnt = *A.list_count;
    if (bid * SPB >= cnt) return;
    const TinyLds<DH> S(true);
    const int i = threadIdx.x & 7, grp = threadIdx.x >> 3, h = grp & 1, sq = threadIdx.x >> 4;
    unsigned padmask;
    const int nmax = tiny_stage<DH>(A, S, cnt, padmask, bid);
    const int t0 = S.meta[sq], n = S.meta[SPB + sq], b = S.meta[2 * SPB + sq];
    const bool act = i < n;
    float q[DH], cf[DH];
    float mi = 0.f, inv = 0.f, rdot = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) { q[c] = 0.f; cf[c] = 0.f; }
    if (act) {
        load_row<DH>(q, A.qkv + (size_t)(t0 + i) * 3 * D + h * DH);
        load_row<DH>(cf, A.dctx + (size_t)(t0 + i) * D + h * DH);
        const float2 st = *reinterpret_cast<const float2*>(A.stat + ((size_t)(t0 + i) * H + h) * 2);
        mi = st.x; inv = st.y;
        rdot = A.rd[(size_t)(t0 + i) * H + h];
    }
    float mk[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) mk[j] = 1.f;
    if (A.training && A.p > 0.f) {
        const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
        const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
        const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
        float4 m0, m1;                                                         // keys 0..7 of query row i: ONE Philox call
        drop8(rk, site, ebase, m0, m1);
        mk[0] = m0.x; mk[1] = m0.y; mk[2] = m0.z; mk[3] = m0.w; mk[4] = m1.x; mk[5] = m1.y; mk[6] = m1.z; mk[7] = m1.w;
    }
    lds_barrier();
    const float scale = 1.0f / sqrtf((float)DH);
    float* Kr = S.Ks + (sq * NMAX) * LD + h * DH;
    float* Vr = S.Vs + (sq * NMAX) * LD + h * DH;
    float* X = S.X + grp * XS;
    // Phase A in two passes so that q, dctx and dq are never live together (192 -> under 128 VGPRs: the kernel can then share a launch
    // with the 16-row list class, attn_mfma.hip k_attn_small_bwd): A1 forms dS and P~ (q and dctx_i live), then dctx_i takes V's place
    // in LDS; A2 accumulates dQ_i = sum_j dS[i][j] K_j (q and dq live), then q takes K's place.  The workgroup is ONE wave: the extra
    // barriers cost nothing.
    float ds[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        float pt = 0.f;
        ds[j] = 0.f;
        if (j < nmax) {
            const float sc = dot_row<DH>(q, Kr + j * LD) * scale;
            const float dp = dot_row<DH>(cf, Vr + j * LD);
            const float p = (act && j <= i && !((padmask >> j) & 1u)) ? __expf(sc - mi) * inv : 0.f;
            ds[j] = p * (dp * mk[j] - rdot) * scale;
            pt = p * mk[j];
        }
        X[j * NMAX + i] = ds[j]; X[NMAX * NMAX + j * NMAX + i] = pt;            // transposed: key lane j reads its 8 query entries contiguously
    }
    lds_barrier();                              // every lane is done with V: the lanes' own dctx rows take its place
#pragma unroll
    for (int c = 0; c < DH; c += 4) st4(Vr + i * LD + c, make_float4(cf[c], cf[c + 1], cf[c + 2], cf[c + 3]));
    {
        float dq[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) dq[c] = 0.f;
#pragma unroll
        for (int j = 0; j < NMAX; ++j)
            if (j < nmax) axpy_row<DH>(dq, ds[j], Kr + j * LD);
        if (act) {
            float* dst = A.dqkv + (size_t)(t0 + i) * 3 * D + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) st4(dst + c, make_float4(dq[c], dq[c + 1], dq[c + 2], dq[c + 3]));
        }
    }
    lds_barrier();                              // every lane is done with K: the lanes' own Q rows take its place
#pragma unroll
    for (int c = 0; c < DH; c += 4) st4(Kr + i * LD + c, make_float4(q[c], q[c + 1], q[c + 2], q[c + 3]));
    lds_barrier();
    {
        float dsr[NMAX], ptr[NMAX];
        { const float4 a = ld4(X + i * NMAX), bb = ld4(X + i * NMAX + 4);
          dsr[0] = a.x; dsr[1] = a.y; dsr[2] = a.z; dsr[3] = a.w; dsr[4] = bb.x; dsr[5] = bb.y; dsr[6] = bb.z; dsr[7] = bb.w; }
        { const float4 a = ld4(X + NMAX * NMAX + i * NMAX), bb = ld4(X + NMAX * NMAX + i * NMAX + 4);
          ptr[0] = a.x; ptr[1] = a.y; ptr[2] = a.z; ptr[3] = a.w; ptr[4] = bb.x; ptr[5] = bb.y; ptr[6] = bb.z; ptr[7] = bb.w; }
        float dk[DH], dv[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
#pragma unroll
        for (int ii = 0; ii < NMAX; ++ii)
            if (ii < nmax) {                    // entries with ii < i (non-causal) or ii >= n are exact zeros in the tile
                axpy_row<DH>(dk, dsr[ii], Kr + ii * LD);
                axpy_row<DH>(dv, ptr[ii], Vr + ii * LD);
            }
        if (act) {
            float* dst = A.dqkv + (size_t)(t0 + i) * 3 * D + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                st4(dst + D + c, make_float4(dk[c], dk[c + 1], dk[c + 2], dk[c + 3]));
                st4(dst + 2 * D + c, make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]));
            }
        }
    }
}

template <int DH> __global__ __launch_bounds__(NT) void k_attn_tiny_fwd(const AttnArgs2 A) { fwd_body<DH>(A, blockIdx.x); }
template <int DH> __global__ __launch_bounds__(NT) void k_attn_tiny_bwd(const AttnArgs2 A) { bwd_body<DH>(A, blockIdx.x); }

}  // namespace tiny
