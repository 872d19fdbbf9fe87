// attn_tile.h (included by linear.hip) — causal 2-head self-attention computed INSIDE the 16-token tile kernels of the latency regime
// (round 4): no attention launches.
//
// Why.  At the reference's batch size (256 sequences, ~1 500 tokens) the step was ten dependent launches of 8–25 µs; four of them were the
// attention (one workgroup per sequence, 4 x ~10 µs of 122), each a chain of round trips in front of a handful of MFMAs.  Attention is
// causal: token t needs the K | V rows of the EARLIER tokens of its own sequence only — and in the packed token stream those are the at
// most L - 1 <= 63 rows in front of it.  So the workgroup that owns tokens [t0, t0 + 16) (k_post_fwd / k_post_mid / k_post_bwd) stages
// the 80 rows [t0 - 64, t0 + 16) of this layer's qkv in LDS and runs the attention of its 16 query rows itself, every query masked to
// the rows of its own sequence (per-token words {first token of the sequence, sequence slot | PAD flag} written by the embedding stage).
//   forward  (head of k_post_fwd / k_post_mid): S^T = K Q^T over the window's key tiles, softmax, P~ V -> the ctx tile lands in the LDS
//             tile the out_proj GEMM reads (and in global memory: the weight gradient of out_proj and the row term read it);
//   backward (tail of k_post_mid / k_post_bwd, where the dctx tile has just been produced in LDS): waves 0, 1 (one per head) recompute P
//             in the transposed orientation and emit dQ of the tile's rows (complete: plain stores); waves 2, 3 recompute it in the natural
//             orientation and ADD dK | dV into the window's rows with fp32 atomics — a key row collects from every tile that holds later
//             tokens of its sequence (at most 5).  The rows are zeroed by the launch that produced this layer's qkv.
// Arithmetic, saved statistics and dropout element indexing ((b*H + h)*64 + i)*64 + j are those of attn_mfma.hip (the one-workgroup-per-
// sequence kernels, still used at scale and as the cross-check: DR4SR_ATTN_SEPARATE=1), so the two forms are interchangeable per launch.
//
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer, /root/reference model/sasrec.py:21-34, attn_mask
// triu(1) :58, key_padding_mask idx == 0 :48.
#pragma once
#include "common.h"
#include "kernels.h"

#define TSTAMP(i, thr) do { if (P.stamps && blockIdx.x == 0 && threadIdx.x == (thr)) P.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
namespace tattn {

constexpr int WR = 80;            // window rows: tokens [t0 - 64, t0 + 16)
constexpr int QR0 = 64;           // window row of the tile's first token
constexpr int MT = WR / 16;

template <int D>
struct Lds {
    static constexpr int LD = D + 4;
    static constexpr int floats = 2 * WR * LD + 16 * LD + 64 + 32 + 2 * WR + 8;
    float *Kw, *Vw, *Qs, *st, *rd; int2* tok; unsigned* pad;
    __device__ __forceinline__ explicit Lds(float* base) {
        Kw = base; Vw = Kw + WR * LD; Qs = Vw + WR * LD;
        st = Qs + 16 * LD;                                    // [H][16][2] row max, 1 / sum
        rd = st + 64;                                         // [16][H]    <dctx, ctx> per head
        tok = reinterpret_cast<int2*>(rd + 32);               // [WR] {first token of the sequence, slot | length << 20 | PAD << 30}
        pad = reinterpret_cast<unsigned*>(tok + WR);          // [MT] PAD flags of the rows of key tile jt as bits
    }
};

__device__ __forceinline__ float xg_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16, 64)), fmaxf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
__device__ __forceinline__ float xg_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }

// rows of one operand: lane (r16, g) reads DH / 4 contiguous floats of row (row0 + r16) at column col0 + g * DH / 4
template <int DH>
__device__ __forceinline__ void frag(float (&f)[DH / 4], const float* __restrict__ base, int ld, int row0, int col0) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const float* p = base + (row0 + r16) * ld + col0 + g * (DH / 4);
#pragma unroll
    for (int c = 0; c < DH / 4; c += 4) {
        const float4 v = ld4(p + c);
        f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w;
    }
}
template <int DH>
__device__ __forceinline__ void frag_g(float (&f)[DH / 4], const float* __restrict__ base, int ld, int t0, int T) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) f[c] = 0.f;
    if (t0 + r16 < T) {
        const float* p = base + (size_t)(t0 + r16) * ld + g * (DH / 4);
#pragma unroll
        for (int c = 0; c < DH / 4; c += 4) {
            const float4 v = ld4(p + c);
            f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w;
        }
    }
}
// C[16 x 16] = A_rows . B_rows^T over DH features: lane (n = l & 15, g) holds C[4 g + e][n]
template <int DH>
__device__ __forceinline__ f32x4 mma_rows(const float (&a)[DH / 4], const float (&b)[DH / 4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) acc = mfma16x4(a[s], b[s], acc);
    return acc;
}

// The 64 keep decisions of query row (b, h, i) (bit j = the probability of key position j is kept): lane (i16, g) runs the Philox calls
// of key groups g and g + 4 of ITS query row — only those the causal mask leaves — and the bytes go round the row's four lanes.
__device__ __forceinline__ void keep_bits(const RngKey& rk, const uint32_t site, const uint64_t ebase, const int i, uint32_t& lo, uint32_t& hi) {
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4;
    unsigned a = 0xffu, b = 0xffu;
    if (i >= 8 * g) a = drop_bits8(rk, site, ebase + 8 * g);
    if (i >= 8 * (g + 4)) b = drop_bits8(rk, site, ebase + 8 * (g + 4));
    lo = 0; hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lo |= ((unsigned)__shfl((int)a, i16 | (k << 4), 64) & 0xffu) << (8 * k);
        hi |= ((unsigned)__shfl((int)b, i16 | (k << 4), 64) & 0xffu) << (8 * k);
    }
}
__device__ __forceinline__ bool keep_bit(const uint32_t lo, const uint32_t hi, const int j) { return (((j & 32) ? hi : lo) >> (j & 31)) & 1u; }

// K | V rows (and the tokens' words) of the window into LDS; WITH_Q: also the tile's Q rows, WITH_STAT: the saved statistics (backward).
// Rows outside [max(0, t0 - (L - 1)), min(T, t0 + 16)) are zero-filled: they are multiplied by zero probabilities.
// Two halves — issue() requests everything into registers, commit() stores to LDS — so that the caller's own first loads (and the
// Philox calls of the keep decisions) sit between them: one global round trip, not two.
template <int D, bool WITH_Q, bool WITH_STAT>
struct Stage {
    static constexpr int LD = D + 4, C4 = 2 * D / 4, N4 = WR * C4, PER = (N4 + 255) / 256, QPER = (16 * D / 4) / 256;
    int2 tw; float4 v[PER]; float4 qv[WITH_Q ? QPER : 1]; float2 sv;
    // rows [r_lo, r_hi) of the window (multiples of 16: whole key tiles)
    __device__ __forceinline__ void issue(const TileAttnArgs& A, const int t0, const int T, const int r_lo = 0, const int r_hi = WR) {
        const int wb = t0 - QR0, lo = max(0, t0 - (A.L - 1)), hi = min(T, t0 + 16);
        tw = make_int2(0x7fffffff, 0);
        if ((int)threadIdx.x >= r_lo && (int)threadIdx.x < r_hi) {
            const int tk = wb + (int)threadIdx.x;
            if (tk >= lo && tk < hi) tw = A.tok[tk];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int f = threadIdx.x + 256 * q, r = f / C4, c4 = f % C4, tk = wb + r;
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r >= r_lo && r < r_hi && tk >= lo && tk < hi) v[q] = ld4(A.qkv + (size_t)tk * 3 * D + D + 4 * c4);
        }
        if constexpr (WITH_Q) {
#pragma unroll
            for (int q = 0; q < QPER; ++q) {
                const int f = threadIdx.x + 256 * q, r = f / (D / 4), c4 = f % (D / 4);
                qv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t0 + r < T) qv[q] = ld4(A.qkv + (size_t)(t0 + r) * 3 * D + 4 * c4);
            }
        }
        sv = make_float2(0.f, 0.f);
        if constexpr (WITH_STAT) {
            if (threadIdx.x < 32) {                          // (token i, head h) = (thread >> 1, thread & 1)
                const int i = threadIdx.x >> 1;
                if (t0 + i < T) sv = *reinterpret_cast<const float2*>(A.stat + ((size_t)(t0 + i) * 2 + (threadIdx.x & 1)) * 2);
            }
        }
    }
    __device__ __forceinline__ void commit(const Lds<D>& S, const int r_lo = 0, const int r_hi = WR) const {
        if ((int)threadIdx.x >= r_lo && (int)threadIdx.x < r_hi) S.tok[threadIdx.x] = tw;
        if (threadIdx.x < 128) {                             // waves 0, 1 hold the 80 rows' words
            const unsigned long long bal = __ballot(((tw.y >> 30) & 1) != 0);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    if (16 * jt >= r_lo && 16 * jt < r_hi) S.pad[jt] = (unsigned)(bal >> (16 * jt)) & 0xffffu;
            }
            if (threadIdx.x == 64 && 64 >= r_lo && 64 < r_hi) S.pad[4] = (unsigned)bal & 0xffffu;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int f = threadIdx.x + 256 * q, r = f / C4, c4 = f % C4;
            if (r >= r_lo && r < r_hi) st4((4 * c4 < D ? S.Kw + r * LD + 4 * c4 : S.Vw + r * LD + 4 * c4 - D), v[q]);
        }
        if constexpr (WITH_Q) {
#pragma unroll
            for (int q = 0; q < QPER; ++q) {
                const int f = threadIdx.x + 256 * q, r = f / (D / 4), c4 = f % (D / 4);
                st4(S.Qs + r * LD + 4 * c4, qv[q]);
            }
        }
        if constexpr (WITH_STAT) {
            if (threadIdx.x < 32) {
                const int i = threadIdx.x >> 1, h = threadIdx.x & 1;
                S.st[(h * 16 + i) * 2] = sv.x; S.st[(h * 16 + i) * 2 + 1] = sv.y;
            }
        }
    }
};

// Short-sequence plans (A.on & 4: the plan expects a mean length of at most 16 tokens) stage the NEAR rows [t0 - 16, t0 + 16) only — half the
// window's bytes, and at these sizes the staging time is bytes x latency (the rows come from another XCD's launch: 0.9 us of k_post_fwd at
// d = 64, 4 us at d = 128) — and fetch the far rows in a second round trip in the tiles whose first token's sequence started more than 16
// tokens earlier (one in ten on a toys-shaped batch).  Call with every thread of the workgroup, behind a barrier that made the near rows'
// words visible; ends with a barrier when it staged something.
constexpr int NEAR0 = WR - 32;
template <int D>
__device__ __forceinline__ void far_rows_if_needed(const TileAttnArgs& A, const Lds<D>& S, const int t0, const int T) {
    if (!(A.on & 4) || S.tok[QR0].x >= t0 - 16) return;       // (workgroup-uniform)
    Stage<D, false, false> st;
    st.issue(A, t0, T, 0, NEAR0);
    st.commit(S, 0, NEAR0);
    lds_barrier();
}

// keep decisions of this lane's query row (tile row l & 15, head (wave & 1)) from the token's word in GLOBAL memory: requested before the
// staging loads, computed while they are in flight
struct Keep { uint32_t lo, hi; };
__device__ __forceinline__ int2 own_word(const TileAttnArgs& A, const int t0, const int T) {
    const int tq = t0 + (int)(threadIdx.x & 15);
    return tq < T ? A.tok[tq] : make_int2(0x7fffffff, 0);
}
__device__ __forceinline__ Keep own_keep(const PostArgs& P, const int2 mq, const int t0, const int T) {
    Keep k{0xffffffffu, 0xffffffffu};
    if (P.training && P.p > 0.f) {
        const int tq = t0 + (int)(threadIdx.x & 15), h = (threadIdx.x >> 6) & 1, i = tq < T ? tq - mq.x : -1;
        const RngKey rk = make_rng(P.seed, (uint32_t)P.state[DR4SR_STATE_RNGSTEP], P.p);
        keep_bits(rk, DR4SR_SITE_ATTN + 4 * P.layer, ((uint64_t)((mq.y & 0xfffff) * 2 + h) * 64 + i) * 64, i, k.lo, k.hi);
    }
    return k;
}

// ------------------------------------------------------------------------------------------------ forward
// Wave w: head h = w & 1 (scores and softmax of the 16 query rows, computed by both waves of a head), output columns block(s) w >> 1.
// Leaves the ctx tile in R0 [16][ldr] (rows >= T zero) and in A.ctx; a workgroup barrier must follow before R0 is read.
// KEEP_QS: k_post_mid runs the backward of the same tile later in the launch — the Q rows and the statistics stay in LDS for it.
// The forward in three pieces, so that the caller can put its own long-latency loads BETWEEN them: fwd_issue() requests the window, the
// tokens' words and the Q rows; fwd_stage() computes the keep decisions while they are in flight and commits the window; fwd_compute().  Loads return in issue
// order: k_post_fwd / k_post_mid prefetch 128 KB (d = 64) / 200 KB (d = 128) of weight fragments per workgroup for their four GEMMs.  At
// d = 128 the window goes first (linear.hip post_fwd_body: no register spill that way, step -4.5 %); at d = 64 the order made no difference
// (measured: NOTEBOOK round 4).
template <int D, bool KEEP_QS>
struct FwdPre { Stage<D, KEEP_QS, false> st; int2 mq_g; float qf[D / 8]; uint32_t klo, khi; };
template <int D, bool KEEP_QS>
__device__ __forceinline__ void fwd_issue(const PostArgs& P, const int t0, const int T, FwdPre<D, KEEP_QS>& F) {
    constexpr int DH = D / 2;
    const TileAttnArgs& A = P.at;
    const int h = (threadIdx.x >> 6) & 1;
    F.mq_g = own_word(A, t0, T);
    if constexpr (!KEEP_QS) frag_g<DH>(F.qf, A.qkv + h * DH, 3 * D, t0, T);       // used once: straight from global
    F.st.issue(A, t0, T, (A.on & 4) ? NEAR0 : 0, WR);
}
// ... fwd_stage(): keep decisions (Philox, while the window is in flight), window -> LDS, workgroup barrier, far rows on demand
template <int D, bool KEEP_QS>
__device__ __forceinline__ void fwd_stage(const PostArgs& P, const int t0, const int T, float* base, FwdPre<D, KEEP_QS>& F) {
    const TileAttnArgs& A = P.at;
    const Lds<D> S(base);
    const int r_lo = (A.on & 4) ? NEAR0 : 0;
    TSTAMP(27, 0);
    __builtin_amdgcn_sched_barrier(0);
    const Keep k0 = own_keep(P, F.mq_g, t0, T);
    F.klo = k0.lo; F.khi = k0.hi;
    __builtin_amdgcn_sched_barrier(0);
    TSTAMP(28, 0);
    F.st.commit(S, r_lo, WR);
    TSTAMP(8, 0);
    lds_barrier();
    far_rows_if_needed<D>(A, S, t0, T);
    TSTAMP(9, 0);
}
// ... fwd_compute(): scores, softmax, P~ V
template <int D, bool KEEP_QS>
__device__ __forceinline__ Keep fwd_compute(const PostArgs& P, const int t0, const int T, float* R0, const int ldr, float* base, FwdPre<D, KEEP_QS>& F) {
    constexpr int DH = D / 2, LD = D + 4, H = 2;
    const TileAttnArgs& A = P.at;
    const Lds<D> S(base);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = w & 1, half = w >> 1, i16 = lane & 15, g = lane >> 4;
    const uint32_t klo_ = F.klo, khi_ = F.khi;
    float qf[DH / 4];
    if constexpr (!KEEP_QS) {
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) qf[c] = F.qf[c];
    }
    if constexpr (KEEP_QS) frag<DH>(qf, S.Qs, LD, 0, h * DH);
    const int tq = t0 + i16, wb = t0 - QR0;
    const int2 mq = S.tok[QR0 + i16];
    const bool qok = tq < T;
    const int qs = mq.x;
    const int jt_lo = (S.tok[QR0].x - wb) >> 4;          // first key tile a query of this tile can see (token t0 < T always)
    const float scale = 1.0f / sqrtf((float)DH);
    f32x4 s[MT];
    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (jt >= jt_lo) {
            float kf[DH / 4];
            frag<DH>(kf, S.Kw, LD, jt * 16, h * DH);
            s[jt] = mma_rows<DH>(kf, qf);
            const unsigned pw = S.pad[jt] >> (4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tk = wb + jt * 16 + 4 * g + r;
                const bool ok = qok && tk >= qs && tk <= tq && !((pw >> r) & 1u);
                const float v = ok ? s[jt][r] * scale : -INFINITY;
                s[jt][r] = v;
                m = fmaxf(m, v);
            }
        }
    }
    TSTAMP(10, 0);
    m = xg_max(m);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
        if (jt >= jt_lo)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(s[jt][r] - m); s[jt][r] = e; sum += e; }
    sum = xg_sum(sum);
    const float inv = 1.0f / sum;
    if (half == 0 && g == 0) {
        if (qok) { float* st = A.stat + ((size_t)tq * H + h) * 2; st[0] = m; st[1] = inv; }
        if constexpr (KEEP_QS) { S.st[(h * 16 + i16) * 2] = m; S.st[(h * 16 + i16) * 2 + 1] = inv; }
    }
    const bool dodrop = P.training && P.p > 0.f;
    const uint32_t klo = klo_, khi = khi_;
    const float keepv = dodrop ? 1.0f / (1.0f - P.p) : 1.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
        if (jt >= jt_lo)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = wb + jt * 16 + 4 * g + r - qs;
                const float mk = keep_bit(klo, khi, j) ? keepv : 0.f;
                s[jt][r] = qok ? s[jt][r] * inv * mk : 0.f;
            }
    TSTAMP(11, 0);
    // out^T[d][i] = sum_j V[j][d] P~[i][j]
#pragma unroll
    for (int q = 0; q < DH / 32; ++q) {
        const int db = half * (DH / 32) + q;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
            if (jt >= jt_lo)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o = mfma16x4(S.Vw[(jt * 16 + 4 * g + r) * LD + h * DH + db * 16 + i16], s[jt][r], o);
        const float4 ov = qok ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (qok) st4(A.ctx + (size_t)tq * D + h * DH + db * 16 + 4 * g, ov);
        st4(R0 + i16 * ldr + h * DH + db * 16 + 4 * g, ov);
    }
    TSTAMP(12, 0);
    return Keep{klo, khi};
}
template <int D, bool KEEP_QS>
__device__ __forceinline__ Keep fwd(const PostArgs& P, const int t0, const int T, float* R0, const int ldr, float* base) {
    FwdPre<D, KEEP_QS> F;
    fwd_issue<D, KEEP_QS>(P, t0, T, F);
    fwd_stage<D, KEEP_QS>(P, t0, T, base, F);
    return fwd_compute<D, KEEP_QS>(P, t0, T, R0, ldr, base, F);
}

// ------------------------------------------------------------------------------------------------ backward
// Cs = the dctx tile [16][ldc] in LDS, S.rd = <dctx, ctx> per (row, head), window / Q rows / statistics staged, and a workgroup barrier
// has passed.  Waves 0, 1: dQ of head w (transposed orientation); waves 2, 3: dK | dV of head w - 2 (natural orientation), added into
// the window's rows with atomics — except the rows whose sequence starts and ends inside this tile (no other tile adds to them):
// plain 16-byte stores.
// Measured and dropped (same box, ms per step at B = 256): every wave both phases in sequence, key tiles / feature blocks split between
// the two waves of a head, so that the atomics are in flight during the dQ phase: 0.1090 against 0.1066; the dK | dV rows through a
// per-wave LDS exchange tile so that an atomic instruction covers ONE row's 64 columns instead of 4-byte pieces of 16 rows: 0.1091 /
// 0.1080 — the isolated launch liked both (19.2 against 20.1 us), the step did not.
// DET (deterministic latency form, kernels.h Workspace::det_lat): the rows another tile adds to as well go to this query tile's partial blocks
// instead (linear.hip det_kv_rows sums them in query-tile order) — a separate instantiation, the default kernels are unchanged.
template <int D, bool DET = false>
__device__ __forceinline__ void bwd(const PostArgs& P, const int t0, const int T, const float* Cs, const int ldc, float* base, const Keep keep) {
    constexpr int DH = D / 2, LD = D + 4;
    const TileAttnArgs& A = P.at;
    const Lds<D> S(base);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = w & 1, i16 = lane & 15, g = lane >> 4;
    const int wb = t0 - QR0, wmin = S.tok[QR0].x, jt_lo = (wmin - wb) >> 4, hi = min(T, t0 + 16);
    const float scale = 1.0f / sqrtf((float)DH);
    const bool dodrop = P.training && P.p > 0.f;
    const float keepv = dodrop ? 1.0f / (1.0f - P.p) : 1.f;
    const uint32_t klo = keep.lo, khi = keep.hi;           // keep decisions of query row i16, head h (own_keep)
    TSTAMP(20, 0); TSTAMP(24, 128);
    float qf[DH / 4], cf[DH / 4];
    frag<DH>(qf, S.Qs, LD, 0, h * DH);
    frag<DH>(cf, Cs, ldc, 0, h * DH);
    if (w < 2) {
        // ---- dQ: lane (query i16, keys 4 g + r)
        const int tq = t0 + i16, qs = S.tok[QR0 + i16].x;
        const bool qok = tq < T;
        const float mi = S.st[(h * 16 + i16) * 2], inv = S.st[(h * 16 + i16) * 2 + 1], rdot = S.rd[i16 * 2 + h];
        f32x4 ds[MT];
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
            ds[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (jt >= jt_lo) {
                float kf[DH / 4];
                frag<DH>(kf, S.Kw, LD, jt * 16, h * DH);
                const f32x4 s = mma_rows<DH>(kf, qf);
                frag<DH>(kf, S.Vw, LD, jt * 16, h * DH);
                const f32x4 dp = mma_rows<DH>(kf, cf);         // dP~^T[j][i] = sum_d V[j][d] dctx[i][d]
                const unsigned pw = S.pad[jt] >> (4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tk = wb + jt * 16 + 4 * g + r;
                    const bool ok = qok && tk >= qs && tk <= tq && !((pw >> r) & 1u);
                    const float p = ok ? __expf(s[r] * scale - mi) * inv : 0.f;
                    const float mk = keep_bit(klo, khi, tk - qs) ? keepv : 0.f;
                    ds[jt][r] = p * (dp[r] * mk - rdot) * scale;
                }
            }
        }
        TSTAMP(21, 0);
        // dQ^T[f][i] = sum_j K[j][f] dS^T[j][i]
#pragma unroll
        for (int fb = 0; fb < DH / 16; ++fb) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < MT; ++jt)
                if (jt >= jt_lo)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o = mfma16x4(S.Kw[(jt * 16 + 4 * g + r) * LD + h * DH + fb * 16 + i16], ds[jt][r], o);
            if (qok) st4(A.dqkv + (size_t)tq * 3 * D + h * DH + fb * 16 + 4 * g, make_float4(o[0], o[1], o[2], o[3]));
        }
        TSTAMP(22, 0);
        return;
    }
    // ---- dK | dV: lane (key i16 of the key tile, queries 4 g + r)
    int qsr[4]; float mr[4], invr[4], rdr[4]; uint32_t qlo[4], qhi[4]; bool qokr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = 4 * g + r;
        qsr[r] = S.tok[QR0 + q].x; qokr[r] = t0 + q < T;
        mr[r] = S.st[(h * 16 + q) * 2]; invr[r] = S.st[(h * 16 + q) * 2 + 1]; rdr[r] = S.rd[q * 2 + h];
        qlo[r] = (uint32_t)__shfl((int)klo, q, 64); qhi[r] = (uint32_t)__shfl((int)khi, q, 64);
    }
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
        if (jt < jt_lo) continue;
        float kf[DH / 4], vf[DH / 4];
        frag<DH>(kf, S.Kw, LD, jt * 16, h * DH);
        frag<DH>(vf, S.Vw, LD, jt * 16, h * DH);
        const f32x4 s = mma_rows<DH>(qf, kf);                  // S[i][j]: rows i = 4 g + r, column j = l & 15
        const f32x4 dp = mma_rows<DH>(cf, vf);                 // dP~[i][j] = sum_d dctx[i][d] V[j][d]
        const int c = jt * 16 + i16, tk = wb + c;
        const int2 kw = S.tok[c];
        const bool kok = !((S.pad[jt] >> i16) & 1u);
        f32x4 pt, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tq = t0 + 4 * g + r;
            const bool ok = qokr[r] && kok && tk >= qsr[r] && tk <= tq;
            const float p = ok ? __expf(s[r] * scale - mr[r]) * invr[r] : 0.f;
            const float mk = keep_bit(qlo[r], qhi[r], tk - qsr[r]) ? keepv : 0.f;
            pt[r] = p * mk;                                    // P~[i][j]
            ds[r] = p * (dp[r] * mk - rdr[r]) * scale;
        }
        const bool live = tk >= wmin && tk < hi;               // a row of one of the tile's sequences
        const bool own = (DET || !(A.on & 2)) && tk >= t0 && kw.x + ((kw.y >> 20) & 0x7f) <= t0 + 16;     // ... that no other tile adds to
        // dK^T[f][j] = sum_i Q[i][f] dS[i][j] ;  dV^T[d][j] = sum_i dctx[i][d] P~[i][j]
#pragma unroll
        for (int fb = 0; fb < DH / 16; ++fb) {
            f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dk = mfma16x4(S.Qs[(4 * g + r) * LD + h * DH + fb * 16 + i16], ds[r], dk);
                dv = mfma16x4(Cs[(4 * g + r) * ldc + h * DH + fb * 16 + i16], pt[r], dv);
            }
            if (live) {
                float* dst = A.dqkv + (size_t)tk * 3 * D + D + h * DH + fb * 16 + 4 * g;
                if (own) {
                    st4(dst, make_float4(dk[0], dk[1], dk[2], dk[3]));
                    st4(dst + D, make_float4(dv[0], dv[1], dv[2], dv[3]));
                } else if constexpr (DET) {                    // this query tile's block for key tile jt
                    float* pb = A.kv_part + (((size_t)(t0 >> 4) * MT + jt) * 16 + i16) * 2 * D + h * DH + fb * 16 + 4 * g;
                    st4(pb, make_float4(dk[0], dk[1], dk[2], dk[3]));
                    st4(pb + D, make_float4(dv[0], dv[1], dv[2], dv[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(dst + e, dk[e]); unsafeAtomicAdd(dst + D + e, dv[e]); }
                }
            }
        }
    }
    TSTAMP(25, 128);
}

}  // namespace tattn
