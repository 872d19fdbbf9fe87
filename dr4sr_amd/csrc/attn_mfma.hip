// attn_mfma.hip — causal 2-head self-attention over ragged sequences on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// One workgroup (4 waves) per sequence; wave w owns head h = w&1 and query row-tiles {0,3} (w>>1 == 0) or {1,2}
// (causal work balance: 1+4 vs 2+3 key tiles).  Everything is computed in the TRANSPOSED orientation
//     S^T[j][i] = sum_k K[j][k] Q[i][k]          (A = K rows, B = Q rows)
// so the C-layout registers of a tile (lane: i = l&15, j = 4*(l>>4)+r) are, as they stand, the B operand of the next
// product  out^T[d][i] = sum_j V[j][d] P[i][j]  (k-permuted: lane group g supplies j = 4g+s at step s) — no LDS
// round trip, no shuffles between softmax and PV.  Row statistics reduce over r in-lane and over the 4 lane groups
// with two shuffles.  The backward pass recomputes P in both orientations: phase A (query-tile owners) produces dQ and
// the row statistics, phase B (key-tile owners) produces dK and dV, so no cross-wave accumulation is needed.
//
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer, attn_mask = triu(ones,1)
// (model/sasrec.py:58), key_padding_mask = (idx == 0) (model/sasrec.py:48), scale 1/sqrt(head_dim), dropout on the
// probabilities.  Dropout element index ((b*H+h)*64 + i)*64 + j, 4 consecutive j per Philox call (= one lane's r=0..3).
#include "common.h"
#include "kernels.h"
#include "attn_args.h"

extern __shared__ __attribute__((aligned(16))) float smem[];


__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float xgroup_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16, 64)), fmaxf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
__device__ __forceinline__ float xgroup_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float lane16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// rows of one operand: lane (r16 = l&15, g = l>>4) reads DH/4 contiguous floats of row (row0 + r16) at column h*DH + g*DH/4
template <int DH>
__device__ __forceinline__ void load_frag(float (&f)[DH / 4], const float* __restrict__ base, int ld, int row0, int col0) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const float* p = base + (row0 + r16) * ld + col0 + g * (DH / 4);
#pragma unroll
    for (int c = 0; c < DH / 4; c += 4) {
        const float4 v = ld4(p + c);
        f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w;
    }
}

// the same fragment straight from global rows (row stride ld floats), rows >= n read as zero: operands that are used once per
// tile need no LDS copy — and every array not staged is a workgroup more per CU
template <int DH>
__device__ __forceinline__ void load_frag_g(float (&f)[DH / 4], const float* __restrict__ base, int ld, int row0, int n) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    if (row0 + r16 < n) {
        const float* p = base + (size_t)(row0 + r16) * ld + g * (DH / 4);
#pragma unroll
        for (int c = 0; c < DH / 4; c += 4) {
            const float4 v = ld4(p + c);
            f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < DH / 4; ++c) f[c] = 0.f;
    }
}

// C[16x16] += A_rows . B_rows^T over DH features (both given as row fragments)
template <int DH>
__device__ __forceinline__ f32x4 mma_rows(const float (&a)[DH / 4], const float (&b)[DH / 4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) acc = mfma16(a[s], b[s], acc);
    return acc;
}

template <int D, int ROWS, int NT>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, int ld, const float* __restrict__ src, int src_ld, int n) {
    for (int i = threadIdx.x; i < ROWS * (D / 4); i += NT) {
        const int r = i / (D / 4), c = (i % (D / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v = ld4(src + (size_t)r * src_ld + c);
        st4(dst + r * ld + c, v);
    }
}

// the same rows in two halves — global -> registers (issue only), registers -> LDS — for the persistent list kernels, which
// request the NEXT sequence's rows before computing the current one
template <int D, int ROWS, int NT, int PER>
__device__ __forceinline__ void rows_to_regs(float4 (&v)[PER], const float* __restrict__ src, int src_ld, int n) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = threadIdx.x + q * NT, r = i / (D / 4), c = (i % (D / 4)) * 4;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < ROWS * (D / 4) && r < n) v[q] = ld4(src + (size_t)r * src_ld + c);
    }
}
template <int D, int ROWS, int NT, int PER>
__device__ __forceinline__ void regs_to_rows(float* __restrict__ dst, int ld, const float4 (&v)[PER]) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = threadIdx.x + q * NT, r = i / (D / 4), c = (i % (D / 4)) * 4;
        if (i < ROWS * (D / 4)) st4(dst + r * ld + c, v[q]);
    }
}
// In the pipelined loops a value requested in iteration i is first TOUCHED in iteration i+1: the compiler places the vmcnt wait at
// the first use, and returns are in order — one early use (a subtraction, a compare) would drain every load requested before it.
struct SeqRaw { int b, c0, c1; int64_t row; };          // exactly what was loaded
struct SeqMeta { int b, t0, n; int64_t row; };
// The words are wave-uniform, and the compiler would move a uniform load's result to an SGPR (v_readfirstlane) right behind the load
// — a use, hence a wait.  `z` is a zero it cannot see through: the addresses look divergent, the results stay in VGPRs until
// seq_final() makes them scalar one iteration later.
__device__ __forceinline__ int opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ int list_raw(const AttnArgs2& A, int k, int z) { return A.list[k + z]; }
__device__ __forceinline__ SeqRaw seq_raw(const AttnArgs2& A, int b_raw, int z) {
    const int b = __builtin_amdgcn_readfirstlane(b_raw);
    SeqRaw m;
    m.b = b; m.c0 = A.cu[b + z]; m.c1 = A.cu[b + 1 + z]; m.row = A.rows ? A.rows[b + z] : (int64_t)b;
    return m;
}
__device__ __forceinline__ SeqMeta seq_final(const SeqRaw& r) {
    const int c0 = __builtin_amdgcn_readfirstlane(r.c0), c1 = __builtin_amdgcn_readfirstlane(r.c1);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)r.row), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)r.row >> 32));
    return SeqMeta{r.b, c0, c1 - c0, (int64_t)(((uint64_t)hi << 32) | lo)};
}

// Dropout keep factors of one query tile's probabilities in the transposed orientation (lane (i, g) holds keys jt * 16 + 4 g + 0..3 of
// query row i, for every key tile jt <= it).  The four keys of a lane are HALF of a Philox call (16-bit decisions, 8 per call) and lanes
// g, g ^ 1 would both compute it; with four key tiles, lane (i, g) instead computes the call of key tile 2 round + (g & 1), key half
// g >> 1, and the 8-bit masks are exchanged inside the lanes of query row i (ds_bpermute): ceil((it + 1) / 2) calls per lane instead of
// it + 1.  MT == 1 (16-row kernels): one tile, nothing to share.
template <int MT>
__device__ __forceinline__ void attn_keep_masks(float4 (&mk)[MT], const RngKey& rk, const uint32_t site, const uint64_t ebase, const int it,
                                                const int lane, const bool dodrop) {
    const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) mk[jt] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (!dodrop) return;
    if constexpr (MT == 1) { mk[0] = drop4(rk, site, ebase + 4 * g); return; }
    else {
#pragma unroll
        for (int rnd = 0; rnd < MT / 2; ++rnd) {
            if (2 * rnd <= it) {                            // wave-uniform
                const unsigned m8 = drop_bits8(rk, site, ebase + (2 * rnd + (g & 1)) * 16 + 8 * (g >> 1));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int jt = 2 * rnd + q;
                    const unsigned mm = __shfl(m8, i16 | ((2 * (g >> 1) + q) << 4), 64) >> (4 * (g & 1));
                    mk[jt] = make_float4((mm & 1u) ? rk.scale : 0.f, (mm & 2u) ? rk.scale : 0.f, (mm & 4u) ? rk.scale : 0.f, (mm & 8u) ? rk.scale : 0.f);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// K, V and the key-padding flags of the sequence are in LDS (and a barrier has passed)
template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_fwd_compute(const AttnArgs2& A, const int b, const int t0, const int n, const uint32_t rngstep) {
    constexpr int D = 2 * DH, LD = D + 4, H = 2, MT = ROWS / 16;
    float* Ks = smem;                    // [ROWS][LD]   (Q fragments come straight from global: used once per query tile)
    float* Vs = Ks + ROWS * LD;
    int* kpad = reinterpret_cast<int*>(Vs + ROWS * LD);      // [64]
    const float* src = A.qkv + (size_t)t0 * 3 * D;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = w & 1, half = w >> 1;
    const int i16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, rngstep, A.p);
    const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
    const float scale = 1.0f / sqrtf((float)DH);
    const int ntile = (n + 15) >> 4;
#pragma unroll
    for (int pass = 0; pass < (MT == 1 ? 1 : 2); ++pass) {
        const int it = pass == 0 ? (half == 0 ? 0 : 1) : (half == 0 ? 3 : 2);
        if (it >= ntile) continue;
        const int i = it * 16 + i16;                       // this lane's query row
        float qf[DH / 4];
        load_frag_g<DH>(qf, src + h * DH, 3 * D, it * 16, n);
        f32x4 s[MT];
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
            if (jt <= it) {
                float kf[DH / 4];
                load_frag<DH>(kf, Ks, LD, jt * 16, h * DH);
                s[jt] = mma_rows<DH>(kf, qf);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = jt * 16 + 4 * g + r;
                    const float v = (j <= i && j < n && !kpad[j]) ? s[jt][r] * scale : -INFINITY;
                    s[jt][r] = v;
                    m = fmaxf(m, v);
                }
            }
        }
        m = xgroup_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
            if (jt <= it)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __expf(s[jt][r] - m); s[jt][r] = e; sum += e; }
        sum = xgroup_sum(sum);
        const float inv = 1.0f / sum;
        if (g == 0 && i < n) { float* st = A.stat + ((size_t)(t0 + i) * H + h) * 2; st[0] = m; st[1] = inv; }
        const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
        float4 mkt[MT];
        attn_keep_masks<MT>(mkt, rk, site, ebase, it, lane, dodrop);
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
            if (jt <= it) {
                const float4 mk = mkt[jt];
                s[jt][0] *= inv * mk.x; s[jt][1] *= inv * mk.y; s[jt][2] *= inv * mk.z; s[jt][3] *= inv * mk.w;
            }
        // out^T[d][i] = sum_j V[j][d] P~[i][j]
#pragma unroll
        for (int db = 0; db < DH / 16; ++db) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < MT; ++jt)
                if (jt <= it)
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx)
                        o = mfma16(Vs[(jt * 16 + 4 * g + sidx) * LD + h * DH + db * 16 + i16], s[jt][sidx], o);
            if (i < n) st4(A.ctx + (size_t)(t0 + i) * D + h * DH + db * 16 + 4 * g, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}


template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_fwd_seq(const AttnArgs2& A, const int b) {
    constexpr int D = 2 * DH, LD = D + 4;
    const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
    if (n <= 0) return;
    float* Ks = smem;
    float* Vs = Ks + ROWS * LD;
    int* kpad = reinterpret_cast<int*>(Vs + ROWS * LD);
    const int64_t row = A.rows ? A.rows[b] : b;
    const float* src = A.qkv + (size_t)t0 * 3 * D;
    stage_rows<D, ROWS, NT>(Ks, LD, src + D, 3 * D, n);
    stage_rows<D, ROWS, NT>(Vs, LD, src + 2 * D, 3 * D, n);
    if (threadIdx.x < 64) kpad[threadIdx.x] = threadIdx.x < n ? (A.idx[row * A.L + threadIdx.x] == 0) : 1;
    lds_barrier();
    attn_fwd_compute<DH, ROWS, NT>(A, b, t0, n, (uint32_t)A.state[DR4SR_STATE_RNGSTEP]);
}

// Persistent loop over a length-class list, software-pipelined three deep: while sequence i is computed, the rows of sequence i+1
// are in flight to registers, the (cu, rows) words of i+2 and the list entry of i+3 are requested — the dependent chain
// list -> cu -> rows of a sequence never sits in front of its compute phase.
template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_fwd_list(const AttnArgs2& A, const int bid, const int G) {
    constexpr int D = 2 * DH, LD = D + 4, PER = (ROWS * (D / 4) + NT - 1) / NT;
    const int cnt = *A.list_count;
    if (bid >= cnt) return;
    const uint32_t rngstep = (uint32_t)A.state[DR4SR_STATE_RNGSTEP];
    float* Ks = smem;
    float* Vs = Ks + ROWS * LD;
    int* kpad = reinterpret_cast<int*>(Vs + ROWS * LD);
    float4 rk[PER], rv[PER];
    int64_t ridx = 1;                                    // raw item id of key position threadIdx.x (1 = "not PAD" placeholder)
    bool rin = false;
    auto request = [&](const SeqMeta& m) {
        const float* src = A.qkv + (size_t)m.t0 * 3 * D;
        rows_to_regs<D, ROWS, NT, PER>(rk, src + D, 3 * D, m.n);
        rows_to_regs<D, ROWS, NT, PER>(rv, src + 2 * D, 3 * D, m.n);
        rin = threadIdx.x < 64 && (int)threadIdx.x < m.n;
        if (rin) ridx = A.idx[m.row * A.L + threadIdx.x];
    };
    const int z = opaque_zero();
    SeqMeta cur = seq_final(seq_raw(A, list_raw(A, bid, z), z));
    request(cur);
    int k1 = bid + G, k2 = k1 + G;
    bool has1 = k1 < cnt, has2 = k2 < cnt;
    SeqRaw nxt_raw = SeqRaw{0, 0, 0, 0};
    if (has1) nxt_raw = seq_raw(A, list_raw(A, k1, z), z);
    int b2 = has2 ? list_raw(A, k2, z) : 0;
    for (;;) {
        regs_to_rows<D, ROWS, NT, PER>(Ks, LD, rk);
        regs_to_rows<D, ROWS, NT, PER>(Vs, LD, rv);
        if (threadIdx.x < 64) kpad[threadIdx.x] = rin ? (ridx == 0) : 1;
        lds_barrier();
        const SeqMeta nxt = seq_final(nxt_raw);                              // loaded one iteration ago
        if (has1) request(nxt);                                              // rows of i+1
        SeqRaw nn_raw = nxt_raw;
        if (has2) nn_raw = seq_raw(A, b2, z);                                // cu / rows words of i+2
        const int k3 = k2 + G;
        const bool has3 = k3 < cnt;
        const int b3 = has3 ? list_raw(A, k3, z) : 0;                        // list entry of i+3
        attn_fwd_compute<DH, ROWS, NT>(A, cur.b, cur.t0, cur.n, rngstep);
        lds_barrier();                                                       // LDS is reused by the next sequence
        if (!has1) break;
        cur = nxt; nxt_raw = nn_raw; has1 = has2; has2 = has3; b2 = b3; k2 = k3;
    }
}

// LIST = false: workgroup b = sequence b;  true: persistent loop over a length-class list (separate instantiations: the
// pipelined loop's staging registers must not weigh on the one-sequence form)
template <int DH, int ROWS, int NT, bool LIST>
__global__ __launch_bounds__(NT) void k_attn2_fwd(const AttnArgs2 A) {
    if constexpr (!LIST) attn_fwd_seq<DH, ROWS, NT>(A, blockIdx.x);
    else attn_fwd_list<DH, ROWS, NT>(A, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------ backward
// No softmax pass: P = exp(s - m) / sum from the statistics the forward saved, and the row term sum_j P dP = <dctx, ctx>
// (per head; also true under dropout) comes from k_post_bwd.  dQ (phase A, query-tile owners) and dK/dV (phase B, key-tile
// owners) are therefore independent and run CONCURRENTLY on two halves of the workgroup: waves [0, NT/128) do phase A,
// waves [NT/128, NT/64) phase B — half the serial chain of a long sequence, twice the waves per workgroup to hide latency.
// LDS map of the backward: Q, K, dctx rows (+ V rows for the 16-row variant, whose LDS budget allows it: one dependent global
// round trip less inside each phase), the per-row statistics and the key-padding flags
template <int DH, int ROWS>
struct BwdLds {
    static constexpr int D = 2 * DH, LD = D + 4, H = 2;
    static constexpr bool VLDS = ROWS == 16;
    float *Qs, *Ks, *Cs, *Vs, *stat; int* kpad;
    __device__ __forceinline__ BwdLds() {
        Qs = smem; Ks = Qs + ROWS * LD; Cs = Ks + ROWS * LD;
        Vs = Cs + ROWS * LD;
        stat = Vs + (VLDS ? ROWS * LD : 0);                 // [H][ROWS][3]  row max, 1/sum, sum_j P dP
        kpad = reinterpret_cast<int*>(stat + H * ROWS * 3);
    }
};

// everything of the sequence is in LDS (and a barrier has passed)
template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_bwd_compute(const AttnArgs2& A, const int b, const int t0, const int n, const uint32_t rngstep) {
    constexpr int D = 2 * DH, LD = D + 4, H = 2, MT = ROWS / 16, NW = NT / 64;
    constexpr bool VLDS = BwdLds<DH, ROWS>::VLDS;
    const BwdLds<DH, ROWS> S;
    float* Qs = S.Qs; float* Ks = S.Ks; float* Cs = S.Cs; float* Vs = S.Vs; float* stat = S.stat; int* kpad = S.kpad;
    const float* src = A.qkv + (size_t)t0 * 3 * D;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = w & 1;
    // NW = 8: phases on separate wave quads (latency regime);  ROWS = 64 with NW = 4: every wave runs phase A then phase B
    // (throughput regime: 4 workgroups' worth of waves per CU instead of 2, no A/B load imbalance);  short variant: one wave per
    // (phase, head)
    constexpr bool SEQ = (ROWS == 64 && NT == 256) || (ROWS == 16 && NT == 128);
    const int half = (NW == 8 || SEQ) ? (w >> 1) & 1 : 0;
    const bool phaseB = !SEQ && w >= NW / 2;
    const int i16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, rngstep, A.p);
    const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
    const float scale = 1.0f / sqrtf((float)DH);
    const int ntile = (n + 15) >> 4;

    if (SEQ || !phaseB) {
        // ---- phase A: query tiles (transposed orientation: lane i = l&15, j = 4g+r) -> dQ
#pragma unroll
        for (int pass = 0; pass < (MT == 1 ? 1 : 2); ++pass) {
            const int it = pass == 0 ? (half == 0 ? 0 : 1) : (half == 0 ? 3 : 2);
            if (it >= ntile) continue;
            const int i = it * 16 + i16;
            float qf[DH / 4], cf[DH / 4];
            load_frag<DH>(qf, Qs, LD, it * 16, h * DH);
            load_frag<DH>(cf, Cs, LD, it * 16, h * DH);
            const float* sti = stat + (h * ROWS + i) * 3;
            const float mi = sti[0], inv = sti[1], rdot = sti[2];
            const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
            f32x4 ds[MT];
            float4 mkt[MT];
            attn_keep_masks<MT>(mkt, rk, site, ebase, it, lane, dodrop);
#pragma unroll
            for (int jt = 0; jt < MT; ++jt)
                if (jt <= it) {
                    float kf[DH / 4];
                    load_frag<DH>(kf, Ks, LD, jt * 16, h * DH);
                    const f32x4 s = mma_rows<DH>(kf, qf);
                    if constexpr (VLDS) load_frag<DH>(kf, Vs, LD, jt * 16, h * DH);
                    else load_frag_g<DH>(kf, src + 2 * D + h * DH, 3 * D, jt * 16, n);
                    const f32x4 dp = mma_rows<DH>(kf, cf);         // dP~^T[j][i] = sum_d V[j][d] dctx[i][d]
                    const float4 mk = mkt[jt];
                    const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + 4 * g + r;
                        const float p = (j <= i && j < n && !kpad[j]) ? __expf(s[r] * scale - mi) * inv : 0.f;
                        ds[jt][r] = p * (dp[r] * mkv[r] - rdot) * scale;   // dS^T[j][i]
                    }
                }
            // dQ^T[f][i] = sum_j K[j][f] dS^T[j][i]
#pragma unroll
            for (int fb = 0; fb < DH / 16; ++fb) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jt = 0; jt < MT; ++jt)
                    if (jt <= it)
#pragma unroll
                        for (int sidx = 0; sidx < 4; ++sidx)
                            o = mfma16(Ks[(jt * 16 + 4 * g + sidx) * LD + h * DH + fb * 16 + i16], ds[jt][sidx], o);
                if (i < n) st4(A.dqkv + (size_t)(t0 + i) * 3 * D + h * DH + fb * 16 + 4 * g, make_float4(o[0], o[1], o[2], o[3]));
            }
        }
        if (!SEQ) return;
    }
    // ---- phase B: key tiles (natural orientation: lane j = l&15, i = 4g+r) -> dK, dV
#pragma unroll
    for (int pass = 0; pass < (MT == 1 ? 1 : 2); ++pass) {
        const int jt = pass == 0 ? (half == 0 ? 0 : 1) : (half == 0 ? 3 : 2);     // key tile 0 has 4 query tiles, 3 has 1
        if (jt >= ntile) continue;
        const int j = jt * 16 + i16;                       // this lane's key row
        float kf[DH / 4], vf[DH / 4];
        load_frag<DH>(kf, Ks, LD, jt * 16, h * DH);
        if constexpr (VLDS) load_frag<DH>(vf, Vs, LD, jt * 16, h * DH);
        else load_frag_g<DH>(vf, src + 2 * D + h * DH, 3 * D, jt * 16, n);
        const bool jok = j < n && !kpad[j];
        f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
        for (int fb = 0; fb < DH / 16; ++fb) { dk[fb] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[fb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int it = 0; it < MT; ++it) {
            if (it >= jt && it < ntile) {
                float qf[DH / 4];
                load_frag<DH>(qf, Qs, LD, it * 16, h * DH);
                f32x4 s = mma_rows<DH>(qf, kf);            // S[i][j]: rows i = 4g+r (C layout), col j = l&15
                load_frag<DH>(qf, Cs, LD, it * 16, h * DH);
                f32x4 dp = mma_rows<DH>(qf, vf);           // dP~[i][j] = sum_d dctx[i][d] V[j][d]
                f32x4 pt, ds;
                // dropout decisions of this (query tile, key tile): element (i, j) sits in lane j, rows i = 4 g + r — eight lanes would
                // each recompute the Philox call that covers (i, 8 keys).  Instead lane j computes ONE call, that of query row
                // 4 g + (j & 3) and key half (j >> 2) & 1, and the four decisions a lane needs come from its 16-lane row by ds_bpermute
                // (round 3: one call per lane and tile instead of four — drop1 per element was 60 % of this phase's VALU work).
                unsigned m8 = 0xffu;
                if (dodrop)
                    m8 = drop_bits8(rk, site, ((uint64_t)(b * H + h) * 64 + (it * 16 + 4 * g + (i16 & 3))) * 64 + jt * 16 + 8 * ((i16 >> 2) & 1));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = it * 16 + 4 * g + r;
                    float p = 0.f, mkv = 1.f, rd = 0.f;
                    const unsigned mm = __shfl(m8, (lane & 48) | (r + 4 * (i16 >> 3)), 64);
                    if (dodrop) mkv = ((mm >> (i16 & 7)) & 1u) ? rk.scale : 0.f;
                    if (i < n) {
                        const float* st = stat + (h * ROWS + i) * 3;
                        if (jok && j <= i) p = __expf(s[r] * scale - st[0]) * st[1];
                        rd = st[2];
                    }
                    pt[r] = p * mkv;                       // P~[i][j]
                    ds[r] = p * (dp[r] * mkv - rd) * scale;
                }
                // dK^T[f][j] = sum_i Q[i][f] dS[i][j] ;  dV^T[d][j] = sum_i dctx[i][d] P~[i][j]
#pragma unroll
                for (int fb = 0; fb < DH / 16; ++fb)
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx) {
                        const int irow = (it * 16 + 4 * g + sidx) * LD + h * DH + fb * 16 + i16;
                        dk[fb] = mfma16(Qs[irow], ds[sidx], dk[fb]);
                        dv[fb] = mfma16(Cs[irow], pt[sidx], dv[fb]);
                    }
            }
        }
        if (j < n) {
#pragma unroll
            for (int fb = 0; fb < DH / 16; ++fb) {
                float* base = A.dqkv + (size_t)(t0 + j) * 3 * D + h * DH + fb * 16 + 4 * g;
                st4(base + D, make_float4(dk[fb][0], dk[fb][1], dk[fb][2], dk[fb][3]));
                st4(base + 2 * D, make_float4(dv[fb][0], dv[fb][1], dv[fb][2], dv[fb][3]));
            }
        }
    }
}

template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_bwd_seq(const AttnArgs2& A, const int b) {
    constexpr int D = 2 * DH, LD = D + 4, H = 2;
    const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
    if (n <= 0) return;
    const BwdLds<DH, ROWS> S;
    const int64_t row = A.rows ? A.rows[b] : b;
    const float* src = A.qkv + (size_t)t0 * 3 * D;
    stage_rows<D, ROWS, NT>(S.Qs, LD, src, 3 * D, n);
    stage_rows<D, ROWS, NT>(S.Ks, LD, src + D, 3 * D, n);
    stage_rows<D, ROWS, NT>(S.Cs, LD, A.dctx + (size_t)t0 * D, D, n);
    if constexpr (BwdLds<DH, ROWS>::VLDS) stage_rows<D, ROWS, NT>(S.Vs, LD, src + 2 * D, 3 * D, n);
    if (threadIdx.x < 64) S.kpad[threadIdx.x] = threadIdx.x < n ? (A.idx[row * A.L + threadIdx.x] == 0) : 1;
    for (int i = threadIdx.x; i < H * ROWS; i += NT) {      // (h, row) -> m, 1/sum, rowdot
        const int hh = i / ROWS, r = i % ROWS;
        float m = 0.f, inv = 0.f, rdot = 0.f;
        if (r < n) {
            const float* st = A.stat + ((size_t)(t0 + r) * H + hh) * 2;
            m = st[0]; inv = st[1]; rdot = A.rd[(size_t)(t0 + r) * H + hh];
        }
        S.stat[i * 3] = m; S.stat[i * 3 + 1] = inv; S.stat[i * 3 + 2] = rdot;
    }
    lds_barrier();
    attn_bwd_compute<DH, ROWS, NT>(A, b, t0, n, (uint32_t)A.state[DR4SR_STATE_RNGSTEP]);
}

// persistent list loop, software-pipelined like attn_fwd_list (rows and statistics of sequence i+1 in flight during sequence i)
template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_bwd_list(const AttnArgs2& A, const int bid, const int G) {
    constexpr int D = 2 * DH, LD = D + 4, H = 2, PER = (ROWS * (D / 4) + NT - 1) / NT;
    constexpr bool VLDS = BwdLds<DH, ROWS>::VLDS;
    static_assert(H * ROWS <= NT, "one statistics triple per thread");
    const int cnt = *A.list_count;
    if (bid >= cnt) return;
    const uint32_t rngstep = (uint32_t)A.state[DR4SR_STATE_RNGSTEP];
    const BwdLds<DH, ROWS> S;
    float4 rq[PER], rk[PER], rc[PER], rv[VLDS ? PER : 1];
    float2 sst = make_float2(0.f, 0.f);
    float sr = 0.f;
    int64_t ridx = 1;
    bool rin = false;
    auto request = [&](const SeqMeta& m) {
        const float* src = A.qkv + (size_t)m.t0 * 3 * D;
        rows_to_regs<D, ROWS, NT, PER>(rq, src, 3 * D, m.n);
        rows_to_regs<D, ROWS, NT, PER>(rk, src + D, 3 * D, m.n);
        rows_to_regs<D, ROWS, NT, PER>(rc, A.dctx + (size_t)m.t0 * D, D, m.n);
        if constexpr (VLDS) rows_to_regs<D, ROWS, NT, PER>(rv, src + 2 * D, 3 * D, m.n);
        rin = threadIdx.x < 64 && (int)threadIdx.x < m.n;
        if (rin) ridx = A.idx[m.row * A.L + threadIdx.x];
        sst = make_float2(0.f, 0.f); sr = 0.f;
        const int hh = threadIdx.x / ROWS, r = threadIdx.x % ROWS;
        if ((int)threadIdx.x < H * ROWS && r < m.n) {
            sst = *reinterpret_cast<const float2*>(A.stat + ((size_t)(m.t0 + r) * H + hh) * 2);
            sr = A.rd[(size_t)(m.t0 + r) * H + hh];
        }
    };
    const int z = opaque_zero();
    SeqMeta cur = seq_final(seq_raw(A, list_raw(A, bid, z), z));
    request(cur);
    int k1 = bid + G, k2 = k1 + G;
    bool has1 = k1 < cnt, has2 = k2 < cnt;
    SeqRaw nxt_raw = SeqRaw{0, 0, 0, 0};
    if (has1) nxt_raw = seq_raw(A, list_raw(A, k1, z), z);
    int b2 = has2 ? list_raw(A, k2, z) : 0;
    for (;;) {
        regs_to_rows<D, ROWS, NT, PER>(S.Qs, LD, rq);
        regs_to_rows<D, ROWS, NT, PER>(S.Ks, LD, rk);
        regs_to_rows<D, ROWS, NT, PER>(S.Cs, LD, rc);
        if constexpr (VLDS) regs_to_rows<D, ROWS, NT, PER>(S.Vs, LD, rv);
        if (threadIdx.x < 64) S.kpad[threadIdx.x] = rin ? (ridx == 0) : 1;
        if ((int)threadIdx.x < H * ROWS) { S.stat[threadIdx.x * 3] = sst.x; S.stat[threadIdx.x * 3 + 1] = sst.y; S.stat[threadIdx.x * 3 + 2] = sr; }
        lds_barrier();
        const SeqMeta nxt = seq_final(nxt_raw);
        if (has1) request(nxt);
        SeqRaw nn_raw = nxt_raw;
        if (has2) nn_raw = seq_raw(A, b2, z);
        const int k3 = k2 + G;
        const bool has3 = k3 < cnt;
        const int b3 = has3 ? list_raw(A, k3, z) : 0;
        attn_bwd_compute<DH, ROWS, NT>(A, cur.b, cur.t0, cur.n, rngstep);
        lds_barrier();
        if (!has1) break;
        cur = nxt; nxt_raw = nn_raw; has1 = has2; has2 = has3; b2 = b3; k2 = k3;
    }
}

template <int DH, int ROWS, int NT, bool LIST>
__global__ __launch_bounds__(NT) void k_attn2_bwd(const AttnArgs2 A) {
    if constexpr (!LIST) attn_bwd_seq<DH, ROWS, NT>(A, blockIdx.x);
    else if constexpr (ROWS == 16) attn_bwd_list<DH, ROWS, NT>(A, blockIdx.x, gridDim.x);
    else {                                   // 64-row sequences: the staging registers of the pipelined loop would cost a wave per SIMD
        const int cnt = *A.list_count;
        for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
            attn_bwd_seq<DH, ROWS, NT>(A, A.list[k]);
            lds_barrier();                   // LDS is reused by the next sequence
        }
    }
}

// ------------------------------------------------------------------------------------------------ the two short classes in ONE launch
#include "attn_tiny_body.h"
// At scale an attention call was three launches (1..8 tokens: VALU; 9..16: 16-row MFMA list; longer: 64-row list) whose durations add
// (toys B = 8192: 12 + 7 + 12 us forward, 19 + 10 + 27 us backward), each bound by the latency chains of the sequences in flight,
// not by issue slots.  Graph-level parallel branches cost more in fork / join than they win; here the first gs blocks of
// one launch run the persistent 16-row list loop over gs workgroups and the rest are the tiny-class blocks (one wave each: the other
// waves of such a block exit at once), so the two classes share the CUs without a launch boundary between them.  The 64-row class
// keeps its own launch (its 54 KB of LDS per workgroup would cap the residency of the small ones).
template <int DH>
__global__ __launch_bounds__(128, DH == 32 ? 4 : 2) void k_attn_small_fwd(const AttnArgs2 S, const AttnArgs2 Tn, const int gs) {
    if ((int)blockIdx.x < gs) attn_fwd_list<DH, 16, 128>(S, blockIdx.x, gs);
    else if (threadIdx.x < tiny::NT) tiny::fwd_body<DH>(Tn, blockIdx.x - gs);
}
constexpr int SHORT_BWD_NT_ = 256;
template <int DH>
__global__ __launch_bounds__(SHORT_BWD_NT_) void k_attn_small_bwd(const AttnArgs2 S, const AttnArgs2 Tn, const int gs) {
    if ((int)blockIdx.x < gs) attn_bwd_list<DH, 16, SHORT_BWD_NT_>(S, blockIdx.x, gs);
    else if (threadIdx.x < tiny::NT) tiny::bwd_body<DH>(Tn, blockIdx.x - gs);
}

// grid = worst case (every sequence of the batch tiny); workgroups beyond the list's device-side count exit at once
int launch_attn_tiny(const AttnArgs2& A, int DH, int B, bool bwd, hipStream_t s) {
    using namespace tiny;
    if (!A.desc || !A.list_count) return DR4SR_E_ARG;
    dim3 grid((B + SPB - 1) / SPB), blk(NT);
    if (DH == 32) {
        const size_t lds = TinyLds<32>::bytes(bwd);
        if (bwd) { big_lds(k_attn_tiny_bwd<32>, lds); hipLaunchKernelGGL(k_attn_tiny_bwd<32>, grid, blk, lds, s, A); }
        else hipLaunchKernelGGL(k_attn_tiny_fwd<32>, grid, blk, lds, s, A);
    } else if (DH == 64) {
        const size_t lds = TinyLds<64>::bytes(bwd);
        if (bwd) { big_lds(k_attn_tiny_bwd<64>, lds); hipLaunchKernelGGL(k_attn_tiny_bwd<64>, grid, blk, lds, s, A); }
        else { big_lds(k_attn_tiny_fwd<64>, lds); hipLaunchKernelGGL(k_attn_tiny_fwd<64>, grid, blk, lds, s, A); }
    } else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ launchers
static AttnArgs2 make_args2(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training) {
    AttnArgs2 A;
    const LayerWs& lw = ws.layer[layer];
    A.qkv = lw.qkv; A.ctx = lw.ctx; A.dctx = ws.dctx; A.dqkv = lw.dqkv;
    A.idx = p->in_item_id; A.rows = p->rows; A.cu = ws.cu;
    A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.layer = layer; A.training = training; A.L = p->L;
    A.list = nullptr; A.list_count = nullptr; A.desc = nullptr; A.stat = lw.attn_st; A.rd = ws.attn_rd;
    A.tok = attn_wave_on(p, ws) ? ws.tok : nullptr; A.keep = lw.attn_keep;
    return A;
}

// Large batches (same threshold as tile_rows): two persistent launches over k_prep's length-class lists instead of one
// workgroup per sequence with worst-case LDS.  seq_class = [n_short, n_long, n_tiny, - | tiny_desc[B] int4 | short_list[B] | long_list[B] | tiny_list[B]].
static bool split_by_length(const Workspace& ws) { return ws.attn_split && !DR4SR_ENV("DR4SR_ATTN_NOSPLIT"); }

// short sequences, backward: one wave per head runs phase A then phase B (2 waves per sequence, twice the sequences per CU of the
// 4-wave form — the kernel is bound by how many sequences are in flight, not by issue slots)
constexpr int SHORT_BWD_NT = 256;

// workgroups of `kernel` that fit one CU (fallback: the hand-computed figure)
static int resident_per_cu(const void* kernel, int threads, size_t lds, int fallback) {
    if (DR4SR_XENV("DR4SR_ATTN_GRID_FIXED")) return fallback;
    if (lds > 48 * 1024) big_lds_impl(kernel, lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fallback; }
    return n;
}

template <int DH>
static int attn2_launch(const dr4sr_sasrec_plan* p, const Workspace& ws, AttnArgs2 A, bool bwd, hipStream_t s) {
    const int D = p->D, B = p->B;
    auto lds_of = [&](int rows) { return sizeof(float) * ((bwd ? (rows == 16 ? 4 : 3) : 2) * rows * (D + 4) + (bwd ? 2 * rows * 3 : 0) + 64); };
    if (!split_by_length(ws)) {
        const size_t lds = lds_of(64);
        // backward: 8 waves per sequence for both head widths (head_dim 64 needed 187 VGPRs and ran with 4 waves until the dropout decisions
        // came from one Philox call per lane and tile: 118 now; d = 128, B = 256: 2 x 19.4 -> 2 x 14.7 us).  DR4SR_ATTN_BWD_4WAVE: cross-check
        const bool four = DR4SR_XENV("DR4SR_ATTN_BWD_4WAVE") != nullptr;
        if (bwd && !four) { big_lds(k_attn2_bwd<DH, 64, 512, false>, lds); hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 512, false>), dim3(B), dim3(512), lds, s, A); }
        else if (bwd) { big_lds(k_attn2_bwd<DH, 64, 256, false>, lds); hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 256, false>), dim3(B), dim3(256), lds, s, A); }
        else { big_lds(k_attn2_fwd<DH, 64, 256, false>, lds); hipLaunchKernelGGL((k_attn2_fwd<DH, 64, 256, false>), dim3(B), dim3(256), lds, s, A); }
        return DR4SR_LAUNCH_CHECK();
    }
    if (A.tok) return launch_attn_wave(A, DH, ws.Tmax, p->expected_tokens, bwd, s);     // round 6: one wave per (token tile, head[, phase]), no lists (attn_wave.hip)
    // persistent grids = what is really co-resident (registers, LDS and wave slots, as the runtime computes it): a larger grid
    // runs in two rounds with an unbalanced tail, a smaller one leaves latency-hiding slots empty
    static int per_cu[2][2];                                                 // [bwd][long], per head width
    static int per_cu_gen[2] = {-1, -1};                                     // DR4SR_ATTN_GRID_FIXED may have changed (dr4sr_reload_env)
    if (per_cu_gen[bwd] != dr4sr_env_generation()) { per_cu[bwd][0] = 0; per_cu_gen[bwd] = dr4sr_env_generation(); }
    if (!per_cu[bwd][0]) {
        per_cu[bwd][1] = bwd ? resident_per_cu((const void*)k_attn2_bwd<DH, 64, 256, true>, 256, lds_of(64), 2)
                             : resident_per_cu((const void*)k_attn2_fwd<DH, 64, 256, true>, 256, lds_of(64), 3);
        per_cu[bwd][0] = bwd ? resident_per_cu((const void*)k_attn2_bwd<DH, 16, SHORT_BWD_NT, true>, SHORT_BWD_NT, lds_of(16), 4)
                             : resident_per_cu((const void*)k_attn2_fwd<DH, 16, 128, true>, 128, lds_of(16), 8);
        if (DR4SR_XENV("DR4SR_ATTN_GRID_PRINT")) fprintf(stderr, "attention grids (DH %d, %s): %d short / %d long workgroups per CU\n", DH, bwd ? "bwd" : "fwd", per_cu[bwd][0], per_cu[bwd][1]);
    }
    const int per_cu_s = per_cu[bwd][0], per_cu_l = per_cu[bwd][1];
    const int gs = B < 256 * per_cu_s ? B : 256 * per_cu_s;
    const int gl = B < 256 * per_cu_l ? B : 256 * per_cu_l;
    AttnArgs2 S = A, Lg = A, Tn = A;
    S.list = ws.seq_class + 4 + 4 * B; S.list_count = ws.seq_class;
    Lg.list = ws.seq_class + 4 + 5 * B; Lg.list_count = ws.seq_class + 1;
    Tn.list = ws.seq_class + 4 + 6 * B; Tn.list_count = ws.seq_class + 2; Tn.desc = ws.seq_class + 4;
    const size_t lds_s = lds_of(16), lds_l = lds_of(64);
    // default: the 1..8-token and 9..16-token classes as ONE launch (k_attn_small_*), then the 64-row list.  DR4SR_ATTN_NOMERGE: one
    // launch per class (cross-check; cached until dr4sr_reload_env() like every switch)
    // The length classes are disjoint sets of sequences: their launches are independent and go to side streams (parallel branches of a
    // captured step graph): the class kernels are latency-bound at 1-7 % MFMA utilisation, side by side they take the longest one's time
    const StepFork& fk = step_fork();
    if (!DR4SR_ENV("DR4SR_ATTN_NOTINY") && !DR4SR_XENV("DR4SR_ATTN_NOMERGE")) {
        const int dv = DR4SR_XENV("DR4SR_ATTN_SMALL_DIV") ? atoi(DR4SR_XENV("DR4SR_ATTN_SMALL_DIV")) : 2;      // share of the residency left to the tiny blocks (tuning)
        const int gsm = gs / (dv > 0 ? dv : 1) > 0 ? gs / (dv > 0 ? dv : 1) : 1;
        const size_t lds_t = tiny::TinyLds<DH>::bytes(bwd), lds_m = lds_s > lds_t ? lds_s : lds_t;
        dim3 grid(gsm + (B + tiny::SPB - 1) / tiny::SPB);
        const bool merge_bwd = DR4SR_XENV("DR4SR_ATTN_MERGE_BWD") != nullptr;
        if (bwd && !merge_bwd) {
            // backward: NOT merged by default — the tiny class needs 192 VGPRs, the 16-row list 120; at the merged kernel's 208 the list's
            // workgroups fill the register file two per CU and the tiny blocks queue behind them (32.7 us against 19.2 + 9.8)
            int rc = fk.fork(s, 0, 2);
            if (rc) return rc;
            rc = launch_attn_tiny(Tn, DH, B, true, fk.side(s, 0));
            if (rc) return rc;
            hipLaunchKernelGGL((k_attn2_bwd<DH, 16, SHORT_BWD_NT, true>), dim3(gs), dim3(SHORT_BWD_NT), lds_s, fk.side(s, 1), S);
            big_lds(k_attn2_bwd<DH, 64, 256, true>, lds_l);
            hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 256, true>), dim3(gl), dim3(256), lds_l, s, Lg);
            rc = DR4SR_LAUNCH_CHECK();
            if (rc) return rc;
            return fk.join(s, 0, 2);
        } else if (bwd) {
            int rc = fk.fork(s, 0, 1);
            if (rc) return rc;
            big_lds(k_attn_small_bwd<DH>, lds_m);
            hipLaunchKernelGGL((k_attn_small_bwd<DH>), grid, dim3(SHORT_BWD_NT), lds_m, fk.side(s, 0), S, Tn, gsm);
            big_lds(k_attn2_bwd<DH, 64, 256, true>, lds_l);
            hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 256, true>), dim3(gl), dim3(256), lds_l, s, Lg);
            rc = DR4SR_LAUNCH_CHECK();
            if (rc) return rc;
            return fk.join(s, 0, 1);
        } else {
            int rc = fk.fork(s, 0, 1);
            if (rc) return rc;
            big_lds(k_attn_small_fwd<DH>, lds_m);
            hipLaunchKernelGGL((k_attn_small_fwd<DH>), grid, dim3(128), lds_m, fk.side(s, 0), S, Tn, gsm);
            big_lds(k_attn2_fwd<DH, 64, 256, true>, lds_l);
            hipLaunchKernelGGL((k_attn2_fwd<DH, 64, 256, true>), dim3(gl), dim3(256), lds_l, s, Lg);
            rc = DR4SR_LAUNCH_CHECK();
            if (rc) return rc;
            return fk.join(s, 0, 1);
        }
    }
    // third class, 1..8 tokens: VALU kernels (attn_tiny_body.h).  DR4SR_ATTN_NOTINY (cross-check): the same list through the 16-row MFMA kernels
    if (!DR4SR_ENV("DR4SR_ATTN_NOTINY")) {
        const int rc = launch_attn_tiny(Tn, DH, B, bwd, s);
        if (rc) return rc;
    } else if (bwd) hipLaunchKernelGGL((k_attn2_bwd<DH, 16, SHORT_BWD_NT, true>), dim3(gs), dim3(SHORT_BWD_NT), lds_s, s, Tn);
    else hipLaunchKernelGGL((k_attn2_fwd<DH, 16, 128, true>), dim3(gs), dim3(128), lds_s, s, Tn);
    if (bwd) {
        hipLaunchKernelGGL((k_attn2_bwd<DH, 16, SHORT_BWD_NT, true>), dim3(gs), dim3(SHORT_BWD_NT), lds_s, s, S);
        big_lds(k_attn2_bwd<DH, 64, 256, true>, lds_l);
        hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 256, true>), dim3(gl), dim3(256), lds_l, s, Lg);
    } else {
        hipLaunchKernelGGL((k_attn2_fwd<DH, 16, 128, true>), dim3(gs), dim3(128), lds_s, s, S);
        big_lds(k_attn2_fwd<DH, 64, 256, true>, lds_l);
        hipLaunchKernelGGL((k_attn2_fwd<DH, 64, 256, true>), dim3(gl), dim3(256), lds_l, s, Lg);
    }
    return DR4SR_LAUNCH_CHECK();
}

int launch_attn2_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    const AttnArgs2 A = make_args2(p, ws, layer, training);
    if (p->D == 64) return attn2_launch<32>(p, ws, A, false, s);
    if (p->D == 128) return attn2_launch<64>(p, ws, A, false, s);
    return DR4SR_E_SHAPE;
}

int launch_attn2_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    const AttnArgs2 A = make_args2(p, ws, layer, training);
    if (p->D == 64) return attn2_launch<32>(p, ws, A, true, s);
    if (p->D == 128) return attn2_launch<64>(p, ws, A, true, s);
    return DR4SR_E_SHAPE;
}
