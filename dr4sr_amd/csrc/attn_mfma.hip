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

extern __shared__ __attribute__((aligned(16))) float smem[];

struct AttnArgs2 {
    const float* qkv; float* ctx;
    const float* dctx; float* dqkv;
    const int64_t* idx; const int64_t* rows; const int* cu;
    const int* state; uint64_t seed; float p; int layer; int training; int L;
    float* stat;                 // [T][H][2]  softmax row max, 1/row sum: written by the forward, read by the backward
    const float* rd;             // [T][H]     sum_j P dP = <dctx, ctx> per head, from the epilogue of k_post_bwd
    // large batches: sequences are split by length class (k_prep's seq_class lists) into a short kernel (n <= 16: 16 LDS rows,
    // 2 waves, ~9 workgroups per CU) and a long kernel, each a persistent loop over its list.  list == NULL: block b = sequence b.
    const int* list; const int* list_count;
};

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

// ------------------------------------------------------------------------------------------------ forward
template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_fwd_seq(const AttnArgs2& A, const int b) {
    constexpr int D = 2 * DH, LD = D + 4, H = 2, MT = ROWS / 16;
    const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
    if (n <= 0) return;
    float* Ks = smem;                    // [ROWS][LD]   (Q fragments come straight from global: used once per query tile)
    float* Vs = Ks + ROWS * LD;
    int* kpad = reinterpret_cast<int*>(Vs + ROWS * LD);      // [64]
    const int64_t row = A.rows ? A.rows[b] : b;
    const float* src = A.qkv + (size_t)t0 * 3 * D;
    stage_rows<D, ROWS, NT>(Ks, LD, src + D, 3 * D, n);
    stage_rows<D, ROWS, NT>(Vs, LD, src + 2 * D, 3 * D, n);
    if (threadIdx.x < 64) kpad[threadIdx.x] = threadIdx.x < n ? (A.idx[row * A.L + threadIdx.x] == 0) : 1;
    lds_barrier();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = w & 1, half = w >> 1;
    const int i16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
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
                for (int r = 0; r < 4; ++r) { const float e = expf(s[jt][r] - m); s[jt][r] = e; sum += e; }
        sum = xgroup_sum(sum);
        const float inv = 1.0f / sum;
        if (g == 0 && i < n) { float* st = A.stat + ((size_t)(t0 + i) * H + h) * 2; st[0] = m; st[1] = inv; }
        const uint64_t ebase = ((uint64_t)(b * H + h) * 64 + i) * 64;
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
            if (jt <= it) {
                float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
                if (dodrop) mk = drop4(rk, site, ebase + jt * 16 + 4 * g);
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
__global__ __launch_bounds__(NT) void k_attn2_fwd(const AttnArgs2 A) {
    if (!A.list) { attn_fwd_seq<DH, ROWS, NT>(A, blockIdx.x); return; }
    const int cnt = *A.list_count;
    for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
        attn_fwd_seq<DH, ROWS, NT>(A, A.list[k]);
        __syncthreads();                                   // LDS is reused by the next sequence
    }
}

// ------------------------------------------------------------------------------------------------ backward
// No softmax pass: P = exp(s - m) / sum from the statistics the forward saved, and the row term sum_j P dP = <dctx, ctx>
// (per head; also true under dropout) comes from k_post_bwd.  dQ (phase A, query-tile owners) and dK/dV (phase B, key-tile
// owners) are therefore independent and run CONCURRENTLY on two halves of the workgroup: waves [0, NT/128) do phase A,
// waves [NT/128, NT/64) phase B — half the serial chain of a long sequence, twice the waves per workgroup to hide latency.
template <int DH, int ROWS, int NT>
__device__ __forceinline__ void attn_bwd_seq(const AttnArgs2& A, const int b) {
    constexpr int D = 2 * DH, LD = D + 4, H = 2, MT = ROWS / 16, NW = NT / 64;
    const int t0 = A.cu[b], n = A.cu[b + 1] - t0;
    if (n <= 0) return;
    float* Qs = smem;                                       // (V is only ever used as a row fragment: read from global)
    float* Ks = Qs + ROWS * LD;
    float* Cs = Ks + ROWS * LD;                             // dctx rows
    float* stat = Cs + ROWS * LD;                           // [H][ROWS][3]  row max, 1/sum, sum_j P dP
    int* kpad = reinterpret_cast<int*>(stat + H * ROWS * 3);
    const int64_t row = A.rows ? A.rows[b] : b;
    const float* src = A.qkv + (size_t)t0 * 3 * D;
    stage_rows<D, ROWS, NT>(Qs, LD, src, 3 * D, n);
    stage_rows<D, ROWS, NT>(Ks, LD, src + D, 3 * D, n);
    stage_rows<D, ROWS, NT>(Cs, LD, A.dctx + (size_t)t0 * D, D, n);
    if (threadIdx.x < 64) kpad[threadIdx.x] = threadIdx.x < n ? (A.idx[row * A.L + threadIdx.x] == 0) : 1;
    for (int i = threadIdx.x; i < H * ROWS; i += NT) {      // (h, row) -> m, 1/sum, rowdot
        const int hh = i / ROWS, r = i % ROWS;
        float m = 0.f, inv = 0.f, rdot = 0.f;
        if (r < n) {
            const float* st = A.stat + ((size_t)(t0 + r) * H + hh) * 2;
            m = st[0]; inv = st[1]; rdot = A.rd[(size_t)(t0 + r) * H + hh];
        }
        stat[i * 3] = m; stat[i * 3 + 1] = inv; stat[i * 3 + 2] = rdot;
    }
    lds_barrier();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = w & 1;
    // NW = 8: phases on separate wave quads (latency regime);  ROWS = 64 with NW = 4: every wave runs phase A then phase B
    // (throughput regime: 4 workgroups' worth of waves per CU instead of 2, no A/B load imbalance);  short variant: one wave per
    // (phase, head)
    constexpr bool SEQ = ROWS == 64 && NT == 256;
    const int half = (NW == 8 || SEQ) ? (w >> 1) & 1 : 0;
    const bool phaseB = !SEQ && w >= NW / 2;
    const int i16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
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
#pragma unroll
            for (int jt = 0; jt < MT; ++jt)
                if (jt <= it) {
                    float kf[DH / 4];
                    load_frag<DH>(kf, Ks, LD, jt * 16, h * DH);
                    const f32x4 s = mma_rows<DH>(kf, qf);
                    load_frag_g<DH>(kf, src + 2 * D + h * DH, 3 * D, jt * 16, n);
                    const f32x4 dp = mma_rows<DH>(kf, cf);         // dP~^T[j][i] = sum_d V[j][d] dctx[i][d]
                    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (dodrop) mk = drop4(rk, site, ebase + jt * 16 + 4 * g);
                    const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + 4 * g + r;
                        const float p = (j <= i && j < n && !kpad[j]) ? expf(s[r] * scale - mi) * inv : 0.f;
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
        load_frag_g<DH>(vf, src + 2 * D + h * DH, 3 * D, jt * 16, n);
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
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = it * 16 + 4 * g + r;
                    float p = 0.f, mkv = 1.f, rd = 0.f;
                    if (i < n) {
                        const float* st = stat + (h * ROWS + i) * 3;
                        if (jok && j <= i) p = expf(s[r] * scale - st[0]) * st[1];
                        rd = st[2];
                        if (dodrop) mkv = drop1(rk, site, ((uint64_t)(b * H + h) * 64 + i) * 64 + j);
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
__global__ __launch_bounds__(NT) void k_attn2_bwd(const AttnArgs2 A) {
    if (!A.list) { attn_bwd_seq<DH, ROWS, NT>(A, blockIdx.x); return; }
    const int cnt = *A.list_count;
    for (int k = blockIdx.x; k < cnt; k += gridDim.x) {
        attn_bwd_seq<DH, ROWS, NT>(A, A.list[k]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static AttnArgs2 make_args2(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training) {
    AttnArgs2 A;
    const LayerWs& lw = ws.layer[layer];
    A.qkv = lw.qkv; A.ctx = lw.ctx; A.dctx = ws.dctx; A.dqkv = lw.dqkv;
    A.idx = p->in_item_id; A.rows = p->rows; A.cu = ws.cu;
    A.state = p->state; A.seed = p->seed; A.p = p->p_drop; A.layer = layer; A.training = training; A.L = p->L;
    A.list = nullptr; A.list_count = nullptr; A.stat = lw.attn_st; A.rd = ws.attn_rd;
    return A;
}

// Large batches (same threshold as tile_rows): two persistent launches over k_prep's length-class lists instead of one
// workgroup per sequence with worst-case LDS.  seq_class = [n_short, n_long, short_list[B], long_list[B]].
static bool split_by_length(const Workspace& ws) { return ws.Tmax > 16384 && !getenv("DR4SR_ATTN_NOSPLIT"); }

template <int DH>
static int attn2_launch(const dr4sr_sasrec_plan* p, const Workspace& ws, AttnArgs2 A, bool bwd, hipStream_t s) {
    const int D = p->D, B = p->B;
    auto lds_of = [&](int rows) { return sizeof(float) * ((bwd ? 3 : 2) * rows * (D + 4) + (bwd ? 2 * rows * 3 : 0) + 64); };
    if (!split_by_length(ws)) {
        const size_t lds = lds_of(64);
        if (bwd && DH == 32) { big_lds(k_attn2_bwd<DH, 64, 512>, lds); hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 512>), dim3(B), dim3(512), lds, s, A); }
        else if (bwd) {                              // head_dim 64: the 8-wave variant would spill (256-VGPR cap at 512 threads)
            big_lds(k_attn2_bwd<DH, 64, 256>, lds); hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 256>), dim3(B), dim3(256), lds, s, A);
        }
        else { big_lds(k_attn2_fwd<DH, 64, 256>, lds); hipLaunchKernelGGL((k_attn2_fwd<DH, 64, 256>), dim3(B), dim3(256), lds, s, A); }
        return DR4SR_LAUNCH_CHECK();
    }
    const int gs = B < 256 * 8 ? B : 256 * 8;                                // persistent grids: workgroups per CU that fit by LDS
    const int glmax = bwd ? 256 * 2 : 256 * 3, gl = B < glmax ? B : glmax;
    AttnArgs2 S = A, Lg = A;
    S.list = ws.seq_class + 2; S.list_count = ws.seq_class;
    Lg.list = ws.seq_class + 2 + B; Lg.list_count = ws.seq_class + 1;
    const size_t lds_s = lds_of(16), lds_l = lds_of(64);
    if (bwd) {
        hipLaunchKernelGGL((k_attn2_bwd<DH, 16, 256>), dim3(gs), dim3(256), lds_s, s, S);
        big_lds(k_attn2_bwd<DH, 64, 256>, lds_l);
        hipLaunchKernelGGL((k_attn2_bwd<DH, 64, 256>), dim3(gl), dim3(256), lds_l, s, Lg);
    } else {
        hipLaunchKernelGGL((k_attn2_fwd<DH, 16, 128>), dim3(gs), dim3(128), lds_s, s, S);
        big_lds(k_attn2_fwd<DH, 64, 256>, lds_l);
        hipLaunchKernelGGL((k_attn2_fwd<DH, 64, 256>), dim3(gl), dim3(256), lds_l, s, Lg);
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
