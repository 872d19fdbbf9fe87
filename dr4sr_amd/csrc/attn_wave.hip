// attn_wave.hip — causal 2-head self-attention over the PACKED TOKEN STREAM, one wave per (16-token tile, head[, phase]): the at-scale
// attention of short-sequence plans as ONE launch per layer and direction (round 6).
//
// What it replaces.  At scale the attention of a layer was five launches over k_prep's length-class lists (attn_mfma.hip: 1..8-token VALU
// class + 9..16-token MFMA list forward in one launch, the 64-row list in a second; three launches backward): 26 + 51 us per layer of a
// 511 us step at toys B = 8 192 (profiles/round5_kernels_sasrec_B8192.txt), each launch a latency chain (list entry -> cu / rows ->
// K | V rows -> LDS -> barrier -> compute) on a grid that holds one class.  Round 4's window form (attn_tile_sa.hip) put every tile through
// a 256-thread workgroup with an 80-row LDS window and was slower than the lists.  This form has no lists, no LDS and no barrier:
//   * the wave that owns query tile `it` (tokens [16 it, 16 it + 16) of the packed stream, whatever sequences they belong to) reads the
//     per-token words the embedding stage wrote (Workspace::tok: {first token of the sequence, slot | length << 20 | PAD << 30}),
//     finds the earliest key tile any of its queries needs (a wave-uniform minimum: 1 or 2 tiles for the short sequences that make up
//     95 % of a toys batch, up to 5 for a 50-token sequence) and walks exactly those tiles;
//   * "same sequence, causal, not PAD" is one comparison chain per score: key token tk belongs to query tq's sequence iff
//     s0 <= tk, and causality is tk <= tq — the token order of the packed stream IS the position order inside a sequence;
//   * operands go from global memory straight into MFMA fragments (attn_mfma.hip's transposed orientation: S^T = K Q^T, so the softmax
//     output is the next product's B operand as it stands); rows of other sequences are multiplied by probabilities that are exactly 0;
//   * backward: phase A (a wave per query tile and head) recomputes P^T and writes dQ; phase B (a wave per KEY tile and head) recomputes
//     P for the query tiles [it, last tile any of its keys' sequences reaches] and writes dK | dV — every dqkv row has one writer, no
//     atomics, bit-reproducible (the deterministic mode uses the same launch).
// Arithmetic, saved statistics {row max, 1 / row sum} and dropout elements ((slot H + h) 64 + i) 64 + j are attn_mfma.hip's: the two forms
// are interchangeable per launch and the tests hold them against each other and the oracle.
//
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer as configured at /root/reference model/sasrec.py:21-34
// and called at :65-68: attn_mask = triu(ones, 1) (:58), key_padding_mask = (idx == 0) (:48), scale 1 / sqrt(head_dim), dropout on the
// probabilities.
#include "common.h"
#include "kernels.h"
#include "attn_args.h"
#include "attn_wave_body.h"

namespace {
using namespace awv;

// ------------------------------------------------------------------------------------------------ forward
// 256 threads = 2 query tiles x 2 heads.  Every load whose address does not depend on loaded data — the token words, the Q rows, the K | V
// rows of the tile itself and of the tile in front of it (all a short sequence can need) — is requested before anything is consumed: one
// round trip for 95 % of the tiles; only the tiles of long sequences take the dependent path (key tiles 2 .. 4 back).
template <int DH>
__device__ __forceinline__ void fwd_tile(const int tid, const AttnArgs2& A, const int T, const int it, float* __restrict__ vt0, float* __restrict__ vt1) {
    constexpr int D = 2 * DH, H = 2;
    const int lane = tid & 63, w = tid >> 6, h = w & 1, i16 = lane & 15, g = lane >> 4;
    const int t0 = 16 * it;
    const int tq = t0 + i16;
    const bool qv = tq < T, has1 = it > 0;
    const float* __restrict__ qkv = A.qkv;
    const int2 wq_raw = tok_raw(A.tok, tq, T), w1_raw = tok_raw(A.tok, tq - 16, T);
    float qf[DH / 4], kf0[DH / 4], vf0[DH / 4], kf1[DH / 4], vf1[DH / 4];
    frag_rows<DH>(lane, qf, qkv, 3 * D, t0, h * DH, T);
    frag_rows<DH>(lane, kf0, qkv, 3 * D, t0, D + h * DH, T);
    frag_rows<DH>(lane, vf0, qkv, 3 * D, t0, 2 * D + h * DH, T);
    frag_rows<DH>(lane, kf1, qkv, 3 * D, has1 ? t0 - 16 : t0, D + h * DH, has1 ? T : 0);
    frag_rows<DH>(lane, vf1, qkv, 3 * D, has1 ? t0 - 16 : t0, 2 * D + h * DH, has1 ? T : 0);
    const int2 wq = tok_fix(wq_raw, tq, T);
    const int w1 = has1 ? tok_fix(w1_raw, tq - 16, T).y : 0;          // (has1 is wave-uniform)
    const int s0 = wq.x, nq = (wq.y >> 20) & 0x3ff, bq = wq.y & 0xfffff;
    const int lo = __builtin_amdgcn_readfirstlane(max(min16(qv ? (s0 >> 4) : it), max(it - (MT - 1), 0)));
    const int nk = it - lo + 1;
    const bool need_hi = __ballot(nq > 32) != 0ull;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
    const float scale = 1.0f / sqrtf((float)DH);
    const float kscale = dodrop ? rk.scale : 1.f;          // (all-ones keep words without dropout: eval with p > 0 must not rescale)
    const Keep64 keep = keep_row(lane, rk, site, ((uint64_t)(bq * H + h) * 64 + (uint64_t)(tq - s0)) * 64, need_hi, dodrop);
    // the backward reads the decisions instead of recomputing them: one Philox call is ~130 VALU instructions with 40 quarter-rate integer
    // multiplies (~1 000 SIMD cycles), and the backward needed three per (tile, head) — 38 % of its issue time — for 8 bytes per (token, head)
    if (dodrop && g == 0 && qv) *reinterpret_cast<uint2*>(A.keep + ((size_t)tq * H + h) * 2) = make_uint2(keep.lo, keep.hi);
    f32x4 s[MT];
    float m = -INFINITY;
    auto mask_tile = [&](const int k, const unsigned pad) {
        const int jt = it - k;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tk = 16 * jt + 4 * g + r;
            const bool ok = qv && tk >= s0 && tk <= tq && !((pad >> (4 * g + r)) & 1u);
            const float v = ok ? s[k][r] * scale : -INFINITY;
            s[k][r] = v;
            m = fmaxf(m, v);
        }
    };
    s[0] = mma_rows<DH>(kf0, qf);
    mask_tile(0, pad16(wq.y));
    tile_store<DH>(lane, vt0, vf0);
    if (nk > 1) {
        s[1] = mma_rows<DH>(kf1, qf);
        mask_tile(1, pad16(w1));
        tile_store<DH>(lane, vt1, vf1);
    }
#pragma unroll
    for (int k = 2; k < MT; ++k) {
        if (k < nk) {
            float kf[DH / 4];
            frag_rows<DH>(lane, kf, qkv, 3 * D, 16 * (it - k), D + h * DH, T);
            const unsigned pad = pad_bits(lane, A.tok, it - k, T);
            s[k] = mma_rows<DH>(kf, qf);
            mask_tile(k, pad);
        }
    }
    m = xg_max(m);
    const float mref = m == -INFINITY ? 0.f : m;            // a query whose every key is PAD: probabilities 0 (torch: NaN)
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MT; ++k)
        if (k < nk)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(s[k][r] - mref); s[k][r] = e; sum += e; }
    sum = xg_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    if (g == 0 && qv) *reinterpret_cast<float2*>(A.stat + ((size_t)tq * H + h) * 2) = make_float2(mref, inv);
    f32x4 o[DH / 16];
#pragma unroll
    for (int db = 0; db < DH / 16; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        if (k < nk) {
            const int jt = it - k;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[k][r] *= inv * keep_at(keep, 16 * jt + 4 * g + r - s0, kscale);
            if (k == 0) tile_cols<DH>(lane, o, vt0, s[k]);                                      // out^T[d][i] += sum_j V[j][d] P~[i][j]
            else if (k == 1) tile_cols<DH>(lane, o, vt1, s[k]);
            else mma_cols<DH>(lane, o, qkv, 3 * D, 16 * jt, 2 * D + h * DH, s[k], T);
        }
    }
    if (qv) {
#pragma unroll
        for (int db = 0; db < DH / 16; ++db)
            st4(A.ctx + (size_t)tq * D + h * DH + 16 * db + 4 * g, make_float4(o[db][0], o[db][1], o[db][2], o[db][3]));
    }
}

// Persistent grid in the XCD-aware order: workgroup b runs on XCD b % 8 (observed on MI355X, MI355X_MICROARCH.md "Workgroup dispatch" — for
// speed only, never for correctness), and XCD x owns the CONTIGUOUS units [x per, (x + 1) per).  A tile's neighbour — whose K | V rows
// (forward, phase A) or Q | dctx rows (phase B) it reads too — is then worked on by the same XCD at about the same time: the second read
// is an L2 hit instead of a second trip through the fabric (first cut, tile = blockIdx.x: the forward moved 68 MB for 47 MB of operands).
// unit u of iteration k of workgroup (x = b & 7, j = b >> 3): x per + j + k (gridDim.x / 8).
__device__ __forceinline__ int wave_units(const int n_units) { return (n_units + 7) >> 3; }

template <int DH>
__global__ __launch_bounds__(256) void k_attn_wave_fwd(const AttnArgs2 A) {
    constexpr int TF = WTile<DH>::FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[4][2][TF];
    const int T = A.state[DR4SR_STATE_T];
    const int w = threadIdx.x >> 6;
    const int units = (((T + 15) >> 4) + 1) >> 1, per = wave_units(units), x = (int)blockIdx.x & 7, stride = (int)gridDim.x >> 3;
#pragma unroll 1
    for (int j = (int)blockIdx.x >> 3; j < per; j += stride) {
        // (the thread index through an opaque move: everything lane-dependent is then recomputed per tile instead of being hoisted out of the
        //  loop and held in registers across the whole body — 74 -> 93 VGPRs forward, 89 -> 130 backward without it)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int u = x * per + j, it = 2 * u + (w >> 1);
        if (u < units && 16 * it < T) fwd_tile<DH>(tid, A, T, it, lds[w][0], lds[w][1]);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// 256 threads = 1 tile x 2 heads x {phase A: the tile as QUERY tile -> dQ | phase B: the tile as KEY tile -> dK, dV}.
// No softmax pass: P = exp(s - m) / sum from the saved statistics, the row term sum_j P dP = <dctx, ctx> from the epilogue of the tile
// kernel in front (A.rd), as in attn_mfma.hip.  Loads as in the forward: the tile's own rows and those of its neighbour (in front for
// phase A, behind for phase B) are requested up front; further tiles (long sequences) take the dependent path.
template <int DH>
__device__ __forceinline__ void bwd_tile(const int tid, const AttnArgs2& A, const int T, const int it, float (*__restrict__ lds)[WTile<DH>::FLOATS]) {
    constexpr int D = 2 * DH, H = 2;
    const int lane = tid & 63, w = tid >> 6, h = w & 1, i16 = lane & 15, g = lane >> 4;
    const bool phaseB = w >= 2;
    const int t0 = 16 * it;
    const bool dodrop = A.training && A.p > 0.f;
    const float kscale = dodrop ? 1.0f / (1.0f - A.p) : 1.f;        // keep factor (the decisions themselves are the forward's: A.keep)
    const float scale = 1.0f / sqrtf((float)DH);
    const float* __restrict__ qkv = A.qkv;
    const float* __restrict__ dctx = A.dctx;
    const int tl = t0 + i16;                               // this lane's own token: query row (phase A) / key row (phase B)
    const bool lv = tl < T;
    const int2 wl_raw = tok_raw(A.tok, tl, T);

    if (!phaseB) {
        // ---- phase A: transposed orientation (lane: query i = i16, keys j = 4 g + r)  -> dQ rows of this tile
        float* kt0 = lds[2 * h];
        float* kt1 = lds[2 * h + 1];
        const bool has1 = it > 0;
        const int2 w1_raw = tok_raw(A.tok, tl - 16, T);
        float qf[DH / 4], cf[DH / 4], kf0[DH / 4], vf0[DH / 4], kf1[DH / 4], vf1[DH / 4];
        frag_rows<DH>(lane, qf, qkv, 3 * D, t0, h * DH, T);
        frag_rows<DH>(lane, cf, dctx, D, t0, h * DH, T);
        frag_rows<DH>(lane, kf0, qkv, 3 * D, t0, D + h * DH, T);
        frag_rows<DH>(lane, vf0, qkv, 3 * D, t0, 2 * D + h * DH, T);
        frag_rows<DH>(lane, kf1, qkv, 3 * D, has1 ? t0 - 16 : t0, D + h * DH, has1 ? T : 0);
        frag_rows<DH>(lane, vf1, qkv, 3 * D, has1 ? t0 - 16 : t0, 2 * D + h * DH, has1 ? T : 0);
        float mi, inv, rdot;
        row_stats(A, tl, h, T, mi, inv, rdot);
        const Keep64 keep = keep_load(A, tl, h, T, dodrop);      // the forward's decisions (no Philox in the backward)
        const int2 wl = tok_fix(wl_raw, tl, T);
        const int w1 = has1 ? tok_fix(w1_raw, tl - 16, T).y : 0;
        const int s0 = wl.x;
        const int lo = __builtin_amdgcn_readfirstlane(max(min16(lv ? (s0 >> 4) : it), max(it - (MT - 1), 0)));
        const int nk = it - lo + 1;
        f32x4 o[DH / 16];
#pragma unroll
        for (int fb = 0; fb < DH / 16; ++fb) o[fb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto ds_of = [&](const int jt, const unsigned pad, const f32x4 s, const f32x4 dp) {
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tk = 16 * jt + 4 * g + r;
                const bool ok = lv && tk >= s0 && tk <= tl && !((pad >> (4 * g + r)) & 1u);
                const float p = ok ? __expf(s[r] * scale - mi) * inv : 0.f;
                ds[r] = p * (dp[r] * keep_at(keep, tk - s0, kscale) - rdot) * scale;      // dS^T[j][i]
            }
            return ds;
        };
        {
            const f32x4 s = mma_rows<DH>(kf0, qf), dp = mma_rows<DH>(vf0, cf);                // dP~^T[j][i] = sum_d V[j][d] dctx[i][d]
            tile_store<DH>(lane, kt0, kf0);
            tile_cols<DH>(lane, o, kt0, ds_of(it, pad16(wl.y), s, dp));                             // dQ^T[f][i] += sum_j K[j][f] dS^T[j][i]
        }
        if (nk > 1) {
            const f32x4 s = mma_rows<DH>(kf1, qf), dp = mma_rows<DH>(vf1, cf);
            tile_store<DH>(lane, kt1, kf1);
            tile_cols<DH>(lane, o, kt1, ds_of(it - 1, pad16(w1), s, dp));
        }
#pragma unroll
        for (int k = 2; k < MT; ++k) {
            if (k < nk) {
                const int jt = it - k;
                float kf[DH / 4], vf[DH / 4];
                frag_rows<DH>(lane, kf, qkv, 3 * D, 16 * jt, D + h * DH, T);
                frag_rows<DH>(lane, vf, qkv, 3 * D, 16 * jt, 2 * D + h * DH, T);
                const unsigned pad = pad_bits(lane, A.tok, jt, T);
                const f32x4 s = mma_rows<DH>(kf, qf), dp = mma_rows<DH>(vf, cf);
                mma_cols<DH>(lane, o, qkv, 3 * D, 16 * jt, D + h * DH, ds_of(jt, pad, s, dp), T);
            }
        }
        if (lv) {
#pragma unroll
            for (int fb = 0; fb < DH / 16; ++fb)
                st4(A.dqkv + (size_t)tl * 3 * D + h * DH + 16 * fb + 4 * g, make_float4(o[fb][0], o[fb][1], o[fb][2], o[fb][3]));
        }
        return;
    }
    // ---- phase B: natural orientation (lane: key j = i16, queries i = 4 g + r)  -> dK, dV rows of this tile
    float* qt0 = lds[4 + 4 * h], *ct0 = lds[4 + 4 * h + 1], *qt1 = lds[4 + 4 * h + 2], *ct1 = lds[4 + 4 * h + 3];
    const int last = (T - 1) >> 4;
    const bool has1 = it < last;
    const int t1 = tl + 16;
    const bool v1 = has1 && t1 < T;
    float kf[DH / 4], vf[DH / 4];
    frag_rows<DH>(lane, kf, qkv, 3 * D, t0, D + h * DH, T);
    frag_rows<DH>(lane, vf, qkv, 3 * D, t0, 2 * D + h * DH, T);
    // row data of the query rows of tile it (this lane's own token) and it + 1, one row per lane; the natural orientation needs rows 4 g + r
    // of them per lane: fetched by ds_bpermute below
    float rm0, ri0, rr0, rm1, ri1, rr1;
    row_stats(A, tl, h, T, rm0, ri0, rr0);
    const Keep64 kp0 = keep_load(A, tl, h, T, dodrop);             // the forward's dropout decisions of the query rows (no Philox here)
    const int2 wn_raw = tok_raw(A.tok, t1, T);
    row_stats(A, v1 ? t1 : T, h, T, rm1, ri1, rr1);
    const Keep64 kp1 = keep_load(A, t1, h, T, dodrop);
    {
        float qf[DH / 4], cf[DH / 4], qg[DH / 4], cg[DH / 4];
        frag_rows<DH>(lane, qf, qkv, 3 * D, t0, h * DH, T);
        frag_rows<DH>(lane, cf, dctx, D, t0, h * DH, T);
        frag_rows<DH>(lane, qg, qkv, 3 * D, t0 + 16, h * DH, has1 ? T : 0);
        frag_rows<DH>(lane, cg, dctx, D, t0 + 16, h * DH, has1 ? T : 0);
        tile_store<DH>(lane, qt0, qf); tile_store<DH>(lane, ct0, cf);
        tile_store<DH>(lane, qt1, qg); tile_store<DH>(lane, ct1, cg);
    }
    const int2 wl = tok_fix(wl_raw, tl, T);
    int2 wn = tok_fix(wn_raw, t1, T);
    if (!v1) wn = make_int2(-1, 0);
    const int s0 = wl.x, nl = (wl.y >> 20) & 0x3ff;
    const int hi = __builtin_amdgcn_readfirstlane(min(min(max16(lv ? ((s0 + max(nl, 1) - 1) >> 4) : it), it + (MT - 1)), last));
    const int nqt = hi - it + 1;
    const bool need_hi = __ballot(nl > 32) != 0ull;         // some key of this tile sits in a sequence longer than 32 tokens: its queries' bits 32..63 matter
    const bool jok = lv && !((wl.y >> 30) & 1);
    f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
    for (int fb = 0; fb < DH / 16; ++fb) { dk[fb] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[fb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // one query tile: S, dP~ in the natural orientation, then P~ and dS for the four query rows 4 g + r of this lane.
    // rs0 / rm / ri / rr / kp: sequence start (-1: no such row), row max, 1 / row sum, row term, saved keep bits of query row (lane & 15) of the tile
    auto query_tile = [&](const int qt, const float (&qf)[DH / 4], const float (&cf)[DH / 4], const int rs0, const float rm,
                          const float ri, const float rr, const Keep64 kp, f32x4& pt, f32x4& ds) {
        const f32x4 s = mma_rows<DH>(qf, kf);                  // S[i][j]: rows i = 4 g + r (C layout), column j = i16
        const f32x4 dp = mma_rows<DH>(cf, vf);                 // dP~[i][j] = sum_d dctx[i][d] V[j][d]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = 4 * g + r, ti = 16 * qt + src;
            const int qs0 = __shfl(rs0, src, 64);
            const float qm = __shfl(rm, src, 64), qi = __shfl(ri, src, 64), qr = __shfl(rr, src, 64);
            const int pk = tl - qs0;                           // this key's position inside query row i's sequence (if it is that sequence)
            // query row i's decision for this key: bit pk of its saved words (positions >= 32 only exist in sequences longer than 32: the second word
            // is fetched under a wave-uniform test)
            unsigned kw = (unsigned)__shfl((int)kp.lo, src, 64);
            if (need_hi) { const unsigned kh = (unsigned)__shfl((int)kp.hi, src, 64); kw = pk < 32 ? kw : kh; }
            const float mkv = ((kw >> (pk & 31)) & 1u) ? kscale : 0.f;
            const bool ok = jok && qs0 == s0 && tl <= ti;      // (qs0 == -1: no such query row)
            const float p = ok ? __expf(s[r] * scale - qm) * qi : 0.f;
            pt[r] = p * mkv;                                   // P~[i][j]
            ds[r] = p * (dp[r] * mkv - qr) * scale;            // dS[i][j]
        }
    };
    {
        float qf[DH / 4], cf[DH / 4];
        tile_frag<DH>(lane, qf, qt0); tile_frag<DH>(lane, cf, ct0);
        f32x4 pt, ds;
        query_tile(it, qf, cf, lv ? s0 : -1, rm0, ri0, rr0, kp0, pt, ds);
        tile_cols<DH>(lane, dk, qt0, ds);                            // dK^T[f][j] += sum_i Q[i][f] dS[i][j]
        tile_cols<DH>(lane, dv, ct0, pt);                            // dV^T[d][j] += sum_i dctx[i][d] P~[i][j]
    }
    if (nqt > 1) {
        float qf[DH / 4], cf[DH / 4];
        tile_frag<DH>(lane, qf, qt1); tile_frag<DH>(lane, cf, ct1);
        f32x4 pt, ds;
        query_tile(it + 1, qf, cf, wn.x, rm1, ri1, rr1, kp1, pt, ds);
        tile_cols<DH>(lane, dk, qt1, ds);
        tile_cols<DH>(lane, dv, ct1, pt);
    }
#pragma unroll
    for (int q = 2; q < MT; ++q) {
        if (q < nqt) {
            const int qt = it + q, ti = 16 * qt + i16;
            const bool vi = ti < T;
            float qf[DH / 4], cf[DH / 4];
            frag_rows<DH>(lane, qf, qkv, 3 * D, 16 * qt, h * DH, T);
            frag_rows<DH>(lane, cf, dctx, D, 16 * qt, h * DH, T);
            int2 wi = tok_word(A.tok, ti, T);
            if (!vi) wi = make_int2(-1, 0);
            float rm, ri, rr;
            row_stats(A, ti, h, T, rm, ri, rr);
            f32x4 pt, ds;
            query_tile(qt, qf, cf, wi.x, rm, ri, rr, keep_load(A, ti, h, T, dodrop), pt, ds);
            mma_cols<DH>(lane, dk, qkv, 3 * D, 16 * qt, h * DH, ds, T);
            mma_cols<DH>(lane, dv, dctx, D, 16 * qt, h * DH, pt, T);
        }
    }
    if (lv) {
#pragma unroll
        for (int fb = 0; fb < DH / 16; ++fb) {
            float* base = A.dqkv + (size_t)tl * 3 * D + h * DH + 16 * fb + 4 * g;
            st4(base + D, make_float4(dk[fb][0], dk[fb][1], dk[fb][2], dk[fb][3]));
            st4(base + 2 * D, make_float4(dv[fb][0], dv[fb][1], dv[fb][2], dv[fb][3]));
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void k_attn_wave_bwd(const AttnArgs2 A) {
    constexpr int TF = WTile<DH>::FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 + 2 * 4][TF];      // phase A waves: K tiles k = 0, 1; phase B waves: Q | dctx tiles q = 0, 1
    const int T = A.state[DR4SR_STATE_T];
    const int units = (T + 15) >> 4, per = wave_units(units), x = (int)blockIdx.x & 7, stride = (int)gridDim.x >> 3;
#pragma unroll 1
    for (int j = (int)blockIdx.x >> 3; j < per; j += stride) {
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int it = x * per + j;
        if (it < units) bwd_tile<DH>(tid, A, T, it, lds);
    }
}

}  // namespace

// Persistent grids: 8 x ceil(units / 8) workgroups for the units the plan EXPECTS (expected_tokens; the capacity Tmax without a hint), at most
// what the device holds at once times four; the loop inside covers whatever the batch really has.  The first cut launched one workgroup per
// tile of the CAPACITY: 25 600 workgroups at B = 8 192 of which 2 800 had a tile.
int launch_attn_wave(const AttnArgs2& A, int DH, int Tmax, int Thint, bool bwd, hipStream_t s) {
    if (!A.tok || !A.stat || (bwd && (!A.rd || !A.dctx || !A.dqkv))) return DR4SR_E_ARG;
    const int Te = Thint > 0 && Thint < Tmax ? Thint + Thint / 8 + 64 : Tmax;
    const int tiles = ((Te < Tmax ? Te : Tmax) + 15) / 16, units = bwd ? tiles : (tiles + 1) / 2;
    int grid = 8 * ((units + 7) / 8);
    const int cap = DR4SR_XENV("DR4SR_ATTN_WAVE_GRID") ? atoi(DR4SR_XENV("DR4SR_ATTN_WAVE_GRID")) : 8192;
    if (grid > cap) grid = cap > 8 ? cap / 8 * 8 : 8;
    if (DH == 32) {
        if (bwd) hipLaunchKernelGGL(k_attn_wave_bwd<32>, dim3(grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL(k_attn_wave_fwd<32>, dim3(grid), dim3(256), 0, s, A);
    } else if (DH == 64) {
        if (bwd) hipLaunchKernelGGL(k_attn_wave_bwd<64>, dim3(grid), dim3(256), 0, s, A);
        else hipLaunchKernelGGL(k_attn_wave_fwd<64>, dim3(grid), dim3(256), 0, s, A);
    } else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}
