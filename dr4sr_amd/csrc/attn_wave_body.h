// attn_wave_body.h — device helpers of the wave-per-tile attention (round 6), shared by its launches of their own (attn_wave.hip) and by the
// forward folded into the wave-tile kernels (linear_wave.hip: wt_attn_ctx).  See attn_wave.hip for the design.
#pragma once
#include "common.h"
#include "attn_args.h"

namespace awv {

constexpr int MT = 5;                         // key tiles a query tile can need: a sequence of <= 64 tokens touches at most 5 tiles

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float xg_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16, 64)), fmaxf(__shfl_xor(v, 32, 64), __shfl_xor(v, 48, 64))); }
__device__ __forceinline__ float xg_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ int min16(int v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int max16(int v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// row fragment of an MFMA operand: lane (r16, g) takes DH / 4 consecutive floats of row (row0 + r16) at column col0 + g DH / 4; rows >= T
// read as zero (rows behind the batch's last token hold whatever an earlier, larger batch left there)
// BRANCH-FREE: the address is clamped into the batch and the value selected afterwards — a conditional load is a branch, and the compiler
// drains every load in flight (s_waitcnt vmcnt(0)) at the join of the first one, which serialised the token-word round trip in front of
// the operand round trip in the first cuts.  T <= 0 (no such tile): all zero.
template <int DH>
__device__ __forceinline__ void frag_rows(const int lane, float (&f)[DH / 4], const float* __restrict__ base, const int ld, const int row0, const int col0, const int T) {
    const int r16 = lane & 15, g = lane >> 4;
    const int row = row0 + r16;
    const bool ok = row < T;
    const float* p = base + (size_t)max(min(row, T - 1), 0) * ld + col0 + g * (DH / 4);
#pragma unroll
    for (int c = 0; c < DH / 4; c += 4) {
        const float4 v = ld4(p + c);
        f[c] = ok ? v.x : 0.f; f[c + 1] = ok ? v.y : 0.f; f[c + 2] = ok ? v.z : 0.f; f[c + 3] = ok ? v.w : 0.f;
    }
}
// token word of token t, branch-free: {t, 0} (a sequence of its own, length 0) behind the batch's last token
// tok_raw issues the load (clamped address, always executed); tok_fix, called AFTER the other loads of the flight have been issued, pins
// the value with an opaque move — without it the compiler sinks the load back under the `t < T` branch (the select's only consumer) and
// waits for it before it issues anything else.
__device__ __forceinline__ int2 tok_raw(const int2* __restrict__ tok, const int t, const int T) { return tok[max(min(t, T - 1), 0)]; }
__device__ __forceinline__ int2 tok_fix(int2 w, const int t, const int T) {
    asm volatile("" : "+v"(w.x), "+v"(w.y));
    return t < T ? w : make_int2(t, 0);
}
__device__ __forceinline__ int2 tok_word(const int2* __restrict__ tok, const int t, const int T) { return tok_fix(tok_raw(tok, t, T), t, T); }
// saved softmax statistics {row max, 1 / row sum} and the row term of (token t, head h), branch-free (zeros behind the batch)
__device__ __forceinline__ void row_stats(const AttnArgs2& A, const int t, const int h, const int T, float& m, float& inv, float& rd) {
    const int tc = max(min(t, T - 1), 0);
    const float2 st = *reinterpret_cast<const float2*>(A.stat + ((size_t)tc * 2 + h) * 2);
    const float r = A.rd[(size_t)tc * 2 + h];
    const bool ok = t < T;
    m = ok ? st.x : 0.f; inv = ok ? st.y : 0.f; rd = ok ? r : 0.f;
}
template <int DH>
__device__ __forceinline__ f32x4 mma_rows(const float (&a)[DH / 4], const float (&b)[DH / 4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DH / 4; ++s) acc = mfma16(a[s], b[s], acc);
    return acc;
}
// acc[fb] += sum over the tile's 16 rows of M[row][col0 + 16 fb + i16] * w[row]: the A operand (rows 4 g + s, one column per lane) straight
// from global memory — 16 lanes read 64 consecutive bytes of each of 4 rows
template <int DH>
__device__ __forceinline__ void mma_cols(const int lane, f32x4 (&acc)[DH / 16], const float* __restrict__ base, const int ld, const int row0, const int col0, const f32x4 w,
                                         const int T) {
    const int i16 = lane & 15, g = lane >> 4;
    float v[DH / 16][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = row0 + 4 * g + s;
#pragma unroll
        for (int fb = 0; fb < DH / 16; ++fb) v[fb][s] = row < T ? base[(size_t)row * ld + col0 + 16 * fb + i16] : 0.f;
    }
#pragma unroll
    for (int fb = 0; fb < DH / 16; ++fb)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[fb] = mfma16(v[fb][s], w[s], acc[fb]);
}

// the same product with the OUTPUT rows in the wave-tile kernels' register layout (linear_wave.hip): C tile fb, row 4 g' + r <-> column
// 8 g' + 4 fb + r of the head's block, i.e. lane (token, g') ends up holding the 8 consecutive columns 8 g' .. 8 g' + 7 of its token as
// acc[0] | acc[1] — which weight row feeds which C-tile row is a free permutation of the A operand's rows (here: of M's columns)
template <int DH>
__device__ __forceinline__ void mma_cols_wt(const int lane, f32x4 (&acc)[DH / 16], const float* __restrict__ base, const int ld, const int row0, const int col0,
                                            const f32x4 w, const int T) {
    static_assert(DH == 32, "two C tiles per 32-column head block");
    const int i16 = lane & 15, g = lane >> 4;
    float v[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = row0 + 4 * g + s;
        const float* p = base + (size_t)max(min(row, T - 1), 0) * ld + col0 + 8 * (i16 >> 2) + (i16 & 3);
        const float a = p[0], b = p[4];
        v[0][s] = row < T ? a : 0.f; v[1][s] = row < T ? b : 0.f;
    }
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[fb] = mfma16(v[fb][s], w[s], acc[fb]);
}

// keep decisions of one (query row, head): bit j of lo / hi = key position j / 32 + j is kept.  The four lanes of a query column each
// compute one Philox call (8 decisions) and exchange them; `hi` only when some query of the tile has more than 32 keys (wave-uniform).
struct Keep64 { unsigned lo, hi; };
__device__ __forceinline__ Keep64 keep_row(const int lane, const RngKey& rk, const uint32_t site, const uint64_t ebase, const bool need_hi, const bool dodrop) {
    Keep64 k{0xffffffffu, 0xffffffffu};
    if (!dodrop) return k;
    const int i16 = lane & 15, g = lane >> 4;
    const unsigned m8 = drop_bits8(rk, site, ebase + 8 * g);
    k.lo = __shfl(m8, i16, 64) | (__shfl(m8, i16 | 16, 64) << 8) | (__shfl(m8, i16 | 32, 64) << 16) | (__shfl(m8, i16 | 48, 64) << 24);
    if (need_hi) {
        const unsigned n8 = drop_bits8(rk, site, ebase + 32 + 8 * g);
        k.hi = __shfl(n8, i16, 64) | (__shfl(n8, i16 | 16, 64) << 8) | (__shfl(n8, i16 | 32, 64) << 16) | (__shfl(n8, i16 | 48, 64) << 24);
    }
    return k;
}
__device__ __forceinline__ float keep_at(const Keep64& k, const int pos, const float scale) {
    const unsigned w = pos < 32 ? k.lo : k.hi;
    return ((w >> (pos & 31)) & 1u) ? scale : 0.f;
}

// saved keep bits of (token t, head h), branch-free; all ones without dropout
__device__ __forceinline__ Keep64 keep_load(const AttnArgs2& A, const int t, const int h, const int T, const bool dodrop) {
    const uint2 v = *reinterpret_cast<const uint2*>(A.keep + ((size_t)max(min(t, T - 1), 0) * 2 + h) * 2);
    return dodrop ? Keep64{v.x, v.y} : Keep64{0xffffffffu, 0xffffffffu};
}
// PAD flags (bit 30 of the token words) of key tile jt as 16 bits, wave-uniform
__device__ __forceinline__ unsigned pad_bits(const int lane, const int2* __restrict__ tok, const int jt, const int T) {
    const int t = 16 * jt + (lane & 15);
    const int w = tok_word(tok, t, T).y;
    return (unsigned)(__ballot((w >> 30) & 1) & 0xffffull);
}

// ------------------------------------------------------------------------------------------------ staging helpers
// A wave keeps the 16 x DH tiles it needs BOTH as row fragments (products contracted over the features) and column-wise (products contracted
// over the tile's 16 rows) in a private LDS tile [16][DH + 4]: the row fragment is loaded once from global memory, stored, and every other
// view is an LDS read — the first cut read the column view from global memory again, a second dependent round trip per tile.  Private to the
// wave: LDS operations of one wave execute in order, so no barrier is needed, only the lgkmcnt wait the compiler places.
template <int DH> struct WTile { static constexpr int LD = DH + 4, FLOATS = 16 * LD; };

template <int DH>
__device__ __forceinline__ void tile_store(const int lane, float* __restrict__ t, const float (&f)[DH / 4]) {
    constexpr int LD = WTile<DH>::LD;
    const int r16 = lane & 15, g = lane >> 4;
    float* p = t + r16 * LD + g * (DH / 4);
#pragma unroll
    for (int c = 0; c < DH / 4; c += 4) st4(p + c, make_float4(f[c], f[c + 1], f[c + 2], f[c + 3]));
}
template <int DH>
__device__ __forceinline__ void tile_frag(const int lane, float (&f)[DH / 4], const float* __restrict__ t) {
    constexpr int LD = WTile<DH>::LD;
    const int r16 = lane & 15, g = lane >> 4;
    const float* p = t + r16 * LD + g * (DH / 4);
#pragma unroll
    for (int c = 0; c < DH / 4; c += 4) {
        const float4 v = ld4(p + c);
        f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w;
    }
}
// acc[fb] += sum over the tile's 16 rows of tile[row][16 fb + i16] * w[row]  (w in C layout: register s = row 4 g + s)
template <int DH>
__device__ __forceinline__ void tile_cols(const int lane, f32x4 (&acc)[DH / 16], const float* __restrict__ t, const f32x4 w) {
    constexpr int LD = WTile<DH>::LD;
    const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int fb = 0; fb < DH / 16; ++fb)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[fb] = mfma16(t[(4 * g + s) * LD + 16 * fb + i16], w[s], acc[fb]);
}
__device__ __forceinline__ unsigned pad16(const int word) { return (unsigned)(__ballot((word >> 30) & 1) & 0xffffull); }


}  // namespace awv
