// linear_wave.hip — at-scale forms of the SASRec layer's token-tile kernels (round 3): WAVE-AUTONOMOUS 16-token tiles with the
// layer's weights RESIDENT IN LDS, GEMMs on the bf16 matrix cores as a 3-term split ("bf16x3").
//
// Why.  The 256-thread / 32-row tile kernels of linear.hip stream every weight from L2 as the MFMA B operand of every tile
// (128 KB of weights per 32 tokens) and cross five workgroup barriers per tile; at B = 8 192 their waves spend 42 % of their life in
// s_waitcnt and the MFMA pipe is 29 % busy (profiles/round2_sq_pmc_B8192_toys.txt).  And v_mfma_f32_16x16x4_f32 shares the SIMD's
// fp32 datapath with the VALU: an fp32-MFMA wave and a VALU wave on one SIMD take the SUM of their times
// (tools/probes/mfma_valu_overlap_probe.hip), so MFMA utilisation of an fp32 kernel is capped at MFMA / (MFMA + VALU) cycles — ~55 % for
// these kernels even with no stall at all.  Here
//   * one workgroup of 12 (or 16) waves per CU is PERSISTENT: it builds the layer's weight image in LDS once (145 KB of the CU's 160 KB
//     at d = 64: out_proj 64x64, linear1 128x64, linear2 64x128, next in_proj 192x64) and every wave then walks token tiles on its
//     own: NO barrier after the prologue, no LDS staging of activations;
//   * every GEMM runs in the TRANSPOSED orientation  Y^T[n][t] = sum_k W[n][k] X^T[k][t]: A = a weight fragment from LDS, B = the
//     activation tile held in REGISTERS, and the C tile a wave gets is directly the B operand of the next GEMM.  Lane
//     (t = lane & 15, g = lane >> 4) owns, of token t, the 8 consecutive columns 32 J + 8 g .. + 7 of every 32-column block J (two C
//     tiles per block; which weight row feeds which C-tile row is a free permutation, applied when the LDS image is built).  A token
//     never leaves its four lanes: bias / dropout / GELU are per register, LayerNorm is an in-lane sum + two shuffles, global accesses
//     are 32 contiguous bytes per lane = one full 128-byte line per token and block;
//   * bf16x3: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits), and x·w ~ hi·hi + hi·lo + lo·hi on
//     v_mfma_f32_16x16x32_bf16 with fp32 accumulation — 3 matrix instructions of 8 passes per 32 k against 8 fp32 ones of 8 passes: 2.7x less
//     matrix-pipe time, on a pipe that (unlike the fp32 MFMA) runs BESIDE the VALU.  Max-norm error of a K = 64..192 product:
//     5e-6 (fp32 MFMA: 4e-7) — the precision class JAX calls HIGH; the parity bar is 1e-3 (north star) / 2e-4 (tests).  The
//     EXACT instantiations keep v_mfma_f32_16x16x4_f32 on the same data flow and are the DEFAULT (HBM-bound either way);
//     DR4SR_WT_BF16X3=1 selects the split in the forward kernel (tests run both);
//   * 8 consecutive columns of a token = ONE Philox call (16-bit dropout decisions, common.h), half the calls of the float4 kernels;
//   * saved activations keep their [T, *] layouts: each kernel is a drop-in for its linear.hip counterpart (DR4SR_NO_WAVE_TILES
//     restores those; tests run both).
//
// Reference arithmetic: torch.nn.TransformerEncoderLayer as configured at /root/reference model/sasrec.py:21-34 (post-norm, exact-erf
// GELU, batch_first), called at model/sasrec.py:65-68.
#include "common.h"
#include "kernels.h"
#include "attn_wave_body.h"
#include <cstdlib>

extern __shared__ __attribute__((aligned(16))) float smem[];

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A per-iteration zero the compiler cannot see through.  After the prologue barrier nothing writes the LDS images, so every weight
// fragment / bias / LayerNorm-parameter read of the tile loop is loop-invariant: LICM hoists them out of the loop (hundreds of registers)
// and the allocator spills them back (50-150 dwords per lane per tile, measured).  Adding this zero to the LDS base inside the loop keeps
// the reads where they are used.
__device__ __forceinline__ int wt_opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}
__device__ __forceinline__ f32x4 to4(const float4 v) { return (f32x4){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float4 from4(const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float quad_sum(float v) {       // over the four lanes (t, 0..3) of a token
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// x = hi + lo (+ 2^-17 x): two bf16 parts of 8 consecutive fp32 values
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h0 = (__bf16)a[i], h1 = (__bf16)b[i];
        hi[i] = h0; hi[4 + i] = h1;
        lo[i] = (__bf16)(a[i] - (float)h0); lo[4 + i] = (__bf16)(b[i] - (float)h1);
    }
}

// ---------------------------------------------------------------------------------------------- LDS weight images
// Row m = 16 T + i of an image feeds row i of C tile T; C tile T = 2 J + h holds, in lane group g, the output columns
// 32 J + 8 g + 4 h + q (q = 0..3), i.e. i = 4 g + q  <->  weight row n(T, i) = 32 (T / 2) + 8 (i / 4) + 4 (T & 1) + (i & 3).
__device__ __forceinline__ int wt_image_row(int n) {        // image row of weight row n
    const int J = n >> 5, r = n & 31, gq = r >> 3, h = (r >> 2) & 1, q = r & 3;
    return 16 * (2 * J + h) + 4 * gq + q;
}
// W [N][K] (y = x W^T form).  EXACT: fp32 rows of K + 8 floats; else rows of [K bf16 hi | K bf16 lo | 32 B pad].  Either row pitch is
// 2 mod 16 in 16-byte slots, which makes the ds_read_b128 of lane (i, g) — row i, slot offset g — conflict-free in all four lane
// groups of the instruction (MI355X_MICROARCH.md §LDS; a pitch of 1 slot collides lanes 11/12: measured 41 % conflict cycles).
template <bool EXACT, int N, int K> struct WtImg {
    static constexpr int pitch = EXACT ? (K + 8) * 4 : 4 * K + 32;          // bytes per row
    static constexpr int bytes = N * pitch;
    static __device__ __forceinline__ void load(char* __restrict__ img, const float* __restrict__ W) {
        constexpr int C8 = K / 8;
        for (int i = threadIdx.x; i < N * C8; i += (int)blockDim.x) {
            const int n = i / C8, c = (i % C8) * 8;
            const float4 a = ld4(W + (size_t)n * K + c), b = ld4(W + (size_t)n * K + c + 4);
            char* row = img + wt_image_row(n) * pitch;
            if constexpr (EXACT) {
                st4(reinterpret_cast<float*>(row) + c, a);
                st4(reinterpret_cast<float*>(row) + c + 4, b);
            } else {
                bf16x8 hi, lo;
                split8(to4(a), to4(b), hi, lo);
                *reinterpret_cast<bf16x8*>(row + 2 * c) = hi;
                *reinterpret_cast<bf16x8*>(row + 2 * K + 2 * c) = lo;
            }
        }
    }
    // acc[T] (C tile T, this lane: Y[t][32 (T/2) + 8 g + 4 (T&1) + q]) += sum_k W[n(T, .)][k] X[t][k];  x[2 J + h] = X[t][32 J + 8 g + 4 h + 0..3]
    static __device__ __forceinline__ void gemm(const char* __restrict__ img, const f32x4 (&x)[K / 16], f32x4 (&acc)[N / 16]) {
        const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
        constexpr int NT = N / 16, GT = EXACT ? 4 : 2;      // C tiles per group: their fragment loads, then their MFMAs (sched barrier
        static_assert(NT % GT == 0, "column tiles in groups");     //  between groups: the compiler otherwise hoists a whole GEMM's loads and spills)
        if constexpr (EXACT) {
            const char* base = img + r16 * pitch + 32 * g;
#pragma unroll
            for (int kb = 0; kb < K / 16; ++kb) {           // kb = 2 J + h: k = 32 J + 8 g + 4 h + q
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += GT) {
                    float4 a[GT];
#pragma unroll
                    for (int u = 0; u < GT; ++u)
                        a[u] = *reinterpret_cast<const float4*>(base + (t0 + u) * 16 * pitch + 128 * (kb >> 1) + 16 * (kb & 1));
#pragma unroll
                    for (int u = 0; u < GT; ++u) {
                        acc[t0 + u] = mfma16x4(a[u].x, x[kb][0], acc[t0 + u]);
                        acc[t0 + u] = mfma16x4(a[u].y, x[kb][1], acc[t0 + u]);
                        acc[t0 + u] = mfma16x4(a[u].z, x[kb][2], acc[t0 + u]);
                        acc[t0 + u] = mfma16x4(a[u].w, x[kb][3], acc[t0 + u]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            const char* base = img + r16 * pitch + 16 * g;
            bf16x8 bh[K / 32], bl[K / 32];
#pragma unroll
            for (int J = 0; J < K / 32; ++J) split8(x[2 * J], x[2 * J + 1], bh[J], bl[J]);
#pragma unroll
            for (int J = 0; J < K / 32; ++J) {
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += GT) {
                    bf16x8 ah[GT], al[GT];
#pragma unroll
                    for (int u = 0; u < GT; ++u) {
                        const char* p = base + (t0 + u) * 16 * pitch + 64 * J;
                        ah[u] = *reinterpret_cast<const bf16x8*>(p);
                        al[u] = *reinterpret_cast<const bf16x8*>(p + 2 * K);
                    }
#pragma unroll
                    for (int u = 0; u < GT; ++u) acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[u], bh[J], acc[t0 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < GT; ++u) acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u], bl[J], acc[t0 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < GT; ++u) acc[t0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[u], bh[J], acc[t0 + u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
};

// Data-gradient form on the SAME image (fp32, EXACT):  acc[T'] (dX[t][kcol(T', .)]) += sum_n dY[t][n] W[n][k],  W [NR][KC] as
// WtImg<true, NR, KC>, by[2 J + h] = dY[t][32 J + 8 g + 4 h + 0..3].  A operand of instruction (J, h, q): lane (i, slot g) supplies
// W[32 J + 8 g + 4 h + q][kcol(T', i)] = image row 16 (2 J + h) + 4 g + q, column kcol(T', i) = 32 (T'/2) + 8 (i/4) + 4 (T'&1) + (i&3):
// four ds_read_b32 per four MFMAs (the two lane groups of a half-wave share banks: 2-way conflicts, LDS is not the limiter here).
template <int NR, int KC>
__device__ __forceinline__ void wt_gemm_xw(const char* __restrict__ img, const f32x4 (&by)[NR / 16], f32x4 (&acc)[KC / 16]) {
    constexpr int pitch = WtImg<true, NR, KC>::pitch / 4, KT = KC / 16, GT = 4;
    static_assert(KT % GT == 0, "column tiles in groups of 4");
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const float* base = reinterpret_cast<const float*>(img) + 4 * g * pitch + 8 * (r16 >> 2) + (r16 & 3);
#pragma unroll
    for (int nb = 0; nb < NR / 16; ++nb) {
#pragma unroll
        for (int t0 = 0; t0 < KT; t0 += GT) {
            float a[GT][4];
#pragma unroll
            for (int u = 0; u < GT; ++u) {
                const float* wp = base + 16 * nb * pitch + 32 * ((t0 + u) >> 1) + 4 * ((t0 + u) & 1);
                a[u][0] = wp[0]; a[u][1] = wp[pitch]; a[u][2] = wp[2 * pitch]; a[u][3] = wp[3 * pitch];
            }
#pragma unroll
            for (int u = 0; u < GT; ++u) {
                acc[t0 + u] = mfma16x4(a[u][0], by[nb][0], acc[t0 + u]);
                acc[t0 + u] = mfma16x4(a[u][1], by[nb][1], acc[t0 + u]);
                acc[t0 + u] = mfma16x4(a[u][2], by[nb][2], acc[t0 + u]);
                acc[t0 + u] = mfma16x4(a[u][3], by[nb][3], acc[t0 + u]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// sum over the 16 lanes of a DPP row (the 16 tokens of a tile), result in every lane: four rotate-and-add steps
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// LayerNorm backward of one token (lanes (t, 0..3)): g = upstream grad (in) -> d(LN input) (out); u = LN input; affine partials of the
// tile (sum over its 16 tokens) -> part[0..D) = d gamma, part[D..2D) = d beta, written by the lanes of token 0.  Same arithmetic as
// ln_bwd_row (common.h).  `ok`: rows past T contribute nothing.
template <int NT>
__device__ __forceinline__ void wt_ln_bwd(f32x4 (&gv)[NT], const f32x4 (&u)[NT], float mean, float rstd, const f32x4 (&gam)[NT], bool ok,
                                          float* __restrict__ part, int r16, int g) {
    constexpr float invD = 1.0f / (16 * NT);
    constexpr int D = 16 * NT;
    f32x4 xh[NT], gg[NT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        xh[j] = (u[j] - mean) * rstd;
        if (!ok) gv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gg[j] = gv[j] * gam[j];
        s1 += (gg[j][0] + gg[j][1]) + (gg[j][2] + gg[j][3]);
        s2 += (gg[j][0] * xh[j][0] + gg[j][1] * xh[j][1]) + (gg[j][2] * xh[j][2] + gg[j][3] * xh[j][3]);
    }
    s1 = quad_sum(s1) * invD;
    s2 = quad_sum(s2) * invD;
    // affine partials: column sums over the tile's tokens (rows past T hold zeros)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x4 dg, db;
#pragma unroll
        for (int q = 0; q < 4; ++q) { dg[q] = row16_sum(gv[j][q] * xh[j][q]); db[q] = row16_sum(gv[j][q]); }
        if (r16 == 0) {
            const int c = 32 * (j >> 1) + 8 * g + 4 * (j & 1);
            st4(part + c, from4(dg));
            st4(part + D + c, from4(db));
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) gv[j] = rstd * (gg[j] - s1 - xh[j] * s2);
}

__device__ __forceinline__ void wt_load_v(float* __restrict__ dst, const float* __restrict__ v, int n) {
    for (int i = threadIdx.x; i < n; i += (int)blockDim.x) dst[i] = v[i];
}

// a token row [NC columns] <-> registers: v[2 J + h] = columns 32 J + 8 g + 4 h + 0..3
template <int NC>
__device__ __forceinline__ void wt_row_load(f32x4 (&v)[NC / 16], const float* __restrict__ p, int g) {
#pragma unroll
    for (int J = 0; J < NC / 32; ++J) {
        v[2 * J] = to4(ld4(p + 32 * J + 8 * g));
        v[2 * J + 1] = to4(ld4(p + 32 * J + 8 * g + 4));
    }
}
template <int NC>
__device__ __forceinline__ void wt_row_store(float* __restrict__ p, const f32x4 (&v)[NC / 16], int g) {
#pragma unroll
    for (int J = 0; J < NC / 32; ++J) {
        st4(p + 32 * J + 8 * g, from4(v[2 * J]));
        st4(p + 32 * J + 8 * g + 4, from4(v[2 * J + 1]));
    }
}
// the same with non-temporal stores: activations saved for the BACKWARD pass (nothing reads them before the forward has ended), so that
// they stream past L2 / MALL instead of evicting what the next launches read
template <int NC>
__device__ __forceinline__ void wt_row_store_nt(float* __restrict__ p, const f32x4 (&v)[NC / 16], int g) {
#ifdef WT_NO_NT
    wt_row_store<NC>(p, v, g);
#else
#pragma unroll
    for (int J = 0; J < NC / 32; ++J) {
        __builtin_nontemporal_store(v[2 * J], reinterpret_cast<f32x4*>(p + 32 * J + 8 * g));
        __builtin_nontemporal_store(v[2 * J + 1], reinterpret_cast<f32x4*>(p + 32 * J + 8 * g + 4));
    }
#endif
}
// v *= dropout keep factors of elements e0 + (this lane's columns): one Philox call per 32-column block
template <int NC>
__device__ __forceinline__ void wt_row_drop(f32x4 (&v)[NC / 16], const RngKey& rk, uint32_t site, uint64_t e0, int g) {
#pragma unroll
    for (int J = 0; J < NC / 32; ++J) {
        float4 lo, hi;
        drop8(rk, site, e0 + 32 * J + 8 * g, lo, hi);
        v[2 * J] *= to4(lo);
        v[2 * J + 1] *= to4(hi);
    }
}

// LayerNorm of one token spread over lanes (t, 0..3).  Same arithmetic as ln_stats16 (common.h).
template <int NT>
__device__ __forceinline__ void wt_ln_stats(const f32x4 (&v)[NT], float& mean, float& rstd, float eps) {
    constexpr float invD = 1.0f / (16 * NT);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    mean = quad_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float a = v[j][0] - mean, b = v[j][1] - mean, c = v[j][2] - mean, d = v[j][3] - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    rstd = 1.0f / sqrtf(quad_sum(q) * invD + eps);
}

// LDS carve-up of the forward kernel (bytes)
template <bool EXACT, int D, int F> struct WtFwdLds {
    typedef WtImg<EXACT, D, D> Out;
    typedef WtImg<EXACT, F, D> W1;
    typedef WtImg<EXACT, D, F> W2;
    typedef WtImg<EXACT, 3 * D, D> Nx;
    static constexpr int o_out = 0, o_w1 = o_out + Out::bytes, o_w2 = o_w1 + W1::bytes, o_nx = o_w2 + W2::bytes, o_vec = o_nx + Nx::bytes;
    // vectors (floats): out_b[D] ln1_w[D] ln1_b[D] b1[F] b2[D] ln2_w[D] ln2_b[D] nx_b[3D]
    static constexpr int v_outb = 0, v_ln1w = D, v_ln1b = 2 * D, v_b1 = 3 * D, v_b2 = 3 * D + F, v_ln2w = 4 * D + F, v_ln2b = 5 * D + F,
                         v_nxb = 6 * D + F, n_vec = 9 * D + F;
    static constexpr int total = o_vec + 4 * n_vec;
};

// Round 6 — the layer's attention FORWARD at the head of the tile (PostArgs::wt_attn): the ctx rows of this wave's 16 tokens computed from
// qkv in global memory (written by the launch in front) instead of being read back from an attention launch of its own.  The arithmetic,
// the saved statistics / keep bits and the dropout elements are attn_wave.hip's (shared bodies: attn_wave_body.h); the output lands directly
// in this kernel's register layout — lane (token, g) holds columns 32 J + 8 g .. + 7 of head J as bc[2 J], bc[2 J + 1] — because which V
// column feeds which C-tile row of P V is a free permutation (awv::mma_cols_wt).  ctx is still STORED (the backward's <dctx, ctx> row term and
// the out_proj weight gradient read it); what disappears is one launch per layer and ctx's way back in.  LDS is full of weight images
// (148 of 160 KB), so the V operand comes straight from global memory.  Reference: torch MHA as configured at model/sasrec.py:21-34, called
// :65-68, masks :48, :58.
template <int D>
__device__ __forceinline__ void wt_attn_ctx(const PostArgs& A, f32x4 (&bc)[D / 16], const int tile, const int T, const bool dodrop, const RngKey& rk) {
    static_assert(D == 64, "two heads of 32 columns");
    constexpr int DH = 32, H = 2, MT = awv::MT;
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4;
    const int it = tile, t0 = 16 * it, tq = t0 + i16;
    const bool qv = tq < T, has1 = it > 0;
    const float* __restrict__ qkv = A.at.qkv;
    const int2 wq_raw = awv::tok_raw(A.at.tok, tq, T), w1_raw = awv::tok_raw(A.at.tok, tq - 16, T);
    float qf[H][DH / 4], kf0[H][DH / 4], kf1[H][DH / 4];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        awv::frag_rows<DH>(lane, qf[h], qkv, 3 * D, t0, h * DH, T);
        awv::frag_rows<DH>(lane, kf0[h], qkv, 3 * D, t0, D + h * DH, T);
        awv::frag_rows<DH>(lane, kf1[h], qkv, 3 * D, has1 ? t0 - 16 : t0, D + h * DH, has1 ? T : 0);
    }
    const int2 wq = awv::tok_fix(wq_raw, tq, T);
    const int w1 = has1 ? awv::tok_fix(w1_raw, tq - 16, T).y : 0;
    const int s0 = wq.x, nq = (wq.y >> 20) & 0x3ff, bq = wq.y & 0xfffff;
    const int lo = __builtin_amdgcn_readfirstlane(max(awv::min16(qv ? (s0 >> 4) : it), max(it - (MT - 1), 0)));
    const int nk = it - lo + 1;
    const bool need_hi = __ballot(nq > 32) != 0ull;
    const uint32_t site = DR4SR_SITE_ATTN + 4 * A.layer;
    const float scale = 0.17677669529663687f;               // 1 / sqrt(32)
    const float kscale = dodrop ? rk.scale : 1.f;
    const unsigned pad0 = awv::pad16(wq.y), pad1 = awv::pad16(w1);
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const awv::Keep64 keep = awv::keep_row(lane, rk, site, ((uint64_t)(bq * H + h) * 64 + (uint64_t)(tq - s0)) * 64, need_hi, dodrop);
        if (dodrop && g == 0 && qv) *reinterpret_cast<uint2*>(A.at.keep + ((size_t)tq * H + h) * 2) = make_uint2(keep.lo, keep.hi);
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
        s[0] = awv::mma_rows<DH>(kf0[h], qf[h]);
        mask_tile(0, pad0);
        if (nk > 1) { s[1] = awv::mma_rows<DH>(kf1[h], qf[h]); mask_tile(1, pad1); }
#pragma unroll
        for (int k = 2; k < MT; ++k) {
            if (k < nk) {
                float kf[DH / 4];
                awv::frag_rows<DH>(lane, kf, qkv, 3 * D, 16 * (it - k), D + h * DH, T);
                const unsigned pad = awv::pad_bits(lane, A.at.tok, it - k, T);
                s[k] = awv::mma_rows<DH>(kf, qf[h]);
                mask_tile(k, pad);
            }
        }
        m = awv::xg_max(m);
        const float mref = m == -INFINITY ? 0.f : m;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < MT; ++k)
            if (k < nk)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __expf(s[k][r] - mref); s[k][r] = e; sum += e; }
        sum = awv::xg_sum(sum);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        if (g == 0 && qv) *reinterpret_cast<float2*>(A.at.stat + ((size_t)tq * H + h) * 2) = make_float2(mref, inv);
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            if (k < nk) {
                const int jt = it - k;
#pragma unroll
                for (int r = 0; r < 4; ++r) s[k][r] *= inv * awv::keep_at(keep, 16 * jt + 4 * g + r - s0, kscale);
                awv::mma_cols_wt<DH>(lane, o, qkv, 3 * D, 16 * jt, 2 * D + h * DH, s[k], T);       // out^T += V^T P~^T, rows in this kernel's layout
            }
        }
        bc[2 * h] = o[0]; bc[2 * h + 1] = o[1];
    }
}

// the forward chain for ONE 16-token tile of a wave; z (out): the LayerNorm2 output rows (this lane's columns), also stored to A.z if set
template <int D, int F, bool EXACT, typename Lds>
__device__ __forceinline__ void wt_fwd_tile(const PostArgs& A, const char* lds, const float* vec, f32x4 (&yout)[D / 16], const int tile, const int T,
                                            const bool dodrop, const bool actdrop, const RngKey& rk, float* __restrict__ zout) {
    constexpr int DT = D / 16, FT = F / 16, QT = 3 * D / 16;
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const uint32_t sP = A.sP, sA = A.sA, sF = A.sF;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 y[D / 16];
    {
        const int t = tile * 16 + r16;
        const bool ok = t < T;
        const size_t tl = ok ? t : T - 1;                   // rows past T: a valid row again (a token is one MFMA column: no mixing)
        f32x4 bc[DT];
#ifdef DR4SR_EXPERIMENTS                                    // (measured slower than the attention launch of its own: linear.hip attn_fold_fwd — not in the shipped kernels)
        if constexpr (D == 64) {
            if (A.wt_attn) {                                 // this layer's attention for the tile's 16 queries (no attention launch forward)
                wt_attn_ctx<D>(A, bc, tile, T, dodrop, rk);
                if (ok) wt_row_store<D>(A.at.ctx + (size_t)t * D, bc, g);
                __builtin_amdgcn_sched_barrier(0);
            } else wt_row_load<D>(bc, A.ctx + tl * D, g);
        } else
#endif
        wt_row_load<D>(bc, A.ctx + tl * D, g);
        wt_row_load<D>(y, A.x + tl * D, g);                 // the residual; becomes u1, then y
        // ---- out_proj + dropout1 + residual + LayerNorm1
        {
            f32x4 acc[DT];
#pragma unroll
            for (int j = 0; j < DT; ++j) acc[j] = zero4;
            Lds::Out::gemm(lds + Lds::o_out, bc, acc);
            f32x4 bias[DT];
            wt_row_load<D>(bias, vec + Lds::v_outb, g);
#pragma unroll
            for (int j = 0; j < DT; ++j) acc[j] += bias[j];
            if (dodrop) wt_row_drop<D>(acc, rk, sP, (uint64_t)t * D, g);
#pragma unroll
            for (int j = 0; j < DT; ++j) y[j] += acc[j];
            if (ok) wt_row_store_nt<D>(A.u1 + (size_t)t * D, y, g);
            float mean, rstd;
            wt_ln_stats<DT>(y, mean, rstd, A.eps);
            f32x4 gam[DT], bet[DT];
            wt_row_load<D>(gam, vec + Lds::v_ln1w, g);
            wt_row_load<D>(bet, vec + Lds::v_ln1b, g);
#pragma unroll
            for (int j = 0; j < DT; ++j) y[j] = (y[j] - mean) * rstd * gam[j] + bet[j];
            if (ok) wt_row_store<D>(A.y + (size_t)t * D, y, g);
            if (ok && g == 0) { A.st1[2 * (size_t)t] = mean; A.st1[2 * (size_t)t + 1] = rstd; }
        }
        // ---- linear1 + GELU + dropout
        f32x4 h[FT];
#pragma unroll
        for (int j = 0; j < FT; ++j) h[j] = zero4;
        Lds::W1::gemm(lds + Lds::o_w1, y, h);
        {
            f32x4 bias[FT];
            wt_row_load<F>(bias, vec + Lds::v_b1, g);
#pragma unroll
            for (int j = 0; j < FT; ++j) h[j] += bias[j];
            if (ok && A.a) wt_row_store_nt<F>(A.a + (size_t)t * F, h, g);        // (NULL under DR4SR_WT_RECOMPUTE_A: the backward recomputes a)
#pragma unroll
            for (int j = 0; j < FT; ++j) h[j] = (f32x4){gelu_erf(h[j][0]), gelu_erf(h[j][1]), gelu_erf(h[j][2]), gelu_erf(h[j][3])};
            if (actdrop) wt_row_drop<F>(h, rk, sA, (uint64_t)t * F, g);
            if (ok) wt_row_store_nt<F>(A.h + (size_t)t * F, h, g);
        }
        // ---- linear2 + dropout2 + residual + LayerNorm2
        {
            f32x4 acc[DT];
#pragma unroll
            for (int j = 0; j < DT; ++j) acc[j] = zero4;
            Lds::W2::gemm(lds + Lds::o_w2, h, acc);
            f32x4 bias[DT];
            wt_row_load<D>(bias, vec + Lds::v_b2, g);
#pragma unroll
            for (int j = 0; j < DT; ++j) acc[j] += bias[j];
            if (dodrop) wt_row_drop<D>(acc, rk, sF, (uint64_t)t * D, g);
#pragma unroll
            for (int j = 0; j < DT; ++j) y[j] += acc[j];
            if (ok) wt_row_store_nt<D>(A.u2 + (size_t)t * D, y, g);
            float mean, rstd;
            wt_ln_stats<DT>(y, mean, rstd, A.eps);
            f32x4 gam[DT], bet[DT];
            wt_row_load<D>(gam, vec + Lds::v_ln2w, g);
            wt_row_load<D>(bet, vec + Lds::v_ln2b, g);
#pragma unroll
            for (int j = 0; j < DT; ++j) y[j] = (y[j] - mean) * rstd * gam[j] + bet[j];
            if (ok && zout) wt_row_store<D>(zout + (size_t)t * D, y, g);
            if (ok && g == 0) { A.st2[2 * (size_t)t] = mean; A.st2[2 * (size_t)t + 1] = rstd; }
        }
        // ---- layer-boundary fusion: the next layer's in_proj on the rows still in registers
        if (A.nx_qkv) {
            f32x4 q[QT];
#pragma unroll
            for (int j = 0; j < QT; ++j) q[j] = zero4;
            Lds::Nx::gemm(lds + Lds::o_nx, y, q);
            f32x4 bias[QT];
            wt_row_load<3 * D>(bias, vec + Lds::v_nxb, g);
#pragma unroll
            for (int j = 0; j < QT; ++j) q[j] += bias[j];
            if (ok) wt_row_store<3 * D>(A.nx_qkv + (size_t)t * 3 * D, q, g);
        }
    }
    // (y is a LOCAL array copied out here: written through the reference parameter, the same code keeps 50 more registers alive and spills)
#pragma unroll
    for (int jj = 0; jj < D / 16; ++jj) yout[jj] = y[jj];
}

// ------------------------------------------------------------------------------------------------ k_post_fwd, wave tiles
// ctx -> out_proj -> dropout -> +x -> LayerNorm1 -> linear1 -> GELU -> dropout -> linear2 -> dropout -> +y -> LayerNorm2
// [-> next layer's in_proj], all per wave on 16 tokens, activations in registers.
template <int D, int F, int WT_WAVES, bool EXACT>
__global__ __launch_bounds__(WT_WAVES * 64) void k_wt_post_fwd(const PostArgs A) {
    using Lds = WtFwdLds<EXACT, D, F>;
    constexpr int DT = D / 16;
    const int T = A.state[DR4SR_STATE_T];
    if ((int)blockIdx.x * 16 >= T) return;                  // no tile for this workgroup (the grid covers the device, not the batch)
    char* lds = reinterpret_cast<char*>(smem);
    float* vec = reinterpret_cast<float*>(lds + Lds::o_vec);
    Lds::Out::load(lds + Lds::o_out, A.out_w);
    Lds::W1::load(lds + Lds::o_w1, A.w1);
    Lds::W2::load(lds + Lds::o_w2, A.w2);
    if (A.nx_qkv) { Lds::Nx::load(lds + Lds::o_nx, A.nx_in_w); wt_load_v(vec + Lds::v_nxb, A.nx_in_b, 3 * D); }
    wt_load_v(vec + Lds::v_outb, A.out_b, D); wt_load_v(vec + Lds::v_ln1w, A.ln1_w, D); wt_load_v(vec + Lds::v_ln1b, A.ln1_b, D);
    wt_load_v(vec + Lds::v_b1, A.b1, F); wt_load_v(vec + Lds::v_b2, A.b2, D);
    wt_load_v(vec + Lds::v_ln2w, A.ln2_w, D); wt_load_v(vec + Lds::v_ln2b, A.ln2_b, D);
    __syncthreads();                                        // the only workgroup barrier of the kernel

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const bool actdrop = dodrop && A.sA != 0xffffffffu;

#pragma unroll 1
    for (int tile = w * (int)gridDim.x + (int)blockIdx.x; tile * 16 < T; tile += (int)gridDim.x * WT_WAVES) {
        const int oz = wt_opaque_zero();
        f32x4 z[DT];
        wt_fwd_tile<D, F, EXACT, Lds>(A, lds + oz, vec + oz, z, tile, T, dodrop, actdrop, rk, A.z);
    }
}

// LDS carve-up of the backward kernel (bytes): the three weights of the layer + (layer-boundary fusion) the upper layer's in_proj
template <int D, int F> struct WtBwdLds {
    typedef WtImg<true, D, F> W2;
    typedef WtImg<true, F, D> W1;
    typedef WtImg<true, D, D> Out;
    typedef WtImg<true, 3 * D, D> Up;
    static constexpr int o_w2 = 0, o_w1 = o_w2 + W2::bytes, o_out = o_w1 + W1::bytes, o_up = o_out + Out::bytes, o_vec = o_up + Up::bytes;
    static constexpr int v_ln2w = 0, v_ln1w = D, v_b1 = 2 * D, n_vec = 2 * D + F;           // floats
    static constexpr int total = o_vec + 4 * n_vec;
};

// the exact reverse of the forward chain for ONE 16-token tile of a wave.  gz (in): d z rows of the tile (this lane's columns).
//   LN2' -> du2 (kept: residual branch) ; df = du2 * mask_F -> global ; dh = df W2 ; da = dh * mask_A * gelu'(a) -> global ;
//   dy = da W1 + du2 ; LN1' -> du1 -> global ; dout = du1 * mask_P -> global ; dctx = dout W_out -> global (+ rd = <dctx, ctx> per head)
template <int D, int F, typename Lds>
__device__ __forceinline__ void wt_bwd_tile(const PostArgs& A, const char* lds, const float* vec, const f32x4 (&gz_in)[D / 16], const int tile,
                                            const int T, const bool dodrop, const bool actdrop, const RngKey& rk) {
    constexpr int DT = D / 16, FT = F / 16;
    f32x4 gz[DT];                                           // local copy (see wt_fwd_tile: arrays behind reference parameters cost registers)
#pragma unroll
    for (int j = 0; j < DT; ++j) gz[j] = gz_in[j];
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const int t = tile * 16 + r16;
    const bool ok = t < T;
    const size_t tl = ok ? t : T - 1;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* part = A.ln_part + (size_t)tile * 4 * D;
    // ---- LayerNorm2 backward
    {
        f32x4 u[DT], gam[DT];
        wt_row_load<D>(u, A.u2 + tl * D, g);
        wt_row_load<D>(gam, vec + Lds::v_ln2w, g);
        const float mean = A.st2[2 * tl], rstd = A.st2[2 * tl + 1];
        wt_ln_bwd<DT>(gz, u, mean, rstd, gam, ok, part, r16, g);          // gz = du2
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 da[FT];
    {
        f32x4 df[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) df[j] = gz[j];
        if (dodrop) wt_row_drop<D>(df, rk, A.sF, (uint64_t)t * D, g);
        if (ok) wt_row_store<D>(A.df + (size_t)t * D, df, g);
        // ---- dh = df W2
#pragma unroll
        for (int j = 0; j < FT; ++j) da[j] = zero4;
        wt_gemm_xw<D, F>(lds + Lds::o_w2, df, da);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- da = dh * mask_act * gelu'(a).  A.a == NULL (DR4SR_WT_RECOMPUTE_A, round 4 — the review's "recompute instead of store"):
    // a = y W1^T + b1 is recomputed from the saved LayerNorm1 output (the forward then does not write the [T, F] pre-activations, 512 B
    // per token and layer, and this pass reads 256 B of y instead of 512 B of a) by the forward's own MFMA sequence on the image that is
    // in LDS anyway.  MEASURED, same box, dense B = 8 192 / toys B = 131 072 / toys B = 8 192: k_wt_post_fwd 403 -> 377 / 686 -> 645 /
    // 51 -> 51 us, k_wt_post_bwd 427 -> 463 / 732 -> 798 / 52 -> 58 us, k_wt_post_mid 667 -> 677 / 1297 -> 1326 / 102 -> 105 us; step
    // 3.32 -> 3.34 / 5.86 -> 5.93 / 0.516 -> 0.520 ms.  The backward pays more for 64 more fp32 MFMAs per tile (they share the SIMD's
    // datapath with its VALU work, NOTEBOOK) than the forward gains from 15 % fewer bytes: the kernels sit where the HBM stream and
    // the fp32 datapath are both nearly full, so the saved pre-activations stay (default).
    {
        f32x4 av[FT];
        if (A.a) {
            wt_row_load<F>(av, A.a + tl * F, g);
        } else {
            f32x4 yv[DT];
            wt_row_load<D>(yv, A.y + tl * D, g);
            wt_row_load<F>(av, vec + Lds::v_b1, g);
            Lds::W1::gemm(lds + Lds::o_w1, yv, av);
        }
        if (actdrop) wt_row_drop<F>(da, rk, A.sA, (uint64_t)t * F, g);
#pragma unroll
        for (int j = 0; j < FT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) da[j][q] = ok ? da[j][q] * gelu_erf_grad(av[j][q]) : 0.f;
        if (ok) wt_row_store<F>(A.da + (size_t)t * F, da, g);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dy = da W1 + du2 ; LayerNorm1 backward
    {
        f32x4 acc[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) acc[j] = zero4;
        wt_gemm_xw<F, D>(lds + Lds::o_w1, da, acc);
#pragma unroll
        for (int j = 0; j < DT; ++j) gz[j] += acc[j];
        f32x4 u[DT], gam[DT];
        wt_row_load<D>(u, A.u1 + tl * D, g);
        wt_row_load<D>(gam, vec + Lds::v_ln1w, g);
        const float mean = A.st1[2 * tl], rstd = A.st1[2 * tl + 1];
        wt_ln_bwd<DT>(gz, u, mean, rstd, gam, ok, part + 2 * D, r16, g);  // gz = du1
        if (ok) wt_row_store<D>(A.du1 + (size_t)t * D, gz, g);
        if (dodrop) wt_row_drop<D>(gz, rk, A.sP, (uint64_t)t * D, g);     // gz = dout
        if (ok) wt_row_store<D>(A.dout + (size_t)t * D, gz, g);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dctx = dout W_out
    {
        f32x4 acc[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) acc[j] = zero4;
        wt_gemm_xw<D, D>(lds + Lds::o_out, gz, acc);
        if (ok) wt_row_store<D>(A.dctx + (size_t)t * D, acc, g);
        if (A.rd) {                                         // softmax-backward row term of the attention: <dctx, ctx> per head (H = 2: head = 32-column block)
            f32x4 c[DT];
            wt_row_load<D>(c, A.ctx + tl * D, g);
#pragma unroll
            for (int J = 0; J < D / 32; ++J) {
                float d = 0.f;
#pragma unroll
                for (int hq = 0; hq < 2; ++hq)
                    d += (acc[2 * J + hq][0] * c[2 * J + hq][0] + acc[2 * J + hq][1] * c[2 * J + hq][1])
                       + (acc[2 * J + hq][2] * c[2 * J + hq][2] + acc[2 * J + hq][3] * c[2 * J + hq][3]);
                d = quad_sum(d);
                if (ok && g == 0) A.rd[(size_t)t * A.n_head + J] = d;
            }
        }
    }
}

template <int D, int F, int WT_WAVES>
__global__ __launch_bounds__(WT_WAVES * 64) void k_wt_post_bwd(const PostArgs A) {
    using Lds = WtBwdLds<D, F>;
    constexpr int DT = D / 16, QT = 3 * D / 16;
    const int T = A.state[DR4SR_STATE_T];
    if ((int)blockIdx.x * 16 >= T) return;
    char* lds = reinterpret_cast<char*>(smem);
    float* vec = reinterpret_cast<float*>(lds + Lds::o_vec);
    Lds::W2::load(lds + Lds::o_w2, A.w2);
    Lds::W1::load(lds + Lds::o_w1, A.w1);
    Lds::Out::load(lds + Lds::o_out, A.out_w);
    if (A.up_dqkv) Lds::Up::load(lds + Lds::o_up, A.up_in_w);
    wt_load_v(vec + Lds::v_ln2w, A.ln2_w, D); wt_load_v(vec + Lds::v_ln1w, A.ln1_w, D); wt_load_v(vec + Lds::v_b1, A.b1, F);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const bool actdrop = dodrop && A.sA != 0xffffffffu;
#pragma unroll 1
    for (int tile = w * (int)gridDim.x + (int)blockIdx.x; tile * 16 < T; tile += (int)gridDim.x * WT_WAVES) {
        const int oz = wt_opaque_zero();
        const int t = tile * 16 + r16;
        const size_t tl = t < T ? t : T - 1;
        f32x4 gz[DT];
        if (A.up_dqkv) {                                    // layer-boundary fusion: dz = dqkv(layer+1) W_in(layer+1) + du1(layer+1)
            f32x4 dq[QT];
            wt_row_load<3 * D>(dq, A.up_dqkv + tl * 3 * D, g);
            wt_row_load<D>(gz, A.up_du1 + tl * D, g);
            wt_gemm_xw<3 * D, D>(lds + oz + Lds::o_up, dq, gz);
        } else {
            wt_row_load<D>(gz, A.dz + tl * D, g);
        }
        wt_bwd_tile<D, F, Lds>(A, lds + oz, vec + oz, gz, tile, T, dodrop, actdrop, rk);
    }
}

// ------------------------------------------------------------------------------------------------ k_post_mid, wave tiles
// Last layer: forward tile -> per-token scorer (in-kernel negative draw, tied-embedding dots, BCE, d z, table-gradient records) -> backward
// tile, all on the wave's 16 tokens with the query rows and d z in REGISTERS (model/basemodel.py:204-214, model/loss_func.py:9-38).
// LDS: ONE image of out_proj / linear1 / linear2 serves both directions (wt_gemm_xw reads the forward image), + a per-wave histogram
// for the owner sort of the tile's table-gradient entries.
template <int D, int F> struct WtMidLds {
    typedef WtImg<true, D, D> Out;
    typedef WtImg<true, F, D> W1;
    typedef WtImg<true, D, F> W2;
    typedef WtImg<true, 3 * D, D> Nx;                       // (never loaded: the last layer has no next in_proj)
    static constexpr int o_out = 0, o_w1 = o_out + Out::bytes, o_w2 = o_w1 + W1::bytes, o_nx = 0, o_vec = o_w2 + W2::bytes;
    static constexpr int v_outb = 0, v_ln1w = D, v_ln1b = 2 * D, v_b1 = 3 * D, v_b2 = 3 * D + F, v_ln2w = 4 * D + F, v_ln2b = 5 * D + F,
                         v_nxb = 6 * D + F, n_vec = 6 * D + F;
    static constexpr int o_hist = o_vec + 4 * n_vec;        // [waves][G + 4] ints
    static constexpr int total(int waves, int G) { return o_hist + waves * (G + 4) * 4; }
};

// Owner sort of ONE 16-token tile by ONE wave (the producer side of k_wgrad's owner_job_sorted, see linear.hip tile_sort): the tile's
// <= 48 table-gradient entries — (target, d pos, z_t), (negative, d neg, z_t), (input id, 1, dx0_t) — one per lane, ranked inside their
// owner's bucket by the value an LDS counter returns (lanes of one instruction are served in lane order: the order is the entry index, a
// function of the batch), then an exclusive scan of the G counters by the wave.  rec / idin: this token's record and input id, valid in
// the lanes (t, g = 0).  hist: the wave's private [G + 4] counters, all zero on entry and on exit.
__device__ __forceinline__ void wt_tile_sort(const ScoreTileArgs& S, const int tile, const int T, const int4 rec, const int idin, int* hist) {
    const int lane = threadIdx.x & 63, G = 1 << S.logG, per = G >> 6;
    const int r = lane / 3, kind = lane - 3 * r, t = tile * 16 + r;
    const int4 rr = make_int4(__shfl(rec.x, r & 15, 64), __shfl(rec.y, r & 15, 64), __shfl(rec.z, r & 15, 64), __shfl(rec.w, r & 15, 64));
    const int ii = __shfl(idin, r & 15, 64);
    int k = -1, rank = 0;
    int4 ent = make_int4(0, 0, 0, 0);
    if (lane < 48 && t < T) {
        int id = kind == 2 ? ii : (rr.x > 0 ? (kind ? rr.y : rr.x) : 0);
        const int cf = kind == 2 ? __float_as_int(1.0f) : (kind ? rr.w : rr.z);
        if (id > 0) { k = id & (G - 1); ent = make_int4(t, id >> S.logG, cf, kind == 2); rank = atomicAdd(&hist[k], 1); }
    }
    // exclusive scan of hist[0..G): lane l owns counters [per l, per (l + 1)); two passes over its counters (no register array)
    int sum = 0;
    for (int j = 0; j < per; ++j) sum += hist[lane * per + j];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    int ex = incl - sum;
    for (int j = 0; j < per; ++j) { const int c = hist[lane * per + j]; hist[lane * per + j] = ex; ex += c; }
    if (lane == 63) hist[G] = ex;
    unsigned* offw = reinterpret_cast<unsigned*>(S.off + (size_t)tile * (G + 4));
    for (int q = lane; q < (G + 4) / 4; q += 64)
        offw[q] = (unsigned)hist[4 * q] | ((unsigned)hist[4 * q + 1] << 8) | ((unsigned)hist[4 * q + 2] << 16) | ((unsigned)hist[4 * q + 3] << 24);
    if (k >= 0) S.ent[(size_t)tile * 48 + hist[k] + rank] = ent;
    // leave the counters zero for the wave's next tile (every read above has been issued; LDS serves one wave's operations in order)
    for (int j = 0; j < per; ++j) hist[lane * per + j] = 0;
    if (lane == 63) hist[G] = 0;
}

template <int D, int F, int WT_WAVES>
__global__ __launch_bounds__(WT_WAVES * 64) void k_wt_post_mid(const PostArgs A, const ScoreTileArgs S) {
    using Lds = WtMidLds<D, F>;
    constexpr int DT = D / 16;
    static_assert(D == 64, "scorer layout: 16 columns of a token per lane");
    const int T = A.state[DR4SR_STATE_T];
    if ((int)blockIdx.x * 16 >= T) return;
    float* const zout = S.rec ? A.z : nullptr;              // the query rows leave the kernel only when the owner job will gather them
    char* lds = reinterpret_cast<char*>(smem);
    float* vec = reinterpret_cast<float*>(lds + Lds::o_vec);
    Lds::Out::load(lds + Lds::o_out, A.out_w);
    Lds::W1::load(lds + Lds::o_w1, A.w1);
    Lds::W2::load(lds + Lds::o_w2, A.w2);
    wt_load_v(vec + Lds::v_outb, A.out_b, D); wt_load_v(vec + Lds::v_ln1w, A.ln1_w, D); wt_load_v(vec + Lds::v_ln1b, A.ln1_b, D);
    wt_load_v(vec + Lds::v_b1, A.b1, F); wt_load_v(vec + Lds::v_b2, A.b2, D);
    wt_load_v(vec + Lds::v_ln2w, A.ln2_w, D); wt_load_v(vec + Lds::v_ln2b, A.ln2_b, D);
    const int G = 1 << S.logG;
    int* hist_all = reinterpret_cast<int*>(lds + Lds::o_hist);
    if (S.ent) for (int i = threadIdx.x; i < WT_WAVES * (G + 4); i += WT_WAVES * 64) hist_all[i] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    int* hist = hist_all + w * (G + 4);
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
    const RngKey rk0 = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], 0.f);
    const bool actdrop = dodrop && A.sA != 0xffffffffu;

#pragma unroll 1
    for (int tile = w * (int)gridDim.x + (int)blockIdx.x; tile * 16 < T; tile += (int)gridDim.x * WT_WAVES) {
        const int oz = wt_opaque_zero();
        const int t = tile * 16 + r16;
        const bool ok = t < T;
        // ---- forward half: z = LayerNorm2 output rows of the tile, in registers
        f32x4 z[DT];
        wt_fwd_tile<D, F, true, Lds>(A, lds + oz, vec + oz, z, tile, T, dodrop, actdrop, rk, zout);
        __builtin_amdgcn_sched_barrier(0);
        // ---- the scorer's index chain (after the forward half: carried across it, its ~12 registers cost more than its latency)
        int b = 0, pos = 0, n = 0, idin = 0;
        int64_t row = 0, tgt = 0, ng = 0;
        if (ok) {
            b = find_seq_from(S.cu, S.B, t, S.tile_seq[tile]);
            const int c0 = S.cu[b];
            pos = t - c0; n = S.cu[b + 1] - c0;
            row = S.rows ? S.rows[b] : b;
            tgt = S.target[row * S.L + pos];
            if (S.sample_neg) {
                ng = sample_neg_id(rk0, (uint64_t)b * S.L + pos, S.n_items);
                if (g == 0) S.neg_item[(size_t)b * S.L + pos] = ng;
            } else {
                ng = S.neg_item[(size_t)b * S.L + pos];
            }
            ng = ng < 0 ? 0 : (ng >= S.n_items ? S.n_items - 1 : ng);
            if (S.ent) idin = S.idx32[t];
        }
        const bool on = ok && tgt > 0 && tgt < S.n_items;
        // ---- scorer + BCE (forward and backward) on the register rows
        float lsum = 0.f, cnt = 0.f;
        int4 rec = make_int4(0, 0, 0, 0);
        f32x4 dz[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) dz[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            f32x4 ep[DT], en[DT];
            const size_t et = on ? (size_t)tgt : 0, eg = on ? (size_t)ng : 0;          // (row 0 = PAD: a valid address for the lanes that are off)
            wt_row_load<D>(ep, S.E + et * D, g);
            wt_row_load<D>(en, S.E + eg * D, g);
            float sp = 0.f, sn = 0.f;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                sp += (z[j][0] * ep[j][0] + z[j][1] * ep[j][1]) + (z[j][2] * ep[j][2] + z[j][3] * ep[j][3]);
                sn += (z[j][0] * en[j][0] + z[j][1] * en[j][1]) + (z[j][2] * en[j][2] + z[j][3] * en[j][3]);
            }
            sp = quad_sum(sp); sn = quad_sum(sn);
            if (on) {
                const float lt = softplus_f(-sp) + softplus_f(sn);
                const float dpos = -sigmoid_f(-sp), dneg = sigmoid_f(sn);
                if (g == 0) { lsum += lt; cnt += 1.f; }
#pragma unroll
                for (int j = 0; j < DT; ++j) dz[j] = dpos * ep[j] + dneg * en[j];
                if (!S.rec) {                               // (latency-regime form kept for completeness: fp32 atomics into the table gradient)
#pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        const int c = 32 * (j >> 1) + 8 * g + 4 * (j & 1);
                        float* gp = S.dE + (size_t)tgt * D + c;
                        float* gn = S.dE + (size_t)ng * D + c;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { unsafeAtomicAdd(gp + q, dpos * z[j][q]); unsafeAtomicAdd(gn + q, dneg * z[j][q]); }
                    }
                } else {
                    rec = make_int4((int)tgt, (int)ng, __float_as_int(dpos), __float_as_int(dneg));
                }
            }
            if (S.rec && !S.ent && ok && g == 0) S.rec[t] = rec;
            if (ok && pos == n - 1) {                       // tail positions of this sequence (zero query): loss terms only
                for (int l = n + g; l < S.L; l += 4) {
                    const int64_t tl = S.target[row * S.L + l];
                    if (S.sample_neg) S.neg_item[(size_t)b * S.L + l] = sample_neg_id(rk0, (uint64_t)b * S.L + l, S.n_items);
                    if (tl > 0 && tl < S.n_items) { lsum += 2.0f * 0.69314718055994530942f; cnt += 1.f; }
                }
            }
        }
        cnt = wave_sum(cnt); lsum = wave_sum(lsum);
        if (lane == 0) { S.part[2 * tile] = cnt; S.part[2 * tile + 1] = lsum; }
        __builtin_amdgcn_sched_barrier(0);
        if (S.ent) wt_tile_sort(S, tile, T, rec, idin, hist);
        __builtin_amdgcn_sched_barrier(0);
        // ---- backward half
        wt_bwd_tile<D, F, Lds>(A, lds + oz, vec + oz, dz, tile, T, dodrop, actdrop, rk);
    }
}

// ------------------------------------------------------------------------------------------------ layer-0 fusions, wave tiles
// k_embqkv_fwd: x = drop(E[idx] + P[pos]) gathered straight into registers (a3: sasrec.py:42-48,:61-66), written once to X[0], and
// multiplied by W_in in the same pass.  LDS: the in_proj image only (55 KB).
template <int D, int WT_WAVES, bool EXACT = true>
__global__ __launch_bounds__(WT_WAVES * 64) void k_wt_embqkv_fwd(const EmbQkvArgs A) {
    typedef WtImg<EXACT, 3 * D, D> In;
    constexpr int DT = D / 16, QT = 3 * D / 16;
    const int T = A.state[DR4SR_STATE_T];
    if ((int)blockIdx.x * 16 >= T) return;
    char* lds = reinterpret_cast<char*>(smem);
    float* vec = reinterpret_cast<float*>(lds + In::bytes);
    In::load(lds, A.W);
    wt_load_v(vec, A.bias, 3 * D);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
#pragma unroll 1
    for (int tile = w * (int)gridDim.x + (int)blockIdx.x; tile * 16 < T; tile += (int)gridDim.x * WT_WAVES) {
        const int oz = wt_opaque_zero();
        const int t = tile * 16 + r16;
        const bool ok = t < T;
        const int tt = ok ? t : T - 1;
        const int b = find_seq_from(A.cu, A.B, tt, A.tile_seq[tile]), pos = tt - A.cu[b];
        const int64_t row = A.rows ? A.rows[b] : b;
        int64_t id = A.idx[row * A.L + pos];
        if (A.idx32 && ok && g == 0) A.idx32[t] = (id > 0 && id < A.n_items) ? (int)id : 0;      // as the backward's scatter tests it
        if (A.tok && ok && g == 0) A.tok[t] = make_int2(t - pos, b | ((A.cu[b + 1] - A.cu[b]) << 20) | (id == 0 ? 1 << 30 : 0));     // attn_tile_sa.hip
        id = id < 0 ? 0 : (id >= A.n_items ? A.n_items - 1 : id);
        f32x4 x[DT], pe[DT];
        wt_row_load<D>(x, A.E + (size_t)id * D, g);
        wt_row_load<D>(pe, A.P + (size_t)pos * D, g);
#pragma unroll
        for (int j = 0; j < DT; ++j) x[j] += pe[j];
        if (dodrop) wt_row_drop<D>(x, rk, DR4SR_SITE_EMB, ((uint64_t)b * A.L + pos) * D, g);
        if (ok) wt_row_store<D>(A.X + (size_t)t * D, x, g);
        f32x4 q[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j) q[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        In::gemm(lds + oz, x, q);
        f32x4 bias[QT];
        wt_row_load<3 * D>(bias, vec + oz, g);
#pragma unroll
        for (int j = 0; j < QT; ++j) q[j] += bias[j];
        if (ok) wt_row_store<3 * D>(A.QKV + (size_t)t * 3 * D, q, g);
    }
}

// k_qkv_embed_bwd (at-scale form): dx0 = dqkv W_in + du1, times the embedding-stage dropout mask, stored for k_wgrad's scatter / owner job
template <int D, int WT_WAVES>
__global__ __launch_bounds__(WT_WAVES * 64) void k_wt_qkv_embed_bwd(const QkvEmbBwdArgs A) {
    typedef WtImg<true, 3 * D, D> In;
    constexpr int DT = D / 16, QT = 3 * D / 16;
    const int T = A.state[DR4SR_STATE_T];
    if ((int)blockIdx.x * 16 >= T) return;
    char* lds = reinterpret_cast<char*>(smem);
    In::load(lds, A.W);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
    const bool dodrop = A.training && A.p > 0.f;
    const RngKey rk = make_rng(A.seed, (uint32_t)A.state[DR4SR_STATE_RNGSTEP], A.p);
#pragma unroll 1
    for (int tile = w * (int)gridDim.x + (int)blockIdx.x; tile * 16 < T; tile += (int)gridDim.x * WT_WAVES) {
        const int oz = wt_opaque_zero();
        const int t = tile * 16 + r16;
        const bool ok = t < T;
        const int tt = ok ? t : T - 1;
        f32x4 dq[QT], gx[DT];
        wt_row_load<3 * D>(dq, A.dQKV + (size_t)tt * 3 * D, g);
        wt_row_load<D>(gx, A.dU1 + (size_t)tt * D, g);
        wt_gemm_xw<3 * D, D>(lds + oz, dq, gx);
        if (dodrop) {
            const int b = find_seq_from(A.cu, A.B, tt, A.tile_seq[tile]), pos = tt - A.cu[b];
            wt_row_drop<D>(gx, rk, DR4SR_SITE_EMB, ((uint64_t)b * A.L + pos) * D, g);
        }
        if (ok) wt_row_store<D>(A.gout + (size_t)t * D, gx, g);
    }
}

int wt_grid() {                                             // one workgroup per CU
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
// waves per workgroup of each kernel (one workgroup per CU): 8 = 2 per SIMD (256 VGPRs), 12 = 3 (168), 16 = 4 (128); tuning knobs
int wt_waves(const char* e, int dflt) {
    const int v = e ? atoi(e) : dflt;
    return v == 16 ? 16 : v == 12 ? 12 : 8;
}
bool wt_exact() {                                           // DR4SR_WT_BF16X3: the bf16x3 split instead of fp32 MFMA (forward kernel only)
    const bool v = DR4SR_XENV("DR4SR_WT_BF16X3") == nullptr;
    return v;
}

template <int W, bool EX>
int wt_post_fwd_launch(const PostArgs& A, int grid, hipStream_t s) {
    const size_t lds = WtFwdLds<EX, 64, 128>::total;
    big_lds(k_wt_post_fwd<64, 128, W, EX>, lds);
    hipLaunchKernelGGL((k_wt_post_fwd<64, 128, W, EX>), dim3(grid), dim3(W * 64), lds, s, A);
    return DR4SR_LAUNCH_CHECK();
}

template <int W>
int wt_post_bwd_launch(const PostArgs& A, int grid, hipStream_t s) {
    const size_t lds = WtBwdLds<64, 128>::total;
    big_lds(k_wt_post_bwd<64, 128, W>, lds);
    hipLaunchKernelGGL((k_wt_post_bwd<64, 128, W>), dim3(grid), dim3(W * 64), lds, s, A);
    return DR4SR_LAUNCH_CHECK();
}

template <int W>
int wt_post_mid_launch(const PostArgs& A, const ScoreTileArgs& S, int grid, hipStream_t s) {
    const size_t lds = WtMidLds<64, 128>::total(W, S.ent ? 1 << S.logG : 0);
    big_lds(k_wt_post_mid<64, 128, W>, lds);
    hipLaunchKernelGGL((k_wt_post_mid<64, 128, W>), dim3(grid), dim3(W * 64), lds, s, A, S);
    return DR4SR_LAUNCH_CHECK();
}

template <int W>
int wt_embqkv_launch(const EmbQkvArgs& A, int grid, hipStream_t s) {
    const size_t lds = WtImg<true, 192, 64>::bytes + 4 * 192;
    if (DR4SR_XENV("DR4SR_WT_EMB_BF16X3")) {                 // the in_proj GEMM as a bf16x3 split (same image bytes)
        big_lds(k_wt_embqkv_fwd<64, W, false>, lds);
        hipLaunchKernelGGL((k_wt_embqkv_fwd<64, W, false>), dim3(grid), dim3(W * 64), lds, s, A);
        return DR4SR_LAUNCH_CHECK();
    }
    big_lds(k_wt_embqkv_fwd<64, W>, lds);
    hipLaunchKernelGGL((k_wt_embqkv_fwd<64, W>), dim3(grid), dim3(W * 64), lds, s, A);
    return DR4SR_LAUNCH_CHECK();
}
template <int W>
int wt_qeb_launch(const QkvEmbBwdArgs& A, int grid, hipStream_t s) {
    const size_t lds = WtImg<true, 192, 64>::bytes;
    big_lds(k_wt_qkv_embed_bwd<64, W>, lds);
    hipLaunchKernelGGL((k_wt_qkv_embed_bwd<64, W>), dim3(grid), dim3(W * 64), lds, s, A);
    return DR4SR_LAUNCH_CHECK();
}

}  // namespace

// the wave-tile forms serve the at-scale regime of the d = 64 / FFN 128 encoder (their LDS image of a d = 128 layer does not fit a CU)
bool wave_tiles(const dr4sr_sasrec_plan* p, const Workspace& ws) {
    const bool off = DR4SR_ENV("DR4SR_NO_WAVE_TILES") != nullptr || DR4SR_ENV("DR4SR_NO_FUSE") != nullptr;
    return !off && ws.scale && p->D == 64 && p->F == 128;
}

bool wt_bwd_on() {
    const bool v = DR4SR_ENV("DR4SR_WT_FWD_ONLY") == nullptr;
    return v;
}

int launch_wt_post_fwd(const PostArgs& A, int Tmax, hipStream_t s) {
    int grid = wt_grid();
    const int tiles = (Tmax + 15) / 16;
    if (grid > tiles) grid = tiles;
    const int W = wt_waves(DR4SR_XENV("DR4SR_WT_FWD_WAVES"), 12);
    if (wt_exact()) return W == 16 ? wt_post_fwd_launch<16, true>(A, grid, s) : W == 12 ? wt_post_fwd_launch<12, true>(A, grid, s) : wt_post_fwd_launch<8, true>(A, grid, s);
    return W == 16 ? wt_post_fwd_launch<16, false>(A, grid, s) : W == 12 ? wt_post_fwd_launch<12, false>(A, grid, s) : wt_post_fwd_launch<8, false>(A, grid, s);
}

int launch_wt_post_bwd(const PostArgs& A, int Tmax, hipStream_t s) {
    int grid = wt_grid();
    const int tiles = (Tmax + 15) / 16;
    if (grid > tiles) grid = tiles;
    const int W = wt_waves(DR4SR_XENV("DR4SR_WT_BWD_WAVES"), 12);
    return W == 16 ? wt_post_bwd_launch<16>(A, grid, s) : W == 12 ? wt_post_bwd_launch<12>(A, grid, s) : wt_post_bwd_launch<8>(A, grid, s);
}

int launch_wt_post_mid(const PostArgs& A, const ScoreTileArgs& S, int Tmax, hipStream_t s) {
    int grid = wt_grid();
    const int tiles = (Tmax + 15) / 16;
    if (grid > tiles) grid = tiles;
    const int wm = wt_waves(DR4SR_XENV("DR4SR_WT_MID_WAVES"), 8);
    return wm == 16 ? wt_post_mid_launch<16>(A, S, grid, s) : wm == 12 ? wt_post_mid_launch<12>(A, S, grid, s) : wt_post_mid_launch<8>(A, S, grid, s);
}

int launch_wt_embqkv_fwd(const EmbQkvArgs& A, int Tmax, hipStream_t s) {
    int grid = wt_grid();
    const int tiles = (Tmax + 15) / 16;
    if (grid > tiles) grid = tiles;
    const int W = wt_waves(DR4SR_XENV("DR4SR_WT_EMB_WAVES"), 16);
    return W == 16 ? wt_embqkv_launch<16>(A, grid, s) : W == 12 ? wt_embqkv_launch<12>(A, grid, s) : wt_embqkv_launch<8>(A, grid, s);
}
int launch_wt_qkv_embed_bwd(const QkvEmbBwdArgs& A, int Tmax, hipStream_t s) {
    int grid = wt_grid();
    const int tiles = (Tmax + 15) / 16;
    if (grid > tiles) grid = tiles;
    const int W = wt_waves(DR4SR_XENV("DR4SR_WT_EMB_WAVES"), 16);
    return W == 16 ? wt_qeb_launch<16>(A, grid, s) : W == 12 ? wt_qeb_launch<12>(A, grid, s) : wt_qeb_launch<8>(A, grid, s);
}
