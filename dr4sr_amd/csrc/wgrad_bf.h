// wgrad_bf.h — weight-gradient GEMM tile job on the bf16 matrix cores as a 3-term split (shared by the SASRec launch of linear.hip and,
// round 4, GRU4Rec's k_wgrad64_bf in gru.hip).
#pragma once
#include "common.h"
#include "kernels.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

// The same job on the bf16 matrix cores as a 3-term split (round 3, at scale): g = gh + gl, x = xh + xl with bf16 parts and
// dW ~ gh xh + gh xl + gl xh accumulated in fp32 (max-norm error 5e-6 of the fp32 result; DR4SR_WGRAD_F32 keeps the fp32 MFMA).  The
// fp32 MFMA form above is bound by the matrix pipe — v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate and shares the SIMD's
// datapath with the VALU — at 32 % of its peak; three v_mfma_f32_32x32x16_bf16 per 16 tokens are 5.3x less matrix time and the job
// becomes what it should be: a stream over the saved activations (HBM).  LDS image of a 64-token tile: word [t / 2][n] packs the bf16
// of tokens t, t + 1 (one v_cvt_pk per pair, written as ds_write_b128 over four columns), hi and lo images side by side — the same
// bytes as the fp32 tile; an MFMA operand (8 consecutive tokens of one column) is four ds_read_b32.  Bias sums stay exact fp32: every
// thread sums the columns it loads, over all its tiles, and the partials meet in LDS once at the end.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_pack(const float a, const float b, unsigned& hi, unsigned& lo) {
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    const wg_bf16x2 h = {ha, hb}, l = {(__bf16)(a - (float)ha), (__bf16)(b - (float)hb)};
    hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}
// DEEP: the operand rows of TWO token tiles are in flight (two register sets, the loop unrolled by two): with one set the loads of tile
// i + 1 fly only during the MFMA phase of tile i (a few hundred cycles) and are waited for right behind it
// part (deterministic mode, round 5): instead of adding its [NG x KX] partial product and its NG bias sums into the gradient with fp32 atomics
// — whose arrival order differs from run to run — the workgroup STORES them, row-major + the bias row behind, at `part`; k_wgrad_det_reduce
// (linear.hip) sums the token splits' blocks in split order.
template <int NG, int KX, bool DEEP = false>
__device__ __forceinline__ void wgrad_body_bf(const WgradJob& J, const int* __restrict__ state, float* __restrict__ part = nullptr) {
    constexpr int NT = NG / 32, KT = KX / 32, TPW = (NT * KT) / 4;
    static_assert((NT * KT) % 4 == 0, "tile count must split over 4 waves");
    const int T = state[DR4SR_STATE_T];
    const int ntiles = (T + 63) / 64;
    unsigned* GH = reinterpret_cast<unsigned*>(smem);      // [32][NG] token pairs x columns
    unsigned* GL = GH + 32 * NG;
    unsigned* XH = GL + 32 * NG;                           // [32][KX]
    unsigned* XL = XH + 32 * KX;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, kg = lane >> 5;
    f32x16 acc[TPW];
    acc_zero(acc);
    constexpr int GP = NG / 32, XP = KX / 32;              // (token pair, 4 columns) items per thread per tile
    struct LoadSet { float4 g0[GP], g1[GP], x0[XP], x1[XP]; };
    LoadSet SA, SB;
    float4 bs[GP];
#pragma unroll
    for (int u = 0; u < GP; ++u) bs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](LoadSet& S, int tt) {                 // rows past T: the last row again, zeroed in commit
        const int t0 = tt * 64;
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int i = threadIdx.x + 256 * u, pr = i / (NG / 4), c = (i % (NG / 4)) * 4;
            const int ta = min(t0 + 2 * pr, T - 1), tb = min(t0 + 2 * pr + 1, T - 1);
            S.g0[u] = ld4(J.G + (size_t)ta * J.ldg + J.gcol + c);
            S.g1[u] = ld4(J.G + (size_t)tb * J.ldg + J.gcol + c);
        }
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int i = threadIdx.x + 256 * u, pr = i / (KX / 4), c = (i % (KX / 4)) * 4;
            const int ta = min(t0 + 2 * pr, T - 1), tb = min(t0 + 2 * pr + 1, T - 1);
            S.x0[u] = ld4(J.X + (size_t)ta * J.ldx + c);
            S.x1[u] = ld4(J.X + (size_t)tb * J.ldx + c);
        }
    };
    auto commit = [&](const LoadSet& S, int tt) {
        const int t0 = tt * 64;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < GP; ++u) {
            const int i = threadIdx.x + 256 * u, pr = i / (NG / 4), c = (i % (NG / 4)) * 4;
            const float4 a = t0 + 2 * pr < T ? S.g0[u] : z4, b = t0 + 2 * pr + 1 < T ? S.g1[u] : z4;
            bs[u].x += a.x + b.x; bs[u].y += a.y + b.y; bs[u].z += a.z + b.z; bs[u].w += a.w + b.w;
            uint4 h, l;
            wg_pack(a.x, b.x, h.x, l.x); wg_pack(a.y, b.y, h.y, l.y); wg_pack(a.z, b.z, h.z, l.z); wg_pack(a.w, b.w, h.w, l.w);
            *reinterpret_cast<uint4*>(GH + pr * NG + c) = h;
            *reinterpret_cast<uint4*>(GL + pr * NG + c) = l;
        }
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int i = threadIdx.x + 256 * u, pr = i / (KX / 4), c = (i % (KX / 4)) * 4;
            const float4 a = t0 + 2 * pr < T ? S.x0[u] : z4, b = t0 + 2 * pr + 1 < T ? S.x1[u] : z4;
            uint4 h, l;
            wg_pack(a.x, b.x, h.x, l.x); wg_pack(a.y, b.y, h.y, l.y); wg_pack(a.z, b.z, h.z, l.z); wg_pack(a.w, b.w, h.w, l.w);
            *reinterpret_cast<uint4*>(XH + pr * KX + c) = h;
            *reinterpret_cast<uint4*>(XL + pr * KX + c) = l;
        }
    };
    auto frag = [&](const unsigned* img, int ld, int ks, int col) {   // 8 consecutive tokens 16 ks + 8 kg .. + 7 of one column
        const unsigned* p = img + (8 * ks + 4 * kg) * ld + col;
        const uint4 v = make_uint4(p[0], p[ld], p[2 * ld], p[3 * ld]);
        return __builtin_bit_cast(wg_bf16x8, v);
    };
    auto mfma_phase = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int q = w + 4 * i, nt = q / KT, kt = q % KT;
                const wg_bf16x8 ah = frag(GH, NG, ks, nt * 32 + r), al = frag(GL, NG, ks, nt * 32 + r);
                const wg_bf16x8 bh = frag(XH, KX, ks, kt * 32 + r), bl = frag(XL, KX, ks, kt * 32 + r);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
            }
        }
    };
    int tt = blockIdx.x;
    if (tt >= ntiles) return;
    const int st = (int)gridDim.x;
    issue(SA, tt);
    if constexpr (DEEP) {
        if (tt + st < ntiles) issue(SB, tt + st);
        for (;;) {
            lds_barrier();                                 // previous MFMA phase has finished reading LDS
            commit(SA, tt);
            lds_barrier();
            if (tt + 2 * st < ntiles) issue(SA, tt + 2 * st);
            mfma_phase();
            tt += st;
            if (tt >= ntiles) break;
            lds_barrier();
            commit(SB, tt);
            lds_barrier();
            if (tt + 2 * st < ntiles) issue(SB, tt + 2 * st);
            mfma_phase();
            tt += st;
            if (tt >= ntiles) break;
        }
    } else {
        for (; tt < ntiles; tt += st) {
            lds_barrier();                                 // previous MFMA phase has finished reading LDS
            commit(SA, tt);
            lds_barrier();
            if (tt + st < ntiles) issue(SA, tt + st);
            mfma_phase();
        }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int q = w + 4 * i, nt = q / KT, kt = q % KT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (part) part[(size_t)row * KX + kt * 32 + r] = acc[i][e];
            else unsafeAtomicAdd(J.dW + (size_t)row * (J.ldw ? J.ldw : KX) + kt * 32 + r, acc[i][e]);
        }
    }
    if (!J.db) return;                                     // (uniform per workgroup)
    // bias: this thread's column sums -> LDS [256 / (NG / 4)][NG] -> one atomic per column
    lds_barrier();
    float* bl = smem;
#pragma unroll
    for (int u = 0; u < GP; ++u) {
        const int i = threadIdx.x + 256 * u, pr = i / (NG / 4), c = (i % (NG / 4)) * 4;
        st4(bl + pr * NG + c, bs[u]);                      // pr < 32 distinct (pair row, column quad) slots per u... one slot per item
    }
    lds_barrier();
    for (int n = threadIdx.x; n < NG; n += 256) {
        float sum = 0.f;
        for (int pr = 0; pr < 32; ++pr) sum += bl[pr * NG + n];
        if (part) part[(size_t)NG * KX + n] = sum;
        else unsafeAtomicAdd(J.db + n, sum);
    }
}

