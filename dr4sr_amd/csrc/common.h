// common.h — device helpers shared by the gfx950 kernels (wave64, MFMA f32 32x32x2, Philox).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dr4sr_hip.h"
#include "../../include/dr4sr_hip_hooks.h"
#include <stdlib.h>
#include <string.h>

// Environment switches: every site caches its variable's value and re-reads it when dr4sr_reload_env() (hooks header) has bumped the
// generation — process-lifetime constants in production, switchable in-process by the tests.  NOTE for callers: changing os.environ /
// setenv AFTER the first call of an entry point has no effect until dr4sr_reload_env() is called (dr4sr_amd/_lib.py: set_env(name,
// value) does both).  The value is COPIED (a later setenv may free the string getenv returned; values longer than 255 bytes are cut —
// DR4SR_LIB_PATH-like paths are read by the binding, not here).  The refresh is serialised by one mutex (two host threads entering the
// same site after a reload), the fast path is one relaxed atomic load.  DR4SR_ENV("X") -> const char* or nullptr.
#include <atomic>
#include <mutex>
int dr4sr_env_generation();                                   // step.hip
std::mutex& dr4sr_env_mutex();                                // step.hip
#define DR4SR_ENV(NAME) ([]() -> const char* {                                                   \
    static std::atomic<int> gen_{-1}; static bool set_ = false; static char buf_[256];           \
    const int cur_ = dr4sr_env_generation();                                                     \
    if (gen_.load(std::memory_order_acquire) != cur_) {                                          \
        std::lock_guard<std::mutex> lk_(dr4sr_env_mutex());                                      \
        if (gen_.load(std::memory_order_relaxed) != cur_) { const char* e_ = getenv(NAME); set_ = e_ != nullptr; \
            if (e_) { strncpy(buf_, e_, sizeof(buf_) - 1); buf_[sizeof(buf_) - 1] = 0; }         \
            gen_.store(cur_, std::memory_order_release); } }                                     \
    return set_ ? buf_ : nullptr; }())

// Experiment / tuning switches (round 6): read only by a library built with -DDR4SR_EXPERIMENTS (`make EXPERIMENTS=1` -> libdr4sr_hip_exp.so).
// In the shipped build they are a compile-time nullptr: the rejected experiments and the tuning knobs (the defaults are the measured
// optima) carry neither their branches nor their kernels' ISA into the product, and cost no run-time lookups on the launch path.  The
// cross-check switches the tests use stay run-time (DR4SR_ENV).  SWITCHES.md lists both kinds; dr4sr_build_flags() bit 0 says which build.
#ifdef DR4SR_EXPERIMENTS
#define DR4SR_XENV(NAME) DR4SR_ENV(NAME)
#else
#define DR4SR_XENV(NAME) (static_cast<const char*>(nullptr))
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DR4SR_WAVE 64

// ---------------------------------------------------------------------------------- Philox4x32-10
// Counter-based RNG: the mask of element e at (seed, step, site) is a pure function, so backward
// kernels regenerate forward masks instead of storing them.  4 consecutive elements share one call.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per constant (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: integer multiplies are
        // quarter-rate, and this loop is the largest VALU item of every kernel that drops out (round 6: 40 -> 20 of them per call)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

struct RngKey {
    uint32_t seed_lo, seed_hi, step;
    float p;           // drop probability
    float scale;       // 1/(1-p)
    uint32_t thresh;   // 16-bit decisions: keep iff r16 >= thresh (thresh = round(p * 65536), <= 65535)
};

__device__ __forceinline__ RngKey make_rng(uint64_t seed, uint32_t step, float p) {
    RngKey k;
    k.seed_lo = (uint32_t)seed;
    k.seed_hi = (uint32_t)(seed >> 32);
    k.step = step;
    k.p = p;
    k.scale = 1.0f / (1.0f - p);
    const float t = p * 65536.0f + 0.5f;
    k.thresh = t >= 65535.0f ? 65535u : (uint32_t)t;
    return k;
}

// random words of call `call` of stream (site)
__device__ __forceinline__ uint4 rng_call(const RngKey& k, uint32_t site, uint64_t call) {
    return philox4x32_10(make_uint4((uint32_t)call, (uint32_t)(call >> 32), site, k.step),
                         make_uint2(k.seed_lo, k.seed_hi));
}

// Dropout decisions are 16 bits wide (round 3): one Philox call covers EIGHT consecutive elements — element e takes half (e & 1) of
// word ((e >> 1) & 3) of call e >> 3 — instead of four 32-bit ones.  Philox is the largest VALU item of the token-tile kernels
// (40 quarter-rate integer multiplies per call; fp32 MFMA and VALU share a SIMD's datapath on gfx950 — tools/probes/
// mfma_valu_overlap_probe.hip — so every VALU cycle is a cycle the matrix pipe idles): kernels whose lanes own 8 consecutive
// columns (csrc/linear_wave.hip) halve their calls, the float4 kernels keep one call per float4.  The drop probability is quantised
// to 1 / 65 536 (|p' - p| < 8e-6; p = 0.5 exact); the keep scale stays the reference's 1 / (1 - p).
__device__ __forceinline__ float keep16(const RngKey& k, uint32_t w, int half) {
    return ((half ? (w >> 16) : (w & 0xffffu)) >= k.thresh) ? k.scale : 0.f;
}
// multiplicative keep factors (0 or 1/(1-p)) for 8 consecutive elements starting at e (e % 8 == 0): lo = e..e+3, hi = e+4..e+7
__device__ __forceinline__ void drop8(const RngKey& k, uint32_t site, uint64_t e, float4& lo, float4& hi) {
    const uint4 r = rng_call(k, site, e >> 3);
    lo = make_float4(keep16(k, r.x, 0), keep16(k, r.x, 1), keep16(k, r.y, 0), keep16(k, r.y, 1));
    hi = make_float4(keep16(k, r.z, 0), keep16(k, r.z, 1), keep16(k, r.w, 0), keep16(k, r.w, 1));
}
// ... for 4 consecutive elements starting at e (e % 4 == 0): the matching half of the call
__device__ __forceinline__ float4 drop4(const RngKey& k, uint32_t site, uint64_t e) {
    const uint4 r = rng_call(k, site, e >> 3);
    const bool up = (e >> 2) & 1;
    const uint32_t w0 = up ? r.z : r.x, w1 = up ? r.w : r.y;
    return make_float4(keep16(k, w0, 0), keep16(k, w0, 1), keep16(k, w1, 0), keep16(k, w1, 1));
}

// the 8 keep decisions of the call that covers elements e..e+7 (e % 8 == 0) as a bit mask: bit k = element e + k is kept
__device__ __forceinline__ unsigned drop_bits8(const RngKey& k, uint32_t site, uint64_t e) {
    const uint4 r = rng_call(k, site, e >> 3);
    unsigned m = 0;
    m |= ((r.x & 0xffffu) >= k.thresh) ? 1u : 0u;   m |= ((r.x >> 16) >= k.thresh) ? 2u : 0u;
    m |= ((r.y & 0xffffu) >= k.thresh) ? 4u : 0u;   m |= ((r.y >> 16) >= k.thresh) ? 8u : 0u;
    m |= ((r.z & 0xffffu) >= k.thresh) ? 16u : 0u;  m |= ((r.z >> 16) >= k.thresh) ? 32u : 0u;
    m |= ((r.w & 0xffffu) >= k.thresh) ? 64u : 0u;  m |= ((r.w >> 16) >= k.thresh) ? 128u : 0u;
    return m;
}

// keep factor of a single element (recomputes the shared call; use only off the hot path)
__device__ __forceinline__ float drop1(const RngKey& k, uint32_t site, uint64_t e) {
    const uint4 r = rng_call(k, site, e >> 3);
    const uint32_t c = (uint32_t)((e >> 1) & 3);
    const uint32_t w = c == 0 ? r.x : c == 1 ? r.y : c == 2 ? r.z : r.w;
    return keep16(k, w, (int)(e & 1));
}

// a2 negative sampler (basemodel.py:50-61): element e of the (seed, step) stream
#define DR4SR_SITE_NEG 0x4e454721u     // RNG stream of the negative sampler

__device__ __forceinline__ int64_t sample_neg_id(const RngKey& rk, uint64_t e, int n_items) {
    const uint4 r = rng_call(rk, DR4SR_SITE_NEG, e >> 2);
    const uint32_t c = (uint32_t)(e & 3);
    const uint32_t w = c == 0 ? r.x : c == 1 ? r.y : c == 2 ? r.z : r.w;
    return 1 + (int64_t)__umulhi(w, (uint32_t)(n_items - 1));       // uniform on [1, n_items-1]
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (global stores count on CDNA4), which
// makes every phase of the tile kernels wait ~1-2 us for its activation stores to be acknowledged; nothing in these kernels
// communicates between waves through global memory, so LDS ordering (lgkmcnt) + s_barrier is sufficient.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sequence slot of packed token t: the b with cu[b] <= t < cu[b+1] (binary search; cu is L1/L2 resident)
__device__ __forceinline__ int find_seq(const int* __restrict__ cu, int B, int t) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// same, starting from a hint b0 <= answer (k_prep's tile_seq[t >> 4]): a short forward walk over L1-resident entries instead
// of log2(B) dependent loads
__device__ __forceinline__ int find_seq_from(const int* __restrict__ cu, int B, int t, int b0) {
    int b = b0;
    while (b + 1 < B && cu[b + 1] <= t) ++b;
    return b;
}

// ---------------------------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32-rounding class) on v_rcp/v_exp: ~15 VALU instead of
// libm erff's ~60 — the GELU row passes are VALU-latency-bound with one wave per SIMD.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// GRU gate nonlinearities on v_exp (the recurrences are latency-bound: libm's expf/tanhf cost 3-4x the instructions)
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { const float e = __expf(-2.0f * fabsf(x)); return copysignf((1.0f - e) / (1.0f + e), x); }
// softplus(x) = log(1 + e^x), stable
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------- MFMA tile GEMM
// C[64 x N] = A[64 x K] * W^T, A in LDS (row stride lda floats, 16-B aligned rows), W global [N][K].
// 256 threads = 4 waves; wave w owns row-half rh = w&1 and column tiles ct = (w>>1) + 2*i, i<NTW,
// N = 64*NTW.  v_mfma_f32_32x32x2_f32: lane (r = l&31, g = l>>5) feeds A[r][k], B[k][r] for one k
// per instruction; the k ORDER is permuted (g takes k in [g*K/2, (g+1)*K/2)) so every lane streams
// contiguous floats (ds_read_b128 / global_load_dwordx4) — the sum over k is order-free.
template <int K, int NTW>
__device__ __forceinline__ void mma_64xN(const float* __restrict__ As, int lda,
                                         const float* __restrict__ W, f32x16 (&acc)[NTW]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5, rh = w & 1, cg = w >> 1;
    const float* arow = As + (rh * 32 + r) * lda + g * (K / 2);
    const float* wrow[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) wrow[i] = W + (size_t)((cg + 2 * i) * 32 + r) * K + g * (K / 2);
#pragma unroll
    for (int c = 0; c < K / 2; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + c);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(wrow[i] + c);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
        }
    }
}

// same as mma_64xN with an explicit weight row stride (K-chunked GEMMs: W points at the chunk's first column)
template <int K, int NTW>
__device__ __forceinline__ void mma_64xN_ld(const float* __restrict__ As, int lda, const float* __restrict__ W, int ldw,
                                            f32x16 (&acc)[NTW]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5, rh = w & 1, cg = w >> 1;
    const float* arow = As + (rh * 32 + r) * lda + g * (K / 2);
#pragma unroll
    for (int c = 0; c < K / 2; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + c);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(W + (size_t)((cg + 2 * i) * 32 + r) * ldw + g * (K / 2) + c);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
        }
    }
}

// Data-gradient form:  C[64 x N] = A[64 x KR] * W  with W global [KR][ldw] row-major (the forward weight [out][in] used
// as-is: KR = out features = reduction, N = in features).  B operand lane (r, g) at step kk reads W[g*KR/2 + kk][ct*32 + r]:
// a 128-B-coalesced dword per half-wave; A as in mma_64xN.  Removes the per-step transposed weight copies.
template <int KR, int NTW>
__device__ __forceinline__ void mma_64xN_wT(const float* __restrict__ As, int lda, const float* __restrict__ W, int ldw,
                                            f32x16 (&acc)[NTW]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5, rh = w & 1, cg = w >> 1;
    const float* arow = As + (rh * 32 + r) * lda + g * (KR / 2);
    const float* wcol = W + (size_t)(g * (KR / 2)) * ldw + cg * 32 + r;
#pragma unroll
    for (int c = 0; c < KR / 2; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + c);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const float* wp = wcol + (size_t)c * ldw + 64 * i;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wp[0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wp[ldw], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wp[2 * ldw], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wp[3 * ldw], acc[i], 0, 0, 0);
        }
    }
}

// accumulators (+ bias[n]) -> LDS C tile [64][ldc].  C/D map of 32x32: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <int NTW>
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[NTW], float* __restrict__ Cs, int ldc,
                                           const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5, rh = w & 1, cg = w >> 1;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int col = (cg + 2 * i) * 32 + r;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = rh * 32 + (q & 3) + 8 * (q >> 2) + 4 * g;
            Cs[row * ldc + col] = acc[i][q] + bv;
        }
    }
}

// accumulators (+ bias) -> global C rows directly (C layout: 32 consecutive columns per half-wave = 128-B segments)
template <int NTW>
__device__ __forceinline__ void acc_to_global(const f32x16 (&acc)[NTW], float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                              int t0, int T) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5, rh = w & 1, cg = w >> 1;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int col = (cg + 2 * i) * 32 + r;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int t = t0 + rh * 32 + (q & 3) + 8 * (q >> 2) + 4 * g;
            if (t < T) C[(size_t)t * ldc + col] = acc[i][q] + bv;
        }
    }
}

template <int NTW>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NTW]) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
}

// ---------------------------------------------------------------------------------- tile-size abstraction
// TileAcc<BM, N>: accumulators of a [BM x N] output tile owned by a 256-thread workgroup.
//   BM = 64     : v_mfma_f32_32x32x2_f32, wave = (row half, column group), f32x16 per 32x32 tile       (throughput tiles)
//   BM = 16 / 32: v_mfma_f32_16x16x4_f32, wave w owns column tiles w + 4i of every 16-row tile         (latency / occupancy tiles:
//                 4x / 2x less serial MFMA work per workgroup, 4x / 2x less LDS => more workgroups per CU)
__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int BM, int N> struct TileAcc { f32x4 a[BM / 16][N / 64]; };
template <int N> struct TileAcc<64, N> { f32x16 a[N / 64]; };

template <int BM, int N> __device__ __forceinline__ void tile_zero(TileAcc<BM, N>& t) {
    if constexpr (BM == 64) acc_zero(t.a);
    else {
#pragma unroll
        for (int r = 0; r < BM / 16; ++r)
#pragma unroll
            for (int i = 0; i < N / 64; ++i) t.a[r][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
// C += A[BM x K] * W^T, W global [N][ldw]
template <int BM, int K, int N>
__device__ __forceinline__ void tile_mma_xwT(const float* __restrict__ As, int lda, const float* __restrict__ W, int ldw, TileAcc<BM, N>& t) {
    if constexpr (BM == 64) mma_64xN_ld<K, N / 64>(As, lda, W, ldw, t.a);
    else {
        constexpr int KQ = K / 4, RT = BM / 16, CW = N / 64;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int c = 0; c < KQ; c += 4) {
            float4 a[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = ld4(As + (r * 16 + r16) * lda + g * KQ + c);
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                const float4 b = ld4(W + (size_t)((w + 4 * i) * 16 + r16) * ldw + g * KQ + c);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    t.a[r][i] = mfma16x4(a[r].x, b.x, t.a[r][i]); t.a[r][i] = mfma16x4(a[r].y, b.y, t.a[r][i]);
                    t.a[r][i] = mfma16x4(a[r].z, b.z, t.a[r][i]); t.a[r][i] = mfma16x4(a[r].w, b.w, t.a[r][i]);
                }
            }
        }
    }
}
// C += A[BM x KR] * W, W global [KR][ldw] (data-gradient form: forward weight read column-wise)
template <int BM, int KR, int N>
__device__ __forceinline__ void tile_mma_xw(const float* __restrict__ As, int lda, const float* __restrict__ W, int ldw, TileAcc<BM, N>& t) {
    if constexpr (BM == 64) mma_64xN_wT<KR, N / 64>(As, lda, W, ldw, t.a);
    else {
        constexpr int KQ = KR / 4, RT = BM / 16, CW = N / 64;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int c = 0; c < KQ; c += 4) {
            float4 a[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) a[r] = ld4(As + (r * 16 + r16) * lda + g * KQ + c);
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                const float* wp = W + (size_t)(g * KQ + c) * ldw + (w + 4 * i) * 16 + r16;
                const float b0 = wp[0], b1 = wp[ldw], b2 = wp[2 * ldw], b3 = wp[3 * ldw];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    t.a[r][i] = mfma16x4(a[r].x, b0, t.a[r][i]); t.a[r][i] = mfma16x4(a[r].y, b1, t.a[r][i]);
                    t.a[r][i] = mfma16x4(a[r].z, b2, t.a[r][i]); t.a[r][i] = mfma16x4(a[r].w, b3, t.a[r][i]);
                }
            }
        }
    }
}

// ---- the same tile GEMM on the bf16 matrix cores as a 3-term split (round 5; d = 128 at scale).  The 256-thread tile kernels of a
// d = 128 layer are bound by the fp32 matrix pipe — v_mfma_f32_16x16x4_f32 issues every 32 cycles per SIMD and shares the datapath with
// the VALU: issuing HALF of them (timing probe) took the B = 8 192 step from 1.382 to 1.236 ms.  a = ah + al, w = wh + wl with bf16 parts,
// a w ~ al wh + ah wl + ah wh accumulated in fp32 (the split k_wgrad_bf has used since round 3: max-norm error 5e-6 of the fp32 product):
// three v_mfma_f32_16x16x32_bf16 (17 cycles each) replace EIGHT fp32 MFMAs (256 cycles).  The A operand is split in registers from the
// fp32 LDS tile (the row passes keep reading that tile as fp32); the weights come pre-split from an image k_wsplit writes once per step:
// bf16 high parts of the [N][K] matrix in fragment-major order (bf3_frag_off), the low parts `lo` elements behind, same k-permuted map as
// tile_mma_xwT (lane group g owns k in [g K/4, (g+1) K/4): A and B use the same map, so the contraction is unchanged).  Data-gradient GEMMs (tile_mma_xw: C += A W) run as
// this form on the TRANSPOSED image.
typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf3_split8(const float4 p, const float4 q, t_bf16x8& hi, t_bf16x8& lo) {
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}
// FRAGMENT-MAJOR image: the 8 elements lane (g, r16) feeds to one MFMA — column n = 16 ct + r16, k = g K/4 + 8 q .. + 7 — are contiguous, the 64
// lanes of a (column tile ct, chunk q) pair follow each other: a wave's operand load is ONE contiguous 1 KB block (16 full 64-byte
// sectors).  With the plain [N][K] image the same instruction touched 64 sectors for 1 KB of payload, and the weight stream — every
// 32-token tile re-reads the layer's 393 KB — was what bounded these kernels (loads removed, timing probe: 1.282 -> 1.068 ms).
__host__ __device__ __forceinline__ size_t bf3_frag_off(const int n, const int k, const int K) {
    const int KQ = K / 4, g = k / KQ, rem = k % KQ;
    return ((size_t)((n >> 4) * (KQ / 8) + (rem >> 3)) * 64 + (g * 16 + (n & 15))) * 8 + (rem & 7);
}
// one element of a layer's split-weight images (k_wsplit, and the extra blocks of FMLP's prep launch): element e of a part -> its matrix
// [R][C] (IN 3D x D | OUT D x D | W1 F x D | W2 D x F; o_in < 0: no IN / OUT), bf16 high / low parts into both orientations
__device__ __forceinline__ void wsplit_elem(const float* __restrict__ params, unsigned short* __restrict__ base, const int e, const int64_t o_in,
                                            const int64_t o_out, const int64_t o_w1, const int64_t o_w2, const int64_t layer_off, const int E,
                                            const int D, const int F) {
    if (e < 4 * D * D && o_in < 0) return;
    const float* src; int R, C, m_off;
    if (e < 3 * D * D) { src = params + o_in; R = 3 * D; C = D; m_off = 0; }
    else if (e < 4 * D * D) { src = params + o_out; R = D; C = D; m_off = 3 * D * D; }
    else if (e < 4 * D * D + F * D) { src = params + o_w1; R = F; C = D; m_off = 4 * D * D; }
    else { src = params + o_w2; R = D; C = F; m_off = 4 * D * D + F * D; }
    const int i = e - m_off, r = i / C, c = i % C;
    const float v = src[layer_off + i];
    const __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
    const unsigned short hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
    const size_t a = m_off + bf3_frag_off(r, c, C);                           // as stored: N = R rows, K = C
    base[a] = hb; base[(size_t)E + a] = lb;
    const size_t t = (size_t)2 * E + m_off + bf3_frag_off(c, r, R);           // transposed: N = C rows, K = R
    base[t] = hb; base[t + E] = lb;
}
template <int BM, int K, int N>
__device__ __forceinline__ void tile_mma_xwT_bf3(const float* __restrict__ As, int lda, const unsigned short* __restrict__ Wh, const int lo,
                                                 TileAcc<BM, N>& t) {
    static_assert((BM == 16 || BM == 32) && K % 32 == 0, "16x16x32 bf16 path");
    constexpr int KQ = K / 4, RT = BM / 16, CW = N / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int c = 0; c < KQ; c += 8) {
        t_bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const float* ap = As + (r * 16 + r16) * lda + g * KQ + c;
            bf3_split8(ld4(ap), ld4(ap + 4), ah[r], al[r]);
        }
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const unsigned short* wp = Wh + bf3_frag_off((w + 4 * i) * 16 + r16, g * KQ + c, K);     // one contiguous 1 KB block per load instruction
            const t_bf16x8 bh = *reinterpret_cast<const t_bf16x8*>(wp), bl = *reinterpret_cast<const t_bf16x8*>(wp + lo);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                t.a[r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[r], bh, t.a[r][i], 0, 0, 0);
                t.a[r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bl, t.a[r][i], 0, 0, 0);
                t.a[r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r], bh, t.a[r][i], 0, 0, 0);
            }
        }
    }
}
// the split-weight images (Workspace::wsplit): per layer 4 E bf16 with E = 4 D^2 + 2 D F — orientation o (0: [out][in] as stored, 1:
// transposed [in][out]) starts at o * 2 E, high parts first, low parts E behind; matrices at IN 0 | OUT 3 D^2 | W1 4 D^2 | W2 4 D^2 + F D.
// One pointer in the argument blocks (NULL: fp32 MFMA path); layer + 1's block follows at + 4 E.
template <int D, int F> struct WSplitGeo { static constexpr int E = 4 * D * D + 2 * D * F, OUT = 3 * D * D, W1 = 4 * D * D, W2 = 4 * D * D + F * D; };
// C += A W^T (transposed = false: W [N][K] as stored) or C += A W (transposed = true: W [K][N] as stored -> its [N][K] transposed image);
// sp = the layer's split-weight block or NULL, E = elements per part, m_off = the matrix's offset in a part
template <bool BF3, int BM, int K, int N>
__device__ __forceinline__ void tile_gemm(const float* __restrict__ As, int lda, const float* __restrict__ W, int ldw, bool transposed,
                                          const unsigned short* __restrict__ sp, int E, int m_off, TileAcc<BM, N>& t) {
    if constexpr (BF3 && (BM == 16 || BM == 32) && K % 32 == 0) {
        if (sp) { tile_mma_xwT_bf3<BM, K, N>(As, lda, sp + (transposed ? 2 * (size_t)E : 0) + m_off, E, t); return; }
    }
    if (transposed) tile_mma_xw<BM, K, N>(As, lda, W, ldw, t);
    else tile_mma_xwT<BM, K, N>(As, lda, W, ldw, t);
}

// Register-resident B-operand fragments of the BM = 16 / 32 tile GEMMs.  Loading them is decoupled from the MFMA loop so that
// a latency-bound kernel can issue ALL its weight loads up front (they depend on nothing) and overlap their L2 round trips
// with the phases before the GEMM that consumes them.
template <int K, int N> struct WFragT { float4 b[K / 16][N / 64]; };     // for C += A W^T, W [N][ldw]  (tile_mma_xwT addressing)
template <int K, int N> struct WFragC { float4 b[K / 16][N / 64]; };     // for C += A W,   W [K][ldw]  (tile_mma_xw addressing)
// The same fragments from a FRAGMENT-MAJOR fp32 image of the matrix (round 5; latency forms at d = 128): the float4 that lane `lane` of wave w
// feeds to chunk ci of column tile ct sits at ((ct K/16 + ci) 64 + lane) 4 floats, so a load instruction reads one contiguous 1 KB block
// (16 full 64-byte sectors) where the row-major addressing below touches 64 sectors (16 rows x 4 lane groups) — the kernels of the latency
// regime spend microseconds ISSUING their fragment requests (round 4 stamps).  Written by the first launch of the forward pass
// (linear.hip wfrag_image_write).  Matrix offsets inside a layer's image: WSplitGeo (IN 0 | OUT 3 D^2 | W1 4 D^2 | W2 4 D^2 + F D).
template <int K, int N>
__device__ __forceinline__ void wfrag_load_img(WFragT<K, N>& f, const float* __restrict__ img) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int ci = 0; ci < K / 16; ++ci)
#pragma unroll
        for (int i = 0; i < N / 64; ++i) f.b[ci][i] = ld4(img + ((size_t)((w + 4 * i) * (K / 16) + ci) * 64 + lane) * 4);
}
template <int K, int N>
__device__ __forceinline__ void wfrag_load(WFragT<K, N>& f, const float* __restrict__ W, int ldw) {
    constexpr int KQ = K / 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ci = 0; ci < K / 16; ++ci)
#pragma unroll
        for (int i = 0; i < N / 64; ++i) f.b[ci][i] = ld4(W + (size_t)((w + 4 * i) * 16 + r16) * ldw + g * KQ + 4 * ci);
}
template <int K, int N>
__device__ __forceinline__ void wfrag_load(WFragC<K, N>& f, const float* __restrict__ W, int ldw) {
    constexpr int KQ = K / 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ci = 0; ci < K / 16; ++ci)
#pragma unroll
        for (int i = 0; i < N / 64; ++i) {
            const float* wp = W + (size_t)(g * KQ + 4 * ci) * ldw + (w + 4 * i) * 16 + r16;
            f.b[ci][i] = make_float4(wp[0], wp[ldw], wp[2 * ldw], wp[3 * ldw]);
        }
}
template <int BM, int K, int N, typename FRAG>
__device__ __forceinline__ void tile_mma_frag(const float* __restrict__ As, int lda, const FRAG& f, TileAcc<BM, N>& t) {
    static_assert(BM == 16 || BM == 32, "fragment GEMM is the 16x16x4 path");
    constexpr int KQ = K / 4, RT = BM / 16, CW = N / 64;
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ci = 0; ci < K / 16; ++ci) {
        float4 a[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) a[r] = ld4(As + (r * 16 + r16) * lda + g * KQ + 4 * ci);
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const float4 b = f.b[ci][i];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                t.a[r][i] = mfma16x4(a[r].x, b.x, t.a[r][i]); t.a[r][i] = mfma16x4(a[r].y, b.y, t.a[r][i]);
                t.a[r][i] = mfma16x4(a[r].z, b.z, t.a[r][i]); t.a[r][i] = mfma16x4(a[r].w, b.w, t.a[r][i]);
            }
        }
    }
}
template <int BM, int N>
__device__ __forceinline__ void tile_to_lds(const TileAcc<BM, N>& t, float* __restrict__ Cs, int ldc, const float* __restrict__ bias) {
    if constexpr (BM == 64) acc_to_lds(t.a, Cs, ldc, bias);
    else {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < N / 64; ++i) {
            const int col = (w + 4 * i) * 16 + r16;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < BM / 16; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) Cs[(r * 16 + 4 * g + q) * ldc + col] = t.a[r][i][q] + bv;
        }
    }
}
template <int BM, int N>
__device__ __forceinline__ void tile_to_global(const TileAcc<BM, N>& t, float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                               int t0, int T) {
    if constexpr (BM == 64) acc_to_global(t.a, C, ldc, bias, t0, T);
    else {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < N / 64; ++i) {
            const int col = (w + 4 * i) * 16 + r16;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < BM / 16; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tt = t0 + r * 16 + 4 * g + q;
                    if (tt < T) C[(size_t)tt * ldc + col] = t.a[r][i][q] + bv;
                }
        }
    }
}

// cooperative load of a [BM x K] tile (rows t0.. of a [T x ldg] global matrix, zero beyond T) into LDS, row stride lda
template <int BM, int K>
__device__ __forceinline__ void load_tile_bm(float* __restrict__ As, int lda, const float* __restrict__ G, int ldg, int t0, int T) {
    constexpr int C4 = K / 4;
    for (int i = threadIdx.x; i < BM * C4; i += 256) {
        const int row = i / C4, c = (i % C4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 + row < T) v = ld4(G + (size_t)(t0 + row) * ldg + c);
        st4(As + row * lda + c, v);
    }
}

// cooperative load of a [64 x K] tile (rows t0..t0+63 of a [T x ldg] global matrix, zero beyond T)
// into LDS with row stride lda.  256 threads, float4 per thread per pass.
template <int K>
__device__ __forceinline__ void load_tile(float* __restrict__ As, int lda, const float* __restrict__ G,
                                          int ldg, int t0, int T) {
    constexpr int C4 = K / 4;                 // float4 per row
    for (int i = threadIdx.x; i < 64 * C4; i += 256) {
        const int row = i / C4, c = (i % C4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 + row < T) v = ld4(G + (size_t)(t0 + row) * ldg + c);
        st4(As + row * lda + c, v);
    }
}

// LayerNorm statistics of one row held as NV float4 per lane across a 16-lane group (D = 64*NV)
template <int NV>
__device__ __forceinline__ void ln_stats16(const float4 (&v)[NV], float& mean, float& rstd, float eps) {
    constexpr float invD = 1.0f / (64 * NV);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    mean = group16_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    rstd = 1.0f / sqrtf(group16_sum(q) * invD + eps);
}

// LayerNorm backward of one row spread over a 16-lane group.  dzv: upstream grad, uv: LN input.
// Returns du in dzv; accumulates affine partials.
template <int NV>
__device__ __forceinline__ void ln_bwd_row(float4 (&dzv)[NV], const float4 (&uv)[NV], float mean, float rstd,
                                           const float4 (&gam)[NV], float4 (&dgam)[NV], float4 (&dbet)[NV]) {
    constexpr float invD = 1.0f / (64 * NV);
    float4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        xh[j] = make_float4((uv[j].x - mean) * rstd, (uv[j].y - mean) * rstd, (uv[j].z - mean) * rstd, (uv[j].w - mean) * rstd);
        g[j] = make_float4(dzv[j].x * gam[j].x, dzv[j].y * gam[j].y, dzv[j].z * gam[j].z, dzv[j].w * gam[j].w);
        s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
        dgam[j].x += dzv[j].x * xh[j].x; dgam[j].y += dzv[j].y * xh[j].y; dgam[j].z += dzv[j].z * xh[j].z; dgam[j].w += dzv[j].w * xh[j].w;
        dbet[j].x += dzv[j].x; dbet[j].y += dzv[j].y; dbet[j].z += dzv[j].z; dbet[j].w += dzv[j].w;
    }
    s1 = group16_sum(s1) * invD;
    s2 = group16_sum(s2) * invD;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        dzv[j].x = rstd * (g[j].x - s1 - xh[j].x * s2);
        dzv[j].y = rstd * (g[j].y - s1 - xh[j].y * s2);
        dzv[j].z = rstd * (g[j].z - s1 - xh[j].z * s2);
        dzv[j].w = rstd * (g[j].w - s1 - xh[j].w * s2);
    }
}

// dynamic LDS above 64 KiB has to be opted into per kernel (gfx950 has 160 KiB per CU)
void big_lds_impl(const void* kernel, size_t bytes);      // step.hip: remembers what was granted per kernel
template <typename K>
static inline void big_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024) big_lds_impl(reinterpret_cast<const void*>(kernel), bytes);
}

static inline int hip_ret(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// Fork / join of INDEPENDENT launches of one step onto side streams (step.hip).  The launches of a training step form a chain, but not
// every link depends on the one before it: the length classes of an attention launch are disjoint sequence sets, the weight gradients of a
// layer need nothing the layers below still have to compute.  fork(main, first, n) makes side streams first..first+n-1 wait for what `main` holds so far;
// join(main, first, n) makes `main` wait for them.  Under stream capture the same calls become graph EDGES: the independent launches are
// parallel branches of the step graph and run concurrently (tools/probes/graph_branch_probe.py: 4 branches of 85 us in 141 us).
// side() returns `main` itself when side streams are off (the default: DR4SR_STREAMS=1 turns them on — measured slower, step.hip): fork /
// join are no-ops then.
struct StepFork {
    static constexpr int NSIDE = 3;
    hipStream_t side_[NSIDE];
    hipEvent_t fork_ev[NSIDE], join_ev[NSIDE];
    int state;                                              // 0 = not tried, 1 = ready, -1 = unavailable
    bool on() const;
    hipStream_t side(hipStream_t main, int i) const { return on() ? side_[i] : main; }
    int fork(hipStream_t main, int first, int n) const;     // sides first .. first + n - 1 (the attention classes use 0 and 1, the early
    int join(hipStream_t main, int first, int n) const;     //  weight-gradient launches 2: the two forks nest)
};
const StepFork& step_fork();
#define DR4SR_LAUNCH_CHECK() hip_ret(hipGetLastError())
// sum of n values `stride` floats apart, added in index order (the order IS the contract); eight loads in flight per thread
__device__ __forceinline__ float det_sum(const float* __restrict__ p, const int n, const size_t stride) {
    float s = 0.f;
    int x = 0;
    for (; x + 8 <= n; x += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(x + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; x < n; ++x) s += p[(size_t)x * stride];
    return s;
}
