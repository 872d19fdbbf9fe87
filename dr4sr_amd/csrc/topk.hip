// topk.hip — K6 full-item scorer + history mask + top-k for evaluation (single-domain case).
//
// Reference: BaseModel.topk (model/basemodel.py:354-365): real_score = query @ E[:N].T; non-domain
// items (incl. PAD column 0) and the user's history are set to -inf; torch.topk(k).
//
// v1: one workgroup per query row; the N scores of the row live in LDS (N*4 B <= 150 KiB), the top-k
// is k rounds of block-wide arg-max (ties -> lower id).  The [B,N] score matrix (97.7 MB per 2048-row
// batch in the reference) is never materialised.
#include "common.h"
#include "kernels.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

template <int D>
__global__ __launch_bounds__(256) void k_topk(const float* __restrict__ Q, const float* __restrict__ E,
                                              const int64_t* __restrict__ hist, float* __restrict__ out_score,
                                              int64_t* __restrict__ out_item, int n_items, int Lh, int k) {
    float* sc = smem;                                  // [n_items]
    float* qs = smem + ((n_items + 3) & ~3);           // [D]
    float* rv = qs + D;                                // [4] per-wave best value
    int* ri = reinterpret_cast<int*>(rv + 4);          // [4] per-wave best index
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int c = tid; c < D; c += 256) qs[c] = Q[(size_t)b * D + c];
    __syncthreads();
    for (int n = tid; n < n_items; n += 256) {
        const float* e = E + (size_t)n * D;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 ev = ld4(e + c), qv = ld4(qs + c);
            s0 += ev.x * qv.x; s1 += ev.y * qv.y; s2 += ev.z * qv.z; s3 += ev.w * qv.w;
        }
        sc[n] = n == 0 ? -INFINITY : (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int j = tid; j < Lh; j += 256) {
        const int64_t id = hist[(size_t)b * Lh + j];
        if (id >= 0 && id < n_items) sc[id] = -INFINITY;
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int n = tid; n < n_items; n += 256) {
            const float v = sc[n];
            if (v == v && (bi == 0x7fffffff || v > bv)) { bv = v; bi = n; }   // NaN marks "already taken"
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { rv[w] = bv; ri[w] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = rv[0];
            int fi = ri[0];
            for (int x = 1; x < 4; ++x) {
                const float ov = rv[x];
                const int oi = ri[x];
                if (oi != 0x7fffffff && (fi == 0x7fffffff || ov > fv || (ov == fv && oi < fi))) { fv = ov; fi = oi; }
            }
            if (fi == 0x7fffffff) { fv = -INFINITY; fi = 0; }
            else sc[fi] = __builtin_nanf("");
            out_score[(size_t)b * k + r] = fv;
            out_item[(size_t)b * k + r] = fi;
        }
        __syncthreads();
    }
}

extern "C" int dr4sr_full_score_topk(const float* q, const float* E, const int64_t* hist, float* out_score,
                                     int64_t* out_item, int64_t B, int32_t D, int32_t n_items, int32_t Lh, int32_t k,
                                     void* stream) {
    if (!q || !E || !out_score || !out_item || B < 0 || n_items < 2 || k <= 0 || k > 128 || Lh < 0 || (Lh > 0 && !hist))
        return DR4SR_E_ARG;
    if (D != 64 && D != 128) return DR4SR_E_SHAPE;
    const size_t lds = sizeof(float) * (((n_items + 3) & ~3) + D + 8);
    if (lds > 150 * 1024) return DR4SR_E_SHAPE;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64) { big_lds(k_topk<64>, lds); hipLaunchKernelGGL(k_topk<64>, dim3((unsigned)B), dim3(256), lds, s, q, E, hist, out_score, out_item, n_items, Lh, k); }
    else { big_lds(k_topk<128>, lds); hipLaunchKernelGGL(k_topk<128>, dim3((unsigned)B), dim3(256), lds, s, q, E, hist, out_score, out_item, n_items, Lh, k); }
    return DR4SR_LAUNCH_CHECK();
}
