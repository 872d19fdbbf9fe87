// cl.hip — CL4SRec pieces (reference model/cl4srec.py, module/data_augmentation.py:20-95, :305-350, :577-619).
//
//   k_cl_augment   : Item_Crop / Item_Mask / Item_Reorder / Item_Random on the device (the reference loops over the batch in
//                    Python with torch/numpy/random generators; the DISTRIBUTION is reproduced with Philox, not the streams).
//   k_infonce_*    : InfoNCELoss(sim_method='inner_product', neg_type='batch_both'): logits = [x_i x_j^T | x_i x_i^T (diag -inf)] / T,
//                    cross-entropy against the diagonal of the first block; rows with valid[b] == 0 are removed from rows AND
//                    columns (the reference drops sequences of length 1 before the loss, data_augmentation.py:613-615).
#include "common.h"
#include "kernels.h"

namespace {

#define DR4SR_SITE_AUG 0x41554721u

__device__ __forceinline__ uint32_t aug_rand(const RngKey& rk, uint64_t stream, uint32_t k) {
    const uint4 r = rng_call(rk, DR4SR_SITE_AUG, (stream << 8) + (k >> 2));
    const uint32_t c = k & 3;
    return c == 0 ? r.x : c == 1 ? r.y : c == 2 ? r.z : r.w;
}
__device__ __forceinline__ int rand_below(uint32_t r, int n) { return (int)__umulhi(r, (uint32_t)n); }     // uniform on [0, n)

// one thread per sequence; mode 0 crop (tau), 1 mask (gamma), 2 reorder (beta), 3 = one of the three drawn per CALL (Item_Random)
__global__ void k_cl_augment(const int64_t* __restrict__ seq, const int64_t* __restrict__ seqlen, int64_t* __restrict__ out,
                             int64_t* __restrict__ out_len, int B, int L, int mode, double tau, double gamma, double beta,
                             int64_t mask_id, uint64_t seed, uint32_t step, const int32_t* __restrict__ step_dev,
                             int64_t* __restrict__ out2, int64_t* __restrict__ out_len2) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (blockIdx.y) { out = out2; out_len = out_len2; step += 1; }     // second view of the same launch = the next call's draw
    if (step_dev) step += (uint32_t)*step_dev;             // graph replays: the call counter lives on the device
    const RngKey rk = make_rng(seed, step, 0.f);
    if (mode == 3) mode = rand_below(aug_rand(rk, 0xffffffu, 0), 3);          // data_augmentation.py:95: one method for the whole batch
    int n = (int)seqlen[b];
    n = n < 0 ? 0 : (n > L ? L : n);
    const int64_t* src = seq + (size_t)b * L;
    int64_t* dst = out + (size_t)b * L;
    const uint64_t st = (uint64_t)b + 1;
    if (mode == 0) {                                   // Item_Crop :20-41: contiguous sub-sequence of length max(1, int(tau n))
        const int sub = n > 0 ? max(1, (int)(tau * (double)n)) : 0;       // int(tau * n) in double, as Python does
        const int start = n > 0 ? rand_below(aug_rand(rk, st, 0), n - sub + 1) : 0;
        for (int l = 0; l < L; ++l) dst[l] = l < sub ? src[start + l] : 0;
        out_len[b] = sub;
    } else if (mode == 1) {                            // Item_Mask :44-62: int(gamma n) distinct positions -> mask_id
        const int sub = (int)(gamma * (double)n);
        int pos[64];
        for (int l = 0; l < n; ++l) pos[l] = l;
        for (int k = 0; k < sub; ++k) {                // partial Fisher-Yates = np.random.choice(n, sub, replace=False)
            const int j = k + rand_below(aug_rand(rk, st, k), n - k);
            const int t = pos[k]; pos[k] = pos[j]; pos[j] = t;
        }
        for (int l = 0; l < L; ++l) dst[l] = src[l];
        for (int k = 0; k < sub; ++k) dst[pos[k]] = mask_id;
        out_len[b] = n;
    } else {                                           // Item_Reorder :65-85: shuffle a contiguous segment of length int(beta n)
        const int sub = (int)(beta * (double)n);
        const int start = rand_below(aug_rand(rk, st, 0), n - sub + 1);
        int idx[64];
        for (int k = 0; k < sub; ++k) idx[k] = k;
        for (int k = sub - 1; k > 0; --k) {            // Fisher-Yates = random.shuffle
            const int j = rand_below(aug_rand(rk, st, 1 + k), k + 1);
            const int t = idx[k]; idx[k] = idx[j]; idx[j] = t;
        }
        for (int l = 0; l < L; ++l) dst[l] = src[l];
        for (int k = 0; k < sub; ++k) dst[start + k] = src[start + idx[k]];
        out_len[b] = n;
    }
}

// ---- InfoNCE.  One wave per row i: lane j-strided over the 2B logits, dot products over D from global (B*D floats: L2 resident).
template <int D>
__device__ __forceinline__ float dot_rows(const float* __restrict__ a, const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const float4 x = ld4(a + c), y = ld4(b + c);
        s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    return s;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// lse[i] = logsumexp_j logits[i][j], loss_row[i] = lse[i] - logits[i][i] (0 for invalid rows); stats += {n_valid rows, sum loss_row}
template <int D>
__global__ __launch_bounds__(256) void k_infonce_fwd(const float* __restrict__ xi, const float* __restrict__ xj,
                                                     const uint8_t* __restrict__ valid, int B, float inv_t, float* __restrict__ lse,
                                                     float* __restrict__ loss_row, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    if (valid && !valid[i]) { if (lane == 0) { lse[i] = 0.f; loss_row[i] = 0.f; } return; }
    const float* qi = xi + (size_t)i * D;
    float m = -INFINITY, s = 0.f, pos = 0.f;
    for (int j = lane; j < 2 * B; j += 64) {
        const int jj = j < B ? j : j - B;
        if (valid && !valid[jj]) continue;
        if (j >= B && jj == i) continue;                            // sim_ii diagonal = -inf
        const float v = dot_rows<D>(qi, (j < B ? xj : xi) + (size_t)jj * D) * inv_t;
        if (j == i) pos = v;
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
    const float mg = wave_max(m);
    s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - mg));
    pos = wave_sum(pos);
    if (lane == 0) {
        const float l = mg + logf(s);
        lse[i] = l;
        loss_row[i] = l - pos;
        unsafeAtomicAdd(stats, 1.0f);
        unsafeAtomicAdd(stats + 1, l - pos);
    }
}

// d loss_sum / d x  scaled by *scale (device scalar or NULL):  with P = softmax(logits) (rows = valid i)
//   dxi[i] += sum_j P1_ij xj[j] + sum_{j!=i} P2_ij xi[j] - xj[i]        (row pass, this kernel, role 0)
//   dxj[j] += sum_i P1_ij xi[i] - xi[j] ;  dxi[j] += sum_{i!=j} P2_ij xi[i]   (column pass, role 1)
// One wave per (row | column) a.  The other index o is walked in blocks of 64 in two phases: lanes over o compute the two logits of
// (a, o) as full dot products and leave the probabilities in LDS; lanes over D then accumulate the 64 weighted rows (coalesced).
// No cross-lane reduction inside the loop (the first version had two wave reductions per o: a 115 us dependent chain at B = 256).
template <int D>
__global__ __launch_bounds__(256) void k_infonce_bwd(const float* __restrict__ xi, const float* __restrict__ xj,
                                                     const uint8_t* __restrict__ valid, int B, float inv_t,
                                                     const float* __restrict__ lse, const float* __restrict__ scale,
                                                     float* __restrict__ dxi, float* __restrict__ dxj) {
    constexpr int NV = D / 64, LDR = D + 1;                  // +1: the dot phase reads row `lane`, conflict-free with an odd row stride
    __shared__ float ps[4][2][64];
    __shared__ float xa[4][2][D];
    __shared__ float xo[2][64 * LDR];                        // rows o0 .. o0+63 of xi / xj, staged once per block for all 4 waves
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + w;
    const bool live = item < 2 * B;
    const int role = live && item >= B, a = live ? (role ? item - B : item) : 0;
    const bool act = live && (!valid || valid[a]);          // inactive waves still take the barriers
    const float sc = (scale ? *scale : 1.0f) * inv_t;
    float acc_i[NV], acc_j[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        acc_i[v] = 0.f; acc_j[v] = 0.f;
        xa[w][0][lane + 64 * v] = xi[(size_t)a * D + lane + 64 * v];
        xa[w][1][lane + 64 * v] = xj[(size_t)a * D + lane + 64 * v];
    }
    const float la = lse[a];
    for (int o0 = 0; o0 < B; o0 += 64) {
        const int n = B - o0 < 64 ? B - o0 : 64;
        __syncthreads();                                     // previous block's rows are no longer read
        for (int e = threadIdx.x; e < 64 * D; e += 256) {    // coalesced: consecutive threads walk a row
            const int r = e / D, c = e % D;
            const bool ok = r < n;
            xo[0][r * LDR + c] = ok ? xi[(size_t)(o0 + r) * D + c] : 0.f;
            xo[1][r * LDR + c] = ok ? xj[(size_t)(o0 + r) * D + c] : 0.f;
        }
        __syncthreads();
        const int o = o0 + lane;
        float p1 = 0.f, p2 = 0.f;
        if (act && o < B && (!valid || valid[o])) {
            float d1 = 0.f, d2 = 0.f;
            const float* oi = &xo[0][lane * LDR];
            const float* oj = &xo[1][lane * LDR];
#pragma unroll 8
            for (int c = 0; c < D; ++c) {
                const float ai = xa[w][0][c];
                d2 += oi[c] * ai;                                        // sim_ii[a][o] (symmetric)
                d1 += role ? oi[c] * xa[w][1][c] : ai * oj[c];           // sim_ij[o][a] : sim_ij[a][o]
            }
            const float l = role ? lse[o] : la;
            p1 = __expf(d1 * inv_t - l);
            p2 = o == a ? 0.f : __expf(d2 * inv_t - l);
        }
        ps[w][0][lane] = p1; ps[w][1][lane] = p2;
        __syncthreads();
        if (act) {
            for (int k = 0; k < n; ++k) {
                const float q1 = ps[w][0][k], q2 = ps[w][1][k];
                if (q1 == 0.f && q2 == 0.f) continue;                     // masked-out row (uniform over the wave)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float oi = xo[0][k * LDR + lane + 64 * v];
                    if (!role) acc_i[v] += q1 * xo[1][k * LDR + lane + 64 * v] + q2 * oi;
                    else { acc_j[v] += q1 * oi; acc_i[v] += q2 * oi; }
                }
            }
        }
    }
    if (!act) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const size_t e = (size_t)a * D + lane + 64 * v;
        if (!role) unsafeAtomicAdd(dxi + e, sc * (acc_i[v] - xa[w][1][lane + 64 * v]));
        else { unsafeAtomicAdd(dxj + e, sc * (acc_j[v] - xa[w][0][lane + 64 * v])); unsafeAtomicAdd(dxi + e, sc * acc_i[v]); }
    }
}

}  // namespace

extern "C" int dr4sr_cl_augment(const int64_t* seq, const int64_t* seqlen, int64_t* out, int64_t* out_len, int32_t B, int32_t L,
                                int32_t mode, double tau, double gamma, double beta, int64_t mask_id, uint64_t seed, uint32_t step,
                                void* stream) {
    if (!seq || !seqlen || !out || !out_len || B < 0 || L <= 0 || L > 64 || mode < 0 || mode > 3) return DR4SR_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cl_augment, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out, out_len, B, L, mode, tau,
                       gamma, beta, mask_id, seed, step, (const int32_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr);
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_cl_augment_dev(const int64_t* seq, const int64_t* seqlen, int64_t* out, int64_t* out_len, int32_t B, int32_t L,
                                    int32_t mode, double tau, double gamma, double beta, int64_t mask_id, uint64_t seed,
                                    const int32_t* step_dev, uint32_t step_offset, void* stream) {
    if (!seq || !seqlen || !out || !out_len || !step_dev || B < 0 || L <= 0 || L > 64 || mode < 0 || mode > 3) return DR4SR_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cl_augment, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out, out_len, B, L, mode, tau,
                       gamma, beta, mask_id, seed, step_offset, step_dev, (int64_t*)nullptr, (int64_t*)nullptr);
    return DR4SR_LAUNCH_CHECK();
}
// two views in ONE launch: view 0 = the draw of call (*step_dev + step_offset), view 1 = that of the next call — what two consecutive
// dr4sr_cl_augment_dev calls with step_offset, step_offset + 1 produce (the kernel is one serial chain per sequence: the launches add)
extern "C" int dr4sr_cl_augment2_dev(const int64_t* seq, const int64_t* seqlen, int64_t* out_i, int64_t* len_i, int64_t* out_j,
                                     int64_t* len_j, int32_t B, int32_t L, int32_t mode, double tau, double gamma, double beta,
                                     int64_t mask_id, uint64_t seed, const int32_t* step_dev, uint32_t step_offset, void* stream) {
    if (!seq || !seqlen || !out_i || !len_i || !out_j || !len_j || !step_dev || B < 0 || L <= 0 || L > 64 || mode < 0 || mode > 3)
        return DR4SR_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cl_augment, dim3((B + 63) / 64, 2), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out_i, len_i, B, L, mode, tau,
                       gamma, beta, mask_id, seed, step_offset, step_dev, out_j, len_j);
    return DR4SR_LAUNCH_CHECK();
}

// ---- glue of the direct CL4SRec step (model/cl4srec.py: _api_step_body), one launch each instead of a handful of elementwise ones
namespace {
__global__ __launch_bounds__(256) void k_cl_prepare(const int64_t* __restrict__ seqlen, int B, uint8_t* __restrict__ valid,
                                                    float* __restrict__ stats, float* __restrict__ zero, int64_t nzero) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if (i < B) valid[i] = seqlen[i] != 1;                  // data_augmentation.py:613-615: sequences of length 1 are dropped
    if (i < 2) stats[i] = 0.f;
    for (int64_t k = i; k < nzero; k += stride) zero[k] = 0.f;
}
__global__ void k_cl_scalars(const float* __restrict__ tail, const float* __restrict__ stats, float clw, float* __restrict__ scale_out,
                             float* __restrict__ loss_out) {
    const float nv = tail[0], rows = stats[0];
    if (scale_out) *scale_out = rows > 0.f ? clw * nv / rows : 0.f;
    if (loss_out) *loss_out = tail[1] / nv + (rows > 0.f ? clw * stats[1] / rows : 0.f);
}
}  // namespace
/* valid[b] = seqlen[b] != 1, stats[0..1] = 0, zero[0..nzero) = 0 */
extern "C" int dr4sr_cl_prepare(const int64_t* seqlen, int32_t B, uint8_t* valid, float* stats, float* zero, int64_t nzero, void* stream) {
    if (!seqlen || !valid || !stats || B <= 0 || nzero < 0 || (nzero && !zero)) return DR4SR_E_ARG;
    int64_t nb = ((nzero > B ? nzero : B) + 255) / 256;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(k_cl_prepare, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, seqlen, B, valid, stats, zero, nzero);
    return DR4SR_LAUNCH_CHECK();
}
/* the two device scalars of a step whose main pass left {n_valid, loss_sum} in `tail` and whose InfoNCE forward left {rows, loss_sum}
 * in `stats`:  *scale_out = cl_weight * n_valid / rows  (InfoNCE backward scale under an optimizer that divides by n_valid);
 *              *loss_out  = loss_sum / n_valid + cl_weight * stats[1] / rows.   Either output may be NULL. */
extern "C" int dr4sr_cl_scalars(const float* tail, const float* stats, float cl_weight, float* scale_out, float* loss_out, void* stream) {
    if (!tail || !stats || (!scale_out && !loss_out)) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_cl_scalars, dim3(1), dim3(1), 0, (hipStream_t)stream, tail, stats, cl_weight, scale_out, loss_out);
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_infonce_fwd(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                                 float* lse, float* loss_row, float* stats, void* stream) {
    if (!xi || !xj || !lse || !loss_row || !stats || B <= 0 || !(temperature > 0.f)) return DR4SR_E_ARG;
    dim3 grid((B + 3) / 4), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_infonce_fwd<64>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, loss_row, stats);
    else if (D == 128) hipLaunchKernelGGL(k_infonce_fwd<128>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, loss_row, stats);
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_infonce_bwd(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                                 const float* lse, const float* scale, float* dxi, float* dxj, void* stream) {
    if (!xi || !xj || !lse || !dxi || !dxj || B <= 0 || !(temperature > 0.f)) return DR4SR_E_ARG;
    dim3 grid((2 * B + 3) / 4), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_infonce_bwd<64>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, scale, dxi, dxj);
    else if (D == 128) hipLaunchKernelGGL(k_infonce_bwd<128>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, scale, dxi, dxj);
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}
