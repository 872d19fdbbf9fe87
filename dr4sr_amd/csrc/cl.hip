// cl.hip — CL4SRec pieces (reference model/cl4srec.py, module/data_augmentation.py:20-95, :305-350, :577-619).
//
//   k_cl_augment   : Item_Crop / Item_Mask / Item_Reorder / Item_Random on the device (the reference loops over the batch in
//                    Python with torch/numpy/random generators; the DISTRIBUTION is reproduced with Philox, not the streams).
//   k_infonce_*    : InfoNCELoss(sim_method='inner_product', neg_type='batch_both'): logits = [x_i x_j^T | x_i x_i^T (diag -inf)] / T,
//                    cross-entropy against the diagonal of the first block; rows with valid[b] == 0 are removed from rows AND
//                    columns (the reference drops sequences of length 1 before the loss, data_augmentation.py:613-615).  On MFMA tiles.
//   k_cl_prepare / k_cl_scalars : glue of the step composed without autograd (model/cl4srec.py:_api_step_body): valid mask + zeroing;
//                    the device scalars {InfoNCE backward scale, reported loss}.
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

// mode 0 crop (tau), 1 mask (gamma), 2 reorder (beta), 3 = one of the three drawn per CALL (Item_Random).
// One 64-thread workgroup per 16 sequences: the rows go through LDS with coalesced loads / stores (a row is 8 L contiguous bytes), the
// draws of a sequence are one serial chain on lane (sequence) with ONE Philox call per four draws (aug_rand(k) = word k & 3 of call
// k >> 2).  The first form ran one THREAD per sequence straight on global memory — 64 different rows per load instruction, a Philox call
// per draw, 8 waves on the whole device: 17.8 us of CL4SRec's 0.284 ms step for both views.  Same draws.
constexpr int AUG_SPB = 16;
struct AugRng {
    const RngKey& rk; uint64_t stream; uint4 r; int have;
    __device__ __forceinline__ AugRng(const RngKey& k, uint64_t st) : rk(k), stream(st), r(make_uint4(0, 0, 0, 0)), have(-1) {}
    __device__ __forceinline__ uint32_t get(uint32_t k) {
        const int c = (int)(k >> 2);
        if (c != have) { r = rng_call(rk, DR4SR_SITE_AUG, (stream << 8) + (uint64_t)c); have = c; }
        const uint32_t w = k & 3;
        return w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w;
    }
};
__global__ __launch_bounds__(64) void k_cl_augment(const int64_t* __restrict__ seq, const int64_t* __restrict__ seqlen, int64_t* __restrict__ out,
                             int64_t* __restrict__ out_len, int B, int L, int mode, double tau, double gamma, double beta,
                             int64_t mask_id, uint64_t seed, uint32_t step, const int32_t* __restrict__ step_dev,
                             int64_t* __restrict__ out2, int64_t* __restrict__ out_len2, const int64_t* __restrict__ rows = nullptr) {
    __shared__ int64_t src_s[AUG_SPB][64], dst_s[AUG_SPB][64];
    __shared__ int64_t srow_s[AUG_SPB];
    __shared__ unsigned char perm_lds[AUG_SPB * 64];
    const int b0 = blockIdx.x * AUG_SPB, nb = min(AUG_SPB, B - b0);
    if (nb <= 0) return;
    if (blockIdx.y) { out = out2; out_len = out_len2; step += 1; }     // second view of the same launch = the next call's draw
    // the dependent chain is rows[] -> dataset rows; the call counter (and the Philox call that picks Item_Random's method) is requested
    // with the first link and consumed while the rows are in flight
    const uint32_t step_add = step_dev ? (uint32_t)*step_dev : 0u;     // graph replays: the call counter lives on the device
    if ((int)threadIdx.x < nb) srow_s[threadIdx.x] = rows ? rows[b0 + threadIdx.x] : (int64_t)(b0 + threadIdx.x);   // rows != NULL: seq / seqlen are dataset tensors
    __syncthreads();
    int n = 0;
    if ((int)threadIdx.x < nb) n = (int)seqlen[srow_s[threadIdx.x]];
    int64_t stage[(AUG_SPB * 64 + 63) / 64];
#pragma unroll
    for (int k = 0; k < (AUG_SPB * 64 + 63) / 64; ++k) {
        const int i = threadIdx.x + 64 * k;
        stage[k] = i < nb * L ? seq[(size_t)srow_s[i / L] * L + i % L] : 0;
    }
    step += step_add;
    const RngKey rk = make_rng(seed, step, 0.f);
    if (mode == 3) mode = rand_below(aug_rand(rk, 0xffffffu, 0), 3);          // data_augmentation.py:95: one method for the whole batch
#pragma unroll
    for (int k = 0; k < (AUG_SPB * 64 + 63) / 64; ++k) {
        const int i = threadIdx.x + 64 * k;
        if (i < nb * L) src_s[i / L][i % L] = stage[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
        const int q = threadIdx.x, b = b0 + q;
        n = n < 0 ? 0 : (n > L ? L : n);
        const int64_t* src = src_s[q];
        int64_t* dst = dst_s[q];
        AugRng rng(rk, (uint64_t)b + 1);
        int len_out = n;
        if (mode == 0) {                                   // Item_Crop :20-41: contiguous sub-sequence of length max(1, int(tau n))
            const int sub = n > 0 ? max(1, (int)(tau * (double)n)) : 0;       // int(tau * n) in double, as Python does
            const int start = n > 0 ? rand_below(rng.get(0), n - sub + 1) : 0;
            for (int l = 0; l < L; ++l) dst[l] = l < sub ? src[start + l] : 0;
            len_out = sub;
        } else if (mode == 1) {                            // Item_Mask :44-62: int(gamma n) distinct positions -> mask_id
            // a uniformly random subset of size sub = np.random.choice(n, sub, replace=False) as a SET (every member gets mask_id, so
            // the order of the draw is immaterial): selection sampling, position l is taken with probability needed / remaining.
            int need = (int)(gamma * (double)n);
            for (int l = 0; l < L; ++l) {
                int64_t v = src[l];
                if (l < n && need > 0 && rand_below(rng.get(l), n - l) < need) { v = mask_id; --need; }
                dst[l] = v;
            }
        } else {                                           // Item_Reorder :65-85: shuffle a contiguous segment of length int(beta n)
            const int sub = (int)(beta * (double)n);
            const int start = rand_below(rng.get(0), n - sub + 1);
            unsigned char* idx = perm_lds + q * 64;                // this sequence's permutation of the segment (LDS, not scratch)
            for (int k = 0; k < sub; ++k) idx[k] = (unsigned char)k;
            for (int k = sub - 1; k > 0; --k) {            // Fisher-Yates = random.shuffle
                const int j = rand_below(rng.get(1 + k), k + 1);
                const unsigned char t = idx[k]; idx[k] = idx[j]; idx[j] = t;
            }
            for (int l = 0; l < L; ++l) dst[l] = (l >= start && l < start + sub) ? src[start + idx[l - start]] : src[l];
        }
        out_len[b] = len_out;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * L; i += 64) out[(size_t)(b0 + i / L) * L + i % L] = dst_s[i / L][i % L];
}

// ---- InfoNCE('inner_product', 'batch_both') on MFMA (v_mfma_f32_16x16x4_f32, exact fp32).
// logits[i] = [x_i . xj_j | x_i . xi_j, diagonal -inf] / temperature over the 2B "columns" j (first the other view, then the own one).
// One workgroup per 16-row tile; its waves (16 forward, 8 / 4 backward) split the column tiles.  Logit tiles are computed TRANSPOSED, S^T[j][i] (A = the 16
// column rows, B = the 16 query rows), so a lane holds four j of ONE i (i = lane & 15, j = 4 (lane >> 4) + r): the softmax statistics
// of a row reduce in-lane and over the 4 lane groups (two shuffles), and the probabilities are, as they stand, the B operand of the
// next product out^T[d][i] = sum_j X[j][d] P[i][j] (lane group g supplies j = 4g + s at step s) — the scheme of csrc/attn_mfma.hip.
// (The first versions walked rows with per-lane dot products: 26 us forward / 48 us backward at B = 256, now 14 / 23.)
__device__ __forceinline__ f32x4 cl_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// fragment of 16 rows [r0, r0 + 16) of a [n][D] matrix: lane (l16 = lane & 15, g = lane >> 4) holds features g*D/4 .. of row r0 + l16
template <int D>
__device__ __forceinline__ void cl_frag(float (&f)[D / 4], const float* __restrict__ x, int r0, int n) {
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4;
    if (r0 + l16 < n) {
        const float* p = x + (size_t)(r0 + l16) * D + g * (D / 4);
#pragma unroll
        for (int c = 0; c < D / 4; c += 4) { const float4 v = ld4(p + c); f[c] = v.x; f[c + 1] = v.y; f[c + 2] = v.z; f[c + 3] = v.w; }
    } else {
#pragma unroll
        for (int c = 0; c < D / 4; ++c) f[c] = 0.f;
    }
}
template <int D>
__device__ __forceinline__ f32x4 cl_dots(const float (&a)[D / 4], const float (&b)[D / 4]) {      // C[ra][rb] = <row ra of a, row rb of b>
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D / 4; ++s) acc = cl_mfma(a[s], b[s], acc);
    return acc;
}
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
}

// lse[i] = logsumexp_j logits[i][j], loss_row[i] = lse[i] - logits[i][i] (0 for invalid rows); stats += {n_valid rows, sum loss_row}
constexpr int CL_FW = 16;                                  // waves per forward workgroup: the column tiles of a row tile are a latency chain
template <int D>
__global__ __launch_bounds__(CL_FW * 64) void k_infonce_fwd(const float* __restrict__ xi, const float* __restrict__ xj,
                                                     const uint8_t* __restrict__ valid, int B, float inv_t, float* __restrict__ lse,
                                                     float* __restrict__ loss_row, float* __restrict__ stats) {
    __shared__ float red[CL_FW][3][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int i0 = blockIdx.x * 16, i = i0 + l16, T1 = (B + 15) / 16;
    const bool vi = i < B && (!valid || valid[i]);
    float q[D / 4], a[D / 4];
    cl_frag<D>(q, xi, i0, B);
    float m = -INFINITY, sum = 0.f, pos = 0.f;
    for (int t = w; t < 2 * T1; t += CL_FW) {
        const bool own = t >= T1;                          // second half: the own view's rows, diagonal excluded
        const int j0 = (own ? t - T1 : t) * 16;
        cl_frag<D>(a, own ? xi : xj, j0, B);
        bool vj[4];                                        // requested with the rows, ahead of the MFMA chain
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = j0 + 4 * g + r; vj[r] = j < B && (!valid || valid[j]); }
        const f32x4 acc = cl_dots<D>(a, q);                // acc[r] = <X[j0 + 4g + r], xi[i]>
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * g + r;
            if (vi && vj[r] && !(own && j == i)) {
                const float v = acc[r] * inv_t;
                if (!own && j == i) pos = v;
                const float mn = fmaxf(m, v);
                sum = sum * __expf(m - mn) + __expf(v - mn);
                m = mn;
            }
        }
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {                    // the 4 lane groups of a column i
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(sum, o, 64);
        lse_merge(m, sum, m2, s2);
        pos += __shfl_xor(pos, o, 64);
    }
    if (g == 0) { red[w][0][l16] = m; red[w][1][l16] = sum; red[w][2][l16] = pos; }
    __syncthreads();
    if (w == 0 && g == 0) {
#pragma unroll
        for (int k = 1; k < CL_FW; ++k) { lse_merge(m, sum, red[k][0][l16], red[k][1][l16]); pos += red[k][2][l16]; }
        float l = 0.f, lr = 0.f;
        if (vi) { l = m + logf(sum); lr = l - pos; }
        if (i < B) { lse[i] = l; loss_row[i] = lr; }
        float cnt = vi ? 1.f : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); lr += __shfl_xor(lr, o, 64); }
        if (l16 == 0 && cnt > 0.f) { unsafeAtomicAdd(stats, cnt); unsafeAtomicAdd(stats + 1, lr); }
    }
}

// d loss_sum / d x  scaled by *scale (device scalar or NULL):  with P = softmax(logits) (rows = valid i)
//   dxi[i] += sum_j P1_ij xj[j] + sum_{j!=i} P2_ij xi[j] - xj[i]        (row tiles: blockIdx.x < T1)
//   dxj[j] += sum_i P1_ij xi[i] - xi[j] ;  dxi[j] += sum_{i!=j} P2_ij xi[i]   (column tiles: blockIdx.x >= T1)
// Row tile: S^T[j][i] tiles as in the forward, P^T = exp(S^T / t - lse_i) is the B operand of out^T[d][i] += X[j][d] P^T[j][i].
// Column tile: S[i][j] tiles (A = query rows i, B = the tile's 16 column rows j), P[i][j] with lse of the REGISTER's row
// i = 4g + r, then out^T[d][j] += xi[i][d] P[i][j].  The 4 waves split the other index; their out^T accumulators meet in LDS.
template <int D>
__global__ __launch_bounds__(D == 64 ? 512 : 256) void k_infonce_bwd(const float* __restrict__ xi, const float* __restrict__ xj,
                                                     const uint8_t* __restrict__ valid, int B, float inv_t,
                                                     const float* __restrict__ lse, const float* __restrict__ scale,
                                                     float* __restrict__ dxi, float* __restrict__ dxj,
                                                     const float* __restrict__ stats = nullptr, float* __restrict__ tail = nullptr,
                                                     float clw = 0.f, int fold = 0) {
    constexpr int DT = D / 16, NW = D == 64 ? 8 : 4;       // 64 KB of LDS for the accumulators of all waves either way
    __shared__ float red[2][NW][DT][4][64];                // [accumulator][wave][d tile][r][lane]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int T1 = (B + 15) / 16;
    const bool colpass = (int)blockIdx.x >= T1;
    const int c0 = (colpass ? blockIdx.x - T1 : blockIdx.x) * 16, c = c0 + l16;        // this lane's row (row pass) / column (column pass)
    const bool vc = c < B && (!valid || valid[c]);
    // stats != NULL (round 4): the step's two device scalars are computed here instead of by a k_cl_scalars_dp launch in front —
    // backward scale = cl_weight * n_valid / rows from the main pass's tail {n_valid, loss_sum} and the forward's {rows, loss_sum};
    // fold: the contrastive term's share of the reported loss goes into tail[1] (one thread; the others read tail[0] only)
    float sc0 = scale ? *scale : 1.0f;
    if (stats) {
        const float nv = tail[0], nrows = stats[0];
        sc0 = nrows > 0.f ? clw * nv / nrows : 0.f;
        if (fold && blockIdx.x == 0 && threadIdx.x == 0) tail[1] += (nrows > 0.f ? clw * stats[1] / nrows : 0.f) * nv;
    }
    const float sc = sc0 * inv_t;
    float fa[D / 4], fb[D / 4], fq[D / 4];
    f32x4 acc0[DT], acc1[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { acc0[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[dt] = acc0[dt]; }
    if (!colpass) {
        cl_frag<D>(fq, xi, c0, B);
        const float la = vc ? lse[c] : 0.f;
        for (int t = w; t < 2 * T1; t += NW) {
            const bool own = t >= T1;
            const int j0 = (own ? t - T1 : t) * 16;
            const float* X = own ? xi : xj;
            cl_frag<D>(fa, X, j0, B);
            bool vj[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = j0 + 4 * g + r; vj[r] = j < B && (!valid || valid[j]); }
            const f32x4 s = cl_dots<D>(fa, fq);            // S^T[j0 + 4g + r][c]
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + 4 * g + r;
                p[r] = (vc && vj[r] && !(own && j == c)) ? __expf(s[r] * inv_t - la) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // out^T[d][c] += X[j0 + 4g + r][d] * P^T[j0 + 4g + r][c]
                const int j = j0 + 4 * g + r;
                const float* xr = X + (size_t)(j < B ? j : 0) * D + l16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc0[dt] = cl_mfma(j < B ? xr[16 * dt] : 0.f, p[r], acc0[dt]);
            }
        }
    } else {
        cl_frag<D>(fq, xj, c0, B);                         // column rows of the other view (P1) ...
        cl_frag<D>(fb, xi, c0, B);                         // ... and of the own view (P2)
        for (int t = w; t < T1; t += NW) {
            const int i0 = t * 16;
            cl_frag<D>(fa, xi, i0, B);
            float li[4]; bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * g + r;
                ok[r] = vc && i < B && (!valid || valid[i]);
                li[r] = ok[r] ? lse[i] : 0.f;
            }
            const f32x4 s1 = cl_dots<D>(fa, fq), s2 = cl_dots<D>(fa, fb);       // S1[i0 + 4g + r][c], S2[...]
            float p1[4], p2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * g + r;
                p1[r] = ok[r] ? __expf(s1[r] * inv_t - li[r]) : 0.f;
                p2[r] = (ok[r] && i != c) ? __expf(s2[r] * inv_t - li[r]) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // out^T[d][c] += xi[i0 + 4g + r][d] * P[i0 + 4g + r][c]
                const int i = i0 + 4 * g + r;
                const float* xr = xi + (size_t)(i < B ? i : 0) * D + l16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const float xv = i < B ? xr[16 * dt] : 0.f;
                    acc0[dt] = cl_mfma(xv, p1[r], acc0[dt]);
                    acc1[dt] = cl_mfma(xv, p2[r], acc1[dt]);
                }
            }
        }
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[0][w][dt][r][lane] = acc0[dt][r]; red[1][w][dt][r][lane] = acc1[dt][r]; }
    __syncthreads();
    // element (d = 16 dt + 4 g + r, column l16) of out^T: wave w finishes d tile w (D = 128 with 4 waves: tiles w and w + 4)
    for (int dt = w; dt < DT; dt += NW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) { o0 += red[0][k][dt][r][lane]; o1 += red[1][k][dt][r][lane]; }
            const int d = 16 * dt + 4 * g + r;
            if (vc) {
                const size_t e = (size_t)c * D + d;
                if (!colpass) unsafeAtomicAdd(dxi + e, sc * (o0 - xj[e]));
                else { unsafeAtomicAdd(dxj + e, sc * (o0 - xi[e])); unsafeAtomicAdd(dxi + e, sc * o1); }
            }
        }
    }
}

}  // namespace

extern "C" int dr4sr_cl_augment(const int64_t* seq, const int64_t* seqlen, int64_t* out, int64_t* out_len, int32_t B, int32_t L,
                                int32_t mode, double tau, double gamma, double beta, int64_t mask_id, uint64_t seed, uint32_t step,
                                void* stream) {
    if (!seq || !seqlen || !out || !out_len || B < 0 || L <= 0 || L > 64 || mode < 0 || mode > 3) return DR4SR_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cl_augment, dim3((B + AUG_SPB - 1) / AUG_SPB), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out, out_len, B, L, mode, tau,
                       gamma, beta, mask_id, seed, step, (const int32_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr);
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_cl_augment_dev(const int64_t* seq, const int64_t* seqlen, int64_t* out, int64_t* out_len, int32_t B, int32_t L,
                                    int32_t mode, double tau, double gamma, double beta, int64_t mask_id, uint64_t seed,
                                    const int32_t* step_dev, uint32_t step_offset, void* stream) {
    if (!seq || !seqlen || !out || !out_len || !step_dev || B < 0 || L <= 0 || L > 64 || mode < 0 || mode > 3) return DR4SR_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cl_augment, dim3((B + AUG_SPB - 1) / AUG_SPB), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out, out_len, B, L, mode, tau,
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
    hipLaunchKernelGGL(k_cl_augment, dim3((B + AUG_SPB - 1) / AUG_SPB, 2), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out_i, len_i, B, L, mode, tau,
                       gamma, beta, mask_id, seed, step_offset, step_dev, out_j, len_j);
    return DR4SR_LAUNCH_CHECK();
}
// the same on rows rows[0..B) of dataset tensors seq [U,L] / seqlen [U] (round 4: the batch of a captured CL4SRec step is selected on the
// device — dr4sr_sasrec_plan.perm fills rows[] —, so the views are drawn without materialising the batch first); same draws as the
// [B,L] form on the gathered rows (the stream of a sequence is keyed by its batch slot)
extern "C" int dr4sr_cl_augment2_rows_dev(const int64_t* seq, const int64_t* seqlen, const int64_t* rows, int64_t* out_i, int64_t* len_i,
                                          int64_t* out_j, int64_t* len_j, int32_t B, int32_t L, int32_t mode, double tau, double gamma,
                                          double beta, int64_t mask_id, uint64_t seed, const int32_t* step_dev, uint32_t step_offset,
                                          void* stream) {
    if (!seq || !seqlen || !rows || !out_i || !len_i || !out_j || !len_j || !step_dev || B < 0 || L <= 0 || L > 64 || mode < 0 || mode > 3)
        return DR4SR_E_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cl_augment, dim3((B + AUG_SPB - 1) / AUG_SPB, 2), dim3(64), 0, (hipStream_t)stream, seq, seqlen, out_i, len_i, B, L, mode, tau,
                       gamma, beta, mask_id, seed, step_offset, step_dev, out_j, len_j, rows);
    return DR4SR_LAUNCH_CHECK();
}

// ---- glue of the direct CL4SRec step (model/cl4srec.py: _api_step_body), one launch each instead of a handful of elementwise ones
namespace {
__global__ __launch_bounds__(256) void k_cl_prepare(const int64_t* __restrict__ seqlen, int B, uint8_t* __restrict__ valid,
                                                    float* __restrict__ stats, float* __restrict__ zero, int64_t nzero,
                                                    const int64_t* __restrict__ rows = nullptr, int32_t* __restrict__ step_dev = nullptr,
                                                    int32_t step_add = 0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if (i == 0 && step_dev) *step_dev += step_add;           // the augmentation's device call counter (its launches of this step are behind us)
    if (i < B) valid[i] = seqlen[rows ? rows[i] : i] != 1;  // data_augmentation.py:613-615: sequences of length 1 are dropped
    if (i < 2) stats[i] = 0.f;
    for (int64_t k = i; k < nzero; k += stride) zero[k] = 0.f;
}
__global__ void k_cl_scalars(const float* __restrict__ tail, const float* __restrict__ stats, float clw, float* __restrict__ scale_out,
                             float* __restrict__ loss_out) {
    const float nv = tail[0], rows = stats[0];
    if (scale_out) *scale_out = rows > 0.f ? clw * nv / rows : 0.f;
    if (loss_out) *loss_out = tail[1] / nv + (rows > 0.f ? clw * stats[1] / rows : 0.f);
}
// data-parallel / MetaModel form: n_valid is the SUM of per-rank counts (gathered next to the pooled views), and the contrastive term's
// share of the reported loss is folded into the local {n_valid, loss_sum} tail so that (all-reduced) tail[1] / tail[0] is the step's loss
__global__ void k_cl_scalars_dp(const float* __restrict__ nv_parts, int n_parts, int64_t stride, const float* __restrict__ stats, float clw,
                                float* __restrict__ scale_out, float* __restrict__ tail_local) {
    float nv = 0.f;
    for (int r = 0; r < n_parts; ++r) nv += nv_parts[(size_t)r * stride];
    const float rows = stats[0];
    if (scale_out) *scale_out = rows > 0.f ? clw * nv / rows : 0.f;
    if (tail_local) tail_local[1] += (rows > 0.f ? clw * stats[1] / rows : 0.f) * tail_local[0];
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
/* the same with seqlen a dataset tensor addressed through rows[0..B) */
extern "C" int dr4sr_cl_prepare_rows(const int64_t* seqlen, const int64_t* rows, int32_t B, uint8_t* valid, float* stats, float* zero,
                                     int64_t nzero, void* stream) {
    if (!seqlen || !rows || !valid || !stats || B <= 0 || nzero < 0 || (nzero && !zero)) return DR4SR_E_ARG;
    int64_t nb = ((nzero > B ? nzero : B) + 255) / 256;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(k_cl_prepare, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, seqlen, B, valid, stats, zero, nzero, rows);
    return DR4SR_LAUNCH_CHECK();
}
/* ... and advancing the augmentation's device call counter by step_add (data_augmentation.py end_step()) in the same launch */
extern "C" int dr4sr_cl_prepare_rows_step(const int64_t* seqlen, const int64_t* rows, int32_t B, uint8_t* valid, float* stats, float* zero,
                                          int64_t nzero, int32_t* step_dev, int32_t step_add, void* stream) {
    if (!seqlen || !rows || !valid || !stats || !step_dev || B <= 0 || nzero < 0 || (nzero && !zero)) return DR4SR_E_ARG;
    int64_t nb = ((nzero > B ? nzero : B) + 255) / 256;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(k_cl_prepare, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, seqlen, B, valid, stats, zero, nzero, rows, step_dev, step_add);
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

extern "C" int dr4sr_cl_scalars_dp(const float* nv_parts, int32_t n_parts, int64_t stride, const float* stats, float cl_weight,
                                   float* scale_out, float* tail_local, void* stream) {
    if (!nv_parts || n_parts <= 0 || stride < 0 || !stats || (!scale_out && !tail_local)) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_cl_scalars_dp, dim3(1), dim3(1), 0, (hipStream_t)stream, nv_parts, n_parts, stride, stats, cl_weight, scale_out, tail_local);
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_infonce_fwd(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                                 float* lse, float* loss_row, float* stats, void* stream) {
    if (!xi || !xj || !lse || !loss_row || !stats || B <= 0 || !(temperature > 0.f)) return DR4SR_E_ARG;
    dim3 grid((B + 15) / 16), blk(CL_FW * 64);
    if (D == 64) hipLaunchKernelGGL(k_infonce_fwd<64>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, loss_row, stats);
    else if (D == 128) hipLaunchKernelGGL(k_infonce_fwd<128>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, loss_row, stats);
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_infonce_bwd(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                                 const float* lse, const float* scale, float* dxi, float* dxj, void* stream) {
    if (!xi || !xj || !lse || !dxi || !dxj || B <= 0 || !(temperature > 0.f)) return DR4SR_E_ARG;
    dim3 grid(2 * ((B + 15) / 16)), blk(D == 64 ? 512 : 256);
    if (D == 64) hipLaunchKernelGGL(k_infonce_bwd<64>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, scale, dxi, dxj);
    else if (D == 128) hipLaunchKernelGGL(k_infonce_bwd<128>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, scale, dxi, dxj);
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}

/* dr4sr_infonce_bwd with the step's scalars computed inside (what dr4sr_cl_scalars_dp(tail, 1, 0, stats, cl_weight, &scale, fold ? tail : NULL)
 * in front of it would give): scale = cl_weight * tail[0] / stats[0]; fold_tail != 0: tail[1] += cl_weight * stats[1] / stats[0] * tail[0] */
extern "C" int dr4sr_infonce_bwd_scaled(const float* xi, const float* xj, const uint8_t* valid, int32_t B, int32_t D, float temperature,
                                        const float* lse, float* tail, const float* stats, float cl_weight, int32_t fold_tail,
                                        float* dxi, float* dxj, void* stream) {
    if (!xi || !xj || !lse || !dxi || !dxj || !tail || !stats || B <= 0 || !(temperature > 0.f)) return DR4SR_E_ARG;
    dim3 grid(2 * ((B + 15) / 16)), blk(D == 64 ? 512 : 256);
    if (D == 64) hipLaunchKernelGGL(k_infonce_bwd<64>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, (const float*)nullptr, dxi, dxj, stats, tail, cl_weight, fold_tail);
    else if (D == 128) hipLaunchKernelGGL(k_infonce_bwd<128>, grid, blk, 0, (hipStream_t)stream, xi, xj, valid, B, 1.0f / temperature, lse, (const float*)nullptr, dxi, dxj, stats, tail, cl_weight, fold_tail);
    else return DR4SR_E_SHAPE;
    return DR4SR_LAUNCH_CHECK();
}
