// score.hip — K0 negative sampler + K4 tied-embedding scorer with BCE, forward and backward.
//
// Reference: BaseModel._neg_sampling (model/basemodel.py:50-61): uniform over ids 1..N-1 with
// replacement, history NOT excluded, never PAD;  BaseModel.training_step (model/basemodel.py:204-214):
// pos = <q, E[target]>, neg = <q, E[neg]>, pos[target==0] = -inf;  BinaryCrossEntropyLoss
// (model/loss_func.py:9-38): -logsigmoid(pos) + softplus(neg), masked where target == 0, / n_valid.
//
// One wave per (sequence, position): 64 lanes cover the D dims (D/64 floats per lane), the two dot
// products are wave shuffles.  Gradients into the item table are row-sparse fp32 atomics on a dense
// [N,D] buffer (the table is 3 MB: L2-resident), pads skipped.
#include "common.h"
#include "kernels.h"


__global__ void k_neg_sample(int64_t* __restrict__ out, int64_t n, int n_items, uint64_t seed, uint32_t step, const int* step_dev) {
    const RngKey rk = make_rng(seed, step_dev ? (uint32_t)*step_dev : step, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = sample_neg_id(rk, (uint64_t)i, n_items);
}

extern "C" int dr4sr_neg_sample(int64_t* out, int64_t n, int32_t n_items, uint64_t seed, uint32_t step, void* stream) {
    if (!out || n < 0 || n_items < 2) return DR4SR_E_ARG;
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_neg_sample, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, n_items, seed, step, nullptr);
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_neg_sample_dev(int64_t* out, int64_t n, int32_t n_items, uint64_t seed, const int32_t* step_dev, void* stream) {
    if (!out || !step_dev || n < 0 || n_items < 2) return DR4SR_E_ARG;
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_neg_sample, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, n_items, seed, 0u, step_dev);
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Packed training scorer: forward + backward fused (upstream weight 1, un-normalised).
// One workgroup per sequence.  Wave 0 first handles all L positions at once (lane = position): loads the targets,
// draws/loads the negatives, and publishes them to LDS; then the 8 half-waves walk only the positions with a non-PAD
// target (ballot + bit scan), one position per half-wave (32 lanes x D/32 dims, shuffle reduction within the half).
template <int D>
__global__ __launch_bounds__(256) void k_score_packed(const float* __restrict__ Z, const float* __restrict__ E,
                                                      float* __restrict__ dE, float* __restrict__ dZ,
                                                      const int64_t* __restrict__ target, const int64_t* __restrict__ rows,
                                                      const int* __restrict__ cu, int64_t* __restrict__ neg_item,
                                                      int sample_neg, float* __restrict__ part, const int* __restrict__ state,
                                                      uint64_t seed, int n_items, int B, int L, int4* __restrict__ rec) {
    // rec != NULL (deterministic mode of the GRU4Rec step): per token a record {target, negative, dpos, dneg} (zero: no loss term) for the owners
    // of linear.hip launch_table_owner64 instead of the atomics into dE
    constexpr int NV = D / 32;                            // floats per lane (half-wave covers D)
    __shared__ int s_tgt[64], s_neg[64], s_list[64];
    __shared__ unsigned long long s_valid;
    __shared__ float red[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t0 = cu[b], n = cu[b + 1] - t0;
    const int64_t row = rows ? rows[b] : b;
    if (w == 0) {
        int64_t tgt = 0, ng = 1;
        if (lane < L) {
            tgt = target[row * L + lane];
            if (sample_neg) {
                const RngKey rk = make_rng(seed, (uint32_t)state[DR4SR_STATE_RNGSTEP], 0.f);
                ng = sample_neg_id(rk, (uint64_t)b * L + lane, n_items);
                neg_item[(size_t)b * L + lane] = ng;
            } else {
                ng = neg_item[(size_t)b * L + lane];
            }
            ng = ng < 0 ? 0 : (ng >= n_items ? n_items - 1 : ng);
        }
        const bool ok = lane < L && tgt > 0 && tgt < n_items;
        s_tgt[lane] = ok ? (int)tgt : 0;
        s_neg[lane] = (int)ng;
        const unsigned long long m = __ballot(ok);
        if (ok) s_list[__popcll(m & ((1ull << lane) - 1ull))] = lane;     // compacted list of loss positions
        if (lane == 0) s_valid = m;
    }
    __syncthreads();
    // rows < n whose target is PAD receive a zero upstream gradient (they are masked out of the loss)
    const unsigned long long valid = s_valid;
    for (int i = threadIdx.x; i < n * (D / 4); i += 256) {
        const int l = i / (D / 4), c = (i % (D / 4)) * 4;
        if (!((valid >> l) & 1ull)) st4(dZ + (size_t)(t0 + l) * D + c, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (rec && (int)threadIdx.x < n && !((valid >> threadIdx.x) & 1ull)) rec[t0 + threadIdx.x] = make_int4(0, 0, 0, 0);
    const int hw = threadIdx.x >> 5, l32 = threadIdx.x & 31;          // 8 half-waves
    float lsum = 0.f, cnt = 0.f;
    const int nvalid = __popcll(valid);
    for (int k = hw; k < nvalid; k += 8) {                 // half-waves stay in lock-step on consecutive list entries
        const int l = s_list[k];
        const int tgt = s_tgt[l], ng = s_neg[l];
        const bool in = l < n;
        float q[NV], ep[NV], en[NV];
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            q[j] = in ? Z[(size_t)(t0 + l) * D + l32 + 32 * j] : 0.f;
            ep[j] = E[(size_t)tgt * D + l32 + 32 * j];
            en[j] = E[(size_t)ng * D + l32 + 32 * j];
            sp += q[j] * ep[j];
            sn += q[j] * en[j];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sn += __shfl_xor(sn, o, 64); }
        lsum += softplus_f(-sp) + softplus_f(sn);
        cnt += 1.f;
        if (in) {
            const float dpos = -sigmoid_f(-sp), dneg = sigmoid_f(sn);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                dZ[(size_t)(t0 + l) * D + l32 + 32 * j] = dpos * ep[j] + dneg * en[j];
                if (!rec) {
                    unsafeAtomicAdd(dE + (size_t)tgt * D + l32 + 32 * j, dpos * q[j]);
                    unsafeAtomicAdd(dE + (size_t)ng * D + l32 + 32 * j, dneg * q[j]);
                }
            }
            if (rec && l32 == 0) rec[t0 + l] = make_int4(tgt, ng, __float_as_int(dpos), __float_as_int(dneg));
        }
    }
    if (l32 == 0) { red[2 * hw] = cnt; red[2 * hw + 1] = lsum; }
    __syncthreads();
    if (threadIdx.x == 0) {                      // deterministic per-sequence partial; summed by k_wgrad's reduce job
        float c = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { c += red[2 * i]; s2 += red[2 * i + 1]; }
        part[2 * b] = c;
        part[2 * b + 1] = s2;
    }
}

int launch_score_packed_raw(const float* Z, const float* E, float* dE, float* dZ, const int64_t* target, const int64_t* rows,
                            const int* cu, int64_t* neg_item, int sample_neg, float* part, const int* state, uint64_t seed,
                            int n_items, int B, int L, int D, hipStream_t s, int4* rec) {
    dim3 grid(B), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_score_packed<64>, grid, blk, 0, s, Z, E, dE, dZ, target, rows, cu, neg_item, sample_neg, part, state, seed, n_items, B, L, rec);
    else hipLaunchKernelGGL(k_score_packed<128>, grid, blk, 0, s, Z, E, dE, dZ, target, rows, cu, neg_item, sample_neg, part, state, seed, n_items, B, L, rec);
    return DR4SR_LAUNCH_CHECK();
}
int launch_score_packed(const dr4sr_sasrec_plan* p, const Workspace& ws, hipStream_t s) {
    return launch_score_packed_raw(ws.X[p->n_layer], p->params + ws.off[0], p->grads + ws.off[0], ws.dX[p->n_layer], p->item_id,
                                   p->rows, ws.cu, p->neg_item, p->sample_neg, ws.score_part, p->state, p->seed, p->n_items, p->B,
                                   p->L, p->D, s);
}

// ------------------------------------------------------------------------------------------------
// Dense API scorer (query [B,L,D]); one wave per position.  BPR = false: BinaryCrossEntropyLoss (loss_func.py:9-38, masked
// branch): -logsigmoid(pos) + softplus(neg);  BPR = true: BPRLoss (loss_func.py:40-48, K = 1 so softmax(ones) = 1):
// -logsigmoid(pos - neg) = softplus(neg - pos).
template <int D, bool BPR>
__global__ __launch_bounds__(256) void k_score_dense_fwd(const float* __restrict__ Q, const float* __restrict__ E,
                                                         const int64_t* __restrict__ target, const int64_t* __restrict__ neg,
                                                         float* __restrict__ pos_score, float* __restrict__ neg_score,
                                                         float* __restrict__ loss_pos, float* __restrict__ stats, int64_t npos) {
    constexpr int NV = D / 64;
    const int lane = threadIdx.x & 63;
    float lsum = 0.f, cnt = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < npos; i += (int64_t)gridDim.x * 4) {
        const int64_t tgt = target[i], ng = neg[i];
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float q = Q[i * D + lane + 64 * j];
            sp += q * E[tgt * D + lane + 64 * j];
            sn += q * E[ng * D + lane + 64 * j];
        }
        sp = wave_sum(sp);
        sn = wave_sum(sn);
        const bool pad = tgt == 0;
        const float lp = pad ? 0.f : (BPR ? softplus_f(sn - sp) : softplus_f(-sp) + softplus_f(sn));
        if (lane == 0) {
            if (pos_score) pos_score[i] = pad ? -INFINITY : sp;
            if (neg_score) neg_score[i] = sn;
            if (loss_pos) loss_pos[i] = lp;
        }
        if (!pad) { lsum += lp; cnt += 1.f; }
    }
    // one atomic pair per WORKGROUP: same-address fp32 atomics serialise at ~12 ns each (one pair per wave was the whole 36 us of
    // this kernel at B*L = 12 800)
    __shared__ float red[8];
    if (lane == 0) { red[2 * (threadIdx.x >> 6)] = cnt; red[2 * (threadIdx.x >> 6) + 1] = lsum; }
    __syncthreads();
    if (stats && threadIdx.x == 0) {
        const float c = (red[0] + red[2]) + (red[4] + red[6]), l = (red[1] + red[3]) + (red[5] + red[7]);
        if (c > 0.f) { unsafeAtomicAdd(stats + 0, c); unsafeAtomicAdd(stats + 1, l); }
    }
}

template <int D, bool BPR>
__global__ __launch_bounds__(256) void k_score_dense_bwd(const float* __restrict__ Q, const float* __restrict__ E,
                                                         const int64_t* __restrict__ target, const int64_t* __restrict__ neg,
                                                         const float* __restrict__ wgt, const float* __restrict__ scale,
                                                         float* __restrict__ dQ, float* __restrict__ dE, int64_t npos) {
    constexpr int NV = D / 64;
    const int lane = threadIdx.x & 63;
    const float sc = scale ? *scale : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < npos; i += (int64_t)gridDim.x * 4) {
        const int64_t tgt = target[i], ng = neg[i];
        if (tgt == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) dQ[i * D + lane + 64 * j] = 0.f;
            continue;
        }
        float q[NV], ep[NV], en[NV];
        float sp = 0.f, sn = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            q[j] = Q[i * D + lane + 64 * j];
            ep[j] = E[tgt * D + lane + 64 * j];
            en[j] = E[ng * D + lane + 64 * j];
            sp += q[j] * ep[j];
            sn += q[j] * en[j];
        }
        sp = wave_sum(sp);
        sn = wave_sum(sn);
        const float up = sc * (wgt ? wgt[i] : 1.f);
        const float dneg = (BPR ? sigmoid_f(sn - sp) : sigmoid_f(sn)) * up;
        const float dpos = BPR ? -dneg : -sigmoid_f(-sp) * up;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            dQ[i * D + lane + 64 * j] = dpos * ep[j] + dneg * en[j];
            if (dE) {
                unsafeAtomicAdd(dE + tgt * D + lane + 64 * j, dpos * q[j]);
                unsafeAtomicAdd(dE + ng * D + lane + 64 * j, dneg * q[j]);
            }
        }
    }
}

template <bool BPR>
static int score_dense_fwd(const float* query, const float* E, const int64_t* target, const int64_t* neg,
                           float* pos_score, float* neg_score, float* loss_pos, float* stats, int64_t B,
                           int32_t L, int32_t D, void* stream) {
    if (!query || !E || !target || !neg || B < 0 || L <= 0) return DR4SR_E_ARG;
    if (D != 64 && D != 128) return DR4SR_E_SHAPE;
    const int64_t npos = B * L;
    if (npos == 0) return 0;
    int64_t blocks = (npos + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64) hipLaunchKernelGGL((k_score_dense_fwd<64, BPR>), dim3((unsigned)blocks), dim3(256), 0, s, query, E, target, neg, pos_score, neg_score, loss_pos, stats, npos);
    else hipLaunchKernelGGL((k_score_dense_fwd<128, BPR>), dim3((unsigned)blocks), dim3(256), 0, s, query, E, target, neg, pos_score, neg_score, loss_pos, stats, npos);
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_score_bce_fwd(const float* query, const float* E, const int64_t* target, const int64_t* neg,
                                   float* pos_score, float* neg_score, float* loss_pos, float* stats, int64_t B,
                                   int32_t L, int32_t D, void* stream) {
    return score_dense_fwd<false>(query, E, target, neg, pos_score, neg_score, loss_pos, stats, B, L, D, stream);
}
extern "C" int dr4sr_score_bpr_fwd(const float* query, const float* E, const int64_t* target, const int64_t* neg,
                                   float* pos_score, float* neg_score, float* loss_pos, float* stats, int64_t B,
                                   int32_t L, int32_t D, void* stream) {
    return score_dense_fwd<true>(query, E, target, neg, pos_score, neg_score, loss_pos, stats, B, L, D, stream);
}

// Deterministic mode (DR4SR_DETERMINISTIC; round 6): dE of the dense scorer without atomics and without a workspace (the entry point has none).
// Owner wave o of G = 1024 adds, in position order (target term before negative term), the rows whose id is o modulo G: it scans the
// target / negative ids 64 positions at a time, recomputes the two coefficients of the positions it owns a row of (three row loads, two
// wave sums) and read-modify-writes the table-gradient row itself — a row has one writer, its sum one order.
constexpr int SCORE_OWNERS = 1024;
template <int D, bool BPR>
__global__ __launch_bounds__(256) void k_score_dense_owner(const float* __restrict__ Q, const float* __restrict__ E,
                                                           const int64_t* __restrict__ target, const int64_t* __restrict__ neg,
                                                           const float* __restrict__ wgt, const float* __restrict__ scale,
                                                           float* dE, int64_t npos) {
    constexpr int NV = D / 64;
    const int lane = threadIdx.x & 63, owner = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float sc = scale ? *scale : 1.f;
    for (int64_t base = 0; base < npos; base += 64) {
        const int64_t i = base + lane;
        const int64_t tgt = i < npos ? target[i] : 0, ng = i < npos ? neg[i] : 0;
        const bool live = tgt != 0;
        unsigned long long m = __ballot(live && ((int)(tgt & (SCORE_OWNERS - 1)) == owner || (int)(ng & (SCORE_OWNERS - 1)) == owner));
        while (m) {
            const int l = __ffsll((long long)m) - 1; m &= m - 1;
            const int64_t p = base + l;
            const int64_t t = __shfl(tgt, l, 64), n = __shfl(ng, l, 64);          // (64-bit shuffles: two dwords each)
            float q[NV], sp = 0.f, sn = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                q[j] = Q[p * D + lane + 64 * j];
                sp += q[j] * E[t * D + lane + 64 * j];
                sn += q[j] * E[n * D + lane + 64 * j];
            }
            sp = wave_sum(sp);
            sn = wave_sum(sn);
            const float up = sc * (wgt ? wgt[p] : 1.f);
            const float dneg = (BPR ? sigmoid_f(sn - sp) : sigmoid_f(sn)) * up;
            const float dpos = BPR ? -dneg : -sigmoid_f(-sp) * up;
            if ((int)(t & (SCORE_OWNERS - 1)) == owner) {
#pragma unroll
                for (int j = 0; j < NV; ++j) dE[t * D + lane + 64 * j] += dpos * q[j];
            }
            if ((int)(n & (SCORE_OWNERS - 1)) == owner) {
#pragma unroll
                for (int j = 0; j < NV; ++j) dE[n * D + lane + 64 * j] += dneg * q[j];
            }
        }
    }
}

template <bool BPR>
static int score_dense_bwd(const float* query, const float* E, const int64_t* target, const int64_t* neg,
                           const float* w, const float* scale, float* d_query, float* dE, int64_t B, int32_t L,
                           int32_t D, void* stream) {
    if (!query || !E || !target || !neg || !d_query || B < 0 || L <= 0) return DR4SR_E_ARG;
    if (D != 64 && D != 128) return DR4SR_E_SHAPE;
    const int64_t npos = B * L;
    if (npos == 0) return 0;
    int64_t blocks = (npos + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    const char* de = DR4SR_ENV("DR4SR_DETERMINISTIC");
    const bool det = dE && de && atoi(de) != 0;
    float* dE_at = det ? nullptr : dE;                      // deterministic mode: the first launch writes d_query only
    if (D == 64) hipLaunchKernelGGL((k_score_dense_bwd<64, BPR>), dim3((unsigned)blocks), dim3(256), 0, s, query, E, target, neg, w, scale, d_query, dE_at, npos);
    else hipLaunchKernelGGL((k_score_dense_bwd<128, BPR>), dim3((unsigned)blocks), dim3(256), 0, s, query, E, target, neg, w, scale, d_query, dE_at, npos);
    if (det) {
        if (D == 64) hipLaunchKernelGGL((k_score_dense_owner<64, BPR>), dim3(SCORE_OWNERS / 4), dim3(256), 0, s, query, E, target, neg, w, scale, dE, npos);
        else hipLaunchKernelGGL((k_score_dense_owner<128, BPR>), dim3(SCORE_OWNERS / 4), dim3(256), 0, s, query, E, target, neg, w, scale, dE, npos);
    }
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_score_bce_bwd(const float* query, const float* E, const int64_t* target, const int64_t* neg,
                                   const float* w, const float* scale, float* d_query, float* dE, int64_t B, int32_t L,
                                   int32_t D, void* stream) {
    return score_dense_bwd<false>(query, E, target, neg, w, scale, d_query, dE, B, L, D, stream);
}
extern "C" int dr4sr_score_bpr_bwd(const float* query, const float* E, const int64_t* target, const int64_t* neg,
                                   const float* w, const float* scale, float* d_query, float* dE, int64_t B, int32_t L,
                                   int32_t D, void* stream) {
    return score_dense_bwd<true>(query, E, target, neg, w, scale, d_query, dE, B, L, D, stream);
}

// ------------------------------------------------------------------------------------------------
// The two loss modules on SCORE tensors (model/loss_func.py called directly, outside training_step): pos [n] (-inf = padded
// position), neg [n, K].  kind 0 = BinaryCrossEntropyLoss.forward, masked branch (:26-31): -logsigmoid(pos) + sum_k softplus(neg_k)/K;
// kind 1 = BPRLoss.forward (:44-49): -sum_k logsigmoid(pos - neg_k) * softmax(ones)_k = sum_k softplus(neg_k - pos) / K.
// loss_pos [n] UN-normalised per position (0 at padded positions); stats[0] += #valid, stats[1] += sum.  Backward: upstream
// g[n] per position (NULL = 1) times *scale (NULL = 1): d_pos [n], d_neg [n, K] written.
__global__ __launch_bounds__(256) void k_loss_scores_fwd(const float* __restrict__ pos, const float* __restrict__ neg, int K, int kind,
                                                         float* __restrict__ loss_pos, float* __restrict__ stats, int64_t n) {
    float lsum = 0.f, cnt = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sp = pos[i];
        const bool pad = isinf(sp);
        float l = 0.f;
        if (!pad) {
            float a = 0.f;
            for (int k = 0; k < K; ++k) a += kind ? softplus_f(neg[i * K + k] - sp) : softplus_f(neg[i * K + k]);
            l = a / (float)K + (kind ? 0.f : softplus_f(-sp));
            lsum += l; cnt += 1.f;
        }
        if (loss_pos) loss_pos[i] = l;
    }
    __shared__ float red[8];
    lsum = wave_sum(lsum); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = cnt; red[2 * (threadIdx.x >> 6) + 1] = lsum; }
    __syncthreads();
    if (stats && threadIdx.x == 0) {
        const float c = (red[0] + red[2]) + (red[4] + red[6]), l = (red[1] + red[3]) + (red[5] + red[7]);
        if (c > 0.f) { unsafeAtomicAdd(stats + 0, c); unsafeAtomicAdd(stats + 1, l); }
    }
}
__global__ __launch_bounds__(256) void k_loss_scores_bwd(const float* __restrict__ pos, const float* __restrict__ neg, int K, int kind,
                                                         const float* __restrict__ g, const float* __restrict__ scale,
                                                         float* __restrict__ d_pos, float* __restrict__ d_neg, int64_t n) {
    const float sc = scale ? *scale : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sp = pos[i];
        const bool pad = isinf(sp);
        const float up = pad ? 0.f : sc * (g ? g[i] : 1.f);
        float dp = (kind || pad) ? 0.f : -sigmoid_f(-sp) * up;
        for (int k = 0; k < K; ++k) {
            const float x = neg[i * K + k];
            const float dn = pad ? 0.f : (kind ? sigmoid_f(x - sp) : sigmoid_f(x)) * up / (float)K;
            d_neg[i * K + k] = dn;
            if (kind) dp -= dn;
        }
        d_pos[i] = dp;
    }
}
extern "C" int dr4sr_loss_from_scores_fwd(const float* pos, const float* neg, int64_t n, int32_t K, int32_t kind, float* loss_pos,
                                          float* stats, void* stream) {
    if (!pos || !neg || n < 0 || K <= 0 || kind < 0 || kind > 1) return DR4SR_E_ARG;
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_loss_scores_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pos, neg, K, kind, loss_pos, stats, n);
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_loss_from_scores_bwd(const float* pos, const float* neg, int64_t n, int32_t K, int32_t kind, const float* g,
                                          const float* scale, float* d_pos, float* d_neg, void* stream) {
    if (!pos || !neg || !d_pos || !d_neg || n < 0 || K <= 0 || kind < 0 || kind > 1) return DR4SR_E_ARG;
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_loss_scores_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pos, neg, K, kind, g, scale, d_pos, d_neg, n);
    return DR4SR_LAUNCH_CHECK();
}
