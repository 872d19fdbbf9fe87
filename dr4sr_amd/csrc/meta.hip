// meta.hip — DR4SR+ MetaModel pieces (reference model/metamodel.py:52-57, :169-194; utils/utils.py:134-252).
//
//   k_meta_select_fwd / _bwd : weight[p] = gumbel_softmax(meta_module(query[p]), tau)[0] with the two masks of
//                              metamodel.py:180-185, and its backward (d_query accumulated, d_phi as per-block partials
//                              reduced in a fixed order -> deterministic, which the finite-difference hyper-gradient needs).
//   k_fd_*                   : flat-vector helpers of the first-order (finite-difference) form of Hypergrad.grad.
//   k_meta_sgd               : clip_grad_norm_ + SGD(momentum, weight_decay) on the 4 290 meta parameters.
//
// One wave per position, lane j = hidden unit j of the D->D->2 MLP; W1 (both orientations) lives in LDS.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MD = 64;                                   // embed_dim == hidden width of the meta module
constexpr int N_PHI = MD * MD + MD + 2 * MD + 2;         // W1[D,D] b1[D] W2[2,D] b2[2]
constexpr int SEL_WAVES = 4;

struct SelArgs {
    const float* q;            // [n, D]
    const float* phi;          // flat meta parameters
    const float* gumbel;       // [n, 2] or null (Philox: -log(-log(u)))
    const int64_t* user_id;    // [B] or null
    const int64_t* target;     // [n]
    const uint64_t* gate_in;   // [n] frozen ReLU pattern or null
    uint64_t* gate_out;        // [n] or null
    float* weight;             // [n]           (fwd)
    const float* d_weight;     // [n]           (bwd) upstream dL/dweight
    const float* scale;        // device scalar multiplying d_weight, or null
    float* d_query;            // [n, D] accumulated (bwd), may be null
    float* part;               // [gridDim.x, N_PHI] per-block partial d_phi (bwd)
    int64_t n;
    int L;
    float inv_tau;
    uint64_t seed;
    uint32_t step;
    const int* step_dev;       // when non-null the Gumbel step is read from the device (graph replays draw fresh noise)
};

__device__ __forceinline__ float2 gumbel_pair(const SelArgs& A, int64_t p) {
    if (A.gumbel) return make_float2(A.gumbel[2 * p], A.gumbel[2 * p + 1]);
    const uint32_t step = A.step_dev ? (uint32_t)*A.step_dev : A.step;
    const uint4 r = philox4x32_10(make_uint4((uint32_t)p, (uint32_t)(p >> 32), 0x6D657461u, step),
                                  make_uint2((uint32_t)A.seed, (uint32_t)(A.seed >> 32)));
    // u in (0,1): (r + 0.5) / 2^32 ; torch: gumbel = -log(Exp(1)) = -log(-log(u))
    const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return make_float2(-logf(-logf(u0)), -logf(-logf(u1)));
}

// stage W1 ([j][d] row-major in phi) as W1T[d][j] (+ optionally W1 itself) into LDS
__device__ __forceinline__ void stage_w1(const float* __restrict__ phi, float* __restrict__ w1t, float* __restrict__ w1) {
    for (int i = threadIdx.x; i < MD * MD; i += blockDim.x) {
        const float v = phi[i];
        const int j = i / MD, d = i % MD;
        w1t[d * MD + j] = v;
        if (w1) w1[i] = v;
    }
}

// pre-activation of unit `lane` for the position whose query sits in qs[0..D)
__device__ __forceinline__ float unit_pre(const float* __restrict__ w1t, const float* __restrict__ qs, float b1, int lane) {
    float a0 = b1, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int d = 0; d < MD; d += 4) {
        a0 = fmaf(w1t[(d + 0) * MD + lane], qs[d + 0], a0);
        a1 = fmaf(w1t[(d + 1) * MD + lane], qs[d + 1], a1);
        a2 = fmaf(w1t[(d + 2) * MD + lane], qs[d + 2], a2);
        a3 = fmaf(w1t[(d + 3) * MD + lane], qs[d + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(SEL_WAVES * 64) void k_meta_select_fwd(SelArgs A) {
    __shared__ float w1t[MD * MD];
    __shared__ float qs[SEL_WAVES][MD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    stage_w1(A.phi, w1t, nullptr);
    const float b1 = A.phi[MD * MD + lane];
    const float w20 = A.phi[MD * MD + MD + lane], w21 = A.phi[MD * MD + 2 * MD + lane];
    const float b20 = A.phi[MD * MD + 3 * MD], b21 = A.phi[MD * MD + 3 * MD + 1];
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * SEL_WAVES + w; p < A.n; p += (int64_t)gridDim.x * SEL_WAVES) {
        if (A.target[p] == 0) {                                     // metamodel.py:184-185: PAD target -> weight 0
            if (lane == 0) { A.weight[p] = 0.f; if (A.gate_out) A.gate_out[p] = 0; }
            continue;
        }
        qs[w][lane] = A.q[p * MD + lane];
        __builtin_amdgcn_wave_barrier();
        const float pre = unit_pre(w1t, qs[w], b1, lane);
        const uint64_t gate = A.gate_in ? A.gate_in[p] : __ballot(pre > 0.f);
        const float h = ((gate >> lane) & 1) ? pre : 0.f;
        const float s0 = wave_sum(w20 * h) + b20, s1 = wave_sum(w21 * h) + b21;
        const float2 g = gumbel_pair(A, p);
        const float z = ((s0 + g.x) - (s1 + g.y)) * A.inv_tau;
        const bool forced = A.user_id && A.user_id[p / A.L] == 0;    // metamodel.py:180-183: pattern rows -> weight 1
        if (lane == 0) {
            A.weight[p] = forced ? 1.0f : 1.0f / (1.0f + expf(-z));
            if (A.gate_out) A.gate_out[p] = gate;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(SEL_WAVES * 64) void k_meta_select_bwd(SelArgs A) {
    __shared__ float w1t[MD * MD];
    __shared__ float w1[MD * MD];
    __shared__ float qs[SEL_WAVES][MD];
    __shared__ float dhs[SEL_WAVES][MD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    stage_w1(A.phi, w1t, w1);
    const float b1 = A.phi[MD * MD + lane];
    const float w20 = A.phi[MD * MD + MD + lane], w21 = A.phi[MD * MD + 2 * MD + lane];
    const float b20 = A.phi[MD * MD + 3 * MD], b21 = A.phi[MD * MD + 3 * MD + 1];
    const float sc = A.scale ? *A.scale : 1.0f;
    float dW1[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) dW1[d] = 0.f;
    float db1 = 0.f, dw20 = 0.f, dw21 = 0.f, db2 = 0.f;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * SEL_WAVES + w; p < A.n; p += (int64_t)gridDim.x * SEL_WAVES) {
        if (A.target[p] == 0) continue;
        if (A.user_id && A.user_id[p / A.L] == 0) continue;         // forced weight: constant, no gradient
        const float up = A.d_weight[p] * sc;
        if (up == 0.f) continue;
        qs[w][lane] = A.q[p * MD + lane];
        __builtin_amdgcn_wave_barrier();
        const float pre = unit_pre(w1t, qs[w], b1, lane);
        const uint64_t gate = A.gate_in ? A.gate_in[p] : __ballot(pre > 0.f);
        const bool on = (gate >> lane) & 1;
        const float h = on ? pre : 0.f;
        const float s0 = wave_sum(w20 * h) + b20, s1 = wave_sum(w21 * h) + b21;
        const float2 g = gumbel_pair(A, p);
        const float z = ((s0 + g.x) - (s1 + g.y)) * A.inv_tau;
        const float y = 1.0f / (1.0f + expf(-z));
        const float dz = up * y * (1.0f - y) * A.inv_tau;           // d/d logit0 = +dz, d/d logit1 = -dz
        const float dh = on ? (w20 - w21) * dz : 0.f;
        dw20 = fmaf(dz, h, dw20);
        dw21 = fmaf(-dz, h, dw21);
        db2 += dz;
        db1 += dh;
#pragma unroll
        for (int d = 0; d < MD; ++d) dW1[d] = fmaf(dh, qs[w][d], dW1[d]);
        if (A.d_query) {
            dhs[w][lane] = dh;
            __builtin_amdgcn_wave_barrier();
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int j = 0; j < MD; j += 2) {
                a0 = fmaf(w1[j * MD + lane], dhs[w][j], a0);
                a1 = fmaf(w1[(j + 1) * MD + lane], dhs[w][j + 1], a1);
            }
            A.d_query[p * MD + lane] += a0 + a1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // cross-wave reduction in LDS (reusing w1t/w1), then one partial row per block
    __syncthreads();
    float* red = w1t;                                               // [MD][MD] for dW1, w1[] for the small vectors
    for (int ww = 0; ww < SEL_WAVES; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int d = 0; d < MD; ++d) {
                float* r = red + lane * MD + ((d + lane) & (MD - 1));   // skewed: conflict-free
                *r = ww == 0 ? dW1[d] : *r + dW1[d];
            }
            float* sm = w1;
            sm[lane] = ww == 0 ? db1 : sm[lane] + db1;
            sm[MD + lane] = ww == 0 ? dw20 : sm[MD + lane] + dw20;
            sm[2 * MD + lane] = ww == 0 ? dw21 : sm[2 * MD + lane] + dw21;
            sm[3 * MD + lane] = ww == 0 ? db2 : sm[3 * MD + lane] + db2;
        }
        __syncthreads();
    }
    float* out = A.part + (size_t)blockIdx.x * N_PHI;
    for (int i = threadIdx.x; i < MD * MD; i += blockDim.x) {
        const int j = i / MD, d = i % MD;
        out[i] = red[j * MD + ((d + j) & (MD - 1))];
    }
    if (threadIdx.x < 3 * MD) out[MD * MD + threadIdx.x] = w1[threadIdx.x];
    if (threadIdx.x == 0) {                                         // db2: every lane carries the same sum of dz
        out[MD * MD + 3 * MD] = w1[3 * MD];
        out[MD * MD + 3 * MD + 1] = -w1[3 * MD];
    }
}

// d_phi[i] += sum over blocks (fixed order)
// 64 columns per workgroup, the partial rows split over its 4 waves with 8 independent loads in flight (a single thread walking
// all rows of a column is a chain of dependent-latency loads: 32 us for 4 290 columns)
__global__ __launch_bounds__(256) void k_meta_reduce(const float* __restrict__ part, int nblk, float* __restrict__ d_phi) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6, i = blockIdx.x * 64 + c;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    if (i < N_PHI)
        for (int b = w * 8; b < nblk; b += 32)
#pragma unroll
            for (int u = 0; u < 8; ++u) if (b + u < nblk) acc[u] += part[(size_t)(b + u) * N_PHI + i];
    red[w][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w == 0 && i < N_PHI) d_phi[i] += (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ------------------------------------------------------------------------------------------------ FD helpers
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    return t;
}

// e = rel * sqrt( sum theta^2 over {dir != 0} / sum dir^2 ).  Deterministic two-level sum: FD_BLOCKS workgroups leave one partial pair
// each, the last one to finish (ticket) adds them in index order.  (The first version walked all 833 k parameters in ONE workgroup:
// 320 us per call, four calls per outer step.)
constexpr int FD_BLOCKS = 256;
__device__ float g_fd_part[2 * FD_BLOCKS];
__device__ int g_fd_ticket;
__global__ __launch_bounds__(256) void k_fd_step_size(const float* __restrict__ theta, const float* __restrict__ dir, int64_t n,
                                                     float rel, float* __restrict__ out_e, float* __restrict__ part, int* __restrict__ ticket) {
    __shared__ float sm[16];
    __shared__ int last;
    float st = 0.f, sd = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)FD_BLOCKS * 256) {
        const float d = dir[i], t = theta[i];
        if (d != 0.f) { st = fmaf(t, t, st); sd = fmaf(d, d, sd); }
    }
    st = block_sum(st, sm);
    sd = block_sum(sd, sm);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[2 * blockIdx.x], st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part[2 * blockIdx.x + 1], sd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1) == FD_BLOCKS - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float pt = 0.f, pd = 0.f;
    if (threadIdx.x < FD_BLOCKS) {
        pt = __hip_atomic_load(&part[2 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pd = __hip_atomic_load(&part[2 * threadIdx.x + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pt = block_sum(pt, sm);                                // fixed reduction tree over the partials: run-to-run identical
    pd = block_sum(pd, sm);
    if (threadIdx.x == 0) { out_e[0] = pd > 0.f ? rel * sqrtf(pt / pd) : 0.f; *ticket = 0; }
}

// out = x + sign * e * dir
__global__ void k_fd_shift(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ dir,
                           const float* __restrict__ e, float sign, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fmaf(sign * e[0], dir[i], x[i]);
}

// v -= lr * (gp/np - gm/nm) / (2e) ; pacc += v        (utils/utils.py:196-203 with H v by central difference)
__global__ void k_fd_neumann(float* __restrict__ v, float* __restrict__ pacc, const float* __restrict__ gp,
                             const float* __restrict__ gm, const float* __restrict__ np, const float* __restrict__ nm,
                             const float* __restrict__ e, float lr, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ee = e[0];
    const float hv = ee > 0.f ? (gp[i] / np[0] - gm[i] / nm[0]) / (2.0f * ee) : 0.f;
    const float nv = v[i] - lr * hv;
    v[i] = nv;
    pacc[i] += nv;
}

// out = coef * (fp/np - fm/nm) / (2e)
__global__ void k_fd_diff(float* __restrict__ out, const float* __restrict__ fp, const float* __restrict__ fm,
                          const float* __restrict__ np, const float* __restrict__ nm, const float* __restrict__ e, float coef,
                          int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ee = e[0];
    out[i] = ee > 0.f ? coef * (fp[i] / np[0] - fm[i] / nm[0]) / (2.0f * ee) : 0.f;
}

// Richardson-extrapolated central difference from the probes at +-e and +-2e: (4 D(e) - D(2e)) / 3, D(h) = (f(+h) - f(-h)) / 2h
__global__ void k_fd_diff4(float* __restrict__ out, const float* __restrict__ fp1, const float* __restrict__ fm1,
                           const float* __restrict__ fp2, const float* __restrict__ fm2, const float* __restrict__ nv,
                           const float* __restrict__ e, float coef, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ee = e[0];
    const float d1 = (fp1[i] / nv[0] - fm1[i] / nv[1]) / (2.0f * ee);
    const float d2 = (fp2[i] / nv[2] - fm2[i] / nv[3]) / (4.0f * ee);
    out[i] = ee > 0.f ? coef * (4.0f * d1 - d2) * (1.0f / 3.0f) : 0.f;
}

// out = x / *den
__global__ void k_scale_by(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ den, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] / den[0];
}

// clip_grad_norm_(max_norm) + torch.optim.SGD(momentum, weight_decay) (utils/utils.py:240-247, metamodel.py:68-69). Single block.
__global__ __launch_bounds__(1024) void k_meta_sgd(float* __restrict__ phi, const float* __restrict__ g, float* __restrict__ buf,
                                                  int n, float lr, float momentum, float wd, float max_norm,
                                                  int* __restrict__ step_count, float* __restrict__ out_norm) {
    __shared__ float sm[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(g[i], g[i], s);
    const float total = sqrtf(block_sum(s, sm));
    const float coef = max_norm > 0.f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;
    const bool first = step_count[0] == 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float gi = fmaf(wd, phi[i], g[i] * coef);
        const float b = first ? gi : fmaf(momentum, buf[i], gi);
        buf[i] = b;
        phi[i] -= lr * b;
    }
    __syncthreads();
    if (threadIdx.x == 0) { step_count[0] += 1; if (out_norm) out_norm[0] = total; }
}

// The reference's other meta-optimizer choices (metamodel.py:59-81) behind the same clip_grad_norm_: torch.optim.Adam(lr[, weight_decay
// in the else branch]), Adagrad(lr), RMSprop(lr) with torch's defaults (the formulas of k_adam<OPT> in step.hip: single-tensor Adam
// with double-precision bias corrections; Adagrad eps 1e-10, accumulator 0; RMSprop alpha 0.99, eps 1e-8, no momentum, not centered).
// kind = DR4SR_OPT_*; m / v = the optimizer's state vectors (exp_avg | exp_avg_sq / state_sum / square_avg); step_count = t - 1.
__global__ __launch_bounds__(1024) void k_meta_opt(int kind, float* __restrict__ phi, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int n, float lr, float b1, float b2, float eps, float wd,
                                                  float max_norm, int* __restrict__ step_count, float* __restrict__ out_norm) {
    __shared__ float sm[16];
    __shared__ float bc[2];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(g[i], g[i], s);
    const float total = sqrtf(block_sum(s, sm));
    const float coef = max_norm > 0.f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;
    const int t = step_count[0] + 1;
    if (threadIdx.x == 0) {
        bc[0] = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
        bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t)));
    }
    __syncthreads();
    const float step_size = bc[0], inv_sqrt_bc2 = bc[1];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float pe = phi[i];
        const float gi = fmaf(wd, pe, g[i] * coef);
        if (kind == DR4SR_OPT_ADAGRAD) {
            const float ve = v[i] + gi * gi;
            v[i] = ve;
            pe = pe - lr * (gi / (sqrtf(ve) + eps));
        } else if (kind == DR4SR_OPT_RMSPROP) {
            const float ve = v[i] * b2 + (1.0f - b2) * gi * gi;
            v[i] = ve;
            pe = pe - lr * (gi / (sqrtf(ve) + eps));
        } else {
            const float me = m[i] + (gi - m[i]) * (1.0f - b1);
            const float ve = v[i] * b2 + (1.0f - b2) * gi * gi;
            m[i] = me; v[i] = ve;
            pe = pe - step_size * (me / (sqrtf(ve) * inv_sqrt_bc2 + eps));
        }
        phi[i] = pe;
    }
    __syncthreads();
    if (threadIdx.x == 0) { step_count[0] = t; if (out_norm) out_norm[0] = total; }
}

int sel_grid(int64_t n) {
    int64_t g = (n + SEL_WAVES * 8 - 1) / (SEL_WAVES * 8);
    return (int)(g < 1 ? 1 : g > 128 ? 128 : g);
}

}  // namespace

extern "C" int64_t dr4sr_meta_param_count(int32_t D) { return D == MD ? N_PHI : DR4SR_E_SHAPE; }

extern "C" int64_t dr4sr_meta_select_workspace_floats(int64_t n) { return (int64_t)sel_grid(n) * N_PHI; }

extern "C" int dr4sr_meta_select_fwd(const float* query, const float* phi, const float* gumbel, uint64_t seed, uint32_t step,
                                     const int32_t* step_dev, float tau, const int64_t* user_id, const int64_t* target, int64_t B, int32_t L, int32_t D,
                                     const uint64_t* gate_in, uint64_t* gate_out, float* weight, void* stream) {
    if (!query || !phi || !target || !weight || B < 0 || L <= 0 || !(tau > 0.f)) return DR4SR_E_ARG;
    if (D != MD) return DR4SR_E_SHAPE;
    const int64_t n = B * L;
    if (n == 0) return 0;
    SelArgs A{};
    A.q = query; A.phi = phi; A.gumbel = gumbel; A.user_id = user_id; A.target = target; A.gate_in = gate_in; A.gate_out = gate_out;
    A.weight = weight; A.n = n; A.L = L; A.inv_tau = 1.0f / tau; A.seed = seed; A.step = step; A.step_dev = step_dev;
    hipLaunchKernelGGL(k_meta_select_fwd, dim3(sel_grid(n)), dim3(SEL_WAVES * 64), 0, (hipStream_t)stream, A);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_meta_select_bwd(const float* query, const float* phi, const float* gumbel, uint64_t seed, uint32_t step,
                                     const int32_t* step_dev, float tau, const int64_t* user_id, const int64_t* target, int64_t B, int32_t L, int32_t D,
                                     const uint64_t* gate_in, const float* d_weight, const float* scale, float* d_query,
                                     float* d_phi, float* workspace, void* stream) {
    if (!query || !phi || !target || !d_weight || !d_phi || !workspace || B < 0 || L <= 0 || !(tau > 0.f)) return DR4SR_E_ARG;
    if (D != MD) return DR4SR_E_SHAPE;
    const int64_t n = B * L;
    if (n == 0) return 0;
    SelArgs A{};
    A.q = query; A.phi = phi; A.gumbel = gumbel; A.user_id = user_id; A.target = target; A.gate_in = gate_in;
    A.d_weight = d_weight; A.scale = scale; A.d_query = d_query; A.part = workspace; A.n = n; A.L = L; A.inv_tau = 1.0f / tau;
    A.seed = seed; A.step = step; A.step_dev = step_dev;
    const int g = sel_grid(n);
    hipLaunchKernelGGL(k_meta_select_bwd, dim3(g), dim3(SEL_WAVES * 64), 0, (hipStream_t)stream, A);
    hipLaunchKernelGGL(k_meta_reduce, dim3((N_PHI + 63) / 64), dim3(256), 0, (hipStream_t)stream, workspace, g, d_phi);
    return (int)hipGetLastError();
}

extern "C" int64_t dr4sr_fd_step_size_scratch_floats(void) { return 2 * FD_BLOCKS + 4; }
// re-entrant form: the caller owns the reduction scratch (dr4sr_fd_step_size_scratch_floats() floats, zeroed once; one per
// concurrently running call)
extern "C" int dr4sr_fd_step_size_ws(const float* theta, const float* dir, int64_t n, float rel_step, float* out_e, float* scratch,
                                     void* stream) {
    if (!theta || !dir || !out_e || !scratch || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_fd_step_size, dim3(FD_BLOCKS), dim3(256), 0, (hipStream_t)stream, theta, dir, n, rel_step, out_e, scratch,
                       reinterpret_cast<int*>(scratch + 2 * FD_BLOCKS));
    return (int)hipGetLastError();
}
// convenience form on a module-level scratch: NOT re-entrant (two calls must not run concurrently on different streams)
extern "C" int dr4sr_fd_step_size(const float* theta, const float* dir, int64_t n, float rel_step, float* out_e, void* stream) {
    if (!theta || !dir || !out_e || n <= 0) return DR4SR_E_ARG;
    float* part = nullptr; int* ticket = nullptr;
    if (hipGetSymbolAddress((void**)&part, HIP_SYMBOL(g_fd_part)) != hipSuccess || hipGetSymbolAddress((void**)&ticket, HIP_SYMBOL(g_fd_ticket)) != hipSuccess)
        return (int)hipGetLastError();
    hipLaunchKernelGGL(k_fd_step_size, dim3(FD_BLOCKS), dim3(256), 0, (hipStream_t)stream, theta, dir, n, rel_step, out_e, part, ticket);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_fd_shift(float* out, const float* x, const float* dir, const float* e, float sign, int64_t n, void* stream) {
    if (!out || !x || !dir || !e || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_fd_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x, dir, e, sign, n);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_fd_neumann(float* v, float* pacc, const float* gp, const float* gm, const float* np, const float* nm,
                                const float* e, float lr, int64_t n, void* stream) {
    if (!v || !pacc || !gp || !gm || !np || !nm || !e || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_fd_neumann, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, pacc, gp, gm, np, nm,
                       e, lr, n);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_fd_diff(float* out, const float* fp, const float* fm, const float* np, const float* nm, const float* e,
                             float coef, int64_t n, void* stream) {
    if (!out || !fp || !fm || !np || !nm || !e || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_fd_diff, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, fp, fm, np, nm, e,
                       coef, n);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_fd_diff4(float* out, const float* fp1, const float* fm1, const float* fp2, const float* fm2, const float* nv4,
                              const float* e, float coef, int64_t n, void* stream) {
    if (!out || !fp1 || !fm1 || !fp2 || !fm2 || !nv4 || !e || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_fd_diff4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, fp1, fm1, fp2, fm2,
                       nv4, e, coef, n);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_scale_by(float* out, const float* x, const float* den, int64_t n, void* stream) {
    if (!out || !x || !den || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_scale_by, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, x, den, n);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_meta_opt_step(int32_t optimizer, float* phi, const float* grad, float* state_m, float* state_v, int32_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, float max_norm, int32_t* step_count,
                                   float* out_norm, void* stream) {
    if (!phi || !grad || !state_m || !state_v || !step_count || n <= 0) return DR4SR_E_ARG;
    if (optimizer != DR4SR_OPT_ADAM && optimizer != DR4SR_OPT_ADAGRAD && optimizer != DR4SR_OPT_RMSPROP) return DR4SR_E_ARG;   // SGD: dr4sr_meta_sgd_step
    hipLaunchKernelGGL(k_meta_opt, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)optimizer, phi, grad, state_m, state_v, n, lr, beta1,
                       beta2, eps, weight_decay, max_norm, step_count, out_norm);
    return (int)hipGetLastError();
}

extern "C" int dr4sr_meta_sgd_step(float* phi, const float* grad, float* momentum_buf, int32_t n, float lr, float momentum,
                                   float weight_decay, float max_norm, int32_t* step_count, float* out_norm, void* stream) {
    if (!phi || !grad || !momentum_buf || !step_count || n <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_meta_sgd, dim3(1), dim3(1024), 0, (hipStream_t)stream, phi, grad, momentum_buf, n, lr, momentum,
                       weight_decay, max_norm, step_count, out_norm);
    return (int)hipGetLastError();
}
