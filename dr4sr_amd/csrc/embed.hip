// embed.hip — K1 item-embedding gather + position add (dense, bit-exact) and its packed/ragged
// training variants (forward with dropout, backward scatter-add), plus the prefix-scan "prep" kernel.
//
// Reference: model/sasrec.py:43-46,:62-66 (item_encoder(idx) + position_emb(arange(L)), dropout);
// autograd of nn.Embedding(padding_idx=0) (model/basemodel.py:42) for the backward.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// K1 dense: out[b,l,:] = E[idx[b,l],:] + P[l,:] for ALL B*L positions (pads included: E[0] + P[l]).
// HBM-bound stream: 8 B idx read + 4D B row read (L2/MALL resident table) + 4D B write per token.
// One 16-lane group per token and float4 per lane (D=64) => each wave instruction reads/writes
// 4 x 256 B contiguous segments; grid-stride over tokens.
template <int D>
__global__ __launch_bounds__(256) void k_embed_dense(const float* __restrict__ E, const float* __restrict__ P,
                                                         const int64_t* __restrict__ idx, float* __restrict__ out,
                                                         int64_t ntok, int L, int n_items) {
    constexpr int LPT = D / 4;
    constexpr int TPB = 256 / LPT;
    const int sub = threadIdx.x / LPT, c = (threadIdx.x % LPT) * 4;
    for (int64_t t = (int64_t)blockIdx.x * TPB + sub; t < ntok; t += (int64_t)gridDim.x * TPB) {
        int64_t id = idx[t];
        id = id < 0 ? 0 : (id >= n_items ? n_items - 1 : id);
        const int l = (int)(t % L);
        const float4 e = ld4(E + id * D + c);
        const float4 p = ld4(P + (size_t)l * D + c);
        // streaming output: non-temporal store, so the 4.3 GB of rows written per launch do not evict the 3 MB table from L2 / MALL
        const f32x4 v = {e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + t * D + c));
    }
}

extern "C" int dr4sr_embed_gather_posadd(const float* E, const float* P, const int64_t* idx, float* out,
                                         int64_t B, int32_t L, int32_t D, int32_t n_items, void* stream) {
    if (!E || !P || !idx || !out || B < 0 || L <= 0 || n_items <= 0) return DR4SR_E_ARG;
    if (D != 64 && D != 128) return DR4SR_E_SHAPE;
    const int64_t ntok = B * L;
    if (ntok == 0) return 0;
    const int tpb = 256 / (D / 4);
    int64_t blocks = (ntok + tpb - 1) / tpb;
    const int64_t cap = DR4SR_XENV("DR4SR_GATHER_BLOCKS") ? atoll(DR4SR_XENV("DR4SR_GATHER_BLOCKS")) : 65536;     // measured: 16 tokens x 16 iterations per block beats 4096 long-running blocks by 12 %
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64) hipLaunchKernelGGL(k_embed_dense<64>, dim3((unsigned)blocks), dim3(256), 0, s, E, P, idx, out, ntok, L, n_items);
    else hipLaunchKernelGGL(k_embed_dense<128>, dim3((unsigned)blocks), dim3(256), 0, s, E, P, idx, out, ntok, L, n_items);
    return DR4SR_LAUNCH_CHECK();
}

// ids outside [0, n_items): torch's nn.Embedding (the reference's gather, model/sasrec.py:43) raises "index out of range in self" for
// them; the kernels clamp instead (a launch cannot raise), so callers that want the reference's behaviour count the offenders first —
// dr4sr_amd does it once per dataset tensor (model/basemodel.py) and in the dense dispatcher op (ops.py).  *bad += the count.
__global__ __launch_bounds__(256) void k_check_ids(const int64_t* __restrict__ idx, int64_t n, int n_items, int* __restrict__ bad) {
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t id = idx[i];
        c += (id < 0 || id >= n_items) ? 1 : 0;
    }
    c = (int)wave_sum((float)c);                        // (<= 64 * iterations per wave: exact in fp32 up to 2^24)
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(bad, c);
}
extern "C" int dr4sr_check_ids(const int64_t* idx, int64_t n, int32_t n_items, int32_t* bad_count, void* stream) {
    if (!idx || !bad_count || n < 0 || n_items <= 0) return DR4SR_E_ARG;
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_check_ids, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx, n, n_items, bad_count);
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// prep: cu[b] = exclusive prefix sum of clamp(seqlen[row(b)], 0, L); state[T] = total.  One block.
// Also bumps the RNG step so that every fwd_bwd draws fresh dropout masks / negatives.
// Blocks 1.. of the same launch zero the flat gradient (+tail) when `zero` is given (saves a launch per step; a
// hipMemsetAsync graph node is NOT used: on ROCm 7.2 its replay was observed to fill the last 16 bytes with a stale pattern).
#include "prep_body.h"
__global__ __launch_bounds__(1024) void k_prep(const PrepArgs P, float* __restrict__ zero, int64_t zero_n4) {
    if (blockIdx.x > 0) {
        for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < zero_n4; i += (int64_t)(gridDim.x - 1) * 1024)
            st4(zero + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    __shared__ unsigned long long part[1024];
    prep_body<1024>(P, part);
}

static int launch_prep_sel(const int64_t* seqlen, const int64_t* rows, int* cu, int* state, int B, int L, int bump_rng, float* zero,
                           int64_t zero_floats, const PermSel& sel, int* tile_seq, int* seq_class, hipStream_t s) {
    const int64_t n4 = zero ? zero_floats / 4 : 0;
    int zb = (int)((n4 + 1023) / 1024);
    if (zb > 255) zb = 255;
    const PrepArgs P{seqlen, rows, cu, state, B, L, bump_rng, sel, tile_seq, seq_class, nullptr};
    hipLaunchKernelGGL(k_prep, dim3(1 + zb), dim3(1024), 0, s, P, zero, n4);
    return DR4SR_LAUNCH_CHECK();
}
int launch_prep_raw(const int64_t* seqlen, const int64_t* rows, int* cu, int* state, int B, int L, int bump_rng, float* zero,
                    int64_t zero_floats, hipStream_t s) {
    return launch_prep_sel(seqlen, rows, cu, state, B, L, bump_rng, zero, zero_floats, PermSel{nullptr, 0, 0, 0, nullptr}, nullptr, nullptr, s);
}
// the same with the tile -> sequence hints (tile_seq [ceil(B L / 16) + 1]) and, optionally, the batch selection of the fused step
// (sel.perm != NULL: rows[] is FILLED from the epoch permutation first) — GRU4Rec's fused glue launches (gru.hip)
int launch_prep_raw_hints(const int64_t* seqlen, const int64_t* rows, int* cu, int* state, int B, int L, int bump_rng, float* zero,
                          int64_t zero_floats, int* tile_seq, const PermSel& sel, hipStream_t s) {
    return launch_prep_sel(seqlen, rows, cu, state, B, L, bump_rng, zero, zero_floats, sel, tile_seq, nullptr, s);
}
int make_prep_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int bump_rng, PrepArgs* out) {
    PermSel sel{nullptr, 0, 0, 0, nullptr};
    if (p->perm && bump_rng) {                          // selection only in the calls that start a new step
        if (!p->rows || !p->perm_counter || p->n_perm <= 0) return DR4SR_E_ARG;
        sel = PermSel{p->perm, p->n_perm, p->perm_stride, p->perm_offset, p->perm_counter};
    }
    // the length-class lists are only read by the at-scale attention LIST launches (attn_mfma.hip: split_by_length); the wave-per-tile form
    // (attn_wave.hip) needs none of them
    *out = PrepArgs{p->seqlen, p->rows, ws.cu, p->state, p->B, p->L, bump_rng, sel, ws.tile_seq, (ws.attn_split && !attn_wave_on(p, ws)) ? ws.seq_class : nullptr, ws.len_buf};
    return 0;
}
int launch_prep(const dr4sr_sasrec_plan* p, const Workspace& ws, int bump_rng, int zero_grads, hipStream_t s) {
    PrepArgs P;
    const int rc = make_prep_args(p, ws, bump_rng, &P);
    if (rc) return rc;
    return launch_prep_sel(p->seqlen, p->rows, ws.cu, p->state, p->B, p->L, bump_rng, zero_grads ? p->grads : nullptr,
                           ws.n_params + DR4SR_GRAD_TAIL, P.sel, ws.tile_seq, P.seq_class, s);
}

// ------------------------------------------------------------------------------------------------

// Packed forward, token-parallel: token t belongs to sequence slot b = find_seq(t) at pos = t - cu[b]:
//   x[t,:] = drop(E[idx[row(b),pos],:] + P[pos,:])      (P == NULL: no position table, GRU4Rec)
// One 16/32-lane group per token; the grid covers the worst case B*L tokens, groups beyond T exit.
template <int D>
__global__ __launch_bounds__(256) void k_embed_fwd(const float* __restrict__ E, const float* __restrict__ P,
                                                   const int64_t* __restrict__ idx, const int64_t* __restrict__ rows,
                                                   const int* __restrict__ cu, float* __restrict__ X, int B, int L,
                                                   int n_items, const int* __restrict__ state, uint64_t seed, float p,
                                                   int training) {
    constexpr int LPT = D / 4, TPB = 256 / LPT;
    const int T = state[DR4SR_STATE_T];
    const int t = blockIdx.x * TPB + threadIdx.x / LPT, c = (threadIdx.x % LPT) * 4;
    if (t >= T) return;
    const int b = find_seq(cu, B, t), pos = t - cu[b];
    const int64_t row = rows ? rows[b] : b;
    int64_t id = idx[row * L + pos];
    id = id < 0 ? 0 : (id >= n_items ? n_items - 1 : id);
    float4 o = ld4(E + id * D + c);
    if (P) {
        const float4 pe = ld4(P + (size_t)pos * D + c);
        o = make_float4(o.x + pe.x, o.y + pe.y, o.z + pe.z, o.w + pe.w);
    }
    if (training && p > 0.f) {
        const RngKey rk = make_rng(seed, (uint32_t)state[DR4SR_STATE_RNGSTEP], p);
        const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * L + pos) * D + c);
        o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
    }
    st4(X + (size_t)t * D + c, o);
}

int launch_embed_fwd_raw(const float* E, const float* P, const int64_t* idx, const int64_t* rows, const int* cu, float* X, int B,
                         int L, int D, int n_items, const int* state, uint64_t seed, float p, int training, hipStream_t s) {
    dim3 grid(((int64_t)B * L + (256 / (D / 4)) - 1) / (256 / (D / 4))), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_embed_fwd<64>, grid, blk, 0, s, E, P, idx, rows, cu, X, B, L, n_items, state, seed, p, training);
    else hipLaunchKernelGGL(k_embed_fwd<128>, grid, blk, 0, s, E, P, idx, rows, cu, X, B, L, n_items, state, seed, p, training);
    return DR4SR_LAUNCH_CHECK();
}
int launch_embed_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s) {
    return launch_embed_fwd_raw(p->params + ws.off[0], p->params + ws.off[1], p->in_item_id, p->rows, ws.cu, ws.X[0], p->B, p->L,
                                p->D, p->n_items, p->state, p->seed, p->p_drop, training, s);
}

// ------------------------------------------------------------------------------------------------
// Packed backward, token-parallel: g = dX[t,:] * mask;  dE[idx,:] += g (skipped for idx == 0: padding_idx), dP[pos,:] += g.
// Each block handles 64 consecutive tokens; dP is first accumulated in LDS (ds atomics) and flushed once per block.
template <int D>
__global__ __launch_bounds__(256) void k_embed_bwd(const float* __restrict__ dX, const int64_t* __restrict__ idx,
                                                   const int64_t* __restrict__ rows, const int* __restrict__ cu,
                                                   float* __restrict__ dE, float* __restrict__ dP, int B, int L,
                                                   int n_items, const int* __restrict__ state, uint64_t seed, float p,
                                                   int training) {
    constexpr int LPT = D / 4, TPB = 256 / LPT;
    __shared__ float accP[64 * D];                      // [L <= 64][D]
    const int T = state[DR4SR_STATE_T], t0 = blockIdx.x * 64;
    if (t0 >= T) return;
    if (dP) for (int i = threadIdx.x; i < L * D; i += 256) accP[i] = 0.f;
    __syncthreads();
    const int c = (threadIdx.x % LPT) * 4;
    const bool dodrop = training && p > 0.f;
    const RngKey rk = make_rng(seed, (uint32_t)state[DR4SR_STATE_RNGSTEP], p);
    for (int t = t0 + threadIdx.x / LPT; t < min(T, t0 + 64); t += TPB) {
        const int b = find_seq(cu, B, t), pos = t - cu[b];
        const int64_t row = rows ? rows[b] : b;
        float4 g = ld4(dX + (size_t)t * D + c);
        if (dodrop) {
            const float4 m = drop4(rk, DR4SR_SITE_EMB, ((uint64_t)b * L + pos) * D + c);
            g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
        }
        if (dP) {
            float* a = accP + pos * D + c;
            atomicAdd(a, g.x); atomicAdd(a + 1, g.y); atomicAdd(a + 2, g.z); atomicAdd(a + 3, g.w);
        }
        const int64_t id = idx[row * L + pos];
        if (id > 0 && id < n_items) {
            float* d = dE + id * D + c;
            unsafeAtomicAdd(d, g.x); unsafeAtomicAdd(d + 1, g.y); unsafeAtomicAdd(d + 2, g.z); unsafeAtomicAdd(d + 3, g.w);
        }
    }
    if (dP) {
        __syncthreads();
        for (int i = threadIdx.x; i < L * D; i += 256) {
            const float v = accP[i];
            if (v != 0.f) unsafeAtomicAdd(dP + i, v);
        }
    }
}

int launch_embed_bwd_raw(const float* dX, const int64_t* idx, const int64_t* rows, const int* cu, float* dE, float* dP, int B, int L,
                         int D, int n_items, const int* state, uint64_t seed, float p, int training, hipStream_t s) {
    dim3 grid(((int64_t)B * L + 63) / 64), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_embed_bwd<64>, grid, blk, 0, s, dX, idx, rows, cu, dE, dP, B, L, n_items, state, seed, p, training);
    else hipLaunchKernelGGL(k_embed_bwd<128>, grid, blk, 0, s, dX, idx, rows, cu, dE, dP, B, L, n_items, state, seed, p, training);
    return DR4SR_LAUNCH_CHECK();
}
int launch_embed_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int training, hipStream_t s) {
    return launch_embed_bwd_raw(ws.dX[0], p->in_item_id, p->rows, ws.cu, p->grads + ws.off[0], p->grads + ws.off[1], p->B, p->L, p->D,
                                p->n_items, p->state, p->seed, p->p_drop, training, s);
}

// ------------------------------------------------------------------------------------------------
// pack / unpack between the dense API layout and the packed workspace layout
// unpack: out[b,l,:] = l < n_b ? X[cu[b]+l,:] : 0   (mode 0: ORIGIN/NONE)  |  out[b,:] = X[cu[b]+n_b-1,:]  (mode 1: LAST)
//       | out[b,:] = sum_{l<n_b} X[cu[b]+l,:] / n_b  (mode 2: MEAN — module/functional.py:50-55 with keepdim=False)
template <int D>
__global__ __launch_bounds__(256) void k_unpack(const float* __restrict__ X, const int* __restrict__ cu,
                                                float* __restrict__ out, int B, int L, int mode) {
    constexpr int LPT = D / 4, RPB = 256 / LPT;
    __shared__ float red[256 * 4];
    const int sub = threadIdx.x / LPT, c = (threadIdx.x % LPT) * 4;
    const int b = blockIdx.x;
    const int t0 = cu[b], n = cu[b + 1] - t0;
    if (mode == 1) {
        if (sub == 0) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n > 0) v = ld4(X + (size_t)(t0 + n - 1) * D + c);
            st4(out + (size_t)b * D + c, v);
        }
        return;
    }
    if (mode == 2) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = sub; l < n; l += RPB) {
            const float4 v = ld4(X + (size_t)(t0 + l) * D + c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        st4(red + threadIdx.x * 4, a);
        __syncthreads();
        if (sub == 0) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < RPB; ++k) {
                const float4 v = ld4(red + (k * LPT + threadIdx.x) * 4);
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            const float inv = 1.0f / (float)n;           // n = 0 -> inf/nan like the reference's division by seq_len
            st4(out + (size_t)b * D + c, make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv));
        }
        return;
    }
    for (int l = sub; l < L; l += RPB) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < n) v = ld4(X + (size_t)(t0 + l) * D + c);
        st4(out + ((size_t)b * L + l) * D + c, v);
    }
}
// pack (backward of unpack): dX[cu[b]+l,:] = d_out[b,l,:]  |  LAST: only row n_b-1 gets d_out[b,:]  |  MEAN: every row d_out[b,:]/n_b
template <int D>
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ dout, const int* __restrict__ cu,
                                              float* __restrict__ dX, int B, int L, int mode, float* __restrict__ zkv = nullptr) {
    constexpr int LPT = D / 4, RPB = 256 / LPT;
    const int sub = threadIdx.x / LPT, c = (threadIdx.x % LPT) * 4;
    const int b = blockIdx.x;
    const int t0 = cu[b], n = cu[b + 1] - t0;
    // zkv: the last layer's dqkv [T][3D] — its dK | dV rows are ACCUMULATED by the attention backward inside the tile kernels
    // (attn_tile.h); the first launch of a backward pass zeroes them, so a second backward on one forward starts clean
    if (zkv) for (int i = threadIdx.x; i < n * (2 * D / 4); i += 256)
        st4(zkv + (size_t)(t0 + i / (2 * D / 4)) * 3 * D + D + (i % (2 * D / 4)) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int l = sub; l < n; l += RPB) {
        float4 v;
        if (mode == 1) v = (l == n - 1) ? ld4(dout + (size_t)b * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        else if (mode == 2) {
            const float4 g = ld4(dout + (size_t)b * D + c);
            const float inv = 1.0f / (float)n;
            v = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
        } else v = ld4(dout + ((size_t)b * L + l) * D + c);
        st4(dX + (size_t)(t0 + l) * D + c, v);
    }
}

int launch_unpack_raw(const float* X, const int* cu, float* out, int B, int L, int D, int mode, hipStream_t s) {
    dim3 grid(B), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_unpack<64>, grid, blk, 0, s, X, cu, out, B, L, mode);
    else hipLaunchKernelGGL(k_unpack<128>, grid, blk, 0, s, X, cu, out, B, L, mode);
    return DR4SR_LAUNCH_CHECK();
}
static int launch_pack_z(const float* dout, const int* cu, float* dX, int B, int L, int D, int mode, float* zkv, hipStream_t s) {
    dim3 grid(B), blk(256);
    if (D == 64) hipLaunchKernelGGL(k_pack<64>, grid, blk, 0, s, dout, cu, dX, B, L, mode, zkv);
    else hipLaunchKernelGGL(k_pack<128>, grid, blk, 0, s, dout, cu, dX, B, L, mode, zkv);
    return DR4SR_LAUNCH_CHECK();
}
int launch_pack_raw(const float* dout, const int* cu, float* dX, int B, int L, int D, int mode, hipStream_t s) {
    return launch_pack_z(dout, cu, dX, B, L, D, mode, nullptr, s);
}
int launch_unpack(const dr4sr_sasrec_plan* p, const Workspace& ws, const float* X, float* out, int mode, hipStream_t s) {
    return launch_unpack_raw(X, ws.cu, out, p->B, p->L, p->D, mode, s);
}
int launch_pack(const dr4sr_sasrec_plan* p, const Workspace& ws, const float* dout, float* dX, int mode, hipStream_t s) {
    return launch_pack_z(dout, ws.cu, dX, p->B, p->L, p->D, mode, (attn_in_tile(p, ws) || ws.attn_tile_sa) ? ws.layer[p->n_layer - 1].dqkv : nullptr, s);
}

// ------------------------------------------------------------------------------------------------
// test hook: keep-mask materialisation
__global__ void k_dropout_mask(float* __restrict__ out, int64_t n4, float p, uint64_t seed, uint32_t step, uint32_t site) {
    RngKey rk = make_rng(seed, step, p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 m = drop4(rk, site, (uint64_t)i * 4);
        st4(out + i * 4, make_float4(m.x != 0.f, m.y != 0.f, m.z != 0.f, m.w != 0.f));
    }
}
extern "C" int dr4sr_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream) {
    if (!out || n < 0 || (n & 3) || p < 0.f || p >= 1.f) return DR4SR_E_ARG;
    if (n == 0) return 0;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n / 4, p, seed, step, site);
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// a1 device-side batch selection: rows_out[i] = perm[(counter*stride + offset + i) mod n_perm]; counter++.
// Replaces DataLoader(shuffle=True) + 256 x 7 __getitem__ + default_collate (data/dataset.py:105-108,
// :149-164): the batch is never materialised, kernels index the resident dataset tensors via rows[].
__global__ void k_select_rows(const int64_t* __restrict__ perm, int64_t n_perm, int64_t* __restrict__ rows_out, int B,
                              int64_t stride, int64_t offset, int* __restrict__ counter) {
    const int64_t c = *counter;
    for (int i = threadIdx.x; i < B; i += blockDim.x) rows_out[i] = perm[(c * stride + offset + i) % n_perm];
    __syncthreads();
    if (threadIdx.x == 0) *counter = (int)(c + 1);
}
extern "C" int dr4sr_select_rows(const int64_t* perm, int64_t n_perm, int64_t* rows_out, int32_t B, int64_t stride,
                                 int64_t offset, int32_t* counter, void* stream) {
    if (!perm || !rows_out || !counter || n_perm <= 0 || B <= 0) return DR4SR_E_ARG;
    hipLaunchKernelGGL(k_select_rows, dim3(1), dim3(256), 0, (hipStream_t)stream, perm, n_perm, rows_out, B, stride, offset, counter);
    return DR4SR_LAUNCH_CHECK();
}
