// topk.hip — K6 full-item scorer + history mask + top-k for evaluation (single-domain case).
//
// Reference: BaseModel.topk (model/basemodel.py:354-365): real_score = query @ E[:N].T; non-domain
// items (incl. PAD column 0) and the user's history are set to -inf; torch.topk(k).
//
// dr4sr_full_score_topk (no workspace): one workgroup per query row; the N scores of the row live in LDS (N*4 B <= 150 KiB),
// the top-k is k rounds of block-wide arg-max (ties -> lower id).  Simple, but every row re-reads the whole table from L2 and the
// k serial arg-max rounds cost ~3 us each: 1.7 ms per 2048-row batch, more than a training epoch's worth per validation pass.
//
// dr4sr_full_score_topk_ws (with a [B, Npad] fp32 workspace): two launches.
//   k_score_gemm : S = Q E^T on MFMA 32x32x2 (64 rows x 64 items per workgroup: the table is read B/64 times, not B times),
//                  PAD column -> -inf, streamed out with non-temporal stores;
//   k_topk_select: one workgroup per row: row -> LDS as order-preserving uint keys (history -> -inf), 4-pass radix select of the
//                  k-th largest key (256-bin LDS histograms), compaction of the keys above it (+ the lowest-index ties), bitonic
//                  sort of the <= 128 candidates by (score desc, id asc).  Same results as the arg-max rounds, ~15x faster.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

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

// ------------------------------------------------------------------------------------------------ two-phase path
__device__ __forceinline__ unsigned f2key(float v) { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// Fused form (round 3) — the [B, N] score matrix never exists.  A per-row lower bound of the k-th largest VALID score comes from a
// strided SUBSET of the catalog (the k-th largest of any subset is <= the k-th largest of the whole set), the full GEMM then emits
// only the scores >= that bound as (key, ~id) candidates through a per-row counter, and a per-row kernel masks the history and sorts
// them.  With a subset of 1 / stride of the items the expected candidate count is k * stride.
struct FuseArgs {
    int stride;                    // > 0: SUBSET pass — column j of the output is item 1 + j * stride (n_sub columns)
    int n_sub;
    const float* bound;            // != NULL: EMIT pass — candidates >= bound[b] go to cand[b][cnt[b]++] instead of S
    int* cnt; unsigned long long* cand; int cap;
    const int* run_if;             // != NULL: the launch does nothing unless *run_if != 0 (the fall-back pass of an overflowed batch)
};

template <int D>
__global__ __launch_bounds__(256) void k_score_gemm(const float* __restrict__ Q, const float* __restrict__ E, float* __restrict__ S,
                                                    int B, int n_items, int lds_s, const uint8_t* __restrict__ blocked, const FuseArgs Fz) {
    if (Fz.run_if && *Fz.run_if == 0) return;
    constexpr int LD = D + 1;                              // odd stride: the 32 lanes of an MFMA operand read 32 different rows
    float* Qs = smem;                                      // [64][LD]
    float* Es = smem + 64 * LD;                            // [64][LD]
    const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int ncol = Fz.stride > 0 ? Fz.n_sub : n_items;   // columns of this launch
    for (int e = threadIdx.x; e < 64 * (D / 4); e += 256) {
        const int r = e / (D / 4), c = (e % (D / 4)) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), ev = qv;
        if (b0 + r < B) qv = ld4(Q + (size_t)(b0 + r) * D + c);
        const int item = Fz.stride > 0 ? 1 + (n0 + r) * Fz.stride : n0 + r;
        if (n0 + r < ncol && item < n_items) ev = ld4(E + (size_t)item * D + c);
        float* qd = Qs + r * LD + c; qd[0] = qv.x; qd[1] = qv.y; qd[2] = qv.z; qd[3] = qv.w;
        float* ed = Es + r * LD + c; ed[0] = ev.x; ed[1] = ev.y; ed[2] = ev.z; ed[3] = ev.w;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, g = lane >> 5;
    const int rt = w >> 1, ct = w & 1;                     // wave -> 32x32 quadrant (rows rt, items ct)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* ap = Qs + (rt * 32 + r) * LD + g;
    const float* bp = Es + (ct * 32 + r) * LD + g;
#pragma unroll 8
    for (int sidx = 0; sidx < D / 2; ++sidx) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * sidx], bp[2 * sidx], acc, 0, 0, 0);
    const int col = n0 + ct * 32 + r;
    const int n = Fz.stride > 0 ? 1 + col * Fz.stride : col;
    // basemodel.py:358-360: every item outside the evaluated domain (PAD column 0 is never in a domain's item list) -> -inf
    const bool off = n == 0 || n >= n_items || col >= ncol || (blocked && blocked[n]);
    if (Fz.bound) {                                        // EMIT: only what can be in the top-k leaves the kernel
        if (off) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int b = b0 + rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
            if (b < B && acc[e] >= Fz.bound[b]) {
                const int p = atomicAdd(Fz.cnt + b, 1);
                if (p < Fz.cap) Fz.cand[(size_t)b * Fz.cap + p] = ((unsigned long long)f2key(acc[e]) << 32) | (unsigned)(~(unsigned)n);
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int b = b0 + rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        if (b < B && col < lds_s) __builtin_nontemporal_store(off ? -INFINITY : acc[e], S + (size_t)b * lds_s + col);
    }
}

// EMIT pass of the fused form: a workgroup owns 64 rows x a CHUNK of item tiles; what passes a row's bound is staged in LDS through
// per-row LDS counters and leaves with ONE global reservation per (row, chunk) — per-candidate global atomics ran at 1.9 G/s
// (0.84 ms for the 1.6 M candidates of a 2048 x 11 925 batch).
constexpr int EMIT_TILES = 8, EMIT_CAPL = 80;              // 512 items per chunk: ~34 candidates expected per row at k = 100, stride 8
template <int D>
__global__ __launch_bounds__(256) void k_score_emit(const float* __restrict__ Q, const float* __restrict__ E, int B, int n_items,
                                                    const uint8_t* __restrict__ blocked, const float* __restrict__ bound,
                                                    int* __restrict__ gcnt, unsigned long long* __restrict__ cand, int cap,
                                                    int* __restrict__ overflow) {
    constexpr int LD = D + 1;
    float* Qs = smem;                                      // [64][LD]
    float* Es = smem + 64 * LD;                            // [64][LD]
    float* bnd = Es + 64 * LD;                             // [64]
    int* cnt = reinterpret_cast<int*>(bnd + 64);           // [64] candidates of this chunk per row, then the row's global base
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(cnt + 64);       // [64][EMIT_CAPL]
    const int b0 = blockIdx.y * 64, tile0 = blockIdx.x * EMIT_TILES, ntile = (n_items + 63) / 64;
    for (int e = threadIdx.x; e < 64 * (D / 4); e += 256) {
        const int r = e / (D / 4), c = (e % (D / 4)) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + r < B) qv = ld4(Q + (size_t)(b0 + r) * D + c);
        float* qd = Qs + r * LD + c; qd[0] = qv.x; qd[1] = qv.y; qd[2] = qv.z; qd[3] = qv.w;
    }
    if (threadIdx.x < 64) { bnd[threadIdx.x] = b0 + threadIdx.x < B ? bound[b0 + threadIdx.x] : INFINITY; cnt[threadIdx.x] = 0; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, g = lane >> 5;
    const int rt = w >> 1, ct = w & 1;
    for (int tile = tile0; tile < tile0 + EMIT_TILES && tile < ntile; ++tile) {
        const int n0 = tile * 64;
        __syncthreads();                                   // the previous tile's MFMAs have read Es
        for (int e = threadIdx.x; e < 64 * (D / 4); e += 256) {
            const int rr = e / (D / 4), c = (e % (D / 4)) * 4;
            float4 ev = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + rr < n_items) ev = ld4(E + (size_t)(n0 + rr) * D + c);
            float* ed = Es + rr * LD + c; ed[0] = ev.x; ed[1] = ev.y; ed[2] = ev.z; ed[3] = ev.w;
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float* ap = Qs + (rt * 32 + r) * LD + g;
        const float* bp = Es + (ct * 32 + r) * LD + g;
#pragma unroll 8
        for (int sidx = 0; sidx < D / 2; ++sidx) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * sidx], bp[2 * sidx], acc, 0, 0, 0);
        const int n = n0 + ct * 32 + r;
        const bool off = n == 0 || n >= n_items || (blocked && blocked[n]);
        if (!off) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
                if (acc[e] >= bnd[rl]) {
                    const int p = atomicAdd(&cnt[rl], 1);
                    if (p < EMIT_CAPL) stage[rl * EMIT_CAPL + p] = ((unsigned long long)f2key(acc[e]) << 32) | (unsigned)(~(unsigned)n);
                }
            }
        }
    }
    __syncthreads();
    int* base = reinterpret_cast<int*>(Es);                // [64] (the item tile is dead)
    if (threadIdx.x < 64) {
        const int c = cnt[threadIdx.x], b = b0 + threadIdx.x;
        int bs = 0;
        if (c > EMIT_CAPL) atomicExch(overflow, 1);
        if (b < B && c > 0) {
            bs = atomicAdd(gcnt + b, c < EMIT_CAPL ? c : EMIT_CAPL);
            if (bs + c > cap) atomicExch(overflow, 1);
        }
        base[threadIdx.x] = bs;
    }
    __syncthreads();
    for (int rl = w; rl < 64; rl += 4) {                   // a wave per row: coalesced write-out of the row's staged candidates
        const int c = min(cnt[rl], EMIT_CAPL), bs = base[rl], b = b0 + rl;
        for (int i = lane; i < c; i += 64)
            if (bs + i < cap) cand[(size_t)b * cap + bs + i] = stage[rl * EMIT_CAPL + i];
    }
}

// the selection reads the row either from LDS (catalogs up to ~37 k items) or straight from the score workspace (any size)
struct RowKeys {
    const unsigned* lds; const float* glb;
    __device__ __forceinline__ unsigned operator[](int n) const { return lds ? lds[n] : f2key(glb[n]); }
};
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// exact selection by radix: the key T of rank kk (4 passes of 256-bin LDS histograms), then every key above T and the lowest-index
// keys equal to T.  Slow path (the histograms contend on a few bins) — used when ties make the fast path's candidate set overflow.
__device__ __forceinline__ int select_radix(const RowKeys key, int* hst, int* ctl, unsigned long long* cand, int n_items, int k) {
    const int tid = threadIdx.x, lane = tid & 63;
    __syncthreads();
    if (tid == 0) { ctl[1] = k < n_items ? k : n_items; ctl[2] = 0; ctl[3] = 0; }
    // ---- radix select: the key T of rank ctl[1] (1-based, from the top)
    unsigned prefix = 0, pmask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hst[tid] = 0;
        __syncthreads();
        for (int n = tid; n < n_items; n += 256) {
            const unsigned kv = key[n];
            if ((kv & pmask) == prefix) atomicAdd(&hst[(kv >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid < 64) {                                    // wave 0: suffix sums over the 256 bins, 4 bins per lane (descending digits)
            const int rank = ctl[1];
            const int d0 = 255 - 4 * lane;                 // this lane's bins: d0, d0-1, d0-2, d0-3
            const int c0 = hst[d0], c1 = hst[d0 - 1], c2 = hst[d0 - 2], c3 = hst[d0 - 3];
            int tot = c0 + c1 + c2 + c3, incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
            const int before = incl - tot;                 // keys in strictly higher bins than d0
            if (before < rank && rank <= incl) {           // the rank-th key falls into one of this lane's bins
                int acc = before, dsel = d0, rem = rank - before;
                if (acc + c0 >= rank) { dsel = d0; rem = rank - acc; }
                else if (acc + c0 + c1 >= rank) { dsel = d0 - 1; rem = rank - acc - c0; }
                else if (acc + c0 + c1 + c2 >= rank) { dsel = d0 - 2; rem = rank - acc - c0 - c1; }
                else { dsel = d0 - 3; rem = rank - acc - c0 - c1 - c2; }
                ctl[0] = dsel; ctl[1] = rem;
            }
        }
        __syncthreads();
        prefix |= (unsigned)ctl[0] << shift;
        pmask |= 255u << shift;
        __syncthreads();
    }
    const unsigned T = prefix;                             // exactly the k-th largest key; ctl[1] = how many keys == T belong to the top-k
    const int need_eq = ctl[1];
    // ---- candidates: every key above T (any order), then the need_eq lowest-index keys equal to T
    for (int n = tid; n < n_items; n += 256) {
        const unsigned kv = key[n];
        if (kv > T) { const int p = atomicAdd(&ctl[2], 1); cand[p] = ((unsigned long long)kv << 32) | (unsigned)(~(unsigned)n); }
    }
    __syncthreads();
    const int ngt = ctl[2];
    {   // ordered compaction of the ties (usually exactly need_eq of them: one chunk pass finds them)
        int taken = 0;
        for (int n0 = 0; n0 < n_items && taken < need_eq; n0 += 256) {
            const int n = n0 + tid;
            const bool eq = n < n_items && key[n] == T;
            const unsigned long long bal = __ballot(eq);
            const int inwave = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) hst[tid >> 6] = __popcll(bal);
            __syncthreads();
            const int w = tid >> 6;
            int off = 0;
            for (int x = 0; x < w; ++x) off += hst[x];
            const int tot = hst[0] + hst[1] + hst[2] + hst[3];
            const int rnk = taken + off + inwave;
            if (eq && rnk < need_eq) cand[ngt + rnk] = ((unsigned long long)T << 32) | (unsigned)(~(unsigned)n);
            taken += tot;
            __syncthreads();
        }
    }
    return ngt + need_eq;
}

template <int NS>
__device__ __forceinline__ void bitonic_desc(unsigned long long* a) {      // NS power of two <= 512, 256 threads
    const int tid = threadIdx.x;
    for (int sz = 2; sz <= NS; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            if (tid < NS / 2) {
                const int i = 2 * tid - (tid & (st - 1)), j = i + st;
                const bool desc = ((i & sz) == 0);
                const unsigned long long x = a[i], y = a[j];
                if ((x < y) == desc) { a[i] = y; a[j] = x; }
            }
            __syncthreads();
        }
}

// One workgroup per row.  Fast path: the k-th largest of the 256 per-thread maxima is a lower bound of the k-th largest score, so
// only the keys >= that bound (~1.3 k of them for unstructured scores) are candidates; they are sorted as (key, ~id) composites,
// which also orders ties by id.  No histogram, no contended atomics: ~80 barriers per row.
// sub_stride > 0 (fused form, SUBSET pass): the row holds the scores of items 1 + j * sub_stride; the kernel then only publishes
// bound_out[b] = the k-th largest valid score of the subset (-inf if it holds fewer than k valid items) and zeroes the row's counter.
template <bool LDSROW>
__global__ __launch_bounds__(256) void k_topk_select(float* __restrict__ S, const int64_t* __restrict__ hist,
                                                     float* __restrict__ out_score, int64_t* __restrict__ out_item, int n_items,
                                                     int lds_s, int Lh, int k, int sub_stride, float* __restrict__ bound_out,
                                                     int* __restrict__ cnt_out, const int* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;
    constexpr int CAP = 512;
    unsigned* keyl = reinterpret_cast<unsigned*>(smem);    // [n_items] (LDSROW only)
    int* hst = reinterpret_cast<int*>(keyl + (LDSROW ? ((n_items + 3) & ~3) : 0));       // [256] digit histogram (radix path)
    int* ctl = hst + 256;                                  // [8]
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(hst + 256 + 8);     // [CAP] (key << 32) | ~id
    const int b = blockIdx.x, tid = threadIdx.x;
    float* row = S + (size_t)b * lds_s;
    const unsigned kneg = f2key(-INFINITY);
    if (LDSROW) {
        for (int n = tid; n < n_items; n += 256) keyl[n] = f2key(row[n]);
        __syncthreads();
    }
    for (int j = tid; j < Lh; j += 256) {                  // history -> -inf (in the LDS copy, or in the workspace row itself)
        int64_t id = hist[(size_t)b * Lh + j];
        if (sub_stride > 0) id = (id >= 1 && (id - 1) % sub_stride == 0) ? (id - 1) / sub_stride : -1;      // subset column of this item, if any
        if (id >= 0 && id < n_items) { if (LDSROW) keyl[id] = kneg; else row[id] = -INFINITY; }
    }
    if (tid == 0) ctl[4] = 0;
    __syncthreads();
    const RowKeys key{LDSROW ? keyl : nullptr, row};
    const int kk = k < n_items ? k : n_items;
    unsigned lm = 0;
    for (int n = tid; n < n_items; n += 256) lm = max(lm, key[n]);
    cand[tid] = (unsigned long long)lm << 32;
    __syncthreads();
    bitonic_desc<256>(cand);
    const unsigned bound = (unsigned)(cand[kk - 1] >> 32);  // at least kk keys are >= bound (kk distinct thread maxima)
    __syncthreads();
    for (int n = tid; n < n_items; n += 256) {
        const unsigned kv = key[n];
        if (kv >= bound && bound != 0u) {
            const int p = atomicAdd(&ctl[4], 1);
            if (p < CAP) cand[p] = ((unsigned long long)kv << 32) | (unsigned)(~(unsigned)n);
        }
    }
    __syncthreads();
    int nc = ctl[4];
    if (nc > CAP || bound == 0u) {                         // ties piled up at the bound (e.g. k > number of unmasked items): exact radix path
        nc = select_radix(key, hst, ctl, cand, n_items, k);
        for (int i = nc + tid; i < 128; i += 256) cand[i] = 0ull;
        __syncthreads();
        bitonic_desc<128>(cand);
    } else {
        int ns = 128;
        while (ns < nc) ns <<= 1;
        for (int i = nc + tid; i < ns; i += 256) cand[i] = 0ull;
        __syncthreads();
        if (ns == 128) bitonic_desc<128>(cand); else if (ns == 256) bitonic_desc<256>(cand); else bitonic_desc<512>(cand);
        nc = kk;
    }
    if (sub_stride > 0) {                                  // SUBSET pass of the fused form: the k-th largest valid score is all that is wanted
        if (tid == 0) {
            bound_out[b] = (nc >= k && k <= n_items) ? key2f((unsigned)(cand[k - 1] >> 32)) : -INFINITY;
            cnt_out[b] = 0;
        }
        return;
    }
    for (int r = tid; r < k; r += 256) {
        float fv = -INFINITY;
        int64_t fi = 0;
        if (r < nc) { const unsigned long long c = cand[r]; fv = key2f((unsigned)(c >> 32)); fi = (int64_t)(~(unsigned)c); }
        out_score[(size_t)b * k + r] = fv;
        out_item[(size_t)b * k + r] = fi;
    }
}

// ---- wave-per-row selection (fused form): no workgroup barrier anywhere.  A barrier costs ~0.5-1 us with eight workgroups per CU and
// the workgroup forms above cross 50-80 of them per row; one wave serves its LDS operations in order, so a wave-private histogram
// needs none.
// The k-th largest of n 32-bit keys read through key(i): 4 passes over 256-bin histograms.  Returns T; *n_gt = how many keys are > T.
template <typename KeyFn>
__device__ __forceinline__ unsigned wave_radix_kth(KeyFn key, const int n, int rank, int* hst, int* n_gt) {
    const int lane = threadIdx.x & 63;
    unsigned prefix = 0, pmask = 0;
    int above = 0;                                         // keys strictly above the current prefix range
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int j = 0; j < 4; ++j) hst[lane * 4 + j] = 0;
        for (int i = lane; i < n; i += 64) {
            const unsigned kv = key(i);
            if ((kv & pmask) == prefix) atomicAdd(&hst[(kv >> shift) & 255u], 1);
        }
        const int d0 = 255 - 4 * lane;                     // this lane's bins, descending digits: d0, d0-1, d0-2, d0-3
        const int c0 = hst[d0], c1 = hst[d0 - 1], c2 = hst[d0 - 2], c3 = hst[d0 - 3];
        const int tot = c0 + c1 + c2 + c3;
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        const int before = incl - tot;
        int dsel = -1, rem = 0, ab = 0;
        if (before < rank && rank <= incl) {
            if (before + c0 >= rank) { dsel = d0; rem = rank - before; ab = before; }
            else if (before + c0 + c1 >= rank) { dsel = d0 - 1; rem = rank - before - c0; ab = before + c0; }
            else if (before + c0 + c1 + c2 >= rank) { dsel = d0 - 2; rem = rank - before - c0 - c1; ab = before + c0 + c1; }
            else { dsel = d0 - 3; rem = rank - before - c0 - c1 - c2; ab = before + c0 + c1 + c2; }
        }
        const unsigned long long who = __ballot(dsel >= 0);               // exactly one lane (rank <= number of keys in range)
        const int src = who ? __ffsll((long long)who) - 1 : 0;
        dsel = __shfl(dsel, src, 64); rem = __shfl(rem, src, 64); ab = __shfl(ab, src, 64);
        if (!who) { dsel = 0; rem = 1; ab = 0; }
        prefix |= (unsigned)dsel << shift;
        pmask |= 255u << shift;
        above += ab;
        rank = rem;
    }
    *n_gt = above;
    return prefix;
}

// SUBSET pass, wave per row: bound[b] = the k-th largest valid score among the subset's n_sub columns (history masked in the row)
__global__ __launch_bounds__(64) void k_subset_bound_w(float* __restrict__ S, const int64_t* __restrict__ hist, int n_sub, int sub_s, int Lh,
                                                       int k, int stride, float* __restrict__ bound_out, int* __restrict__ cnt_out,
                                                       int* __restrict__ flag_out) {
    __shared__ int hst[256];
    const int b = blockIdx.x, lane = threadIdx.x;
    float* row = S + (size_t)b * sub_s;
    for (int j = lane; j < Lh; j += 64) {
        const int64_t id = hist[(size_t)b * Lh + j];
        if (id >= 1 && (id - 1) % stride == 0 && (id - 1) / stride < n_sub) row[(id - 1) / stride] = -INFINITY;
    }
    __threadfence_block();
    int ngt;
    const unsigned T = n_sub >= k ? wave_radix_kth([&](int i) { return f2key(row[i]); }, n_sub, k, hst, &ngt) : f2key(-INFINITY);
    if (lane == 0) {
        bound_out[b] = key2f(T);
        cnt_out[b] = 0;
        if (b == 0) *flag_out = 0;                         // the batch's overflow flag (no memset node: DESIGN §4a)
    }
}

// last pass, wave per row: history items out, radix-select the k-th key, gather the survivors (everything above it + its ties), sort
// those <= 128 composites (key, ~id) in registers — score descending, ties by ascending id — and write the top k.
template <int CAPC>
__global__ __launch_bounds__(64) void k_cand_select_w(unsigned long long* __restrict__ cand_g, const int* __restrict__ cnt,
                                                      const int64_t* __restrict__ hist, float* __restrict__ out_score,
                                                      int64_t* __restrict__ out_item, int Lh, int k, int* __restrict__ overflow) {
    __shared__ int hst[256];
    __shared__ int hl[128];
    __shared__ unsigned long long srt[128];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nc = cnt[b];
    if (nc > CAPC || *overflow) { if (lane == 0) atomicExch(overflow, 1); return; }       // (an overflowed batch is redone by the two-kernel form)
    unsigned long long* cg = cand_g + (size_t)b * CAPC;
    const int Lc = Lh < 128 ? Lh : 128;
    for (int j = lane; j < Lc; j += 64) hl[j] = (int)hist[(size_t)b * Lh + j];
    for (int i = lane; i < nc; i += 64) {                  // history -> dropped (key 0 sorts below every real score)
        const int id = (int)(~(unsigned)cg[i]);
        bool hit = false;
        for (int j = 0; j < Lc; ++j) hit |= hl[j] == id;
        for (int j = Lc; j < Lh; ++j) hit |= (int)hist[(size_t)b * Lh + j] == id;
        if (hit) cg[i] = 0ull;
    }
    __threadfence_block();
    int ngt = 0;
    const int kk = k < nc ? k : nc;
    unsigned T = 0;
    if (kk > 0) T = wave_radix_kth([&](int i) { return (unsigned)(cg[i] >> 32); }, nc, kk, hst, &ngt);
    // survivors: key > T, and key == T (its ties; T == 0 means fewer than k real candidates: take the real ones only)
    srt[lane] = 0ull; srt[lane + 64] = 0ull;
    int ns = 0;
    for (int i0 = 0; i0 < nc; i0 += 64) {
        const int i = i0 + lane;
        const unsigned long long c = i < nc ? cg[i] : 0ull;
        const unsigned kv = (unsigned)(c >> 32);
        const bool take = c != 0ull && kv >= T && kv != 0u;
        const unsigned long long bal = __ballot(take);
        const int pos = ns + __popcll(bal & ((1ull << lane) - 1ull));
        if (take && pos < 128) srt[pos] = c;
        ns += __popcll(bal);
    }
    if (ns > 128) { if (lane == 0) atomicExch(overflow, 1); return; }                     // a large tie at the k-th score: exact two-kernel form
    // bitonic sort of 128 composites, descending: element e = lane + 64 j (j = 0, 1)
    unsigned long long v0 = srt[lane], v1 = srt[lane + 64];
    for (int sz = 2; sz <= 128; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            if (st == 64) {                                // partner = the lane's other element
                const bool desc = true;                    // (sz == 128: one descending run)
                if ((v0 < v1) == desc) { const unsigned long long t = v0; v0 = v1; v1 = t; }
            } else {
                const unsigned long long p0 = __shfl_xor(v0, st, 64), p1 = __shfl_xor(v1, st, 64);
                const bool low = (lane & st) == 0;         // this lane holds the lower index of the pair
                const bool d0 = ((lane & sz) == 0), d1 = (((lane + 64) & sz) == 0);
                // lower index keeps the larger value in a descending run
                v0 = (low == d0) ? (v0 > p0 ? v0 : p0) : (v0 < p0 ? v0 : p0);
                v1 = (low == d1) ? (v1 > p1 ? v1 : p1) : (v1 < p1 ? v1 : p1);
            }
        }
    for (int r = lane; r < k; r += 64) {
        const unsigned long long c = r < 64 ? v0 : v1;
        float fv = -INFINITY;
        int64_t fi = 0;
        if (r < 128 && c) { fv = key2f((unsigned)(c >> 32)); fi = (int64_t)(~(unsigned)c); }
        out_score[(size_t)b * k + r] = fv;
        out_item[(size_t)b * k + r] = fi;
    }
}

// fused form, last pass: one workgroup per row sorts the row's emitted candidates (history -> dropped) as (key, ~id) composites —
// score descending, ties by ascending id (torch.topk's order on equal scores is unspecified; the per-row kernels use the same rule) —
// and writes the top k.  A row whose candidates overflowed the buffer raises *overflow: the whole batch is then redone by the
// two-kernel form (k_score_gemm / k_topk_select with run_if = overflow).
template <int CAPC>
__global__ __launch_bounds__(256) void k_cand_select(const unsigned long long* __restrict__ cand_g, const int* __restrict__ cnt,
                                                     const int64_t* __restrict__ hist, float* __restrict__ out_score,
                                                     int64_t* __restrict__ out_item, int Lh, int k, int* __restrict__ overflow) {
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem);      // [CAPC]
    int* hl = reinterpret_cast<int*>(cand + CAPC);                               // [Lh] the row's history ids
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nc = cnt[b];
    if (nc > CAPC || *overflow) { if (tid == 0) atomicExch(overflow, 1); return; }      // (an overflowed batch is redone by the two-kernel form)
    for (int j = tid; j < Lh; j += 256) hl[j] = (int)hist[(size_t)b * Lh + j];
    __syncthreads();
    int ns = 128;
    while (ns < nc) ns <<= 1;
    for (int i = tid; i < ns; i += 256) {
        unsigned long long c = i < nc ? cand_g[(size_t)b * CAPC + i] : 0ull;
        if (c) {
            const int id = (int)(~(unsigned)c);
            bool hit = false;
            for (int j = 0; j < Lh; ++j) hit |= hl[j] == id;
            if (hit) c = 0ull;                             // a history item: out (sorts last)
        }
        cand[i] = c;
    }
    __syncthreads();
    for (int sz = 2; sz <= ns; sz <<= 1)                   // bitonic, descending (ns <= CAPC, a power of two)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int t = tid; t < ns / 2; t += 256) {
                const int i = 2 * t - (t & (st - 1)), j = i + st;
                const bool desc = ((i & sz) == 0);
                const unsigned long long x = cand[i], y = cand[j];
                if ((x < y) == desc) { cand[i] = y; cand[j] = x; }
            }
            __syncthreads();
        }
    for (int r = tid; r < k; r += 256) {
        const unsigned long long c = r < ns ? cand[r] : 0ull;
        float fv = -INFINITY;
        int64_t fi = 0;
        if (c) { fv = key2f((unsigned)(c >> 32)); fi = (int64_t)(~(unsigned)c); }
        out_score[(size_t)b * k + r] = fv;
        out_item[(size_t)b * k + r] = fi;
    }
}

extern "C" int64_t dr4sr_full_score_topk_workspace_bytes(int64_t B, int32_t n_items) {
    if (B < 0 || n_items < 2) return DR4SR_E_ARG;
    return B * (int64_t)((n_items + 63) / 64 * 64) * 4 + 256;     // [B][lds_s] scores + the fused form's overflow flag behind them
}

static int topk_ws_impl(const float* q, const float* E, const int64_t* hist, const uint8_t* blocked, float* out_score, int64_t* out_item,
                        int64_t B, int32_t D, int32_t n_items, int32_t Lh, int32_t k, float* workspace,
                        int64_t workspace_bytes, void* stream) {
    if (!q || !E || !out_score || !out_item || !workspace || B < 0 || n_items < 2 || k <= 0 || k > 128 || Lh < 0 || (Lh > 0 && !hist))
        return DR4SR_E_ARG;
    if (D != 64 && D != 128) return DR4SR_E_SHAPE;
    const int lds_s = (n_items + 63) / 64 * 64;
    if (workspace_bytes < B * (int64_t)lds_s * 4) return DR4SR_E_WS;
    const size_t lds_fix = sizeof(int) * (256 + 8) + sizeof(unsigned long long) * 512;
    const size_t lds_row = sizeof(unsigned) * ((n_items + 3) & ~3) + lds_fix;
    const int lds_max_kb = DR4SR_XENV("DR4SR_TOPK_LDS_KB") ? atoi(DR4SR_XENV("DR4SR_TOPK_LDS_KB")) : 24;      // measured: above ~4 k items the LDS copy costs more occupancy than the second row read (0.120 vs 0.104 ms at N = 11 925)
    const bool ldsrow = lds_row <= (size_t)lds_max_kb * 1024;               // (above that the LDS copy costs more occupancy than the second row read)
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(lds_s / 64, (unsigned)((B + 63) / 64));
    const size_t lds_g = sizeof(float) * 2 * 64 * (D + 1);
    FuseArgs Fz{};
    // ---- fused form: catalogs of >= 4 096 items, k * stride candidates expected per row
    // Measured (2048 x 11 925, k = 100, tools/topk_probe.py): the fused form moves ~50 MB instead of 332 MB but takes 167 us against the
    // two-kernel form's 104 us (subset pass 38, emit 85, candidate select 40: per-row LDS counters in the GEMM epilogue and 2048 small
    // selection waves cost more than the 97 MB matrix round trip at 3.2 TB/s); at N = 200 000 1.3-1.8 ms against 1.55 ms.  It is
    // therefore OPT-IN (DR4SR_TOPK_FUSED=1; cached until dr4sr_reload_env()) until the emit epilogue is cheaper; the tests run both forms.
    const bool unfused = DR4SR_ENV("DR4SR_TOPK_FUSED") == nullptr;
    constexpr int CAPC = 2048;
    const int stride = 8, n_sub = (n_items - 1 + stride - 1) / stride, sub_s = (n_sub + 63) / 64 * 64;
    const int64_t need = B * ((int64_t)sub_s * 4 + (int64_t)CAPC * 8 + 8) + 256;
    const int64_t flag_off = B * (int64_t)lds_s * 4;       // BEHIND the [B][lds_s] matrix: the two-kernel fall-back of an overflowed batch
    const bool fused = !unfused && n_items >= 4096 && need <= flag_off && flag_off + 4 <= workspace_bytes && k * stride * 2 <= CAPC;     //  writes its scores over everything else
    const int* run_if = nullptr;
    if (fused) {
        // workspace: [B][sub_s] subset scores | [B] bound | [B] counters | [B][CAPC] candidates  (the two-kernel fall-back, which only
        // runs for an overflowed batch, re-uses the same bytes as its [B][lds_s] score matrix) ... | flag at flag_off
        float* sub = workspace;
        float* bound = sub + B * (int64_t)sub_s;
        int* cnt = reinterpret_cast<int*>(bound + B);
        int* flag = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + flag_off);
        unsigned long long* cand = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + ((B * ((int64_t)sub_s * 4 + 8) + 4 + 255) / 256) * 256);
        Fz.stride = stride; Fz.n_sub = n_sub;
        dim3 gsub(sub_s / 64, (unsigned)((B + 63) / 64));
        if (D == 64) hipLaunchKernelGGL(k_score_gemm<64>, gsub, dim3(256), lds_g, s, q, E, sub, (int)B, n_items, sub_s, blocked, Fz);
        else { big_lds(k_score_gemm<128>, lds_g); hipLaunchKernelGGL(k_score_gemm<128>, gsub, dim3(256), lds_g, s, q, E, sub, (int)B, n_items, sub_s, blocked, Fz); }
        hipLaunchKernelGGL(k_subset_bound_w, dim3((unsigned)B), dim3(64), 0, s, sub, hist, n_sub, sub_s, Lh, k, stride, bound, cnt, flag);
        {
            dim3 ge((unsigned)((lds_s / 64 + EMIT_TILES - 1) / EMIT_TILES), (unsigned)((B + 63) / 64));
            const size_t lds_e = sizeof(float) * (2 * 64 * (D + 1) + 64) + sizeof(int) * 64 + sizeof(unsigned long long) * 64 * EMIT_CAPL;
            if (D == 64) { big_lds(k_score_emit<64>, lds_e); hipLaunchKernelGGL(k_score_emit<64>, ge, dim3(256), lds_e, s, q, E, (int)B, n_items, blocked, bound, cnt, cand, CAPC, flag); }
            else { big_lds(k_score_emit<128>, lds_e); hipLaunchKernelGGL(k_score_emit<128>, ge, dim3(256), lds_e, s, q, E, (int)B, n_items, blocked, bound, cnt, cand, CAPC, flag); }
        }
        hipLaunchKernelGGL(k_cand_select_w<CAPC>, dim3((unsigned)B), dim3(64), 0, s, cand, cnt, hist, out_score, out_item, Lh, k, flag);
        run_if = flag;                                     // the two launches below exit at once unless a row overflowed
        Fz = FuseArgs{};
    }
    Fz.run_if = run_if;
    if (D == 64) hipLaunchKernelGGL(k_score_gemm<64>, grid, dim3(256), lds_g, s, q, E, workspace, (int)B, n_items, lds_s, blocked, Fz);
    else { big_lds(k_score_gemm<128>, lds_g); hipLaunchKernelGGL(k_score_gemm<128>, grid, dim3(256), lds_g, s, q, E, workspace, (int)B, n_items, lds_s, blocked, Fz); }
    if (ldsrow) {
        big_lds(k_topk_select<true>, lds_row);
        hipLaunchKernelGGL(k_topk_select<true>, dim3((unsigned)B), dim3(256), lds_row, s, workspace, hist, out_score, out_item, n_items, lds_s, Lh, k, 0, nullptr, nullptr, run_if);
    } else {
        hipLaunchKernelGGL(k_topk_select<false>, dim3((unsigned)B), dim3(256), lds_fix, s, workspace, hist, out_score, out_item, n_items, lds_s, Lh, k, 0, nullptr, nullptr, run_if);
    }
    return DR4SR_LAUNCH_CHECK();
}

extern "C" int dr4sr_full_score_topk_ws(const float* q, const float* E, const int64_t* hist, float* out_score, int64_t* out_item,
                                        int64_t B, int32_t D, int32_t n_items, int32_t Lh, int32_t k, float* workspace,
                                        int64_t workspace_bytes, void* stream) {
    return topk_ws_impl(q, E, hist, nullptr, out_score, out_item, B, D, n_items, Lh, k, workspace, workspace_bytes, stream);
}
extern "C" int dr4sr_full_score_topk_masked_ws(const float* q, const float* E, const int64_t* hist, const uint8_t* item_blocked,
                                               float* out_score, int64_t* out_item, int64_t B, int32_t D, int32_t n_items, int32_t Lh,
                                               int32_t k, float* workspace, int64_t workspace_bytes, void* stream) {
    return topk_ws_impl(q, E, hist, item_blocked, out_score, out_item, B, D, n_items, Lh, k, workspace, workspace_bytes, stream);
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
