// prep_body.h — the per-step "prep" (batch selection, prefix scan of the sequence lengths, search hints, length classes) as a
// device function of one workgroup, shared by k_prep (embed.hip) and by the Adam kernel that prepares the NEXT step (step.hip).
#pragma once
#include "common.h"

// sequences of 1..DR4SR_TINY_MAX tokens form the third attention length class (8 lanes per (sequence, head), no MFMA)
#define DR4SR_TINY_MAX 8

struct PermSel { const int64_t* perm; int64_t n, stride, offset; int* counter; };

struct PrepArgs {
    const int64_t* seqlen; const int64_t* rows; int* cu; int* state; int B, L, bump_rng; PermSel sel; int* tile_seq; int* seq_class;
    int* len_buf;                              // scratch of the two-phase form (prep_phase1 -> prep_phase2: 4 B + 4 PREP_MAX_BLK words), or NULL
};

// Two-phase form for large batches (the optimizer launch prepares the next step; one workgroup doing 8 192 sequences by itself was
// 41 us as a launch of its own, and 25 us as the serial tail of the optimizer launch when only the selection was spread out).
// Phase 1 (prep_phase1, run by EVERY workgroup of that launch before its share of the sweep): workgroup k owns the CONTIGUOUS
// sequences [k C, (k+1) C) — batch selection, the seqlen gather and a workgroup-local scan of (tokens, class counts); per sequence it
// publishes {local token prefix, local class prefixes, length} (two 8-byte words) and per workgroup its totals, with agent-scope
// (write-through) stores because the reader is another workgroup of the SAME launch.  Phase 2 (prep_phase2, the last workgroup to
// finish, from agent-scope loads): a scan of the <= 1 024 workgroup totals, then every sequence's outputs follow from
// offset[k] + local prefix with no serial dependence between sequences (the one-workgroup form carries `run` through 8 sequences per
// thread and 4 chunks).  Outputs are identical to prep_body's (lists ordered by batch slot).
// len_buf layout: [B] x 16 B per-sequence words, then [PREP_MAX_BLK] x 16 B workgroup totals.
#define PREP_MAX_BLK 1024
__device__ __forceinline__ int prep_clamp_len(int64_t n, int L) { return (int)(n < 0 ? 0 : (n > L ? L : n)); }
__device__ __forceinline__ void prep_pub(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long prep_get(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// block-wide inclusive scan of two packed words (wave shuffles + the wave totals through LDS); returns the exclusive parts and totals
template <int NT>
__device__ __forceinline__ void prep_scan2(unsigned long long s, unsigned long long s2, unsigned long long* part,
                                           unsigned long long& ex, unsigned long long& ex2, unsigned long long& tot, unsigned long long& tot2) {
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long inc = s, inc2 = s2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long v = __shfl_up(inc, o, 64), v2 = __shfl_up(inc2, o, 64);
        if (lane >= o) { inc += v; inc2 += v2; }
    }
    if (lane == 63) { part[wv] = inc; part[NWV + wv] = inc2; }
    __syncthreads();
    unsigned long long before = 0, before2 = 0;
    tot = 0; tot2 = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        const unsigned long long v = part[w], v2 = part[NWV + w];
        before += w < wv ? v : 0ull; before2 += w < wv ? v2 : 0ull;
        tot += v; tot2 += v2;
    }
    ex = before + inc - s; ex2 = before2 + inc2 - s2;
    __syncthreads();                                        // part[] is rewritten by the next call
}

template <int NT>
__device__ __forceinline__ void prep_phase1(const PrepArgs& P, const int blk, const int nblk, unsigned long long* part) {
    const PermSel sel = P.sel;
    const int64_t c = sel.perm ? (int64_t)*sel.counter : 0;
    int64_t* rw = const_cast<int64_t*>(P.rows);
    const int B = P.B, C = (B + nblk - 1) / nblk, b_lo = min(B, blk * C), b_hi = min(B, b_lo + C);
    unsigned long long* pre = reinterpret_cast<unsigned long long*>(P.len_buf);
    unsigned long long carry = 0, carry2 = 0;               // words: tokens | n_short << 32, n_long | n_tiny << 32
    for (int c0 = b_lo; c0 < b_hi; c0 += NT) {
        const int b = c0 + threadIdx.x;
        int nn = 0;
        if (b < b_hi) {
            int64_t row = b;
            if (sel.perm) { row = sel.perm[(c * sel.stride + sel.offset + b) % sel.n]; __hip_atomic_store(rw + b, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (P.rows) row = P.rows[b];
            nn = prep_clamp_len(P.seqlen[row], P.L);
        }
        const unsigned long long s = (unsigned long long)nn + (nn > DR4SR_TINY_MAX && nn <= 16 ? (1ull << 32) : 0ull);
        const unsigned long long s2 = (nn > 16 ? 1ull : 0ull) + (nn > 0 && nn <= DR4SR_TINY_MAX ? (1ull << 32) : 0ull);
        unsigned long long ex, ex2, tot, tot2;
        prep_scan2<NT>(s, s2, part, ex, ex2, tot, tot2);
        if (b < b_hi) {
            prep_pub(pre + 2 * (size_t)b, carry + ex);                                          // local token prefix | local n_short
            prep_pub(pre + 2 * (size_t)b + 1, ((carry2 + ex2) & 0xffffffffull) | ((unsigned long long)(((carry2 + ex2) >> 32) & 0xffffu) << 32)
                                                  | ((unsigned long long)nn << 48));          // local n_long | local n_tiny (16 bits) | length
        }
        carry += tot; carry2 += tot2;
    }
    if (threadIdx.x == 0) { prep_pub(pre + 2 * (size_t)(B + blk), carry); prep_pub(pre + 2 * (size_t)(B + blk) + 1, carry2); }
}

// wg / nwg: the per-sequence part is strided over nwg workgroups (k_prep_phase2: a launch of its own spread over the device — as the tail
// of the optimizer launch, on ONE workgroup, it was 22 us of a B = 8 192 step and 68 us of a B = 32 768 step: one CU's scattered-store
// rate); every workgroup repeats the (cheap) scan of the <= 1 024 totals, workgroup 0 writes the scalars.
template <int NT>
__device__ __forceinline__ void prep_phase2(const PrepArgs& P, const int nblk, unsigned long long* part, int4* boff, const int wg = 0,
                                            const int nwg = 1) {
    const int B = P.B, C = (B + nblk - 1) / nblk;
    const unsigned long long* pre = reinterpret_cast<const unsigned long long*>(P.len_buf);
    int* __restrict__ cu = P.cu; int* __restrict__ tile_seq = P.tile_seq; int* __restrict__ seq_class = P.seq_class;
    const int64_t* rw = P.rows;
    unsigned long long carry = 0, carry2 = 0;
    for (int k0 = 0; k0 < nblk; k0 += NT) {                  // exclusive scan of the workgroup totals
        const int k = k0 + threadIdx.x;
        unsigned long long s = 0, s2 = 0;
        if (k < nblk) { s = prep_get(pre + 2 * (size_t)(B + k)); s2 = prep_get(pre + 2 * (size_t)(B + k) + 1); }
        unsigned long long ex, ex2, tot, tot2;
        prep_scan2<NT>(s, s2, part, ex, ex2, tot, tot2);
        if (k < nblk) {
            const unsigned long long a = carry + ex, a2 = carry2 + ex2;
            boff[k] = make_int4((int)(a & 0xffffffffull), (int)(a >> 32), (int)(a2 & 0xffffffffull), (int)(a2 >> 32));
        }
        carry += tot; carry2 += tot2;
    }
    __syncthreads();
    const bool need_row = seq_class != nullptr && (P.sel.perm || rw);
    constexpr int U = 8;                                    // sequences per thread per round: their agent-scope loads fly together (32: no faster — one CU's scattered stores bound this loop)
    for (int b0 = wg * U * NT + threadIdx.x; b0 < B; b0 += nwg * U * NT) {
        unsigned long long w0[U], w1[U]; int64_t rowv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * NT;
            w0[u] = 0; w1[u] = 0; rowv[u] = b;
            if (b < B) {
                w0[u] = prep_get(pre + 2 * (size_t)b); w1[u] = prep_get(pre + 2 * (size_t)b + 1);
                if (need_row) rowv[u] = __hip_atomic_load(const_cast<int64_t*>(rw) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * NT;
            if (b >= B) break;
            const int4 o = boff[b / C];
            const int run = o.x + (int)(w0[u] & 0xffffffffull), nn = (int)(w1[u] >> 48);
            cu[b] = run;
            if (tile_seq)                                   // sequence slot of the first token of every 16-token tile it starts
                for (int q = (run + 15) >> 4; (q << 4) < run + nn; ++q) tile_seq[q] = b;
            if (seq_class && nn > 0) {
                if (nn <= DR4SR_TINY_MAX) {
                    const int nt = o.w + (int)((w1[u] >> 32) & 0xffffu);
                    seq_class[4 + 6 * B + nt] = b;
                    reinterpret_cast<int4*>(seq_class + 4)[nt] = make_int4(run, nn, b, (int)rowv[u]);
                } else if (nn <= 16) seq_class[4 + 4 * B + o.y + (int)(w0[u] >> 32)] = b;
                else seq_class[4 + 5 * B + o.z + (int)(w1[u] & 0xffffffffull)] = b;
            }
        }
    }
    if (wg != 0) return;
    if (P.sel.perm && threadIdx.x == 0) *P.sel.counter = *P.sel.counter + 1;       // every reader of the counter is past its launch's ticket
    if (threadIdx.x == NT - 1) {
        const int T = (int)(carry & 0xffffffffull);
        cu[B] = T;
        P.state[DR4SR_STATE_T] = T;
        if (P.bump_rng) P.state[DR4SR_STATE_RNGSTEP] += 1;
        if (seq_class) { seq_class[0] = (int)(carry >> 32); seq_class[1] = (int)(carry2 & 0xffffffffull); seq_class[2] = (int)(carry2 >> 32); }
    }
}

// one workgroup of NT threads; `part` = NT words of LDS.
// The batch is walked in chunks of 8 * NT sequences (8 consecutive sequences per thread, their loads issued together, everything
// the second pass needs kept in registers) with block-uniform running totals carried from chunk to chunk.
template <int NT>
__device__ __forceinline__ void prep_body(const PrepArgs& P, unsigned long long* part) {
    const int64_t* __restrict__ seqlen = P.seqlen; const int64_t* rows = P.rows;
    int* __restrict__ cu = P.cu; int* __restrict__ state = P.state; const int B = P.B, L = P.L, bump_rng = P.bump_rng;
    const PermSel sel = P.sel; int* __restrict__ tile_seq = P.tile_seq; int* __restrict__ seq_class = P.seq_class;
    // one packed scan: bits 0-31 tokens, 32-47 short sequences (9..16 tokens), 48-63 long sequences; a second word counts the tiny
    // sequences (1..8 tokens: the VALU attention class, attn_tiny_body.h)
    constexpr int KEEP = 8, NWV = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t c = sel.perm ? (int64_t)*sel.counter : 0;       // a1: this step's batch = a slice of the epoch permutation
    int64_t* rw = const_cast<int64_t*>(rows);
    unsigned long long carry = 0;                       // running totals of the chunks already done (block-uniform)
    unsigned int carry2 = 0;
    // sequences per thread per chunk: as few as the batch allows (B <= NT: ONE per thread — at B = 256 this workgroup is the critical
    // path of the optimizer launch, and 8 per thread would leave 7/8 of it idle behind 8-long serial chains)
    const int kr = min(KEEP, (B + NT - 1) / NT);
    for (int chunk0 = 0; chunk0 < B; chunk0 += NT * kr) {
        const int b0 = chunk0 + tid * kr, b1 = min(B, b0 + kr);
        int keep[KEEP]; int64_t krow[KEEP];
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            keep[k] = 0; krow[k] = 0;
            if (b0 + k < b1) {
                const int b = b0 + k;
                if (sel.perm) { krow[k] = sel.perm[(c * sel.stride + sel.offset + b) % sel.n]; rw[b] = krow[k]; }   // every thread selects
                else krow[k] = rows ? rows[b] : (int64_t)b;                                                     //  the rows of ITS chunk
            }
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) if (b0 + k < b1) keep[k] = prep_clamp_len(seqlen[krow[k]], L);
        unsigned long long s = 0;
        unsigned int s2 = 0;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int nn = keep[k];
            s += (unsigned long long)nn + (nn > DR4SR_TINY_MAX && nn <= 16 ? (1ull << 32) : 0ull) + (nn > 16 ? (1ull << 48) : 0ull);
            s2 += nn > 0 && nn <= DR4SR_TINY_MAX ? 1u : 0u;
        }
        // inclusive scan: shuffles inside the wave, then the <= 16 wave totals through LDS (one barrier instead of 2 log2(NT))
        unsigned long long inc = s;
        unsigned int inc2 = s2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long v = __shfl_up(inc, o, 64);
            const unsigned int v2 = __shfl_up(inc2, o, 64);
            if (lane >= o) { inc += v; inc2 += v2; }
        }
        if (lane == 63) { part[wv] = inc; part[NWV + wv] = inc2; }
        __syncthreads();
        unsigned long long before = 0, tot = 0;
        unsigned int before2 = 0, tot2 = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const unsigned long long v = part[w];
            const unsigned int v2 = (unsigned int)part[NWV + w];
            before += w < wv ? v : 0ull;
            before2 += w < wv ? v2 : 0u;
            tot += v;
            tot2 += v2;
        }
        const unsigned long long ex = carry + before + inc - s;     // exclusive prefix of this thread's 8 sequences (tokens never carry
        int run = (int)(ex & 0xffffffffull), ns = (int)((ex >> 32) & 0xffff), nl = (int)(ex >> 48);     //  into bit 32: T < 2^32)
        int nt = (int)(carry2 + before2 + inc2 - s2);
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            if (b0 + k < b1) {
                const int b = b0 + k, nn = keep[k];
                cu[b] = run;
                if (tile_seq)                               // sequence slot of the first token of every 16-token tile it starts
                    for (int q = (run + 15) >> 4; (q << 4) < run + nn; ++q) tile_seq[q] = b;
                if (seq_class && nn > 0) {                  // length classes for the split attention launches
                    if (nn <= DR4SR_TINY_MAX) {             // tiny class: list entry + a 16-byte descriptor {t0, n, slot, dataset row}, so that
                        seq_class[4 + 6 * B + nt] = b;      // the VALU kernels start from ONE load instead of the list -> cu -> rows chain
                        reinterpret_cast<int4*>(seq_class + 4)[nt] = make_int4(run, nn, b, (int)krow[k]);
                        ++nt;
                    } else if (nn <= 16) seq_class[4 + 4 * B + ns++] = b;
                    else seq_class[4 + 5 * B + nl++] = b;
                }
                run += nn;
            }
        }
        carry += tot;
        carry2 += tot2;
        __syncthreads();                                    // part[] is rewritten by the next chunk
    }
    if (sel.perm && tid == 0) *sel.counter = *sel.counter + 1;      // every reader of the counter is past a barrier
    if (tid == NT - 1) {
        cu[B] = (int)(carry & 0xffffffffull);
        state[DR4SR_STATE_T] = (int)(carry & 0xffffffffull);
        if (bump_rng) state[DR4SR_STATE_RNGSTEP] += 1;
        if (seq_class) { seq_class[0] = (int)((carry >> 32) & 0xffff); seq_class[1] = (int)(carry >> 48); seq_class[2] = (int)carry2; }
    }
}
