// prep_body.h — the per-step "prep" (batch selection, prefix scan of the sequence lengths, search hints, length classes) as a
// device function of one workgroup, shared by k_prep (embed.hip) and by the Adam kernel that prepares the NEXT step (step.hip).
#pragma once
#include "common.h"

// sequences of 1..DR4SR_TINY_MAX tokens form the third attention length class (8 lanes per (sequence, head), no MFMA)
#define DR4SR_TINY_MAX 8

struct PermSel { const int64_t* perm; int64_t n, stride, offset; int* counter; };

struct PrepArgs {
    const int64_t* seqlen; const int64_t* rows; int* cu; int* state; int B, L, bump_rng; PermSel sel; int* tile_seq; int* seq_class;
    int* len_buf;                              // [B] scratch of the two-phase form (prep_select -> prep_body<NT, true>), or NULL
};

// Two-phase form for large batches.  Phase 1 (prep_select, run by EVERY workgroup of a launch that has many — the optimizer launch):
// batch selection + the seqlen gather, coalesced over the whole grid; rows[] and len_buf[] are published with agent-scope
// (write-through) stores because their reader is another workgroup of the SAME launch: the last one to finish, which runs
// prep_body<NT, true> (the scan and everything that depends on it) from agent-scope loads.  One workgroup doing the selection of 8 192
// sequences by itself is bound by a single CU's scattered-access rate: 41 us as a launch of its own, every step.
__device__ __forceinline__ int prep_clamp_len(int64_t n, int L) { return (int)(n < 0 ? 0 : (n > L ? L : n)); }
template <int NT>
__device__ __forceinline__ void prep_select(const PrepArgs& P, int blk, int nblk) {
    const PermSel sel = P.sel;
    const int64_t c = sel.perm ? (int64_t)*sel.counter : 0;
    int64_t* rw = const_cast<int64_t*>(P.rows);
    for (int b = blk * NT + threadIdx.x; b < P.B; b += nblk * NT) {
        int64_t row = b;
        if (sel.perm) {
            row = sel.perm[(c * sel.stride + sel.offset + b) % sel.n];
            __hip_atomic_store(rw + b, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (P.rows) row = P.rows[b];
        __hip_atomic_store(P.len_buf + b, prep_clamp_len(P.seqlen[row], P.L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one workgroup of NT threads; `part` = NT words of LDS.  PRE: phase 2 of the two-phase form (lengths and rows come from prep_select).
// The batch is walked in chunks of 8 * NT sequences (8 consecutive sequences per thread, their loads issued together, everything
// the second pass needs kept in registers) with block-uniform running totals carried from chunk to chunk.
template <int NT, bool PRE = false>
__device__ __forceinline__ void prep_body(const PrepArgs& P, unsigned long long* part) {
    const int64_t* __restrict__ seqlen = P.seqlen; const int64_t* rows = P.rows;
    int* __restrict__ cu = P.cu; int* __restrict__ state = P.state; const int B = P.B, L = P.L, bump_rng = P.bump_rng;
    const PermSel sel = P.sel; int* __restrict__ tile_seq = P.tile_seq; int* __restrict__ seq_class = P.seq_class;
    // one packed scan: bits 0-31 tokens, 32-47 short sequences (9..16 tokens), 48-63 long sequences; a second word counts the tiny
    // sequences (1..8 tokens: the VALU attention class, attn_tiny_body.h)
    constexpr int KEEP = 8, NWV = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t c = (sel.perm && !PRE) ? (int64_t)*sel.counter : 0;       // a1: this step's batch = a slice of the epoch permutation
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
                if (PRE) {
                    krow[k] = (sel.perm || rows) ? __hip_atomic_load(rw + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (int64_t)b;
                    keep[k] = __hip_atomic_load(P.len_buf + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (sel.perm) { krow[k] = sel.perm[(c * sel.stride + sel.offset + b) % sel.n]; rw[b] = krow[k]; }   // every thread selects
                    else krow[k] = rows ? rows[b] : (int64_t)b;                                                     //  the rows of ITS chunk
                }
            }
        }
        if (!PRE) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) if (b0 + k < b1) keep[k] = prep_clamp_len(seqlen[krow[k]], L);
        }
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
    if (sel.perm && tid == 0) *sel.counter = *sel.counter + 1;      // every reader of the counter is past a barrier (PRE: past its launch's ticket)
    if (tid == NT - 1) {
        cu[B] = (int)(carry & 0xffffffffull);
        state[DR4SR_STATE_T] = (int)(carry & 0xffffffffull);
        if (bump_rng) state[DR4SR_STATE_RNGSTEP] += 1;
        if (seq_class) { seq_class[0] = (int)((carry >> 32) & 0xffff); seq_class[1] = (int)(carry >> 48); seq_class[2] = (int)carry2; }
    }
}
