// prep_body.h — the per-step "prep" (batch selection, prefix scan of the sequence lengths, search hints, length classes) as a
// device function of one workgroup, shared by k_prep (embed.hip) and by the Adam kernel that prepares the NEXT step (step.hip).
#pragma once
#include "common.h"

// sequences of 1..DR4SR_TINY_MAX tokens form the third attention length class (8 lanes per (sequence, head), no MFMA)
#define DR4SR_TINY_MAX 8

struct PermSel { const int64_t* perm; int64_t n, stride, offset; int* counter; };

struct PrepArgs {
    const int64_t* seqlen; const int64_t* rows; int* cu; int* state; int B, L, bump_rng; PermSel sel; int* tile_seq; int* seq_class;
};
// one workgroup of NT threads; `part` = NT words of LDS
template <int NT>
__device__ __forceinline__ void prep_body(const PrepArgs& P, unsigned long long* part) {
    const int64_t* __restrict__ seqlen = P.seqlen; const int64_t* rows = P.rows;
    int* __restrict__ cu = P.cu; int* __restrict__ state = P.state; const int B = P.B, L = P.L, bump_rng = P.bump_rng;
    const PermSel sel = P.sel; int* __restrict__ tile_seq = P.tile_seq; int* __restrict__ seq_class = P.seq_class;
    // one packed scan: bits 0-31 tokens, 32-47 short sequences (9..16 tokens), 48-63 long sequences; a second word counts the tiny
    // sequences (1..8 tokens: the VALU attention class, attn_tiny.hip)
    const int tid = threadIdx.x;
    const int per = (B + NT - 1) / NT;
    const int b0 = tid * per, b1 = min(B, b0 + per);
    if (sel.perm) {                                     // a1: this step's batch = a slice of the epoch permutation.  Every thread
        const int64_t c = *sel.counter;                 // selects the rows of ITS chunk (it is their only reader below): no barrier,
        int64_t* rw = const_cast<int64_t*>(rows);       // no second round trip; the counter is bumped behind the scan's barrier
        for (int b = b0; b < b1; ++b) rw[b] = sel.perm[(c * sel.stride + sel.offset + b) % sel.n];
    }
    unsigned long long s = 0;
    unsigned int s2 = 0;
    constexpr int KEEP = 8;                             // lengths of the first 8 sequences of the chunk stay in registers (B <= 8192):
    int keep[KEEP];                                     // independent loads issued together instead of 3 x per dependent chains
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        int nn = 0;
        if (b0 + k < b1) {
            const int64_t n = seqlen[rows ? rows[b0 + k] : b0 + k];
            nn = (int)(n < 0 ? 0 : (n > L ? L : n));
        }
        keep[k] = nn;
    }
    auto len_of = [&](int b) -> int {
        const int k = b - b0;
        if (k < KEEP) {
            int v = 0;
#pragma unroll
            for (int q = 0; q < KEEP; ++q) v = k == q ? keep[q] : v;
            return v;
        }
        const int64_t n = seqlen[rows ? rows[b] : b];
        return (int)(n < 0 ? 0 : (n > L ? L : n));
    };
    for (int b = b0; b < b1; ++b) {
        const int nn = len_of(b);
        s += (unsigned long long)nn + (nn > DR4SR_TINY_MAX && nn <= 16 ? (1ull << 32) : 0ull) + (nn > 16 ? (1ull << 48) : 0ull);
        s2 += nn > 0 && nn <= DR4SR_TINY_MAX ? 1u : 0u;
    }
    // inclusive scan: shuffles inside the wave, then the <= 16 wave totals through LDS (one barrier instead of 2 log2(NT))
    const int lane = tid & 63, wv = tid >> 6;
    unsigned long long inc = s;
    unsigned int inc2 = s2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long v = __shfl_up(inc, o, 64);
        const unsigned int v2 = __shfl_up(inc2, o, 64);
        if (lane >= o) { inc += v; inc2 += v2; }
    }
    if (lane == 63) { part[wv] = inc; part[NT / 64 + wv] = inc2; }
    __syncthreads();
    if (sel.perm && tid == 0) *sel.counter = *sel.counter + 1;      // every thread has read the counter before the barrier
    unsigned long long before = 0, tot = 0;
    unsigned int before2 = 0, tot2 = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const unsigned long long v = part[w];
        const unsigned int v2 = (unsigned int)part[NT / 64 + w];
        before += w < wv ? v : 0ull;
        before2 += w < wv ? v2 : 0u;
        tot += v;
        tot2 += v2;
    }
    const unsigned long long ex = before + inc - s;     // exclusive prefix of this thread's chunk
    int run = (int)(ex & 0xffffffffull), ns = (int)((ex >> 32) & 0xffff), nl = (int)(ex >> 48), nt = (int)(before2 + inc2 - s2);
    for (int b = b0; b < b1; ++b) {
        cu[b] = run;
        const int nn = len_of(b);
        if (tile_seq)                                   // sequence slot of the first token of every 16-token tile it starts
            for (int k = (run + 15) >> 4; (k << 4) < run + nn; ++k) tile_seq[k] = b;
        if (seq_class && nn > 0) {                      // length classes for the split attention launches
            if (nn <= DR4SR_TINY_MAX) {               // tiny class: list entry + a 16-byte descriptor {t0, n, slot, dataset row}, so that
                seq_class[4 + 6 * B + nt] = b;        // the VALU kernels start from ONE load instead of the list -> cu -> rows chain
                reinterpret_cast<int4*>(seq_class + 4)[nt] = make_int4(run, nn, b, (int)(rows ? rows[b] : (int64_t)b));
                ++nt;
            } else if (nn <= 16) seq_class[4 + 4 * B + ns++] = b;
            else seq_class[4 + 5 * B + nl++] = b;
        }
        run += nn;
    }
    if (tid == NT - 1) {
        cu[B] = (int)(tot & 0xffffffffull);
        state[DR4SR_STATE_T] = (int)(tot & 0xffffffffull);
        if (bump_rng) state[DR4SR_STATE_RNGSTEP] += 1;
        if (seq_class) { seq_class[0] = (int)((tot >> 32) & 0xffff); seq_class[1] = (int)(tot >> 48); seq_class[2] = (int)tot2; }
    }
}

