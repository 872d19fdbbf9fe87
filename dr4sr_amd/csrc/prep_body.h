// prep_body.h — the per-step "prep" (batch selection, prefix scan of the sequence lengths, search hints, length classes) as a
// device function of one workgroup, shared by k_prep (embed.hip) and by the Adam kernel that prepares the NEXT step (step.hip).
#pragma once
#include "common.h"

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
    // one packed scan: bits 0-31 tokens, 32-47 short sequences (1..16 tokens), 48-63 long sequences
    const int tid = threadIdx.x;
    if (sel.perm) {                                     // a1: this step's batch = a slice of the epoch permutation
        const int64_t c = *sel.counter;
        int64_t* rw = const_cast<int64_t*>(rows);
        for (int i = tid; i < B; i += NT) rw[i] = sel.perm[(c * sel.stride + sel.offset + i) % sel.n];
        __syncthreads();
        if (tid == 0) *sel.counter = (int)(c + 1);
    }
    const int per = (B + NT - 1) / NT;
    const int b0 = tid * per, b1 = min(B, b0 + per);
    unsigned long long s = 0;
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
        s += (unsigned long long)nn + (nn > 0 && nn <= 16 ? (1ull << 32) : 0ull) + (nn > 16 ? (1ull << 48) : 0ull);
    }
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {               // Hillis-Steele inclusive scan
        const unsigned long long v = tid >= o ? part[tid - o] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long ex = part[tid] - s;        // exclusive prefix of this thread's chunk
    int run = (int)(ex & 0xffffffffull), ns = (int)((ex >> 32) & 0xffff), nl = (int)(ex >> 48);
    for (int b = b0; b < b1; ++b) {
        cu[b] = run;
        const int nn = len_of(b);
        if (tile_seq)                                   // sequence slot of the first token of every 16-token tile it starts
            for (int k = (run + 15) >> 4; (k << 4) < run + nn; ++k) tile_seq[k] = b;
        if (seq_class && nn > 0) {                      // length classes for the split attention launches
            if (nn <= 16) seq_class[2 + ns++] = b; else seq_class[2 + B + nl++] = b;
        }
        run += nn;
    }
    if (tid == NT - 1) {
        const unsigned long long tot = part[NT - 1];
        cu[B] = (int)(tot & 0xffffffffull);
        state[DR4SR_STATE_T] = (int)(tot & 0xffffffffull);
        if (bump_rng) state[DR4SR_STATE_RNGSTEP] += 1;
        if (seq_class) { seq_class[0] = (int)((tot >> 32) & 0xffff); seq_class[1] = (int)(tot >> 48); }
    }
}

