// attn_args.h — argument block shared by the MFMA attention kernels (attn_mfma.hip) and the tiny-sequence VALU class (attn_tiny_body.h)
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct AttnArgs2 {
    const float* qkv; float* ctx;
    const float* dctx; float* dqkv;
    const int64_t* idx; const int64_t* rows; const int* cu;
    const int* state; uint64_t seed; float p; int layer; int training; int L;
    float* stat;                 // [T][H][2]  softmax row max, 1/row sum: written by the forward, read by the backward
    const float* rd;             // [T][H]     sum_j P dP = <dctx, ctx> per head, from the epilogue of k_post_bwd
    // large batches: sequences are split by length class (k_prep's seq_class lists) into a short kernel (n <= 16: 16 LDS rows,
    // 2 waves, ~9 workgroups per CU) and a long kernel, each a persistent loop over its list.  list == NULL: block b = sequence b.
    const int* list; const int* list_count;
    const int* desc;             // tiny class: int4 {t0, n, slot, dataset row} per list entry (k_prep), 16-byte aligned
    unsigned* keep;              // [T][H][2] dropout keep bits per (token, head), written by the wave-per-tile forward and read by its backward (attn_wave.hip)
    const int2* tok;             // [T] per-token words of the embedding stage {first token of the sequence, slot | length << 20 | PAD << 30}: the wave-per-tile form (attn_wave.hip); NULL: lists
};

// at scale, short-sequence plans (round 6): one wave per (16-token tile, head[, phase]) of the packed stream, no lists, no LDS — attn_wave.hip
int launch_attn_wave(const AttnArgs2& A, int DH, int Tmax, int Thint, bool bwd, hipStream_t s);

// tiny-sequence class (1..DR4SR_TINY_MAX tokens), attn_tiny_body.h: one wave per 4 list entries
int launch_attn_tiny(const AttnArgs2& A, int DH, int B, bool bwd, hipStream_t s);
