// gru_coop.hip — multi-CU cooperative GRU recurrence for SMALL batches (B <= 384 per launch, up to 1536 in chunks): latency instead of throughput.
//
// The single-workgroup recurrence of gru.hip (16 sequences per workgroup, W_hh streamed from L2) is bound by one CU's fp32
// MFMA pipe: 6.3 MFLOP per time step = ~11 us, and a B = 256 batch occupies 16 of the 256 CUs.  Here a group of 16 sequences
// is spread over NS workgroups (NS CUs): slice s owns hidden units [s*H/NS, (s+1)*H/NS) of all three gates, keeps its
// 3*H/NS rows of W_hh RESIDENT in LDS for the whole launch and needs 1/NS of the MFMA work per step; the price is one
// all-gather of h_t (forward) / one reduce-scatter of the dh partials (backward) per time step between the NS workgroups of a
// group, INSIDE the launch.  NS = 8 (98 KB of weights per slice, one workgroup per CU, up to 24 groups) or, for batches of at
// most 16 groups at H = 256, NS = 16 (49 KB, 4 waves per slice): the step's MFMA chain halves (1.28 -> 0.64 us of a 3.5 us
// step) and 16 groups x 16 slices put one workgroup on every CU of an MI355X — B = 256: 288 k -> 343 k sequences/s.  When a slice
// has ONE 16-unit tile (NS = 16 at H = 256, NS = 8 at H = 128) the forward does not stage h in LDS at all: every wave polls the
// granules that are its MFMA A fragments (355 k sequences/s); both kernels need one workgroup barrier per time step.
//
// Exchange protocol (cdna_hip_programming.md §6 Guideline 16, form R2 "the data is the flag"): every exchanged float travels
// as ONE aligned 8-byte granule {tag, value} written with a relaxed agent-scope store (sc1, write-through) and polled with
// relaxed agent-scope loads until the tag matches this step's epoch — no fences, no separate flags, correct for any
// placement of the workgroups over XCDs.  Two granule buffers alternate by step parity (a slice can only produce step t+2
// after every slice has consumed step t).  Epochs are unique across launches: epoch = 64 * launch_counter + step + 1 with the
// launch counter kept in a device word that the last workgroup to finish increments, so nothing has to be re-zeroed per call
// (the exchange area must be zero once, when the workspace is created).  All NS*ceil(B/16) workgroups must be co-resident
// (NS = 8: one per CU, ~140 KB LDS; NS = 16: 79 KB, two fit a CU): the launcher only takes this path when they fit with
// margin (coop_slices); spins are bounded and a timeout raises a device error word instead of hanging.
#include "common.h"
#include "kernels.h"

#include <cstdlib>
#include <mutex>
#include <unordered_map>

extern __shared__ __attribute__((aligned(16))) float smem[];

namespace {

// slices (workgroups) per group of 16 sequences: 8, or 16 when the batch has at most 16 groups and H = 256 (a step's MFMA chain —
// 6.3 MFLOP on one group's CUs, 1.28 us on 8 of them — is the largest item of the 3.5 us step; 16 groups x 16 slices = one
// workgroup per CU of an MI355X)
constexpr int coop_threads(int H, int NS) { return ((H / NS) / 16) * 4 * 64; }
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef unsigned long long u64;

struct CoopArgs {
    const float* gi; const float* whh; const int* cu;
    float* r; float* z; float* n; float* ghn; float* hprev; float* hout;          // saved per token [T,H]
    const float* dhout; float* dgi; float* dgh;                                   // backward
    u64* xch;                    // granules: fwd [grp][2][16][H] | bwd [grp][2][NS][16][H]
    int* ctl;                    // [0] launch counter, [1] finish ticket, [2] error word
    int B;
};

__device__ __forceinline__ f32x4 mfma16c(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void put_granule(u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// poll until the tag matches (bounded; a timeout anywhere releases everybody through the error word)
__device__ __forceinline__ float get_granule(const u64* g, unsigned tag, int* err) {
    unsigned spins = 0;
    for (;;) {
        const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == tag) return __uint_as_float((unsigned)x);
        if ((++spins & 1023u) == 0) {
            if (spins >= SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 0.f;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// all N granules of a lane are requested together and re-read every pass until every tag matches (the guide's sweep): one L2
// round trip per pass instead of one per granule
template <int N>
__device__ __forceinline__ void sweep_granules(const u64* g, int stride, unsigned tag, float (&v)[N], int* err) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u64 x = __hip_atomic_load(g + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = __uint_as_float((unsigned)x);
            ok &= (unsigned)(x >> 32) == tag;
        }
        if (ok) return;
        if ((++spins & 255u) == 0) {
            if (spins >= SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

__device__ __forceinline__ void finish_launch(int* ctl) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(&ctl[1], 1);
        if (t == (int)gridDim.x - 1) { ctl[1] = 0; ctl[0] += 1; }          // visible to the next launch (kernel boundary)
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int H, int NS>
__global__ __launch_bounds__(coop_threads(H, NS)) void k_gru_fwd_coop(const CoopArgs A) {
    constexpr int US = H / NS, UTL = US / 16, NW = UTL * 4, NT = NW * 64, LDW = H + 4;
    static_assert(US % 16 == 0 && NT == coop_threads(H, NS) && 16 * US <= NT && (16 * H) % NT == 0, "slice geometry");
    float* Ws = smem;                                     // [3*US][LDW]  this slice's rows of W_hh (gate-major)
    // one unit tile per slice (UTL == 1: every h value is multiplied by exactly one wave): the waves poll their own A fragments;
    // two unit tiles (H = 256 on 8 slices): h_t is gathered once per workgroup into LDS (two waves would poll each granule)
    constexpr bool DIRECT = UTL == 1;
    constexpr int LDH = H + 4;
    float* hA = Ws + 3 * US * LDW;                        // !DIRECT: [16][LDH] h_{t-1} of the group's 16 sequences
    float* part = DIRECT ? hA : hA + 16 * LDH;            // DIRECT: [2 parity][4 kq][3][16][US], else [4 kq][3][16][US]
    int* meta = reinterpret_cast<int*>(part + (DIRECT ? 2 : 1) * 4 * 3 * 16 * US);   // [16] t0, [16] n
    // speed-only placement (block b is observed on XCD b % 8): the 8 slices of a group sit on ONE XCD, so their per-step exchange
    // stays inside that XCD's L2; correctness does not depend on it (agent-scope granules)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int grp = (jx / NS) * 8 + xcd, sl = jx % NS, b0 = grp * 16;
    if (b0 >= A.B) { finish_launch(A.ctl); return; }
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    // W_hh rows of the slice.  DIRECT: columns PERMUTED inside each K quarter of H/4 — the MFMA step s of lane group g multiplies K
    // index kq H/4 + 4 s + g (so that the h granules one poll instruction reads are 64 consecutive ones, below); a lane's float4
    // number c must then hold the columns 16 c + 4 j + g, j = 0..3, at position g H/16 + 4 c + j.
    for (int i = threadIdx.x; i < 3 * US * H; i += NT) {
        const int lr = i / H, k = i % H, gate = lr / US, u = lr % US;
        const int kq_ = k / (H / 4), kl = k % (H / 4), cc = kl / 16, jj = (kl % 16) / 4, gg = kl % 4;
        const int pos = DIRECT ? kq_ * (H / 4) + gg * (H / 16) + 4 * cc + jj : k;
        Ws[lr * LDW + pos] = A.whh[(size_t)(gate * H + sl * US + u) * H + k];
    }
    if constexpr (!DIRECT) { for (int i = threadIdx.x; i < 16 * LDH; i += NT) hA[i] = 0.f; }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const unsigned base = (unsigned)A.ctl[0] * 64u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int ct = w >> 2, kq = w & 3;                    // wave = (unit tile, K quarter)
    // element owned by this thread in the gate math: (sequence es, unit eu of the slice)
    const bool own = (int)threadIdx.x < 16 * US;
    const int es = own ? threadIdx.x / US : 0, eu = own ? threadIdx.x % US : 0;
    const int tq = meta[es], nq = own ? meta[16 + es] : 0, gu = sl * US + eu;     // global unit
    float hown = 0.f;
    u64* xg = A.xch + (size_t)grp * 2 * 16 * H;
    if constexpr (DIRECT) {
        // A operand of the step's MFMAs, in registers: lane (l16, g) of K-quarter wave kq supplies h_{t-1}[sequence l16][kq H/4 + 4 s + g]
        // at MFMA step s — sixteen granules of the exchange buffer, 64 apart.  Each wave polls exactly the granules it multiplies with (no
        // staging of h in LDS, no second barrier per step); the `part` tiles alternate by step parity so ONE barrier per step is enough.
        constexpr int KV = H / 16;
        float hv[KV];
    #pragma unroll
        for (int k = 0; k < KV; ++k) hv[k] = 0.f;
        for (int t = 0; t < nmax; ++t) {
            const bool act = t < nq;
            float gir = 0.f, giz = 0.f, gin = 0.f;
            if (act) { const float* gip = A.gi + (size_t)(tq + t) * 3 * H + gu; gir = gip[0]; giz = gip[H]; gin = gip[2 * H]; }
            if (t > 0)                                         // all-gather of h_{t-1}: this lane's slice of the A operand
                sweep_granules<KV>(xg + (size_t)((t - 1) & 1) * 16 * H + (size_t)(kq * (H / 16) * 16 + l16) * 4 + g, 64, base + t, hv, A.ctl + 2);
            // gh partial over this wave's K quarter for its 16 units, all three gates
            f32x4 acc[3];
    #pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) acc[q3] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* br = Ws + (ct * 16 + l16) * LDW + kq * (H / 4) + g * KV;
    #pragma unroll
            for (int c = 0; c < KV; c += 4) {
    #pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) {
                    const float4 b = ld4(br + q3 * US * LDW + c);
                    acc[q3] = mfma16c(hv[c], b.x, acc[q3]); acc[q3] = mfma16c(hv[c + 1], b.y, acc[q3]);
                    acc[q3] = mfma16c(hv[c + 2], b.z, acc[q3]); acc[q3] = mfma16c(hv[c + 3], b.w, acc[q3]);
                }
            }
            float* pt = part + (t & 1) * (4 * 3 * 16 * US);
    #pragma unroll
            for (int q3 = 0; q3 < 3; ++q3)
    #pragma unroll
                for (int r = 0; r < 4; ++r) pt[((kq * 3 + q3) * 16 + 4 * g + r) * US + ct * 16 + l16] = acc[q3][r];
            __syncthreads();
            if (own) {
                float gh[3];
    #pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) {
                    const float* pp = pt + (q3 * 16 + es) * US + eu;
                    gh[q3] = (pp[0] + pp[3 * 16 * US]) + (pp[2 * 3 * 16 * US] + pp[3 * 3 * 16 * US]);
                }
                if (act) {
                    const float rr = sigm(gir + gh[0]), zz = sigm(giz + gh[1]), nn = tanh_f(gin + rr * gh[2]);
                    const float hnew = (1.0f - zz) * nn + zz * hown;
                    const size_t o = (size_t)(tq + t) * H + gu;
                    A.r[o] = rr; A.z[o] = zz; A.n[o] = nn; A.ghn[o] = gh[2]; A.hprev[o] = hown; A.hout[o] = hnew;
                    hown = hnew;
                }
                // exchange layout [K / 4][sequence][K % 4]: the 64 lanes of a poll instruction (16 sequences x 4 lane groups, one MFMA
                // step) read 512 contiguous bytes
                if (t + 1 < nmax) put_granule(xg + (size_t)(t & 1) * 16 * H + (size_t)((gu >> 2) * 16 + es) * 4 + (gu & 3), base + t + 1, hown);
            }
        }
    } else {
        for (int t = 0; t < nmax; ++t) {
            const bool act = t < nq;
            float gir = 0.f, giz = 0.f, gin = 0.f;
            if (act) { const float* gip = A.gi + (size_t)(tq + t) * 3 * H + gu; gir = gip[0]; giz = gip[H]; gin = gip[2 * H]; }
            // gh partial over this wave's K quarter for its 16 units, all three gates
            f32x4 acc[3];
    #pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) acc[q3] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* ar = hA + l16 * LDH + kq * (H / 4) + g * (H / 16);
            const float* br = Ws + (ct * 16 + l16) * LDW + kq * (H / 4) + g * (H / 16);
    #pragma unroll
            for (int c = 0; c < H / 16; c += 4) {
                const float4 a = ld4(ar + c);
    #pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) {
                    const float4 b = ld4(br + q3 * US * LDW + c);
                    acc[q3] = mfma16c(a.x, b.x, acc[q3]); acc[q3] = mfma16c(a.y, b.y, acc[q3]);
                    acc[q3] = mfma16c(a.z, b.z, acc[q3]); acc[q3] = mfma16c(a.w, b.w, acc[q3]);
                }
            }
    #pragma unroll
            for (int q3 = 0; q3 < 3; ++q3)
    #pragma unroll
                for (int r = 0; r < 4; ++r) part[((kq * 3 + q3) * 16 + 4 * g + r) * US + ct * 16 + l16] = acc[q3][r];
            __syncthreads();
            if (own) {
                float gh[3];
    #pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) {
                    const float* pp = part + (q3 * 16 + es) * US + eu;
                    gh[q3] = (pp[0] + pp[3 * 16 * US]) + (pp[2 * 3 * 16 * US] + pp[3 * 3 * 16 * US]);
                }
                if (act) {
                    const float rr = sigm(gir + gh[0]), zz = sigm(giz + gh[1]), nn = tanh_f(gin + rr * gh[2]);
                    const float hnew = (1.0f - zz) * nn + zz * hown;
                    const size_t o = (size_t)(tq + t) * H + gu;
                    A.r[o] = rr; A.z[o] = zz; A.n[o] = nn; A.ghn[o] = gh[2]; A.hprev[o] = hown; A.hout[o] = hnew;
                    hown = hnew;
                }
                if (t + 1 < nmax) put_granule(xg + ((size_t)(t & 1) * 16 + es) * H + gu, base + t + 1, hown);
            }
            if (t + 1 < nmax) {                                // all-gather h_t of the 8 slices into the A operand tile
                const u64* src = xg + (size_t)(t & 1) * 16 * H;
                float hv[(16 * H) / NT];                       // 8 granules per thread, stride NT
                sweep_granules<(16 * H) / NT>(src + threadIdx.x, NT, base + t + 1, hv, A.ctl + 2);
    #pragma unroll
                for (int k = 0; k < (16 * H) / NT; ++k) { const int i = threadIdx.x + k * NT; hA[(i / H) * LDH + (i % H)] = hv[k]; }
            }
            __syncthreads();
        }
    }
    finish_launch(A.ctl);
}

// ------------------------------------------------------------------------------------------------ backward (BPTT)
// Per step (t descending): dh = dhout[t] + carry for the slice's own units; gate derivatives -> the slice's dgh tile [16][3*US]
// (LDS) and dgi/dgh rows (global); partial[16][H] = dgh_tile . W_hh[slice rows][:] on MFMA (W rows read column-wise);
// reduce-scatter over the 8 slices: carry'[own units] = dh*z + sum_slices partial[:, own units].
template <int H, int NS>
__global__ __launch_bounds__(coop_threads(H, NS)) void k_gru_bwd_coop(const CoopArgs A) {
    constexpr int US = H / NS, UTL = US / 16, NW = UTL * 4, NT = NW * 64, LDW = H + 4, KL = 3 * US, LDG = KL + 4;
    constexpr int CTW = (H / 16) / NW;                    // output column tiles per wave
    float* Ws = smem;                                     // [3*US][LDW]
    float* dgl0 = Ws + 3 * US * LDW;                      // [2 parity][16][LDG]  dgh tile of this slice (A operand): two tiles by step
    int* meta = reinterpret_cast<int*>(dgl0 + 2 * 16 * LDG);   // parity, so that ONE barrier per time step is enough
    // speed-only placement (block b is observed on XCD b % 8): the 8 slices of a group sit on ONE XCD, so their per-step exchange
    // stays inside that XCD's L2; correctness does not depend on it (agent-scope granules)
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int grp = (jx / NS) * 8 + xcd, sl = jx % NS, b0 = grp * 16;
    if (b0 >= A.B) { finish_launch(A.ctl); return; }
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    for (int i = threadIdx.x; i < 3 * US * (H / 4); i += NT) {
        const int lr = i / (H / 4), c = (i % (H / 4)) * 4, gate = lr / US, u = lr % US;
        st4(Ws + lr * LDW + c, ld4(A.whh + (size_t)(gate * H + sl * US + u) * H + c));
    }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const unsigned base = (unsigned)A.ctl[0] * 64u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const bool own = (int)threadIdx.x < 16 * US;
    const int es = own ? threadIdx.x / US : 0, eu = own ? threadIdx.x % US : 0;
    const int tq = meta[es], nq = own ? meta[16 + es] : 0, gu = sl * US + eu;
    float carry = 0.f;
    u64* xg = A.xch + (size_t)grp * 2 * NS * 16 * H;
    float sv[6];
    auto load_saved = [&](int t) {
        const bool a = own && t >= 0 && t < nq;
        const size_t o = (size_t)(tq + (a ? t : 0)) * H + gu;
        sv[0] = a ? A.dhout[o] : 0.f; sv[1] = a ? A.r[o] : 0.f; sv[2] = a ? A.z[o] : 0.f;
        sv[3] = a ? A.n[o] : 0.f; sv[4] = a ? A.ghn[o] : 0.f; sv[5] = a ? A.hprev[o] : 0.f;
    };
    load_saved(nmax - 1);
    for (int t = nmax - 1; t >= 0; --t) {
        const unsigned tag = base + (unsigned)(nmax - 1 - t) + 1u;
        const int par = (nmax - 1 - t) & 1;
        float* dgl = dgl0 + par * 16 * LDG;
        float keep = 0.f;
        if (own) {
            float dr = 0.f, dz = 0.f, dnr = 0.f;
            if (t < nq) {
                const float dh = sv[0] + carry, rr = sv[1], zz = sv[2], nn = sv[3], gh = sv[4], hp = sv[5];
                const float dn = dh * (1.0f - zz) * (1.0f - nn * nn);
                dz = dh * (hp - nn) * zz * (1.0f - zz);
                dr = dn * gh * rr * (1.0f - rr);
                dnr = dn * rr;
                keep = dh * zz;
                float* gp = A.dgi + (size_t)(tq + t) * 3 * H + gu;
                gp[0] = dr; gp[H] = dz; gp[2 * H] = dn;
                float* hp2 = A.dgh + (size_t)(tq + t) * 3 * H + gu;
                hp2[0] = dr; hp2[H] = dz; hp2[2 * H] = dnr;
            }
            float* row = dgl + es * LDG + eu;
            row[0] = dr; row[US] = dz; row[2 * US] = dnr;
        }
        __syncthreads();
        load_saved(t - 1);
        if (t > 0) {
            // partial[16][H] = dgl[16][KL] . Ws[KL][H]   (B[k = local row][n = column] read column-wise)
            const float* ar = dgl + l16 * LDG + g * (KL / 4);
#pragma unroll
            for (int ci = 0; ci < CTW; ++ci) {
                const int col = (w * CTW + ci) * 16 + l16;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* wc = Ws + (size_t)(g * (KL / 4)) * LDW + col;
#pragma unroll
                for (int c = 0; c < KL / 4; c += 4) {
                    const float4 a = ld4(ar + c);
                    acc = mfma16c(a.x, wc[(c + 0) * LDW], acc); acc = mfma16c(a.y, wc[(c + 1) * LDW], acc);
                    acc = mfma16c(a.z, wc[(c + 2) * LDW], acc); acc = mfma16c(a.w, wc[(c + 3) * LDW], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    put_granule(xg + (((size_t)par * NS + sl) * 16 + 4 * g + r) * H + col, tag, acc[r]);
            }
            if (own) {                                     // reduce-scatter: the 8 partials of this thread's (sequence, unit)
                float pv[NS];
                sweep_granules<NS>(xg + ((size_t)par * NS * 16 + es) * H + gu, 16 * H, tag, pv, A.ctl + 2);
                float s = keep;
#pragma unroll
                for (int src = 0; src < NS; ++src) s += pv[src];
                carry = s;
            }
        }
    }
    finish_launch(A.ctl);
}

template <int H, int NS> size_t coop_lds(bool bwd) {
    constexpr int US = H / NS;
    return sizeof(float) * (3 * US * (H + 4) + (bwd ? 2 * 16 * (3 * US + 4) : (US == 16 ? 2 * 4 * 3 * 16 * US : 16 * (H + 4) + 4 * 3 * 16 * US))) + 32 * sizeof(int);
}

}  // namespace

// Compute units of the CURRENT device (cached): the cooperative recurrence needs every workgroup of the launch resident at once,
// so its budget follows the device — a CPX / DPX partition or a smaller part gets a smaller budget or none.
static int device_cus() {
    static int cus[64];
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!known[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
        known[dev] = true;
    }
    return cus[dev];
}
// Slices per group for a batch (0 = the batch does not qualify for the cooperative path).
//   16 slices (H = 256, at most 16 groups): 79 KB of LDS per workgroup, two fit a CU, so 256 workgroups are resident on a 256-CU
//     device with room to spare (budget: one per CU);
//    8 slices: 140 KB, one per CU — budget three quarters of the CUs, at most 192 workgroups (24 groups).
static int coop_slices(int B, int H) {
    if (getenv("DR4SR_GRU_NOCOOP") || (H != 128 && H != 256)) return 0;
    const int groups = (B + 15) / 16, g8 = ((groups + 7) / 8) * 8, cus = device_cus();
    static const bool no16 = getenv("DR4SR_GRU_NS8") != nullptr;             // cross-check switch: always 8 slices
    if (H == 256 && !no16 && g8 * 16 <= cus) return 16;
    int b8 = cus * 3 / 4;
    if (b8 > 192) b8 = 192;
    return g8 * 8 <= b8 ? 8 : 0;
}

// Sequences per cooperative launch for a batch of B (0 = the batch takes the single-workgroup recurrence).  A batch that does not fit
// one cooperative launch runs as CONSECUTIVE launches over chunks of 256 sequences while that beats the single-workgroup form:
// a launch of either kind lasts (longest sequence) x (time per step), 21 us per step for a single workgroup streaming W_hh from L2
// against ~3 us cooperatively, so up to 6 chunks (B <= 1536) win — B = 512 trains at the speed of B = 256 instead of 2.8x slower.
constexpr int COOP_CHUNK = 256, COOP_MAX_CHUNKS = 6;
static int coop_chunk(int B, int H) {
    if (coop_slices(B, H)) return B;
    static const bool nochunk = getenv("DR4SR_GRU_NOCHUNK") != nullptr;         // cross-check switch
    if (!nochunk && B <= COOP_CHUNK * COOP_MAX_CHUNKS && coop_slices(COOP_CHUNK, H)) return COOP_CHUNK;
    return 0;
}
// granule words needed by the cooperative path for a batch of B sequences (0 = the batch does not qualify)
int64_t gru_coop_words(int B, int H) {
    const int cb = coop_chunk(B, H);
    if (!cb) return 0;
    const int groups = (cb + 15) / 16, ns = coop_slices(cb, H);
    return (int64_t)groups * 2 * ns * 16 * H;
}
extern "C" int dr4sr_gru4rec_uses_cooperative(int32_t B, int32_t H) { return gru_coop_words(B, H) != 0; }

// returns -100 when the batch does not qualify (caller falls back to the single-workgroup recurrence)
int launch_gru_rec_coop(const float* gi, const float* whh, const int* cu, float* r, float* z, float* n, float* ghn, float* hprev,
                        float* hout, const float* dhout, float* dgi, float* dgh, unsigned long long* xch, int* ctl, int B, int H, bool bwd,
                        hipStream_t s) {
    const int cb = coop_chunk(B, H);
    if (!xch || !ctl || cb == 0) return -100;
    if (cb < B) {                                          // consecutive launches over chunks of sequences (same granule area: epochs are per launch)
        for (int c0 = 0; c0 < B; c0 += cb) {
            const int rc = launch_gru_rec_coop(gi, whh, cu + c0, r, z, n, ghn, hprev, hout, dhout, dgi, dgh, xch, ctl, B - c0 < cb ? B - c0 : cb, H, bwd, s);
            if (rc) return rc;
        }
        return 0;
    }
    const int ns = coop_slices(B, H);
    CoopArgs A;
    A.gi = gi; A.whh = whh; A.cu = cu; A.r = r; A.z = z; A.n = n; A.ghn = ghn; A.hprev = hprev; A.hout = hout;
    A.dhout = dhout; A.dgi = dgi; A.dgh = dgh; A.xch = xch; A.ctl = ctl; A.B = B;
    const int groups = (B + 15) / 16;
    dim3 grid(((groups + 7) / 8) * 8 * ns), blk(coop_threads(H, ns));       // groups rounded up to a multiple of 8 (XCD placement); extra blocks exit
    // the runtime must be able to place such workgroups on a CU at all (register / LDS limits of THIS device); asked once per kernel
    auto resident = [&](const void* k, size_t lds, int per_cu) {
        static std::unordered_map<const void*, int> okmap;
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        auto it = okmap.find(k);
        if (it == okmap.end()) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, blk.x, lds) != hipSuccess) nb = 0;
            it = okmap.emplace(k, nb).first;
        }
        return it->second >= per_cu;
    };
#define COOP_LAUNCH(H_, NS_, PER_CU) do { \
        const size_t lds = coop_lds<H_, NS_>(bwd); \
        if (!bwd) { big_lds(k_gru_fwd_coop<H_, NS_>, lds); if (!resident((const void*)k_gru_fwd_coop<H_, NS_>, lds, PER_CU)) return -100; \
                    hipLaunchKernelGGL((k_gru_fwd_coop<H_, NS_>), grid, blk, lds, s, A); } \
        else { big_lds(k_gru_bwd_coop<H_, NS_>, lds); if (!resident((const void*)k_gru_bwd_coop<H_, NS_>, lds, PER_CU)) return -100; \
               hipLaunchKernelGGL((k_gru_bwd_coop<H_, NS_>), grid, blk, lds, s, A); } } while (0)
    // 16 slices: the budget counted one workgroup per CU, but the launch must not depend on a perfectly even placement: require
    // room for two per CU
    if (H == 256 && ns == 16) COOP_LAUNCH(256, 16, 2);
    else if (H == 256) COOP_LAUNCH(256, 8, 1);
    else if (H == 128) COOP_LAUNCH(128, 8, 1);
    else return DR4SR_E_SHAPE;
#undef COOP_LAUNCH
    return DR4SR_LAUNCH_CHECK();
}
