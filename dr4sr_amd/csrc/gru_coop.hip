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

__device__ __forceinline__ void sleep_units(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); }   // n x 64 cycles
// Cheap "probably there" wait of a whole wave on ONE granule (one 8-byte request per pass instead of a sweep's 8 KB), with sleeps: used
// by the follower layer of the wavefront kernels, which has a step of slack.  Only a hint: the tag-checked sweep still follows.
__device__ __forceinline__ void wait_hint(const u64* g, unsigned tag, int* err) {
    unsigned spins = 0;
    for (;;) {
        const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == tag) return;
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 255u) == 0) {
            if (spins >= SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
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

// ------------------------------------------------------------------------------------------------ two layers, one launch (layer wavefront)
// With two GRU layers the step ran four cooperative recurrences one after the other (2 x 131 us forward, 2 x 144 us backward at
// B = 256, max length 50) although layer 2's step t only needs layer 1's step t.  Here ONE launch holds both layers' workgroups
// (16 groups x 16 slices x 2 layers = 512 workgroups of 4 waves, two per CU): the second layer runs behind the first and takes its
// input from the granules layer 1 publishes for its own all-gather, so the chain is max(seqlen) + 2 steps instead of 2 max(seqlen),
// and the token-parallel GEMMs between the layers (gi_2 = h_1 W_ih2^T forward, dh_1 = dgi_2 W_ih2 backward) disappear:
//   forward : layer-2 slice = its 48 rows of W_hh2 AND of W_ih2 as MFMA B operands in REGISTERS; acc = W_ih2 x_t (x_t = h_1[t],
//             polled) + W_hh2 h_2[t-1];
//   backward: layer 2 leads and publishes its dgi_2[t] (three granules per owner thread); the layer-1 slice forms
//             dh_1[t][own 16 units] = dgi_2[t][0..3H) W_ih2[:, own units] (K = 3H split over its 4 waves, B operand in registers)
//             while its own partials of step t + 1 are in flight.
// The follower only READS what the leader wrote, into per-time-step slots (no ring, no back-pressure: 16 H granules per step and
// group forward, 48 H backward), so the leader never waits for the follower and a follower whose workgroups become resident late
// (or after the leader has finished) still completes: no circular wait between the layers.
// Both layers at once are 302 MFLOP per time step: on v_mfma_f32_16x16x4_f32 (157 TF/s) that alone is 1.9 us per step — the
// first version of this kernel (fp32 MFMA) ran at 4.4 us per step, slower than two launches.  The products therefore run on the
// bf16 matrix cores as a 3-term split (x = hi + lo, hi = bf16(x), lo = bf16(x - hi); x w ~ lo hi + hi lo + hi hi with fp32
// accumulation: 3 v_mfma_f32_16x16x32_bf16 per 32 k against 8 fp32 ones, error 2^-16 per product, as k_wgrad_bf): weights split
// once into registers (same 48 VGPRs as fp32), the polled A operands split per step.  Exchange layouts follow that instruction's
// operand shape (a lane owns 8 consecutive k of one sequence): granule (k, seq) of a step sits at
// ((k / 32) 8 + k % 8) 64 + ((k / 8) % 4) 16 + seq, so the 64 lanes of one poll instruction read 512 contiguous bytes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& lo) {     // x: 8 values in registers (constant indices)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const __bf16 h = (__bf16)x[i]; hi[i] = h; lo[i] = (__bf16)(x[i] - (float)h); }
}
__device__ __forceinline__ f32x4 mfma_x3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
    c = mfma_bf(al, bh, c); c = mfma_bf(ah, bl, c); return mfma_bf(ah, bh, c);
}
// Exchanged activations travel ALREADY SPLIT: the granule's 32-bit payload is bf16 hi | bf16 lo << 16 of the value (the producer splits
// once; every consumer — 16 slices — would otherwise spend three VALU instructions per polled value on it)
__device__ __forceinline__ unsigned pack_split(float x) {
    const __bf16 h = (__bf16)x, l = (__bf16)(x - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// eight payloads -> the hi and lo bf16x8 operands (v_perm_b32 per pair)
__device__ __forceinline__ void unpack8(const unsigned* v, bf16x8& hi, bf16x8& lo) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __builtin_amdgcn_perm(v[2 * i + 1], v[2 * i], 0x05040100u);     // lo halves of the two words
        l[i] = __builtin_amdgcn_perm(v[2 * i + 1], v[2 * i], 0x07060302u);     // hi halves
    }
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ void put_granule_u(u64* g, unsigned tag, unsigned payload) {
    __hip_atomic_store(g, ((u64)tag << 32) | (u64)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sweep of N payloads (as sweep_granules, raw 32-bit payloads)
template <int N>
__device__ __forceinline__ void sweep_payloads(const u64* g, int stride, unsigned tag, unsigned (&v)[N], int* err) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u64 x = __hip_atomic_load(g + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = (unsigned)x;
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
// ... keeping the granules themselves (the payload is the low word: no second register per value)
template <int N>
__device__ __forceinline__ void sweep_u64(const u64* g, int stride, unsigned tag, u64 (&x)[N], int* err) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            x[k] = __hip_atomic_load(g + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok &= (unsigned)(x[k] >> 32) == tag;
        }
        if (ok) return;
        if ((++spins & 255u) == 0) {
            if (spins >= SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
// two sets (own strides and tags) in ONE pass: set A is re-read only until it is complete
template <int NA, int NB>
__device__ __forceinline__ void sweep_u64_2(const u64* ga, int stridea, unsigned taga, u64 (&xa)[NA], const u64* gb, int strideb, unsigned tagb, u64 (&xb)[NB],
                                            int* err) {
    unsigned spins = 0;
    bool adone = false;
    for (;;) {
        bool oka = true, okb = true;
        if (!adone) {
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                xa[k] = __hip_atomic_load(ga + (size_t)k * stridea, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                oka &= (unsigned)(xa[k] >> 32) == taga;
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            xb[k] = __hip_atomic_load(gb + (size_t)k * strideb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            okb &= (unsigned)(xb[k] >> 32) == tagb;
        }
        adone = adone || __all(oka);
        if (adone && okb) return;
        if ((++spins & 255u) == 0) {
            if (spins >= SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
__device__ __forceinline__ void unpack8(const u64* v, bf16x8& hi, bf16x8& lo) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __builtin_amdgcn_perm((unsigned)v[2 * i + 1], (unsigned)v[2 * i], 0x05040100u);
        l[i] = __builtin_amdgcn_perm((unsigned)v[2 * i + 1], (unsigned)v[2 * i], 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
// two sets in ONE pass (their round trips overlap): set A is re-read only until it is complete
template <int N>
__device__ __forceinline__ void sweep_payloads2(const u64* ga, unsigned taga, unsigned (&va)[N], const u64* gb, unsigned tagb, unsigned (&vb)[N],
                                                int stride, int* err) {
    unsigned spins = 0;
    bool adone = false;
    for (;;) {
        bool oka = true, okb = true;
        if (!adone) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const u64 x = __hip_atomic_load(ga + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                va[k] = (unsigned)x;
                oka &= (unsigned)(x >> 32) == taga;
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u64 x = __hip_atomic_load(gb + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vb[k] = (unsigned)x;
            okb &= (unsigned)(x >> 32) == tagb;
        }
        adone = adone || __all(oka);
        if (adone && okb) return;
        if ((++spins & 255u) == 0) {
            if (spins >= SPIN_LIMIT) atomicExch(err, 1);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
__device__ __forceinline__ size_t gran_idx(int k, int seq) { return (size_t)(((k >> 5) * 8 + (k & 7)) * 64 + ((k >> 3) & 3) * 16 + seq); }

struct WaveArgs {
    const float* gi1;                                     // layer-1 input projection [T][3H]
    const float* whh[2]; const float* wih2;
    const int* cu;
    float* r[2]; float* z[2]; float* n[2]; float* ghn[2]; float* hprev[2]; float* hout[2];
    const float* dhout; float* dgi[2]; float* dgh[2];     // backward: dL/d(top output), gate gradients
    u64* xch; int* ctl; int B; int L;
    int presleep, solo, stamp;                                   // tuning / diagnosis (DR4SR_GRU_WAVE_PRESLEEP, DR4SR_GRU_WAVE_SOLO)
    int order;                                                   // block index -> role order (wave_who)
};
// granules per group: per-step slots [L][48 H] (forward: h_1[t] in the first 16 H; backward: dgi_2[t], all 3H gate rows) | forward h_2
// ring [2][16 H] | backward: layer-2 partial ring [2][NS][16][H] | layer-1 partial ring [2][NS][16][H]
template <int H, int NS> struct WaveArea {
    static constexpr size_t SLOT = (size_t)16 * 3 * H, PR = (size_t)NS * 16 * H;
    static constexpr size_t ring_f(int L) { return (size_t)L * SLOT; }
    static constexpr size_t ring_b(int L, int layer) { return ring_f(L) + 2 * 16 * H + (size_t)(1 - layer) * 2 * PR; }   // layer 1 (the leader) first
    static constexpr size_t words(int L) { return ring_f(L) + 2 * 16 * H + 4 * PR; }
};
template <int H, int NS> constexpr size_t wave_group_words(int L) { return WaveArea<H, NS>::words(L); }

// role / group / slice of a workgroup: 8 consecutive blocks = 8 XCDs (speed-only placement, as above).  Blocks jx and jx + 32 of an XCD
// land on the same CU (32 CUs per XCD, round-robin): roles alternate with jx / NS so that a CU holds one leader and one follower
// slice (the follower issues twice the matrix work of the leader)
// DR4SR_GRU_WAVE_STAMP (tools/gru_stamp_probe.py): shader-clock stamps of one middle time step of group 0, slice 0, wave 0, into the
// control block's spare words: ctl[8 + 16 role + i]
#define WAVE_STAMP(i) do { if (A.stamp && st_on) { if (lane == 0) A.ctl[8 + 16 * who.role + (i)] = (int)__builtin_amdgcn_s_memtime(); } } while (0)
struct WaveWho { int grp, sl, role; };
template <int NS> __device__ __forceinline__ WaveWho wave_who(const int order) {
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, q = jx / NS;         // q: 0..3 per 16 groups = (role, group half)
    WaveWho w;
    w.sl = jx % NS;
    // 16 groups: q = 0, 1, 2, 3 -> leader (groups 0..7), leader (8..15), follower (8..15), follower (0..7); 8 groups: leader, follower.
    // EVERY leader has a lower block index than its follower (workgroups are dispatched in index order, so a follower never holds a
    // slot its leader still waits for — ADVICE r3; round 3's order leader, follower, follower, leader had the second half's followers
    // first), and blocks jx / jx + 32 still pair a leader with a follower of the OTHER group half on a CU: a leader sharing its CU with
    // its own follower (order leader, leader, follower(0..7), follower(8..15)) measured 4 % slower — the two wait for each other, so
    // their phases never overlap (0.505 against 0.484 ms per GRU4Rec step)
    const int halves = (int)gridDim.x / (2 * 8 * NS);         // groups of the launch / 8 (grid = 2 roles x 8 halves groups x NS slices)
    if (order == 0 && halves == 2) {                       // round 3's order: leader, follower, follower, leader
        w.grp = ((q >> 1) & 1) * 8 + xcd;
        w.role = (q ^ (q >> 1)) & 1;
        return w;
    }
    w.role = q >= halves;
    w.grp = (w.role ? (order == 2 ? q - halves : 2 * halves - 1 - q) : q) * 8 + xcd;      // order 2: leader, leader, follower(0..7), follower(8..15)
    return w;
}

template <int H, int NS>
__global__ __launch_bounds__(256, 2) void k_gru_fwd_wave(const WaveArgs A) {
    constexpr int US = H / NS;
    static_assert(US == 16 && H == 256, "one 16-unit tile per slice, four K-quarter waves of two 32-k MFMAs");
    float* part = smem;                                   // [2 parity][4 kq][4][16 unit][16 seq]
    int* meta = reinterpret_cast<int*>(part + 2 * 4 * 4 * 16 * 16);
    const WaveWho who = wave_who<NS>(A.order);
    const int grp = who.grp, sl = who.sl, b0 = grp * 16, layer = who.role;       // forward: layer 1 leads
    if (b0 >= A.B || (A.solo == 1 && who.role)) { finish_launch(A.ctl); return; }
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    if (A.stamp && threadIdx.x == 0 && blockIdx.x < 24) {   // where the dispatcher put the first 24 blocks: XCC id (HW_REG_XCC_ID = 20), CU / SE id (HW_REG_HW_ID = 4)
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        A.ctl[40 + blockIdx.x] = (int)((xcc & 0xf) | (hw << 4));
    }
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    // B operands: lane (l16, g) of K-quarter wave kq holds W[gate row of unit l16][kq 64 + 32 m + 8 g + j], j = 0..7, split hi | lo
    bf16x8 whh[3][2], whl[3][2], wih[3][2], wil[3][2];
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const size_t o = (size_t)(q3 * H + sl * US + l16) * H + kq * 64 + m * 32 + 8 * g;
            float x[8];
            { const float4 a = ld4(A.whh[layer] + o), b = ld4(A.whh[layer] + o + 4); x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w; }
            split8(x, whh[q3][m], whl[q3][m]);
            if (layer) { const float4 a = ld4(A.wih2 + o), b = ld4(A.wih2 + o + 4); x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w; }
            split8(x, wih[q3][m], wil[q3][m]);
        }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const unsigned base = (unsigned)A.ctl[0] * 64u;
    const int es = threadIdx.x & 15, eu = threadIdx.x >> 4;                       // gate math: (sequence, unit of the slice); sequences fastest:
    const int tq = meta[es], nq = meta[16 + es], gu = sl * US + eu;               // a wave's granule store is four full 128-byte lines
    u64* xg = A.xch + (size_t)grp * wave_group_words<H, NS>(A.L);
    u64* slots = xg;                                       // h_1[t]: slot t
    u64* ring = xg + WaveArea<H, NS>::ring_f(A.L);         // h_2[t]: [2][16 H]
    const size_t lane_off = (size_t)kq * 1024 + lane;      // + 64 (8 m + j): k = kq 64 + 32 m + 8 g + j of sequence l16
    const size_t own_off = gran_idx(gu, es);
    constexpr size_t SLOT = WaveArea<H, NS>::SLOT;
    float* const R = A.r[layer]; float* const Z = A.z[layer]; float* const N = A.n[layer]; float* const G = A.ghn[layer];
    float* const HP = A.hprev[layer]; float* const HO = A.hout[layer];
    float hown = 0.f;
    unsigned hv[16];
    if (layer == 0) {
        for (int t = 0; t < nmax; ++t) {
            const bool act = t < nq;
            float gir = 0.f, giz = 0.f, gin = 0.f;
            if (act) { const float* gip = A.gi1 + (size_t)(tq + t) * 3 * H + gu; gir = gip[0]; giz = gip[H]; gin = gip[2 * H]; }
            f32x4 acc[3];
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) acc[q3] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bool st_on = grp == 0 && sl == 0 && kq == 0 && t == nmax / 2;
            WAVE_STAMP(0);
            if (t > 0) {
                if (A.presleep) sleep_units(A.presleep);
                sweep_payloads<16>(slots + (size_t)(t - 1) * SLOT + lane_off, 64, base + t, hv, A.ctl + 2);
                WAVE_STAMP(1);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf16x8 ah, al;
                    unpack8(hv + 8 * m, ah, al);
#pragma unroll
                    for (int q3 = 0; q3 < 3; ++q3) acc[q3] = mfma_x3(ah, al, whh[q3][m], whl[q3][m], acc[q3]);
                }
            }
            float* pt = part + (t & 1) * (4 * 4 * 256);
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) *reinterpret_cast<f32x4*>(pt + ((kq * 4 + q3) * 16 + l16) * 16 + 4 * g) = acc[q3];
            WAVE_STAMP(2);
            __syncthreads();
            WAVE_STAMP(3);
            float gh[3];
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) {
                const float* pp = pt + (q3 * 16 + eu) * 16 + es;
                gh[q3] = (pp[0] + pp[4 * 256]) + (pp[2 * 4 * 256] + pp[3 * 4 * 256]);
            }
            float rr = 0.f, zz = 0.f, nn = 0.f;
            const float hold = hown;
            if (act) {
                rr = sigm(gir + gh[0]); zz = sigm(giz + gh[1]); nn = tanh_f(gin + rr * gh[2]);
                hown = (1.0f - zz) * nn + zz * hold;
            }
            // the exchange first (the other slices wait for it), the saved tensors of the backward after it
            put_granule_u(slots + (size_t)t * SLOT + own_off, base + t + 1, pack_split(hown));      // also the last step: layer 2 reads it
            WAVE_STAMP(4);
            if (act) {
                const size_t o = (size_t)(tq + t) * H + gu;
                R[o] = rr; Z[o] = zz; N[o] = nn; G[o] = gh[2]; HP[o] = hold; HO[o] = hown;
            }
            WAVE_STAMP(5);
        }
    } else {
        unsigned xv[16];
        for (int t = 0; t < nmax; ++t) {
            const bool act = t < nq;
            const bool st_on = grp == 0 && sl == 0 && kq == 0 && t == nmax / 2;
            WAVE_STAMP(0);
            // x_t = h_1[t] (normally long there: layer 1 does not wait for this layer) and h_2[t-1] (what this step waits for) are polled
            // in ONE pass, so their round trips overlap
            if (t > 0) {
                if (A.presleep) sleep_units(A.presleep);
                sweep_payloads2<16>(slots + (size_t)t * SLOT + lane_off, base + t + 1, xv, ring + (size_t)((t - 1) & 1) * 16 * H + lane_off, base + t, hv,
                                    64, A.ctl + 2);
            } else sweep_payloads<16>(slots + (size_t)t * SLOT + lane_off, 64, base + t + 1, xv, A.ctl + 2);
            WAVE_STAMP(1);
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 ah, al;
                unpack8(xv + 8 * m, ah, al);
                acc[0] = mfma_x3(ah, al, wih[0][m], wil[0][m], acc[0]);
                acc[1] = mfma_x3(ah, al, wih[1][m], wil[1][m], acc[1]);
                acc[3] = mfma_x3(ah, al, wih[2][m], wil[2][m], acc[3]);
            }
            if (t > 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf16x8 ah, al;
                    unpack8(hv + 8 * m, ah, al);
#pragma unroll
                    for (int q3 = 0; q3 < 3; ++q3) acc[q3] = mfma_x3(ah, al, whh[q3][m], whl[q3][m], acc[q3]);
                }
            }
            float* pt = part + (t & 1) * (4 * 4 * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(pt + ((kq * 4 + q) * 16 + l16) * 16 + 4 * g) = acc[q];
            WAVE_STAMP(2);
            __syncthreads();
            WAVE_STAMP(3);
            float gs[4];                                   // r and z pre-activations (input + recurrent), gh_n, gi_n
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* pp = pt + (q * 16 + eu) * 16 + es;
                gs[q] = (pp[0] + pp[4 * 256]) + (pp[2 * 4 * 256] + pp[3 * 4 * 256]);
            }
            float rr = 0.f, zz = 0.f, nn = 0.f;
            const float hold = hown;
            if (act) {
                rr = sigm(gs[0]); zz = sigm(gs[1]); nn = tanh_f(gs[3] + rr * gs[2]);
                hown = (1.0f - zz) * nn + zz * hold;
            }
            if (t + 1 < nmax) put_granule_u(ring + (size_t)(t & 1) * 16 * H + own_off, base + t + 1, pack_split(hown));
            WAVE_STAMP(4);
            if (act) {
                const size_t o = (size_t)(tq + t) * H + gu;
                R[o] = rr; Z[o] = zz; N[o] = nn; G[o] = gs[2]; HP[o] = hold; HO[o] = hown;
            }
            WAVE_STAMP(5);
        }
    }
    finish_launch(A.ctl);
}

// Single-layer forward of the 16-slice form on the bf16 matrix cores (= the layer-1 half of k_gru_fwd_wave with a two-entry ring instead
// of per-step slots): plans the wavefront does not take (one, three or four layers; DR4SR_GRU_NOWAVE).  k_gru_fwd_coop<256, 16> is the
// fp32 form (DR4SR_GRU_FWD_F32).
template <int H, int NS>
__global__ __launch_bounds__(256) void k_gru_fwd_coop_bf(const CoopArgs A) {
    constexpr int US = H / NS;
    static_assert(US == 16 && H == 256, "one 16-unit tile per slice, four K-quarter waves of two 32-k MFMAs");
    float* part = smem;                                   // [2 parity][4 kq][3][16 unit][16 seq]
    int* meta = reinterpret_cast<int*>(part + 2 * 4 * 3 * 256);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int grp = (jx / NS) * 8 + xcd, sl = jx % NS, b0 = grp * 16;
    if (b0 >= A.B) { finish_launch(A.ctl); return; }
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    bf16x8 whh[3][2], whl[3][2];                          // W_hh[gate row of unit l16][kq 64 + 32 m + 8 g + j], split hi | lo
#pragma unroll
    for (int q3 = 0; q3 < 3; ++q3)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const size_t o = (size_t)(q3 * H + sl * US + l16) * H + kq * 64 + m * 32 + 8 * g;
            float x[8];
            const float4 a = ld4(A.whh + o), b = ld4(A.whh + o + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
            split8(x, whh[q3][m], whl[q3][m]);
        }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const unsigned base = (unsigned)A.ctl[0] * 64u;
    const int es = threadIdx.x & 15, eu = threadIdx.x >> 4;
    const int tq = meta[es], nq = meta[16 + es], gu = sl * US + eu;
    u64* ring = A.xch + (size_t)grp * 2 * 16 * H;         // h[t]: [2][16 H], granule (k, seq) at gran_idx
    const size_t lane_off = (size_t)kq * 1024 + lane, own_off = gran_idx(gu, es);
    float hown = 0.f;
    unsigned hv[16];
    for (int t = 0; t < nmax; ++t) {
        const bool act = t < nq;
        float gir = 0.f, giz = 0.f, gin = 0.f;
        if (act) { const float* gip = A.gi + (size_t)(tq + t) * 3 * H + gu; gir = gip[0]; giz = gip[H]; gin = gip[2 * H]; }
        f32x4 acc[3];
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) acc[q3] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            sweep_payloads<16>(ring + (size_t)((t - 1) & 1) * 16 * H + lane_off, 64, base + t, hv, A.ctl + 2);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                bf16x8 ah, al;
                unpack8(hv + 8 * m, ah, al);
#pragma unroll
                for (int q3 = 0; q3 < 3; ++q3) acc[q3] = mfma_x3(ah, al, whh[q3][m], whl[q3][m], acc[q3]);
            }
        }
        float* pt = part + (t & 1) * (4 * 3 * 256);
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) *reinterpret_cast<f32x4*>(pt + ((kq * 3 + q3) * 16 + l16) * 16 + 4 * g) = acc[q3];
        __syncthreads();
        float gh[3];
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
            const float* pp = pt + (q3 * 16 + eu) * 16 + es;
            gh[q3] = (pp[0] + pp[3 * 256]) + (pp[2 * 3 * 256] + pp[3 * 3 * 256]);
        }
        float rr = 0.f, zz = 0.f, nn = 0.f;
        const float hold = hown;
        if (act) {
            rr = sigm(gir + gh[0]); zz = sigm(giz + gh[1]); nn = tanh_f(gin + rr * gh[2]);
            hown = (1.0f - zz) * nn + zz * hold;
        }
        if (t + 1 < nmax) put_granule_u(ring + (size_t)(t & 1) * 16 * H + own_off, base + t + 1, pack_split(hown));     // the exchange first
        if (act) {
            const size_t o = (size_t)(tq + t) * H + gu;
            A.r[o] = rr; A.z[o] = zz; A.n[o] = nn; A.ghn[o] = gh[2]; A.hprev[o] = hold; A.hout[o] = hown;
        }
    }
    finish_launch(A.ctl);
}

// Single-layer BPTT of the 16-slice form on the bf16 matrix cores: k_gru_bwd_coop with the slice's W_hh rows as split (hi | lo) MFMA B
// operands in REGISTERS instead of an fp32 LDS image read column-wise (48 ds_read_b32 per lane and step on the recurrent chain) and the
// partial product as 3-term bf16 (24 MFMAs of 8 passes instead of 48).  Same exchange, same granules, same ring.
template <int H, int NS>
__global__ __launch_bounds__(256) void k_gru_bwd_coop_bf(const CoopArgs A) {
    constexpr int US = H / NS, KL = 3 * US, LDG = KL + 4, CTW = (H / 16) / 4;
    static_assert(US == 16 && CTW == 4 && H == 256, "slice geometry");
    float* dgl0 = smem;                                   // [2 parity][16][LDG]   dgh tile of this slice (A operand)
    int* meta = reinterpret_cast<int*>(dgl0 + 2 * 16 * LDG);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int grp = (jx / NS) * 8 + xcd, sl = jx % NS, b0 = grp * 16;
    if (b0 >= A.B) { finish_launch(A.ctl); return; }
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    bf16x8 wbh[CTW][2], wbl[CTW][2];                      // local rows 32 m + 8 g + j (zero beyond KL) of column (4 w + ci) 16 + l16
#pragma unroll
    for (int ci = 0; ci < CTW; ++ci)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lr = 32 * m + 8 * g + j, gate = lr / US, u = lr % US;
                x[j] = lr < KL ? A.whh[(size_t)(gate * H + sl * US + u) * H + (w * CTW + ci) * 16 + l16] : 0.f;
            }
            split8(x, wbh[ci][m], wbl[ci][m]);
        }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const unsigned base = (unsigned)A.ctl[0] * 64u;
    const int es = threadIdx.x / US, eu = threadIdx.x % US;
    const int tq = meta[es], nq = meta[16 + es], gu = sl * US + eu;
    float carry = 0.f;
    u64* xg = A.xch + (size_t)grp * 2 * NS * 16 * H;
    float sv[6];
    auto load_saved = [&](int t) {
        const bool a = t >= 0 && t < nq;
        const size_t o = (size_t)(tq + (a ? t : 0)) * H + gu;
        sv[0] = a ? A.dhout[o] : 0.f; sv[1] = a ? A.r[o] : 0.f; sv[2] = a ? A.z[o] : 0.f;
        sv[3] = a ? A.n[o] : 0.f; sv[4] = a ? A.ghn[o] : 0.f; sv[5] = a ? A.hprev[o] : 0.f;
    };
    load_saved(nmax - 1);
    for (int t = nmax - 1; t >= 0; --t) {
        const unsigned tag = base + (unsigned)(nmax - 1 - t) + 1u;
        const int par = (nmax - 1 - t) & 1;
        float* dgl = dgl0 + par * 16 * LDG;
        float keep = 0.f, dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f;
        if (t < nq) {
            const float dh = sv[0] + carry, rr = sv[1], zz = sv[2], nn = sv[3], gh = sv[4], hp = sv[5];
            dn = dh * (1.0f - zz) * (1.0f - nn * nn);
            dz = dh * (hp - nn) * zz * (1.0f - zz);
            dr = dn * gh * rr * (1.0f - rr);
            dnr = dn * rr;
            keep = dh * zz;
        }
        {
            float* row = dgl + es * LDG + eu;
            row[0] = dr; row[US] = dz; row[2 * US] = dnr;
        }
        __syncthreads();
        if (t > 0) {
            // partial[16][H] = dgl[16][KL] . W_hh[slice rows][:], the exchange first, the saved gate gradients after it
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float x[8];
                const int k0 = 32 * m + 8 * g;
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = a;
                if (k0 < KL) { a = *reinterpret_cast<const f32x4*>(dgl + l16 * LDG + k0); b = *reinterpret_cast<const f32x4*>(dgl + l16 * LDG + k0 + 4); }
                x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
                split8(x, ah[m], al[m]);
            }
#pragma unroll
            for (int ci = 0; ci < CTW; ++ci) {
                const int col = (w * CTW + ci) * 16 + l16;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 2; ++m) acc = mfma_x3(ah[m], al[m], wbh[ci][m], wbl[ci][m], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) put_granule(xg + (((size_t)par * NS + sl) * 16 + 4 * g + r) * H + col, tag, acc[r]);
            }
        }
        if (t < nq) {
            float* gp = A.dgi + (size_t)(tq + t) * 3 * H + gu;
            gp[0] = dr; gp[H] = dz; gp[2 * H] = dn;
            float* hp2 = A.dgh + (size_t)(tq + t) * 3 * H + gu;
            hp2[0] = dr; hp2[H] = dz; hp2[2 * H] = dnr;
        }
        load_saved(t - 1);
        if (t > 0) {
            float pv[NS];
            sweep_granules<NS>(xg + ((size_t)par * NS * 16 + es) * H + gu, 16 * H, tag, pv, A.ctl + 2);
            float s = keep;
#pragma unroll
            for (int src = 0; src < NS; ++src) s += pv[src];
            carry = s;
        }
    }
    finish_launch(A.ctl);
}

// Backward: ONE workgroup of 8 waves per CU holds a slice of BOTH layers — waves 0..3 the slice of layer 2 (the leader: the cooperative
// BPTT above with its W_hh2 operands in registers; its owner threads also publish their three dgi_2[t] values, split as the forward's
// h, into the slot of step t), waves 4..7 the same slice of layer 1 (the follower).  dL/dh_1[t] never exists in memory: the follower
// forms its 16 columns dgi_2[t][0..3H) W_ih2[:, own units] from 48 polled granules per lane (K = 3H over its 4 waves) while its own
// partials of step t + 1 are in flight.  That needs the registers for the sweep, so BOTH of the follower's B operands are LDS images
// (W_hh1 64 KB padded, W_ih2 48 KB) — which is why the two slices share a workgroup: as two workgroups per CU (the first three forms,
// NOTEBOOK) the kernel-wide LDS size would have to hold the images twice.  The halves never wait for each other inside the loop:
// each has its own 4-wave barrier (an LDS counter; s_barrier is workgroup-wide on gfx950).
// The slice's 48 local rows (K of the recurrent partial product) are padded to 64 = two 32-k MFMAs (v_mfma_f32_16x16x16_bf16 for
// the last 16 rows is miscompiled behind a 32-k MFMA on gfx950: tools/probes/mfma16_layout_probe.hip).
__device__ __forceinline__ void group_barrier(int* cnt, int& target) {          // the 4 waves that share `cnt`
    target += 4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int H, int NS>
__global__ __launch_bounds__(512) void k_gru_bwd_pair(const WaveArgs A) {
    constexpr int US = H / NS, KL = 3 * US, LDG = KL + 4, CTW = (H / 16) / 4, NM = 3 * H / 4 / 32;      // NM: 32-k MFMAs of a wave's quarter of K = 3H
    static_assert(US == 16 && CTW == 4 && H == 256, "slice geometry");
    using Area = WaveArea<H, NS>;
    float* dglb = smem;                                   // [2 roles][2 parity][16][LDG]   dgh tiles (A operands)
    float* part2 = dglb + 2 * 2 * 16 * LDG;               // [2 parity][4 waves][16][US]   layer 1: K-quarter partials of dh_1
    int* meta = reinterpret_cast<int*>(part2 + 2 * 4 * 16 * US);
    int* bar = meta + 32;                                 // [2] the halves' barrier counters
    bf16x8* wimg = reinterpret_cast<bf16x8*>(bar + 4);    // layer 1: W_ih2 operands [m][hi | lo][256 lanes]
    bf16x8* wbimg = wimg + 2 * NM * 256;                  // layer 1: W_hh1 operands [ci][m][hi | lo][256 lanes]
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int grp = (jx / NS) * 8 + xcd, sl = jx % NS, b0 = grp * 16;
    const int role = threadIdx.x >> 8, layer = 1 - role, tid = threadIdx.x & 255;      // waves 0..3: layer 2 (leads)
    if (b0 >= A.B) { finish_launch(A.ctl); return; }
    if (threadIdx.x < 16) {
        const int b = b0 + threadIdx.x;
        meta[threadIdx.x] = b < A.B ? A.cu[b] : 0;
        meta[16 + threadIdx.x] = b < A.B ? A.cu[b + 1] - A.cu[b] : 0;
    }
    if (threadIdx.x < 2) bar[threadIdx.x] = 0;
    const int lane = tid & 63, w = tid >> 6, l16 = lane & 15, g = lane >> 4;
    // recurrent B operand: partial[16][col] = dgl[16][KL] . W_hh[slice rows][col]: lane (l16, g) holds local rows 32 m + 8 g + j (zero
    // beyond KL) of column (4 w + ci) 16 + l16
    bf16x8 wbh[CTW][2], wbl[CTW][2];
#pragma unroll
    for (int ci = 0; ci < CTW; ++ci)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lr = 32 * m + 8 * g + j, gate = lr / US, u = lr % US;
                x[j] = lr < KL ? A.whh[layer][(size_t)(gate * H + sl * US + u) * H + (w * CTW + ci) * 16 + l16] : 0.f;
            }
            split8(x, wbh[ci][m], wbl[ci][m]);
            if (role == 1) {
                wbimg[((ci * 2 + m) * 2) * 256 + tid] = wbh[ci][m];
                wbimg[((ci * 2 + m) * 2 + 1) * 256 + tid] = wbl[ci][m];
            }
        }
    if (role == 1) {
        // input-gradient B operand: dh_1[16][own units] = dgi_2[16][3H] . W_ih2[3H][own units], K split over the 4 waves:
        // lane (l16, g) of wave w holds rows w 3H/4 + 32 m + 8 g + j of column sl US + l16
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = A.wih2[(size_t)(w * (3 * H / 4) + 32 * m + 8 * g + j) * H + sl * US + l16];
            bf16x8 xh, xl;
            split8(x, xh, xl);
            wimg[(2 * m) * 256 + tid] = xh;
            wimg[(2 * m + 1) * 256 + tid] = xl;
        }
    }
    __syncthreads();
    int nmax = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) nmax = max(nmax, meta[16 + s]);
    const unsigned base = (unsigned)A.ctl[0] * 64u;
    const int es = tid / US, eu = tid % US;
    const int tq = meta[es], nq = meta[16 + es], gu = sl * US + eu;
    u64* xg = A.xch + (size_t)grp * Area::words(A.L);
    u64* slots = xg;                                       // dgi_2[t]: granule (k, seq), k in [0, 3H)
    u64* ring = xg + Area::ring_b(A.L, layer);             // this layer's partial ring [2][NS][16][H]
    float* dgl0 = dglb + role * 2 * 16 * LDG;
    int* mybar = bar + role;
    int btarget = 0;
    const float* const Rr = A.r[layer]; const float* const Zz = A.z[layer]; const float* const Nn = A.n[layer];
    const float* const Gh = A.ghn[layer]; const float* const Hp = A.hprev[layer];
    float* const DGI = A.dgi[layer]; float* const DGH = A.dgh[layer];
    float carry = 0.f, keep = 0.f;
    float sv[6];
    auto load_saved = [&](int t) {
        const bool a = t >= 0 && t < nq;
        const size_t o = (size_t)(tq + (a ? t : 0)) * H + gu;
        sv[0] = (a && layer == 1) ? A.dhout[o] : 0.f; sv[1] = a ? Rr[o] : 0.f; sv[2] = a ? Zz[o] : 0.f;
        sv[3] = a ? Nn[o] : 0.f; sv[4] = a ? Gh[o] : 0.f; sv[5] = a ? Hp[o] : 0.f;
    };
    // one BPTT step's gate derivatives for this thread's (sequence, unit); returns dh z (the part of the carry that needs no exchange)
    auto gates = [&](int t, float dh, float* dgl, float& dr, float& dz, float& dn) -> float {
        float dnr = 0.f, kp = 0.f;
        dr = 0.f; dz = 0.f; dn = 0.f;
        if (t < nq) {
            const float rr = sv[1], zz = sv[2], nn = sv[3], gh = sv[4], hp = sv[5];
            dn = dh * (1.0f - zz) * (1.0f - nn * nn);
            dz = dh * (hp - nn) * zz * (1.0f - zz);
            dr = dn * gh * rr * (1.0f - rr);
            dnr = dn * rr;
            kp = dh * zz;
            float* gp = DGI + (size_t)(tq + t) * 3 * H + gu;
            gp[0] = dr; gp[H] = dz; gp[2 * H] = dn;
            float* hp2 = DGH + (size_t)(tq + t) * 3 * H + gu;
            hp2[0] = dr; hp2[H] = dz; hp2[2 * H] = dnr;
        }
        float* row = dgl + es * LDG + eu;
        row[0] = dr; row[US] = dz; row[2 * US] = dnr;
        return kp;
    };
    // partial[16][H] of this slice's dgh tile -> granules of ring[par]; LDSB: the B operands come from the LDS image (layer 1)
    auto rec_partials = [&](const float* dgl, int par, unsigned tag, const bool ldsb) {
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float x[8];
            const int k0 = 32 * m + 8 * g;                  // A operand: dgl[sequence l16][k0 .. k0 + 8), zero beyond KL
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = a;
            if (k0 < KL) { a = *reinterpret_cast<const f32x4*>(dgl + l16 * LDG + k0); b = *reinterpret_cast<const f32x4*>(dgl + l16 * LDG + k0 + 4); }
            x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
            split8(x, ah[m], al[m]);
        }
#pragma unroll
        for (int ci = 0; ci < CTW; ++ci) {
            const int col = (w * CTW + ci) * 16 + l16;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (ldsb) acc = mfma_x3(ah[m], al[m], wbimg[((ci * 2 + m) * 2) * 256 + tid], wbimg[((ci * 2 + m) * 2 + 1) * 256 + tid], acc);
                else acc = mfma_x3(ah[m], al[m], wbh[ci][m], wbl[ci][m], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) put_granule(ring + (((size_t)par * NS + sl) * 16 + 4 * g + r) * H + col, tag, acc[r]);
        }
    };
    auto sweep_partials = [&](int par, unsigned tag) -> float {
        float pv[NS];
        if (A.presleep) sleep_units(A.presleep);
        sweep_granules<NS>(ring + ((size_t)par * NS * 16 + es) * H + gu, 16 * H, tag, pv, A.ctl + 2);
        float s = 0.f;
#pragma unroll
        for (int src = 0; src < NS; ++src) s += pv[src];
        return s;
    };
#define PAIR_STAMP(i) do { if (A.stamp && st_on) { if (lane == 0) A.ctl[8 + 16 * role + (i)] = (int)__builtin_amdgcn_s_memtime(); } } while (0)
    if (role == 0) {
        load_saved(nmax - 1);
        for (int t = nmax - 1; t >= 0; --t) {
            const int idx = nmax - 1 - t, par = idx & 1;
            const unsigned tag = base + (unsigned)idx + 1u;
            float* dgl = dgl0 + par * 16 * LDG;
            const bool st_on = grp == 0 && sl == 0 && w == 0 && t == nmax / 2;
            PAIR_STAMP(0);
            float dr, dz, dn;
            keep = gates(t, sv[0] + carry, dgl, dr, dz, dn);
            PAIR_STAMP(1);
            group_barrier(mybar, btarget);
            PAIR_STAMP(2);
            load_saved(t - 1);
            if (t > 0) rec_partials(dgl, par, tag, false);  // the recurrent chain first
            PAIR_STAMP(3);
            {   // dgi_2[t] of this (sequence, unit), all three gates (zeros beyond the sequence's length): what layer 1 contracts with W_ih2
                u64* sl_t = slots + (size_t)t * Area::SLOT;
                put_granule_u(sl_t + gran_idx(gu, es), tag, pack_split(dr));
                put_granule_u(sl_t + gran_idx(H + gu, es), tag, pack_split(dz));
                put_granule_u(sl_t + gran_idx(2 * H + gu, es), tag, pack_split(dn));
            }
            if (t > 0) carry = keep + sweep_partials(par, tag);
            PAIR_STAMP(4);
        }
    } else {
        const size_t lane_off = (size_t)w * NM * 8 * 64 + lane;                 // + 64 (8 m + j): K index w 3H/4 + 32 m + 8 g + j of sequence l16
        // dh_1[t] partials of this wave's K quarter from ONE sweep of 48 granules
        auto input_part = [&](int t) {
            u64 xv[NM * 8];
            {   // one optimistic pass; when layer 2 has not published the step yet, wait on ONE granule with sleeps (a spinning 48-granule
                // sweep on every lane starves the leader half on the same SIMDs), then read again
                const u64* gs = slots + (size_t)t * Area::SLOT + lane_off;
                const unsigned tg = base + (unsigned)(nmax - 1 - t) + 1u;
                for (int pass = 0;; ++pass) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < NM * 8; ++k) {
                        xv[k] = __hip_atomic_load(gs + (size_t)k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (unsigned)(xv[k] >> 32) == tg;
                    }
                    if (__all(ok)) break;
                    if (pass > 64 && __hip_atomic_load(A.ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    wait_hint(slots + (size_t)t * Area::SLOT + gran_idx(sl * US + 4 * w, 15), tg, A.ctl + 2);
                }
            }
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                bf16x8 ah, al;
                unpack8(xv + 8 * m, ah, al);
                const bf16x8 bh = wimg[(2 * m) * 256 + tid], bl = wimg[(2 * m + 1) * 256 + tid];
                acc1 = mfma_bf(al, bh, acc1); acc1 = mfma_bf(ah, bl, acc1);       // two chains: the small terms, the hi hi term
                acc0 = mfma_bf(ah, bh, acc0);
            }
            float* p2 = part2 + ((t & 1) * 4 + w) * 16 * US;
#pragma unroll
            for (int r = 0; r < 4; ++r) p2[(4 * g + r) * US + l16] = acc0[r] + acc1[r];
        };
        input_part(nmax - 1);
        load_saved(nmax - 1);
        group_barrier(mybar, btarget);
        for (int t = nmax - 1; t >= 0; --t) {
            const int idx = nmax - 1 - t, par = idx & 1;
            const unsigned tag = base + (unsigned)idx + 1u;
            float* dgl = dgl0 + par * 16 * LDG;
            const bool st_on = grp == 0 && sl == 0 && w == 0 && t == nmax / 2;
            PAIR_STAMP(0);
            if (t > 0) input_part(t - 1);                   // while the partials of step t + 1 are in flight
            PAIR_STAMP(1);
            if (idx > 0) carry = keep + sweep_partials(par ^ 1, tag - 1u);
            PAIR_STAMP(2);
            const float* p2 = part2 + (t & 1) * 4 * 16 * US + es * US + eu;
            const float dh1 = (p2[0] + p2[16 * US]) + (p2[2 * 16 * US] + p2[3 * 16 * US]);
            float dr, dz, dn;
            keep = gates(t, dh1 + carry, dgl, dr, dz, dn);
            PAIR_STAMP(3);
            group_barrier(mybar, btarget);
            PAIR_STAMP(4);
            load_saved(t - 1);
            if (t > 0) rec_partials(dgl, par, tag, true);
            PAIR_STAMP(5);
        }
    }
#undef PAIR_STAMP
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
    if (DR4SR_ENV("DR4SR_GRU_NOCOOP") || (H != 128 && H != 256)) return 0;
    const int groups = (B + 15) / 16, g8 = ((groups + 7) / 8) * 8, cus = device_cus();
    const bool no16 = DR4SR_XENV("DR4SR_GRU_NS8") != nullptr;             // cross-check switch: always 8 slices
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
    const bool nochunk = DR4SR_ENV("DR4SR_GRU_NOCHUNK") != nullptr;         // cross-check switch
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
    const bool bwd_f32 = DR4SR_ENV("DR4SR_GRU_BWD_F32") != nullptr;         // cross-check switch: the fp32-MFMA BPTT with W_hh in LDS
    const bool fwd_f32 = DR4SR_ENV("DR4SR_GRU_FWD_F32") != nullptr;         // ... and the fp32-MFMA forward
    if (H == 256 && ns == 16 && !bwd && !fwd_f32) {
        const size_t lds = sizeof(float) * 2 * 4 * 3 * 256 + 32 * sizeof(int);
        if (!resident((const void*)k_gru_fwd_coop_bf<256, 16>, lds, 1)) return -100;
        hipLaunchKernelGGL((k_gru_fwd_coop_bf<256, 16>), grid, blk, lds, s, A);
    } else if (H == 256 && ns == 16 && bwd && !bwd_f32) {
        const size_t lds = sizeof(float) * 2 * 16 * (3 * 16 + 4) + 32 * sizeof(int);
        if (!resident((const void*)k_gru_bwd_coop_bf<256, 16>, lds, 1)) return -100;
        hipLaunchKernelGGL((k_gru_bwd_coop_bf<256, 16>), grid, blk, lds, s, A);
    } else if (H == 256 && ns == 16) COOP_LAUNCH(256, 16, 2);
    else if (H == 256) COOP_LAUNCH(256, 8, 1);
    else if (H == 128) COOP_LAUNCH(128, 8, 1);
    else return DR4SR_E_SHAPE;
#undef COOP_LAUNCH
    return DR4SR_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ two-layer wavefront launch
// Applies to n_layer == 2, H == 256 and batches the 16-slice cooperative form takes (at most 16 groups per launch: B <= 256, or chunks of
// 256 up to 1536): 2 layers x 16 groups x 16 slices = 512 workgroups, two per CU.  DR4SR_GRU_NOWAVE (cross-check): one launch per layer.
static bool wave_ok(int B, int H, int n_layer, int L) {
    const bool off = DR4SR_ENV("DR4SR_GRU_NOWAVE") != nullptr;
    if (off || n_layer != 2 || H != 256 || L > 64) return false;
    const int cb = coop_chunk(B, H);
    return cb != 0 && coop_slices(cb, H) == 16 && 2 * 16 * (((cb + 15) / 16 + 7) / 8) * 8 <= 2 * device_cus();
}
// Granule words the workspace reserves for a plan of at most B sequences: an upper bound of what ANY batch of at most B sequences needs
// (a smaller batch can need more than a larger one — 256 sequences take 16 slices per group and the wavefront's per-step slots, 300 take
// 8 slices and no wavefront — and the last batch of an epoch runs in the workspace sized for the full one), monotone in B.
int64_t gru_xch_words(int B, int H, int L, int n_layer) {
    if (gru_coop_words(B < 16 ? B : 16, H) == 0) return 0;                     // no cooperative path at all on this device / under DR4SR_GRU_NOCOOP
    const int64_t g = (B + 15) / 16;
    const int64_t coop = (g < 24 ? g : 24) * 2 * 16 * 16 * H;                  // at most 24 groups per launch, at most 16 slices per group
    const int64_t wave = wave_ok(B < 256 ? B : 256, H, n_layer, L) ? (g < 16 ? g : 16) * (int64_t)wave_group_words<256, 16>(L) : 0;
    return coop > wave ? coop : wave;
}
// returns -100 when the plan does not qualify (caller: one cooperative launch per layer)
int launch_gru_wave(const GruWaveArgs& G, unsigned long long* xch, int* ctl, int B, int H, int L, bool bwd, hipStream_t s) {
    if (!xch || !ctl || !wave_ok(B, H, 2, L)) return -100;
    const bool wave_bwd = DR4SR_XENV("DR4SR_GRU_WAVE_BWD") != nullptr;
    if (bwd && !wave_bwd) return -100;
    const int cb = coop_chunk(B, H);
    WaveArgs A;
    A.gi1 = G.gi1; A.wih2 = G.wih2; A.dhout = G.dhout; A.xch = xch; A.ctl = ctl; A.L = L;
    const int presleep = DR4SR_XENV("DR4SR_GRU_WAVE_PRESLEEP") ? atoi(DR4SR_XENV("DR4SR_GRU_WAVE_PRESLEEP")) : 0;
    const int solo = DR4SR_XENV("DR4SR_GRU_WAVE_SOLO") ? atoi(DR4SR_XENV("DR4SR_GRU_WAVE_SOLO")) : 0;            // diagnosis only: the follower layer does not run (wrong results)
    const int stamp = DR4SR_XENV("DR4SR_GRU_WAVE_STAMP") ? 1 : 0;
    A.presleep = presleep; A.solo = solo; A.stamp = stamp;
    A.order = DR4SR_XENV("DR4SR_GRU_WAVE_ORDER") ? atoi(DR4SR_XENV("DR4SR_GRU_WAVE_ORDER")) : 1;
    for (int l = 0; l < 2; ++l) {
        A.whh[l] = G.whh[l]; A.r[l] = G.r[l]; A.z[l] = G.z[l]; A.n[l] = G.n[l]; A.ghn[l] = G.ghn[l]; A.hprev[l] = G.hprev[l]; A.hout[l] = G.hout[l];
        A.dgi[l] = G.dgi[l]; A.dgh[l] = G.dgh[l];
    }
    const size_t lds = bwd ? sizeof(float) * (2 * 2 * 16 * (3 * 16 + 4) + 2 * 4 * 16 * 16) + 36 * sizeof(int) + (size_t)(12 + 16) * 256 * 16
                           : sizeof(float) * (2 * 4 * 4 * 16 * 16) + 32 * sizeof(int);
    const void* k = bwd ? (const void*)k_gru_bwd_pair<256, 16> : (const void*)k_gru_fwd_wave<256, 16>;
    // forward: both layers' workgroups of a CU must be resident together, two per CU; backward: one 8-wave workgroup per CU
    static int fits[2] = {-1, -1};
    if (bwd) big_lds(k_gru_bwd_pair<256, 16>, lds); else big_lds(k_gru_fwd_wave<256, 16>, lds);
    if (fits[bwd] < 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, bwd ? 512 : 256, lds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        fits[bwd] = nb >= (bwd ? 1 : 2);
    }
    if (!fits[bwd]) return -100;
    for (int c0 = 0; c0 < B; c0 += cb) {                    // chunks of sequences: consecutive launches on the same granule area (epochs are per launch)
        A.cu = G.cu + c0; A.B = B - c0 < cb ? B - c0 : cb;
        const int groups = (A.B + 15) / 16, g8 = ((groups + 7) / 8) * 8;
        if (bwd) hipLaunchKernelGGL((k_gru_bwd_pair<256, 16>), dim3(g8 * 16), dim3(512), lds, s, A);
        else hipLaunchKernelGGL((k_gru_fwd_wave<256, 16>), dim3(g8 * 2 * 16), dim3(256), lds, s, A);
    }
    return DR4SR_LAUNCH_CHECK();
}
extern "C" int dr4sr_gru4rec_uses_wavefront(int32_t B, int32_t H, int32_t n_layer, int32_t L) { return wave_ok(B, H, n_layer, L) ? 1 : 0; }
