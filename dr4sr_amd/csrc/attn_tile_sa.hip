// attn_tile_sa.hip — the window attention of attn_tile.h as launches of its own: at scale, short sequences, d = 64 (round 4).
// OPT-IN (DR4SR_ATTN_WINDOW=1): built for the round-3 review's "fold the short classes into the token tiles", oracle-tested, measured
// SLOWER than the lists it would replace — toys histogram, us per layer forward / backward: B = 4 096 15.6 / 39.5 against 16.6 / 35.0,
// B = 8 192 25 / 68 against 24 / 49, B = 32 768 92 / 239 against 63 / 159.  A 16-query x 32-key window computes ~9x the (query, key)
// pairs the sequences of a toys batch hold (mean length 5.45), every tile pays its own staging address arithmetic and Philox calls,
// and with three workgroups per CU the launch is bound by instruction issue (~1 200 VALU + 60..280 fp32-MFMA instructions per wave
// and tile; two workgroups per CU: same backward time), not by a latency chain that co-residency could hide.  NOTEBOOK round 4.
//
// Why it was tried.  At scale the attention ran as length-class lists (attn_mfma.hip: a 1..8-token VALU class, a 16-row and a 64-row MFMA class): two
// launches per layer forward, three backward, each bound by its own latency chain (list entry -> descriptor -> rows) on a grid that
// holds one class only — 26 + 51 us per layer of a 516 us step at B = 8 192 on the toys histogram (profiles/round4_kernels_sasrec_B8192.txt),
// 30 % of the step at 1–7 % matrix-pipe utilisation.  The in-tile form of the latency regime (attn_tile.h) needs no lists at all: the
// workgroup that owns tokens [t0, t0 + 16) of the packed stream stages the rows in front of them and attends for its 16 query rows,
// whatever sequences they belong to.  Inside the wave-tile kernels (linear_wave.hip) there is no LDS left for a window (their weight
// images fill it), so at scale the same bodies run here as ONE launch per layer and direction: a persistent loop over the token
// tiles, three workgroups per CU (53 KB of LDS each).  Short-sequence plans only (expected mean length <= 16: most tiles need the near
// half of the window, tattn::far_rows_if_needed fetches the rest on demand); long-sequence batches keep the 64-row list kernels,
// whose K | V rows are read once per sequence instead of once per tile.
//
// Arithmetic, statistics and dropout elements are those of attn_mfma.hip (tattn::fwd / tattn::bwd are shared with the in-tile form and
// were written to be interchangeable with the lists per launch); the tests hold the two forms against each other and the oracle.
//   forward : qkv [T][3D] -> ctx [T][D], statistics [T][H][2]; the top layer's launch also zeroes the K | V rows of its dqkv
//   backward: dctx [T][D], <dctx, ctx> [T][H] (epilogue of k_wt_post_bwd / k_wt_post_mid) -> dqkv: dQ rows complete, dK | dV rows added
//             with fp32 atomics into zeroed rows (tile-private rows: plain stores); the launch of layer l zeroes the rows of layer l - 1
//             (a second backward pass on one forward starts clean: the top layer's rows are zeroed by k_pack, embed.hip).
//
// Reference arithmetic: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer, /root/reference model/sasrec.py:21-34, attn_mask
// triu(1) :58, key_padding_mask idx == 0 :48.
#include "common.h"
#include "kernels.h"
#ifdef DR4SR_EXPERIMENTS      // a rejected experiment (measured slower than the lists, and round 6's attn_wave.hip replaced both): not in the shipped build
#include "attn_tile.h"

extern __shared__ __attribute__((aligned(16))) float smem[];

namespace {

template <int D>
__device__ __forceinline__ void zero_kv_tile(float* dqkv, const int t0, const int T) {
    constexpr int C4 = 2 * D / 4;
    for (int i = threadIdx.x; i < 16 * C4; i += 256) {
        const int r = i / C4, c = (i % C4) * 4;
        if (t0 + r < T) st4(dqkv + (size_t)(t0 + r) * 3 * D + D + c, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

template <int D>
__global__ __launch_bounds__(256, 3) void k_attn_tile_fwd(const PostArgs P, float* __restrict__ zero_dqkv) {
    constexpr int LD = D + 4;
    const int T = P.state[DR4SR_STATE_T];
    float* R0 = smem + tattn::Lds<D>::floats;               // [16][LD] ctx tile (tattn::fwd leaves it there for a consumer in the same launch: none here)
#pragma unroll 1
    for (int tile = blockIdx.x; tile * 16 < T; tile += gridDim.x) {
        const int t0 = tile * 16;
        if (zero_dqkv) zero_kv_tile<D>(zero_dqkv, t0, T);
        tattn::fwd<D, false>(P, t0, T, R0, LD, smem);
        lds_barrier();                                       // the next tile's window overwrites what the slowest wave may still read
    }
}

template <int D>
__global__ __launch_bounds__(256, 3) void k_attn_tile_bwd(const PostArgs P, const float* __restrict__ dctx, const float* __restrict__ rd,
                                                       float* __restrict__ zero_dqkv) {
    constexpr int LD = D + 4, QPER = (16 * D / 4) / 256;
    const int T = P.state[DR4SR_STATE_T];
    const tattn::Lds<D> S(smem);
    float* Cs = smem + tattn::Lds<D>::floats;               // [16][LD] dctx tile
    const int r_lo = (P.at.on & 4) ? tattn::NEAR0 : 0;
#pragma unroll 1
    for (int tile = blockIdx.x; tile * 16 < T; tile += gridDim.x) {
        const int t0 = tile * 16;
        if (zero_dqkv) zero_kv_tile<D>(zero_dqkv, t0, T);
        const int2 mq = tattn::own_word(P.at, t0, T);
        tattn::Stage<D, true, true> st;
        st.issue(P.at, t0, T, r_lo, tattn::WR);
        float4 cv[QPER];
#pragma unroll
        for (int q = 0; q < QPER; ++q) {
            const int f = threadIdx.x + 256 * q, r = f / (D / 4), c4 = f % (D / 4);
            cv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + r < T) cv[q] = ld4(dctx + (size_t)(t0 + r) * D + 4 * c4);
        }
        float rdv = 0.f;
        if (threadIdx.x < 32 && t0 + (int)(threadIdx.x >> 1) < T) rdv = rd[(size_t)(t0 + (threadIdx.x >> 1)) * 2 + (threadIdx.x & 1)];
        __builtin_amdgcn_sched_barrier(0);
        const tattn::Keep keep = tattn::own_keep(P, mq, t0, T);         // Philox calls while the window is in flight
        __builtin_amdgcn_sched_barrier(0);
        st.commit(S, r_lo, tattn::WR);
#pragma unroll
        for (int q = 0; q < QPER; ++q) {
            const int f = threadIdx.x + 256 * q, r = f / (D / 4), c4 = f % (D / 4);
            st4(Cs + r * LD + 4 * c4, cv[q]);
        }
        if (threadIdx.x < 32) S.rd[(threadIdx.x >> 1) * 2 + (threadIdx.x & 1)] = rdv;
        lds_barrier();
        tattn::far_rows_if_needed<D>(P.at, S, t0, T);
        tattn::bwd<D>(P, t0, T, Cs, LD, smem, keep);
        lds_barrier();
    }
}

size_t sa_lds_bytes(int D) { return sizeof(float) * ((D == 64 ? tattn::Lds<64>::floats : tattn::Lds<128>::floats) + 16 * (D + 4)); }

int sa_grid(const Workspace& ws, int D) {
    int dev = 0, cus = 256;
    hipDeviceProp_t pr;
    static int cached = 0;
    if (!cached) {
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        cached = cus;
    }
    const int per_cu = (int)((160 * 1024) / sa_lds_bytes(D));
    int wpc = per_cu > 0 ? per_cu : 1;
    if (const char* e = DR4SR_XENV("DR4SR_ATTN_SA_WPC")) wpc = atoi(e) > 0 ? atoi(e) : wpc;       // sweeps: workgroups per CU of the persistent grid
    const int tiles = (ws.Tmax + 15) / 16, g = cached * wpc;
    return tiles < g ? tiles : g;
}

PostArgs sa_args(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training) {
    PostArgs A = make_post_args(p, ws, layer, training);
    A.at.on = 1;
    if (!DR4SR_XENV("DR4SR_ATTN_TILE_FULL")) A.at.on |= 4;        // short-sequence plans by construction: the near half of the window first
    if (DR4SR_ENV("DR4SR_ATTN_TILE_ATOMICS")) A.at.on |= 2;      // cross-check: every dK | dV row through atomics
    A.stamps = nullptr;
    return A;
}

}  // namespace

int launch_attn_tile_fwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    if (p->D != 64 || p->H != 2 || p->L > 64) return DR4SR_E_SHAPE;
    const PostArgs A = sa_args(p, ws, layer, training);
    float* zero = layer == p->n_layer - 1 ? ws.layer[layer].dqkv : nullptr;
    const size_t lds = sa_lds_bytes(64);
    big_lds(k_attn_tile_fwd<64>, lds);
    hipLaunchKernelGGL(k_attn_tile_fwd<64>, dim3(sa_grid(ws, 64)), dim3(256), lds, s, A, zero);
    return DR4SR_LAUNCH_CHECK();
}

int launch_attn_tile_bwd(const dr4sr_sasrec_plan* p, const Workspace& ws, int layer, int training, hipStream_t s) {
    if (p->D != 64 || p->H != 2 || p->L > 64) return DR4SR_E_SHAPE;
    const PostArgs A = sa_args(p, ws, layer, training);
    float* zero = layer > 0 ? ws.layer[layer - 1].dqkv : nullptr;
    const size_t lds = sa_lds_bytes(64);
    big_lds(k_attn_tile_bwd<64>, lds);
    hipLaunchKernelGGL(k_attn_tile_bwd<64>, dim3(sa_grid(ws, 64)), dim3(256), lds, s, A, ws.dctx, ws.attn_rd, zero);
    return DR4SR_LAUNCH_CHECK();
}
#else
// shipped build: Workspace::attn_tile_sa is never set (DR4SR_ATTN_WINDOW is a compile-time nullptr, common.h DR4SR_XENV)
int launch_attn_tile_fwd(const dr4sr_sasrec_plan*, const Workspace&, int, int, hipStream_t) { return DR4SR_E_SHAPE; }
int launch_attn_tile_bwd(const dr4sr_sasrec_plan*, const Workspace&, int, int, hipStream_t) { return DR4SR_E_SHAPE; }
#endif
