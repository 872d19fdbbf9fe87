// atomic_scope_probe.hip — how fast are fp32 atomic adds into a 3 MB table at agent scope (resolved memory-side on a multi-XCD part)
// versus workgroup scope into a PER-XCD private copy (resolved in the issuing XCD's L2), and is the latter exact?
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_scope_probe atomic_scope_probe.hip ; run on the GPU box.  DESIGN.md §4a.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// every thread adds 1.0f to K pseudo-random rows' element (lane & 63) — the access pattern of a 64-float row scatter
template <int SCOPE, bool PRIVATE>
__global__ void k_add(float* table, int n_rows, int K, unsigned* xcc_hist) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + x], 1u);
    float* t = PRIVATE ? table + (size_t)x * n_rows * 64 : table;
    const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < K; ++k) {
        const unsigned r = mix(gw * 977u + k) % (unsigned)n_rows;
        __hip_atomic_fetch_add(&t[(size_t)r * 64 + lane], 1.0f, __ATOMIC_RELAXED, SCOPE);
    }
}

int main() {
    const int n_rows = 11925, K = 64, blocks = 4096, threads = 256;
    const size_t tbl = (size_t)n_rows * 64;
    float* d; unsigned* h;
    hipMalloc(&d, tbl * 8 * sizeof(float)); hipMalloc(&h, 128 * sizeof(unsigned));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double total = (double)blocks * threads * K;
    auto run = [&](const char* name, auto kern, bool priv) {
        hipMemset(d, 0, tbl * 8 * sizeof(float)); hipMemset(h, 0, 128 * sizeof(unsigned));
        kern<<<blocks, threads>>>(d, n_rows, K, h);                    // warm
        hipMemset(d, 0, tbl * 8 * sizeof(float)); hipMemset(h, 0, 128 * sizeof(unsigned));
        hipDeviceSynchronize();
        hipEventRecord(a); kern<<<blocks, threads>>>(d, n_rows, K, h); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<float> host(tbl * 8);
        hipMemcpy(host.data(), d, tbl * 8 * sizeof(float), hipMemcpyDeviceToHost);
        double sum = 0; for (size_t i = 0; i < tbl * (priv ? 8 : 1); ++i) sum += host[i];
        printf("%-44s %8.3f ms  %7.1f G atomics/s   sum %.0f of %.0f (%s)\n", name, ms, total / ms / 1e6, sum, total, sum == total ? "exact" : "LOST UPDATES");
    };
    run("agent scope, one table", k_add<__HIP_MEMORY_SCOPE_AGENT, false>, false);
    run("workgroup scope, per-XCD private tables", k_add<__HIP_MEMORY_SCOPE_WORKGROUP, true>, true);
    run("agent scope, per-XCD private tables", k_add<__HIP_MEMORY_SCOPE_AGENT, true>, true);
    run("workgroup scope, one table (expected to lose)", k_add<__HIP_MEMORY_SCOPE_WORKGROUP, false>, false);
    unsigned hist[128]; hipMemcpy(hist, h, sizeof(hist), hipMemcpyDeviceToHost);
    printf("blockIdx %% 8 -> XCC_ID histogram (rows = blockIdx %% 8):\n");
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 8; ++j) printf("%6u", hist[i * 16 + j]); printf("\n"); }
    return 0;
}
