// Where does the dispatcher put the workgroups of 1-D and 3-D grids?  Prints, per grid shape, how many blocks sit on XCC (linear id % 8)
// with linear id = x + gx * (y + gy * z), and the same for x % 8 — also with most blocks exiting at once (as the token-tile kernels' dead blocks do).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcd_dispatch_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(int* out, int live_every, int spin) {
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[lin] = (int)(xcc & 0xf);
    }
    if (live_every > 1 && (lin % live_every) != 0) return;
    // some work so that live blocks stay resident while the rest are dispatched
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[0] = -1;
}

static void run(dim3 g, int live_every, int spin) {
    const int n = g.x * g.y * g.z;
    int* d; hipMalloc(&d, n * sizeof(int));
    hipMemset(d, 0xff, n * sizeof(int));
    hipLaunchKernelGGL(k_probe, g, dim3(256), 0, 0, d, live_every, spin);
    hipDeviceSynchronize();
    std::vector<int> h(n);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    int ok_lin = 0, ok_x = 0;
    for (int z = 0; z < (int)g.z; ++z) for (int y = 0; y < (int)g.y; ++y) for (int x = 0; x < (int)g.x; ++x) {
        const int lin = x + g.x * (y + g.y * z);
        ok_lin += h[lin] == lin % 8; ok_x += h[lin] == x % 8;
    }
    printf("grid (%d,%d,%d) live 1/%d spin %d: xcc == lin %% 8 for %d / %d blocks, xcc == x %% 8 for %d / %d; first 16 xcc:", g.x, g.y, g.z, live_every, spin,
           ok_lin, n, ok_x, n);
    for (int i = 0; i < 16 && i < n; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}

int main() {
    run(dim3(800), 1, 2000); run(dim3(800), 8, 20000); run(dim3(48, 17, 3), 1, 2000); run(dim3(48, 17, 3), 2, 20000);
    run(dim3(50, 17, 3), 1, 2000); run(dim3(256), 1, 20000); run(dim3(768), 1, 20000);
    return 0;
}
