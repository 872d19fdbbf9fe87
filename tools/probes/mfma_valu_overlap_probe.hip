// Does v_mfma_f32_16x16x4_f32 (fp32-in MFMA, "runs at the fp32 vector rate") overlap with VALU work of ANOTHER wave on the same SIMD,
// or do the two share the SIMD's fp32 datapath?  512-thread workgroups put waves w and w + 4 on the same SIMD: waves 0-3 run an MFMA
// loop, waves 4-7 a VALU loop (fp32 fma / integer mul_hi / integer xor-add / bf16 MFMA for comparison).  Prints the time of MFMA alone,
// VALU alone and both: both ~ max -> separate pipes, both ~ sum -> shared.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap_probe mfma_valu_overlap_probe.hip && ./mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, bool BF16>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int do_mfma, int do_valu) {
    const int w = threadIdx.x >> 6;
    if (w < 4) {
        if (!do_mfma) return;
        if (BF16) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            bf16x8 x, y;
            for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)1.0f; }
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
            }
            out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        } else {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            const float x = (float)threadIdx.x, y = 1.0f;
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
            out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else {
        if (!do_valu) return;
        if (KIND == 0) {                                   // fp32 fma, 8 independent chains
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = (float)(threadIdx.x + j);
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
            float s = 0;
            for (int j = 0; j < 8; ++j) s += v[j];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        } else if (KIND == 1) {                            // integer mul_hi (Philox's expensive op)
            uint32_t v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 2654435761u + j;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __umulhi(v[j], 0xD2511F53u) ^ (uint32_t)i;
            uint32_t s = 0;
            for (int j = 0; j < 8; ++j) s += v[j];
            out[blockIdx.x * 512 + threadIdx.x] = (float)s;
        } else {                                           // integer xor / add
            uint32_t v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 2654435761u + j;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (v[j] ^ 0x9E3779B9u) + (uint32_t)i;
            uint32_t s = 0;
            for (int j = 0; j < 8; ++j) s += v[j];
            out[blockIdx.x * 512 + threadIdx.x] = (float)s;
        }
    }
}

template <int KIND, bool BF16>
static void run(const char* name, float* out, int n_mfma, int n_valu) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
        const int dm = mode != 1, dv = mode != 0;
        hipLaunchKernelGGL((k<KIND, BF16>), dim3(256), dim3(512), 0, 0, out, n_mfma, n_valu, dm, dv);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k<KIND, BF16>), dim3(256), dim3(512), 0, 0, out, n_mfma, n_valu, dm, dv);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms[mode], a, b);
    }
    printf("%-34s mfma alone %.3f ms, valu alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", name, ms[0], ms[1], ms[2], ms[0] + ms[1],
           ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    // ~equal alone-times: 4 x 16x16x4 f32 MFMAs = 128 cycles per iteration; 8 VALU ops per iteration
    run<0, false>("f32 MFMA | fp32 fma", out, 20000, 80000);
    run<1, false>("f32 MFMA | int mul_hi", out, 20000, 20000);
    run<2, false>("f32 MFMA | int xor+add", out, 20000, 40000);
    run<0, true>("bf16 MFMA 16x16x32 | fp32 fma", out, 80000, 80000);
    run<1, true>("bf16 MFMA 16x16x32 | int mul_hi", out, 80000, 20000);
    return 0;
}
