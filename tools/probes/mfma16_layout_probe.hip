// Checks (1) the operand layout assumed for v_mfma_f32_16x16x16_bf16 (A[i][k]: lane i + 16 (k / 4), element k % 4; B[k][n]: lane n + 16 (k / 4);
// C[row 4 (lane / 16) + r][col lane % 16]) with integer matrices, (2) a 48-k product as one 32-k + one 16-k bf16x3 chain on the same
// accumulator (csrc/gru_coop.hip, partials) against fp32.   hipcc --offload-arch=gfx950 -O3 mfma16_layout_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out) {
    const int lane = threadIdx.x, l16 = lane & 15, g = lane >> 4;
    s16x4 a, b;
    for (int j = 0; j < 4; ++j) {
        const int kk = 4 * g + j;
        const __bf16 av = (__bf16)(float)((l16 * 3 + kk) % 5 - 2), bv = (__bf16)(float)((kk * 7 + l16) % 9 - 4);
        a[j] = __builtin_bit_cast(short, av); b[j] = __builtin_bit_cast(short, bv);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + l16] = c[r];
}
__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { const __bf16 h = (__bf16)x[i]; hi[i] = h; lo[i] = (__bf16)(x[i] - (float)h); }
}
__device__ __forceinline__ void split4(const float* x, s16x4& hi, s16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const __bf16 h = (__bf16)x[i], l = (__bf16)(x[i] - (float)h); hi[i] = __builtin_bit_cast(short, h); lo[i] = __builtin_bit_cast(short, l); }
}
// A [16][48], B [48][64] (4 column tiles), C [16][64]
__global__ void k2(const float* A, const float* B, float* C, int mode) {
    const int lane = threadIdx.x, l16 = lane & 15, g = lane >> 4;
    bf16x8 ah, al; s16x4 a2h, a2l;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = A[l16 * 48 + 8 * g + j];
    split8(x, ah, al);
    for (int j = 0; j < 4; ++j) x[j] = A[l16 * 48 + 32 + 4 * g + j];
    split4(x, a2h, a2l);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        bf16x8 bh, bl; s16x4 b2h, b2l;
        for (int j = 0; j < 8; ++j) x[j] = B[(8 * g + j) * 64 + ci * 16 + l16];
        split8(x, bh, bl);
        for (int j = 0; j < 4; ++j) x[j] = B[(32 + 4 * g + j) * 64 + ci * 16 + l16];
        split4(x, b2h, b2l);
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, c2 = c;
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
        if (mode == 0) {
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2l, b2h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2h, b2l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2h, b2h, c, 0, 0, 0);
        } else if (mode == 1) {
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2l, b2h, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2h, b2l, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2h, b2h, c2, 0, 0, 0);
            c += c2;
        } else {
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2h, b2h, c2, 0, 0, 0);
            c += c2;
        }
        for (int r = 0; r < 4; ++r) C[(4 * g + r) * 64 + ci * 16 + l16] = c[r];
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
        float s = 0;
        for (int kk = 0; kk < 16; ++kk) s += (float)((i * 3 + kk) % 5 - 2) * (float)((kk * 7 + n) % 9 - 4);
        if (s != h[i * 16 + n]) ++bad;
    }
    printf("mfma 16x16x16 bf16 layout: %d mismatches of 256\n", bad);
    static float A[16 * 48], B[48 * 64], C[16 * 64];
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    float *dA, *dB, *dC; hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dC, sizeof C);
    (void)hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    double worst = 0;
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode);
        (void)hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
        double wm = 0;
        for (int i = 0; i < 16; ++i) for (int n = 0; n < 64; ++n) {
            double s = 0;
            for (int kk = 0; kk < 48; ++kk) s += (double)A[i * 48 + kk] * B[kk * 64 + n];
            wm = fmax(wm, fabs(s - C[i * 64 + n]));
        }
        printf("48-k product, mode %d (0: one chain 32-k then 16-k, 1: separate accumulators, 2: separate, hi hi only): max abs error %.3g\n", mode, wm);
        if (mode == 0) worst = wm;
    }
    return bad != 0 || worst > 1e-3;
}
