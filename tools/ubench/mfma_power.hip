// Sustained MFMA throughput under the package power limit, by instruction shape: every SIMD of the chip runs a loop of
// independent MFMAs on random bf16 operands (8 A x 8 B fragments in registers, so consecutive instructions see different
// data, as in a GEMM main loop), 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OP>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ src, float* out, int iters, int zero) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(i * 256 + threadIdx.x) & 4095];
        b[i] = src[(2048 + i * 256 + threadIdx.x) & 4095];
        if (zero) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)0.f;
    }
    float s = 0.f;
    if (OP == 0) {
        f32x16 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + j) & 7], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[16] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)   // 64 x (16x16x32) = the flops of 32 x (32x32x16)
                    acc[(i & 1) * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[(i & 1) * 8 + j], 0, 0, 0);
        }
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, const bf16x8* src, float* out, int zero) {
    const int iters = 4000, grid = 256 * 2;
    const double flop_per_wg = 4.0 * (double)iters * 32 * 32768.0;   // 4 waves x iters x 32 MFMA-equivalents x 32768 flop
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, src, out, iters / 8, zero);   // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int l = 0; l < 40; ++l) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, src, out, iters, zero);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s zero_A=%d  %.0f TFLOP/s  (%.1f ms per launch)\n", name, zero, 40 * grid * flop_per_wg / ms / 1e9, ms / 40);
    }
}

int main() {
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) {   // random bf16 in roughly [-2, 2]
        float f = ((rand() & 0xffff) / 32768.0f - 1.0f) * 2.0f;
        unsigned u; memcpy(&u, &f, 4);
        h[i] = (unsigned short)(u >> 16);
    }
    bf16x8* src; float* out;
    hipMalloc(&src, 4096 * 16); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    run<0>("v_mfma_f32_32x32x16_bf16", src, out, 0);
    run<1>("v_mfma_f32_16x16x32_bf16", src, out, 0);
    run<0>("v_mfma_f32_32x32x16_bf16", src, out, 1);
    run<1>("v_mfma_f32_16x16x32_bf16", src, out, 1);
    run<0>("v_mfma_f32_32x32x16_bf16", src, out, 0);
    run<1>("v_mfma_f32_16x16x32_bf16", src, out, 0);
    return 0;
}
