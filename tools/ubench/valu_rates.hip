// Instruction-rate microbenchmark for gfx950: cycles per wave-instruction for the VALU ops of the attention softmax,
// alone and next to MFMAs, at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP 64
#define ITERS 256

template <int OP>
__global__ void k(float* out, long long* cyc, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
                if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i & 6)]));
                if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
                if (OP == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i & 6)]));
                if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i & 6)]));
                if (OP == 8) {  // 1 MFMA + 7 fma
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
                }
                if (OP == 9) {  // 1 MFMA + 7 exp
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
                if (OP == 10) {  // MFMA only (2 accumulators)
                    if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                }
                if (OP == 11) {  // 1 MFMA + 3 exp + 4 fma
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else if (i < 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
                }
                if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                if (OP == 13) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(3));
                if (OP == 14) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
                if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(0x07060302));
                if (OP == 16) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(0xffff0000), "v"(seed));
                if (OP == 17) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(a[i]));
                if (OP == 18) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
                if (OP == 19) {  // the attention mix per MFMA: 2 exp + 1 cvt_pk + 2 add (+ 1 slot)
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else if (i < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    else if (i == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                    else if (i < 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
                }
                if (OP == 20) {  // trimmed mix: 2 exp + 1 perm + 1 dot2c
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else if (i < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    else if (i == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(0x07060302));
                    else if (i == 4) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[3]));
                }
                if (OP == 21) {  // 2 exp only per MFMA
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else if (i < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, long long* cyc) {
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        long long c[16];
        int nw = threads / 64;
        hipMemcpy(c, cyc, 8 * nw, hipMemcpyDeviceToHost);
        long long mx = 0, mn = 1ll << 60;
        for (int i = 0; i < nw; ++i) { mx = c[i] > mx ? c[i] : mx; mn = c[i] < mn ? c[i] : mn; }
        double per = (double)mx / (ITERS * REP), permin = (double)mn / (ITERS * REP);
        printf("%-26s waves/SIMD=%d  ticks/instr: slowest wave %.2f fastest %.2f -> per-SIMD issue interval %.2f\n", name,
               threads / 256, per, permin, per / (threads / 256));
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 256);
    // calibrate the counter: s_memtime/readcyclecounter ticks vs wall
    run<0>("v_fma_f32", out, cyc);
    run<12>("v_mul_f32", out, cyc);
    run<5>("v_add_f32", out, cyc);
    run<1>("v_exp_f32", out, cyc);
    run<2>("v_pk_fma_f32", out, cyc);
    run<6>("v_pk_add_f32", out, cyc);
    run<7>("v_pk_mul_f32", out, cyc);
    run<3>("v_cvt_pk_bf16_f32", out, cyc);
    run<4>("v_max3_f32", out, cyc);
    run<13>("v_ldexp_f32", out, cyc);
    run<10>("mfma 32x32x16 only", out, cyc);
    run<8>("1 mfma + 7 fma", out, cyc);
    run<9>("1 mfma + 7 exp", out, cyc);
    run<11>("1 mfma + 3 exp + 4 fma", out, cyc);
    run<14>("v_dot2c_f32_bf16", out, cyc);
    run<15>("v_perm_b32", out, cyc);
    run<16>("v_and_or_b32", out, cyc);
    run<17>("v_lshrrev_b32", out, cyc);
    run<18>("v_exp_f16", out, cyc);
    run<19>("mfma+2exp+cvt+2add (x8/6)", out, cyc);
    run<20>("mfma+2exp+perm+dot2c (x8/5)", out, cyc);
    run<21>("mfma+2exp (x8/3)", out, cyc);
    // wall-clock calibration of the counter
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %lld ticks in <= %.3f ms (incl. launch)\n", c, ms);
    return 0;
}
