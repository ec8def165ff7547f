// What bounds the 256 x 256 x 64 GEMM main loop: the L2 -> LDS fill (LDS-DMA) rate of a CU, alone and beside the loop's other LDS
// traffic (fragment reads) and its MFMAs.  One 512-thread workgroup per CU, 128 KiB of LDS (two 64 KiB stages, as the product
// kernel); per iteration ("K tile") a workgroup moves 64 KiB by global_load_lds_dwordx4 -- 256 rows x 128 B for each of two
// operands, row stride `ld` bytes, the K offset advancing 128 B per iteration -- and/or issues the product kernel's 24
// ds_read_b128 and 64 v_mfma_f32_16x16x32_bf16 per wave.  `share` workgroups of an XCD read the same rows (L2 hits for all but
// the first), as the tiles of a GEMM do.  No synchronisation, no correctness: rates only.
//     hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o lds_dma_rate && ./lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MODE bit 0: DMA, bit 1: fragment reads, bit 2: MFMAs
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* __restrict__ src, float* out, int iters, long ld, int share, long wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const char* base = src + ((long)xcd * 32 + slot / share) * wg_stride;      // `share` workgroups of an XCD read the same rows
    const char* a_src[4];
    const char* w_src[4];
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3), c = lane & 7;
        a_src[i] = base + (long)r * ld + ((c ^ (r & 7)) << 4);
        w_src[i] = base + (long)(256 + r) * ld + ((c ^ (r & 7)) << 4);
    }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fr[12];
    for (int i = 0; i < 12; ++i) for (int e = 0; e < 8; ++e) fr[i][e] = (__bf16)(0.001f * (lane + i + e));
    const unsigned fo = (lane & 15) * 128 + ((lane >> 4) << 4);
    const bool blocked = ld == 128;        // K-blocked operands: a K tile of an operand is one contiguous 32 KiB block
    const long krow = blocked ? 48 : ld / 128;            // K tiles before wrapping
    for (int it = 0; it < iters; ++it) {
        const int st = (it & 1) * 65536;
        if (MODE & 1) {
            const long ko = (long)(it % krow) * (blocked ? 65536 : 128);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                glds16(a_src[i] + ko, smem + st + wave * 4096 + i * 1024);
                glds16(w_src[i] + ko, smem + st + 32768 + wave * 4096 + i * 1024);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    fr[i] = *reinterpret_cast<const bf16x8*>(smem + (st ^ 65536) + ((wave * 12 + i) * 2048 + half * 1024 + fo) % 65536);
            }
            if (MODE & 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[8 + j], fr[i], acc[i * 4 + j], 0, 0, 0);
            } else if (MODE & 2) {
                for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(fr[i]));
            }
        }
        if (MODE & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // one K tile may stay in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += (float)fr[i][0];
    out[blockIdx.x * 512 + tid] = s;
}

// The attention kernel's pattern: 256 threads, two workgroups per CU, per 64-key tile 8 KiB of K (contiguous rows) and 8 KiB of V^T
// (64 channel rows of 128 B, `vstride` bytes apart: S_pad * 2 in the product layout [B, H, 64, S_pad]; 128 = key-blocked).  The
// workgroups of an XCD stream the same head.
__global__ __launch_bounds__(256, 2) void katt(const char* __restrict__ src, float* out, int tiles, long vstride, long head_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7;
    const char* kb = src + (long)xcd * 2 * head_bytes;
    const char* vb = kb + head_bytes;
    const char* ks[2];
    const char* vs[2];
    for (int i = 0; i < 2; ++i) {
        const int L = (wave * 2 + i) * 64 + lane, r = L >> 3, c = L & 7;
        ks[i] = kb + (long)r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
        vs[i] = vb + (long)r * vstride + ((c ^ ((r >> 1) & 7)) << 4);
    }
    const long vtile = vstride == 128 ? 8192 : 128;     // bytes to the next 64-key tile of V^T
    for (int t = 0; t < tiles; ++t) {
        char* st = smem + (t & 1) * 16384 + wave * 2048;
        for (int i = 0; i < 2; ++i) {
            glds16(ks[i] + (long)t * 8192, st + i * 1024);
            glds16(vs[i] + (long)t * vtile, st + 8192 + i * 1024);
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + tid] = (float)smem[tid];
}

void run_att(const char* name, const char* src, float* out, long vstride, int tiles) {
    const long head_bytes = (long)tiles * 8192 + 64 * 128;      // K rows then V^T rows of one head (tiles * 64 keys)
    hipFuncSetAttribute((const void*)katt, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(katt, dim3(512), dim3(256), 32768, 0, src, out, tiles, vstride, head_bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(katt, dim3(512), dim3(256), 32768, 0, src, out, tiles, vstride, head_bytes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us_tile = ms * 1e3 / 5 / tiles;
        printf("{\"what\": \"%s\", \"vt_row_stride_B\": %ld, \"us_per_64_key_tile_per_WG\": %.3f, \"dma_GBps_per_CU\": %.1f, \"needed_at_1300_TFLOPs_us\": 0.83}\n", name,
               vstride, us_tile, 2 * 16384.0 / us_tile / 1e3);
    }
}

// The row-slab convolution's pattern (conv3d_cl_row16_k32_kernel): a slab = 514 voxels x 32 channels = LDS rows of 64 B, one voxel
// `vstride` bytes after the other (channels-last activations: C * 2 bytes; 64 = channel-blocked [C / 32][voxels][32]).  512 threads,
// one workgroup per CU, 33 pieces of 1 KiB (16 rows) per slab; `share` workgroups of an XCD read the same rows.
__global__ __launch_bounds__(512, 2) void kslab(const char* __restrict__ src, float* out, int slabs, long vstride, int share) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long row_bytes = 1024l * vstride;                               // one image row of 1024 voxels
    const char* base = src + ((long)xcd * 32 + slot / share) * 8 * row_bytes;
    for (int t = 0; t < slabs; ++t) {
        const char* rowp = base + (long)(t % 7) * row_bytes + (t / 7 % 4) * 64 * (vstride == 64 ? 0 : 1);
        char* st = smem + (t & 1) * 34816;
        for (int q = wave; q < 33; q += 8) {
            const int r = q * 16 + (lane >> 2), c = lane & 3;
            glds16(rowp + (long)r * vstride + ((c ^ ((r >> 1) & 3)) << 4), st + q * 1024);
        }
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + tid] = (float)smem[tid];
}

void run_slab(const char* name, const char* src, float* out, long vstride, int share) {
    const int slabs = 4000;
    hipFuncSetAttribute((const void*)kslab, hipFuncAttributeMaxDynamicSharedMemorySize, 69632);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kslab, dim3(256), dim3(512), 69632, 0, src, out, slabs / 4, vstride, share);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(kslab, dim3(256), dim3(512), 69632, 0, src, out, slabs, vstride, share);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 5 / slabs;
        printf("{\"what\": \"%s\", \"voxel_stride_B\": %ld, \"share\": %d, \"us_per_slab\": %.3f, \"dma_GBps_per_CU\": %.1f, \"mfma_time_of_a_slab_us\": 0.75}\n", name, vstride, share, us,
               33 * 1024.0 / us / 1e3);
    }
}

template <int MODE>
void run(const char* name, const char* src, float* out, long ld, int share, long wg_stride) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 131072, 0, src, out, iters / 4, ld, share, wg_stride);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 131072, 0, src, out, iters, ld, share, wg_stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us_it = ms * 1e3 / 5 / iters;
        printf("{\"what\": \"%s\", \"row_stride_B\": %ld, \"share\": %d, \"us_per_K_tile\": %.3f, \"dma_GBps_per_CU\": %.1f, \"mfma_TFLOPs_equiv\": %.0f}\n", name, ld, share,
               us_it, (MODE & 1) ? 65536.0 / us_it / 1e3 : 0.0, (MODE & 4) ? 256.0 * 2.0 * 256 * 256 * 64 / us_it / 1e6 : 0.0);
    }
}

int main() {
    const long bytes = 4l << 30;
    char* src;
    float* out;
    hipMalloc(&src, bytes);
    hipMemset(src, 0x11, bytes);
    hipMalloc(&out, 256 * 512 * 4);
    for (long ld : {128l, 6144l, 24576l}) {
        const long wg_stride = ld == 128 ? 48 * 65536 : 512 * ld;                 // one workgroup's 2 x 256 rows (K-blocked: 48 K tiles of 64 KiB)
        for (int share : {4, 1}) {
            if (256 * wg_stride > bytes) continue;
            run<1>("DMA only", src, out, ld, share, wg_stride);
            run<5>("DMA + 64 MFMA per wave", src, out, ld, share, wg_stride);
        }
    }
    run<4>("MFMA only", src, out, 6144, 4, 512 * 6144);
    for (int share : {1, 4}) {
        run_slab("conv slab, channels-last C = 128", src, out, 256, share);
        run_slab("conv slab, channels-last C = 256", src, out, 512, share);
        run_slab("conv slab, channel-blocked [C/32][voxels][32]", src, out, 64, share);
    }
    run_att("attention DMA pattern, V^T [64][S_pad]", src, out, 53504l * 2, 836);
    run_att("attention DMA pattern, V^T key-blocked", src, out, 128, 836);
    return 0;
}
