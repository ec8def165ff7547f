// HBM-bound VAE kernels on channels-last (NDHWC) activations: per-frame GroupNorm (+SiLU), row softmax for the
// mid-block spatial attention, NCDHW <-> NDHWC layout changes.
// reference: vaemodules/common.py:301-319 (per-frame GroupNorm under spatial_group_norm, eps 1e-6, SiLU),
// vaemodules/attention_processors.py:76-139 (softmax(QK^T * C^-1/2)), omnigen_enc_dec.py:258-265, 601-611.
#include "ea_common.h"

namespace {

// ---- GroupNorm statistics: deterministic two-level reduction (no atomics) -------------------------------------
// x [T, HW, C]; partial[t][blk][C/4][2] = (sum, sumsq) over the block's voxel slab for each 4-channel bundle.
__global__ __launch_bounds__(256) void gn_partial_kernel(const unsigned short* __restrict__ x, float* __restrict__ partial,
                                                         int64_t hw, int C, int nblk) {
    __shared__ float red[256][4];
    const int t = blockIdx.y, blk = blockIdx.x;
    const int nvec = C >> 3;                 // 8-channel vectors per voxel
    const int vcol = threadIdx.x % nvec;
    const int vrow = threadIdx.x / nvec;
    const int rows_per_it = 256 / nvec;      // nvec in {8,16,32,64} -> 32..4 voxels per iteration
    const int64_t per_blk = (hw + nblk - 1) / nblk;
    const int64_t v0 = (int64_t)blk * per_blk;
    int64_t v1 = v0 + per_blk;
    v1 = v1 < hw ? v1 : hw;
    const unsigned short* xf = x + (int64_t)t * hw * C;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    if (vrow < rows_per_it) {
        for (int64_t v = v0 + vrow; v < v1; v += rows_per_it) {
            const u16x8 raw = *reinterpret_cast<const u16x8*>(xf + v * C + vcol * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf16_bits_to_f32(raw[j]), b = bf16_bits_to_f32(raw[4 + j]);
                s0 += a; q0 += a * a;
                s1 += b; q1 += b * b;
            }
        }
    }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = q0; red[threadIdx.x][2] = s1; red[threadIdx.x][3] = q1;
    __syncthreads();
    if (threadIdx.x < nvec) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = 0; r < rows_per_it; ++r) {  // fixed order: bitwise reproducible
            const float* e = red[r * nvec + threadIdx.x];
            a0 += e[0]; a1 += e[1]; a2 += e[2]; a3 += e[3];
        }
        float* dst = partial + (((int64_t)t * nblk + blk) * (C >> 2) + threadIdx.x * 2) * 2;
        dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
    }
}

// stats[t][g] = (mean, rstd); one thread per (t, g), fixed summation order, fp64 combine
__global__ void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int T, int groups,
                                   int C, int nblk, int64_t hw, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * groups) return;
    const int t = i / groups, g = i % groups;
    const int cpg = C / groups, bundles = cpg >> 2;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const float* src = partial + (((int64_t)t * nblk + b) * (C >> 2) + g * bundles) * 2;
        for (int u = 0; u < bundles; ++u) {
            s += src[u * 2];
            q += src[u * 2 + 1];
        }
    }
    const double n = (double)hw * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var < 0 ? 0 : var;
    stats[i * 2] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// The same for the many small partial blocks a convolution epilogue writes (ea_conv3d_cl_stats_bf16: nblk is in the
// thousands): one workgroup per (t, g); thread i adds blocks i, i + 256, ... in that order (fp64), then a fixed tree.
__global__ __launch_bounds__(256) void gn_finalize_wide_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                               int groups, int C, int nblk, int64_t hw, float eps) {
    __shared__ double rs[256], rq[256];
    const int t = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups, bundles = cpg >> 2;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float* src = partial + (((int64_t)t * nblk + b) * (C >> 2) + g * bundles) * 2;
        for (int u = 0; u < bundles; ++u) {
            s += src[u * 2];
            q += src[u * 2 + 1];
        }
    }
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            rs[threadIdx.x] += rs[threadIdx.x + o];
            rq[threadIdx.x] += rq[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)hw * cpg;
        const double mean = rs[0] / n;
        double var = rq[0] / n - mean * mean;
        var = var < 0 ? 0 : var;
        stats[blockIdx.x * 2] = (float)mean;
        stats[blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// y = act((x - mean) * rstd * gamma + beta), act 1 = SiLU
__global__ __launch_bounds__(256) void gn_apply_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int64_t hw, int C, int groups,
                                                       int act, int64_t total_vec) {
    const int nvec = C >> 3;
    const int cpg = C / groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const int vc = (int)(i % nvec);
        const int64_t vox = i / nvec;
        const int t = (int)(vox / hw);
        const int c0 = vc * 8;
        const u16x8 raw = *reinterpret_cast<const u16x8*>(x + i * 8);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
        const float* st0 = stats + ((int64_t)t * groups + c0 / cpg) * 2;
        const float* st1 = stats + ((int64_t)t * groups + (c0 + 4) / cpg) * 2;
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = (bf16_bits_to_f32(raw[j]) - st0[0]) * st0[1] * g0[j] + b0[j];
            float b = (bf16_bits_to_f32(raw[4 + j]) - st1[0]) * st1[1] * g1[j] + b1[j];
            if (act == 1) {
                a = silu_f(a);
                b = silu_f(b);
            }
            o[j] = f32_to_bf16_bits(a);
            o[4 + j] = f32_to_bf16_bits(b);
        }
        *reinterpret_cast<u16x8*>(y + i * 8) = o;
    }
}

// Fast path of the same for C/8 dividing 256 (every V5 VAE layer: C = 64 .. 512): a thread keeps ONE 8-channel chunk
// and walks the voxels of ONE frame, so the per-(frame, channel) affine a = rstd*gamma, b = beta - mean*a is built once
// per thread (16 registers) and an element costs one fma (+ 4 ops of SiLU as x * rcp(1 + 2^(-x log2 e))); no integer
// division in the loop.  HBM-bound (4 B per element).
// Round 5: the activation is a template parameter (the run-time `act` cost a select per element and the SiLU arithmetic on layers
// that have none) and the arithmetic is packed fp32 on channel pairs (v_pk_fma / v_pk_mul / v_pk_add: the same IEEE operations per
// element, bit-identical): 353 -> ~210 instructions per 32 elements.  That alone changed nothing (the kernel runs as fast without
// SiLU as with it); what did is the order in which it walks memory, see the loop.
typedef float gn_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gn_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gn_u32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ gn_u32x4 gn_apply_chunk(const gn_u32x4 raw, const gn_f32x2 (&a)[4], const gn_f32x2 (&b)[4]) {
    gn_u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const gn_f32x2 xv = {__uint_as_float(raw[j] << 16), __uint_as_float(raw[j] & 0xffff0000u)};
        gn_f32x2 r = __builtin_elementwise_fma(xv, a[j], b[j]);
        if (ACT == 1) {
            const gn_f32x2 m = r * -1.4426950408889634f;
            gn_f32x2 e = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
            e = 1.0f + e;
            const gn_f32x2 q = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
            r = r * q;
        }
        o[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, gn_bf16x2));
    }
    return o;
}

// CB (round 6): the output is written CHANNEL-BLOCKED, [C / 32][T][hw][32] -- the layout the four-wave row-slab convolutions read
// their slabs from in 1 KiB pieces (conv3d_cl_row16_w4a_kernel<.., .., true>).  A thread's 16-byte chunk of 8 channels lands in the
// 64-byte record of its voxel inside its channel block: a wave's store is C / 32 runs of (2048 / C) consecutive voxels.
template <int ACT, bool CB = false>
__global__ __launch_bounds__(256) void gn_apply_frame_kernel(const unsigned short* __restrict__ x,
                                                             unsigned short* __restrict__ y,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int64_t hw, int C, int groups) {
    const int nvec = C >> 3;
    const int cpg = C / groups;
    const int vc = threadIdx.x % nvec, vr = threadIdx.x / nvec, rows = 256 / nvec;
    const int t = blockIdx.y, c0 = vc * 8;
    gn_f32x2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* st = stats + ((int64_t)t * groups + (c0 + j) / cpg) * 2;
        const float aj = st[1] * gamma[c0 + j];
        a[j >> 1][j & 1] = aj;
        b[j >> 1][j & 1] = beta[c0 + j] - st[0] * aj;
    }
    const unsigned short* xf = x + (int64_t)t * hw * C + c0;
    const int64_t ys = CB ? 32 : C;                       // elements between consecutive voxels of the output
    unsigned short* yf = CB ? y + (((int64_t)(vc >> 2) * gridDim.y + t) * hw) * 32 + (vc & 3) * 8 : y + (int64_t)t * hw * C + c0;
    // four voxels per trip: all four loads are in flight before the first is used (one 16-byte load per thread and trip left
    // the kernel latency-bound at 2.6 TB/s); streaming loads / stores -- nothing here is read twice.
    // a trip covers 4 * rows CONSECUTIVE voxels: a wave's four loads are 4 x 1 KiB side by side.  The first version of this loop
    // kept them a grid stride (16 MiB) apart -- four streams per wave: 4.3 TB/s at 4 x 1024^2 x 128 against 6.2 TB/s now (5.0 -> 6.3
    // at 512^2 x 256, 5.4 -> 5.95 at 256^2 x 512; same bytes out, profiles/r05v_gn_apply_contiguous_trips_ab.jsonl).  With or
    // without SiLU the rate is the same: the kernel is bound by how the memory system is addressed, not by its arithmetic.
    const int64_t grp = 4 * rows, ngrp = hw / grp;
    for (int64_t g = blockIdx.x; g < ngrp; g += gridDim.x) {
        const int64_t v = g * grp + vr;
        gn_u32x4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const gn_u32x4*>(xf + (v + u * rows) * C));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_nontemporal_store(gn_apply_chunk<ACT>(raw[u], a, b), reinterpret_cast<gn_u32x4*>(yf + (v + u * rows) * ys));
    }
    for (int64_t v = ngrp * grp + (int64_t)blockIdx.x * rows + vr; v < hw; v += (int64_t)gridDim.x * rows)
        *reinterpret_cast<gn_u32x4*>(yf + v * ys) = gn_apply_chunk<ACT>(*reinterpret_cast<const gn_u32x4*>(xf + v * C), a, b);
}

// ---- row softmax: y = softmax(x * scale) per row, bf16 in/out, fp32 math; one block per row --------------------
template <int NV, bool F32IN>  // NV 8-element vectors per thread kept in registers (cols <= NV * 2048)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const void* __restrict__ xv, unsigned short* __restrict__ y,
                                                           int cols, float scale_log2e) {
    __shared__ float redm[4], reds[4];
    const int64_t row = blockIdx.x;
    unsigned short* yr = y + row * cols;
    const int nvec = cols >> 3;
    float v[NV][8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 256 + threadIdx.x;
        if (vi < nvec) {
            if (F32IN) {
                const float* xr = reinterpret_cast<const float*>(xv) + row * cols + vi * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(xr), b = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[i][j] = a[j] * scale_log2e;
                    v[i][4 + j] = b[j] * scale_log2e;
                }
            } else {
                const u16x8 raw = *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(xv) + row * cols + vi * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = bf16_bits_to_f32(raw[j]) * scale_log2e;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[i][j]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 256 + threadIdx.x;
        if (vi < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = __builtin_amdgcn_exp2f(v[i][j] - mx);
                s += v[i][j];
            }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) reds[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (reds[0] + reds[1] + reds[2] + reds[3]);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 256 + threadIdx.x;
        if (vi < nvec) {
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16_bits(v[i][j] * inv);
            *reinterpret_cast<u16x8*>(yr + vi * 8) = o;
        }
    }
}

// ---- layout changes ----------------------------------------------------------------------------------------------
// src [C,T,H,W] (fp32 or bf16) -> dst bf16 [T,H,W,Cp] (channels >= C zero-filled); one thread per (voxel, channel)
template <bool BF16>
__global__ void ncdhw_to_ndhwc_kernel(const void* __restrict__ src, unsigned short* __restrict__ dst, int C, int Cp,
                                      int64_t vox) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vox * Cp) return;
    const int c = (int)(i % Cp);
    const int64_t v = i / Cp;
    unsigned short o = 0;
    if (c < C) {
        if (BF16) o = reinterpret_cast<const unsigned short*>(src)[(int64_t)c * vox + v];
        else o = f32_to_bf16_bits(reinterpret_cast<const float*>(src)[(int64_t)c * vox + v]);
    }
    dst[i] = o;
}

// src bf16 [T,H,W,Cs] -> dst [C,T,H,W] (first C channels); post: 0 none, 1 = clamp(-1,1) -> x/2+0.5 -> clamp(0,1)
// (pipeline_easyanimate.py:731,739).  One thread per (channel, voxel), voxel-fastest so writes are coalesced.
template <bool BF16>
__global__ void ndhwc_to_ncdhw_kernel(const unsigned short* __restrict__ src, void* __restrict__ dst, int C, int Cs,
                                      int64_t vox, int post) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vox * C) return;
    const int c = (int)(i / vox);
    const int64_t v = i % vox;
    float f = bf16_bits_to_f32(src[v * Cs + c]);
    if (post == 1) {
        f = fminf(fmaxf(f, -1.f), 1.f);
        f = fminf(fmaxf(f * 0.5f + 0.5f, 0.f), 1.f);
    }
    if (BF16) reinterpret_cast<unsigned short*>(dst)[i] = f32_to_bf16_bits(f);
    else reinterpret_cast<float*>(dst)[i] = f;
}

// ---- narrow-N 3x3x3 convolution, second half (decoder conv_out 128 -> 3) ---------------------------------------
// A 3x3x3 convolution with C_out <= 8 wastes a 128-wide MFMA tile on 3 columns.  It is split instead into
//   (1) ONE plain GEMM with the re-packed weight as the row operand: z[tap*C_out + co, v] = sum_ci w[co, ci, tap] * x[v, ci]
//       (27*C_out rows, K = C_in: 27x fewer MFMA flops than the tile-per-tap form, HBM-bound), fp32 out, VOXEL-MINOR:
//       one plane of `voxels` floats per (tap, co);
//   (2) this kernel: y[v, co] = bias[co] + sum_tap z[tap*C_out + co, v + offset(tap)] with the causal replicate padding
//       in time (frame index clamped at 0) and zero padding in space -- every z element is consumed exactly once.
// One thread per output voxel: the lanes of a wave read 64 consecutive floats of a plane per (tap, co) -- coalesced (the
// first version kept z voxel-major and spent 21 ms moving 128-byte lines for 12 useful bytes each).
template <int CO>
__global__ __launch_bounds__(256) void conv_tap_gather_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              unsigned short* __restrict__ y, int T, int H, int W, int64_t ld,
                                                              int c_pad) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)T * H * W;
    if (v >= total) return;
    const int w = (int)(v % W);
    const int h = (int)((v / W) % H);
    const int t = (int)(v / ((int64_t)W * H));
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        int ti = t + dt - 2;
        ti = ti < 0 ? 0 : ti;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hh = h + dh - 1;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int ww = w + dw - 1;
                if (ww < 0 || ww >= W) continue;
                const float* src = z + (int64_t)(((dt * 3 + dh) * 3 + dw) * CO) * ld + (((int64_t)ti * H + hh) * W + ww);
#pragma unroll
                for (int c = 0; c < CO; ++c) acc[c] += src[c * ld];
            }
        }
    }
    unsigned short* dst = y + v * c_pad;
#pragma unroll
    for (int c = 0; c < CO; ++c) dst[c] = f32_to_bf16_bits(acc[c]);
    for (int c = CO; c < c_pad; ++c) dst[c] = 0;
}

}  // namespace

extern "C" int ea_groupnorm_stats_bf16(const ea_bf16* x, float* partial, float* stats, int T, int64_t hw, int C,
                                       int groups, int nblk, float eps, void* stream) {
    EA_REQUIRE(x && partial && stats, "ea_groupnorm_stats_bf16: null tensor");
    EA_REQUIRE(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "ea_groupnorm_stats_bf16: C=%d unsupported", C);
    EA_REQUIRE(groups > 0 && C % groups == 0 && (C / groups) % 4 == 0, "ea_groupnorm_stats_bf16: channels per group must be a multiple of 4");
    EA_REQUIRE(T > 0 && T <= 65535 && hw > 0 && nblk > 0, "ea_groupnorm_stats_bf16: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, T), dim3(256), 0, st, x, partial, hw, C, nblk);
    int rc = ea_check_launch("ea_groupnorm_stats_bf16(partial)");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((T * groups + 127) / 128), dim3(128), 0, st, partial, stats, T, groups, C,
                       nblk, hw, eps);
    return ea_check_launch("ea_groupnorm_stats_bf16(finalize)");
}

extern "C" int ea_groupnorm_finalize_bf16(const float* partial, float* stats, int T, int64_t hw, int C, int groups, int nblk,
                                          float eps, void* stream) {
    EA_REQUIRE(partial && stats, "ea_groupnorm_finalize_bf16: null tensor");
    EA_REQUIRE(groups > 0 && C % groups == 0 && (C / groups) % 4 == 0 && T > 0 && nblk > 0 && hw > 0 && (int64_t)T * groups < (1ll << 31),
               "ea_groupnorm_finalize_bf16: bad sizes");
    hipLaunchKernelGGL(gn_finalize_wide_kernel, dim3((unsigned)(T * groups)), dim3(256), 0, (hipStream_t)stream, partial, stats,
                       groups, C, nblk, hw, eps);
    return ea_check_launch("ea_groupnorm_finalize_bf16");
}

extern "C" int ea_groupnorm_apply_bf16(const ea_bf16* x, ea_bf16* y, const float* stats, const float* gamma,
                                       const float* beta, int T, int64_t hw, int C, int groups, int act, void* stream) {
    EA_REQUIRE(x && y && stats && gamma && beta, "ea_groupnorm_apply_bf16: null tensor");
    EA_REQUIRE(C % 8 == 0 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0, "ea_groupnorm_apply_bf16: bad channels/groups");
    EA_REQUIRE((act & ~3) == 0, "ea_groupnorm_apply_bf16: act is 0 / 1 (SiLU), + 2 for a channel-blocked output");
    const int blocked = act >> 1;
    act &= 1;
    const int nvec = C / 8;
    EA_REQUIRE(!blocked || (C % 32 == 0 && nvec <= 256 && 256 % nvec == 0 && T <= 65535 && x != y),
               "ea_groupnorm_apply_bf16: the channel-blocked output needs C a multiple of 32 dividing 2048, T <= 65535, and y != x");
    if (blocked) {
        const int rows = 256 / nvec;
        int64_t bx = (hw + rows - 1) / rows;
        bx = bx > 4096 ? 4096 : bx;
        if (act == 1)
            hipLaunchKernelGGL((gn_apply_frame_kernel<1, true>), dim3((unsigned)bx, (unsigned)T), dim3(256), 0, (hipStream_t)stream, x, y, stats, gamma,
                               beta, hw, C, groups);
        else
            hipLaunchKernelGGL((gn_apply_frame_kernel<0, true>), dim3((unsigned)bx, (unsigned)T), dim3(256), 0, (hipStream_t)stream, x, y, stats, gamma,
                               beta, hw, C, groups);
        return ea_check_launch("ea_groupnorm_apply_bf16");
    }
    if (nvec <= 256 && 256 % nvec == 0 && T <= 65535) {
        const int rows = 256 / nvec;
        int64_t bx = (hw + rows - 1) / rows;
        // up to 4096 workgroups per frame, whatever T: the ~2000 resident at a time then sit in ONE frame and sweep it as one moving
        // window.  (The first version divided 16384 workgroups over the T frames: at T = 49 the resident ones spread over six frames --
        // 5.47 TB/s at 49 x 1024^2 x 128 against 5.99 with this; one workgroup per trip, cap 16384, costs the small shapes a third.
        // profiles/r05v_gn_apply_blocks_per_frame.jsonl)
        const int64_t cap = 4096;
        bx = bx > cap ? cap : bx;
        if (act == 1)
            hipLaunchKernelGGL(gn_apply_frame_kernel<1>, dim3((unsigned)bx, (unsigned)T), dim3(256), 0, (hipStream_t)stream, x, y, stats, gamma,
                               beta, hw, C, groups);
        else
            hipLaunchKernelGGL(gn_apply_frame_kernel<0>, dim3((unsigned)bx, (unsigned)T), dim3(256), 0, (hipStream_t)stream, x, y, stats, gamma,
                               beta, hw, C, groups);
        return ea_check_launch("ea_groupnorm_apply_bf16");
    }
    const int64_t total = (int64_t)T * hw * (C / 8);
    int64_t blocks = (total + 255) / 256;
    blocks = blocks > 65536 ? 65536 : blocks;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, stats, gamma, beta, hw,
                       C, groups, act, total);
    return ea_check_launch("ea_groupnorm_apply_bf16");
}

template <bool F32IN>
static int softmax_launch(const void* x, ea_bf16* y, int64_t rows, int cols, float scale, void* stream, const char* who) {
    EA_REQUIRE(x && y && rows > 0 && rows < (1ll << 31), "%s: bad arguments", who);
    EA_REQUIRE(cols > 0 && cols % 8 == 0 && cols <= 16 * 2048, "%s: cols=%d must be a multiple of 8 and <= 32768", who, cols);
    const float sl = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    const int nv = (cols / 8 + 255) / 256;
    if (nv <= 2) hipLaunchKernelGGL((softmax_rows_kernel<2, F32IN>), dim3((unsigned)rows), dim3(256), 0, st, x, y, cols, sl);
    else if (nv <= 8) hipLaunchKernelGGL((softmax_rows_kernel<8, F32IN>), dim3((unsigned)rows), dim3(256), 0, st, x, y, cols, sl);
    else hipLaunchKernelGGL((softmax_rows_kernel<16, F32IN>), dim3((unsigned)rows), dim3(256), 0, st, x, y, cols, sl);
    return ea_check_launch(who);
}

extern "C" int ea_softmax_rows_bf16(const ea_bf16* x, ea_bf16* y, int64_t rows, int cols, float scale, void* stream) {
    return softmax_launch<false>(x, y, rows, cols, scale, stream, "ea_softmax_rows_bf16");
}

extern "C" int ea_softmax_rows_f32in(const float* x, ea_bf16* y, int64_t rows, int cols, float scale, void* stream) {
    return softmax_launch<true>(x, y, rows, cols, scale, stream, "ea_softmax_rows_f32in");
}

extern "C" int ea_ncdhw_to_ndhwc(const void* src, ea_bf16* dst, int C, int C_pad, int64_t voxels, int src_is_bf16,
                                 void* stream) {
    EA_REQUIRE(src && dst && C > 0 && C_pad >= C && voxels > 0, "ea_ncdhw_to_ndhwc: bad arguments");
    const int64_t n = voxels * C_pad;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (src_is_bf16) hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, C, C_pad, voxels);
    else hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, C, C_pad, voxels);
    return ea_check_launch("ea_ncdhw_to_ndhwc");
}

extern "C" int ea_ndhwc_to_ncdhw(const ea_bf16* src, void* dst, int C, int C_src, int64_t voxels, int dst_is_bf16, int post,
                                 void* stream) {
    EA_REQUIRE(src && dst && C > 0 && C_src >= C && voxels > 0, "ea_ndhwc_to_ncdhw: bad arguments");
    const int64_t n = voxels * C;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dst_is_bf16) hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, C, C_src, voxels, post);
    else hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, C, C_src, voxels, post);
    return ea_check_launch("ea_ndhwc_to_ncdhw");
}

extern "C" int ea_conv3d_tap_gather_f32(const float* z, const float* bias, ea_bf16* y, int T, int H, int W, int64_t ld,
                                        int C_out, int C_pad, void* stream) {
    EA_REQUIRE(z && y, "ea_conv3d_tap_gather_f32: null tensor");
    const int64_t total = (int64_t)T * H * W;
    EA_REQUIRE(T > 0 && H > 0 && W > 0 && C_out >= 1 && C_out <= 4 && C_pad >= C_out && ld >= total,
               "ea_conv3d_tap_gather_f32: C_out must be 1..4, ld >= T*H*W");
    EA_REQUIRE((total + 255) / 256 < (1ll << 31), "ea_conv3d_tap_gather_f32: grid too large");
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    switch (C_out) {
        case 1: hipLaunchKernelGGL(conv_tap_gather_kernel<1>, grid, dim3(256), 0, st, z, bias, y, T, H, W, ld, C_pad); break;
        case 2: hipLaunchKernelGGL(conv_tap_gather_kernel<2>, grid, dim3(256), 0, st, z, bias, y, T, H, W, ld, C_pad); break;
        case 3: hipLaunchKernelGGL(conv_tap_gather_kernel<3>, grid, dim3(256), 0, st, z, bias, y, T, H, W, ld, C_pad); break;
        default: hipLaunchKernelGGL(conv_tap_gather_kernel<4>, grid, dim3(256), 0, st, z, bias, y, T, H, W, ld, C_pad); break;
    }
    ea_count("conv_narrow_gemm_tap_gather");
    return ea_check_launch("ea_conv3d_tap_gather_f32");
}
