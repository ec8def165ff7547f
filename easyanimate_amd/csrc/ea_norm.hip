// HBM-bound row kernels: FP32 LayerNorm + adaLN modulation, RMSNorm, small-M linear, sinusoid.
// One wave (64 lanes) owns one row; each lane keeps its 16-byte vectors of the row in registers, so a
// row is read once and written once (roofline: 4 bytes/element, SURVEY.md 2.2 "1 read, 1 write").
#include "ea_common.h"

namespace {

constexpr int LN_MAXV = 16;  // up to 16 x (64 lanes x 8 elems) = 8192 columns

// reference: easyanimate/models/norm.py:16-26 (FP32LayerNorm) + :164-165 (modulation)
// y = LN(x) * gamma + beta, then * (1 + scale) + shift, applied as one multiply-add per element with
// A = gamma * (1 + scale), B = beta * (1 + scale) + shift.  A workgroup builds A and B once in LDS (the four parameter
// vectors are 8x the bytes of a row: re-reading them per row made the kernel L2-bound at 3.1 TB/s) and each of its four
// waves walks LN_ROWS rows, the next row's loads in flight while the current one is reduced and written.
constexpr int LN_ROWS = 8;
template <int NV>
__global__ __launch_bounds__(256) void layernorm_modulate_kernel(
    const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale, const float* __restrict__ shift,
    int64_t mod_stride, int rows, int dim, int64_t xbs, int64_t ybs, float eps) {
    extern __shared__ __attribute__((aligned(16))) float ln_ab[];   // A[dim] | B[dim]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int nvec = dim >> 3;
    {
        const float* sc = scale ? scale + b * mod_stride : nullptr;
        const float* sh = shift ? shift + b * mod_stride : nullptr;
        // the reference rounds the LN output to bf16 before the modulation (FP32LayerNorm .to(dtype)); we keep fp32
        // through the modulation: strictly closer to the fp32 oracle.
        for (int c = threadIdx.x * 4; c < dim; c += 1024) {
            f32x4 a = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
            if (gamma) {
                a = *reinterpret_cast<const f32x4*>(gamma + c);
                bb = *reinterpret_cast<const f32x4*>(beta + c);
            }
            if (sc) {
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(sc + c);
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(sh + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] *= 1.0f + s1[j];
                    bb[j] = bb[j] * (1.0f + s1[j]) + h1[j];
                }
            }
            *reinterpret_cast<f32x4*>(ln_ab + c) = a;
            *reinterpret_cast<f32x4*>(ln_ab + dim + c) = bb;
        }
    }
    __syncthreads();
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * LN_ROWS;
    if (row0 >= rows) return;
    const int nrow = rows - row0 < LN_ROWS ? (int)(rows - row0) : LN_ROWS;
    const unsigned short* xr = x + b * xbs + row0 * dim;
    unsigned short* yr = y + b * ybs + row0 * dim;

    u16x8 raw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 64 + lane;
        if (vi < nvec) raw[i] = *reinterpret_cast<const u16x8*>(xr + vi * 8);
    }
    for (int rr = 0; rr < nrow; ++rr, xr += dim, yr += dim) {
        float v[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = i * 64 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = vi < nvec ? bf16_bits_to_f32(raw[i][j]) : 0.f;
                s += v[i][j];
            }
        }
        if (rr + 1 < nrow) {   // next row in flight under this row's reductions
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int vi = i * 64 + lane;
                if (vi < nvec) raw[i] = *reinterpret_cast<const u16x8*>(xr + dim + vi * 8);
            }
        }
        const float mean = wave_sum(s) / (float)dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = i * 64 + lane;
            if (vi < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[i][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = i * 64 + lane;
            if (vi < nvec) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ln_ab + vi * 8), a1 = *reinterpret_cast<const f32x4*>(ln_ab + vi * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_ab + dim + vi * 8), b1 = *reinterpret_cast<const f32x4*>(ln_ab + dim + vi * 8 + 4);
                u16x8 out;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    out[j] = f32_to_bf16_bits((v[i][j] - mean) * rstd * a0[j] + b0[j]);
                    out[4 + j] = f32_to_bf16_bits((v[i][4 + j] - mean) * rstd * a1[j] + b1[j]);
                }
                *reinterpret_cast<u16x8*>(yr + vi * 8) = out;
            }
        }
    }
}

// Round 6: the same arithmetic, another ORDER of memory accesses (what GroupNorm-apply taught in round 5: these streaming passes are
// bound by how the resident waves are spread over the tensor, not by their arithmetic).  In layernorm_modulate_kernel a workgroup owns
// 32 consecutive rows and a launch's ~1 500 resident workgroups touch the whole 0.65 GB tensor at ~6 000 separate points for the
// kernel's whole life.  Here the workgroups SWEEP: trip k of workgroup g covers the four consecutive rows (k * G + g) * 4 .. + 3 (one
// per wave), so the G resident workgroups read and write one window of G * 4 rows (24 MiB at G = 1024, d = 3072) that moves through
// the tensor; the A / B table is built once per workgroup for all its trips; loads and stores are streaming (nontemporal: the 0.65 GB
// in and out pass through a 4 MiB L2 nobody re-reads them from).  Same operations on the same values in the same order per row:
// bit-identical to layernorm_modulate_kernel.
template <int NV, int NT>
__global__ __launch_bounds__(256) void layernorm_modulate_sweep_kernel(
    const unsigned short* __restrict__ x, unsigned short* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale, const float* __restrict__ shift,
    int64_t mod_stride, int rows, int dim, int64_t xbs, int64_t ybs, float eps) {
    extern __shared__ __attribute__((aligned(16))) float ln_ab[];   // A[dim] | B[dim]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int nvec = dim >> 3;
    {
        const float* sc = scale ? scale + b * mod_stride : nullptr;
        const float* sh = shift ? shift + b * mod_stride : nullptr;
        for (int c = threadIdx.x * 4; c < dim; c += 1024) {
            f32x4 a = {1.f, 1.f, 1.f, 1.f}, bb = {0.f, 0.f, 0.f, 0.f};
            if (gamma) {
                a = *reinterpret_cast<const f32x4*>(gamma + c);
                bb = *reinterpret_cast<const f32x4*>(beta + c);
            }
            if (sc) {
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(sc + c);
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(sh + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] *= 1.0f + s1[j];
                    bb[j] = bb[j] * (1.0f + s1[j]) + h1[j];
                }
            }
            *reinterpret_cast<f32x4*>(ln_ab + c) = a;
            *reinterpret_cast<f32x4*>(ln_ab + dim + c) = bb;
        }
    }
    __syncthreads();
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int64_t step = (int64_t)gridDim.x * 4;
    const unsigned short* xb = x + b * xbs;
    unsigned short* yb = y + b * ybs;
    typedef unsigned ln_u32x4 __attribute__((ext_vector_type(4)));
    auto ld = [&](const unsigned short* p) -> u16x8 {
        if (NT & 1) return __builtin_bit_cast(u16x8, __builtin_nontemporal_load(reinterpret_cast<const ln_u32x4*>(p)));
        return *reinterpret_cast<const u16x8*>(p);
    };

    u16x8 raw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 64 + lane;
        if (vi < nvec) raw[i] = ld(xb + row * dim + vi * 8);
    }
    for (; row < rows; row += step) {
        float v[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = i * 64 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = vi < nvec ? bf16_bits_to_f32(raw[i][j]) : 0.f;
                s += v[i][j];
            }
        }
        if (row + step < rows) {   // the wave's next row in flight under this row's reductions
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int vi = i * 64 + lane;
                if (vi < nvec) raw[i] = ld(xb + (row + step) * dim + vi * 8);
            }
        }
        const float mean = wave_sum(s) / (float)dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = i * 64 + lane;
            if (vi < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[i][j] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
        unsigned short* yr = yb + row * dim;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = i * 64 + lane;
            if (vi < nvec) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ln_ab + vi * 8), a1 = *reinterpret_cast<const f32x4*>(ln_ab + vi * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_ab + dim + vi * 8), b1 = *reinterpret_cast<const f32x4*>(ln_ab + dim + vi * 8 + 4);
                u16x8 out;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    out[j] = f32_to_bf16_bits((v[i][j] - mean) * rstd * a0[j] + b0[j]);
                    out[4 + j] = f32_to_bf16_bits((v[i][4 + j] - mean) * rstd * a1[j] + b1[j]);
                }
                if (NT & 2) __builtin_nontemporal_store(__builtin_bit_cast(ln_u32x4, out), reinterpret_cast<ln_u32x4*>(yr + vi * 8));
                else *reinterpret_cast<u16x8*>(yr + vi * 8) = out;
            }
        }
    }
}

// reference: easyanimate/models/norm.py:28-42 (EasyAnimateRMSNorm)
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const unsigned short* __restrict__ x,
                                                      unsigned short* __restrict__ y,
                                                      const float* __restrict__ w, int rows, int dim,
                                                      float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const unsigned short* xr = x + row * dim;
    unsigned short* yr = y + row * dim;
    const int nvec = dim >> 3;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 64 + lane;
        if (vi < nvec) {
            u16x8 raw = *reinterpret_cast<const u16x8*>(xr + vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = bf16_bits_to_f32(raw[j]);
                s += v[i][j] * v[i][j];
            }
        }
    }
    const float r = rsqrtf(wave_sum(s) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = i * 64 + lane;
        if (vi < nvec) {
            const int c0 = vi * 8;
            u16x8 out;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // weight * (x*rsqrt).to(input_dtype): round once to bf16, multiply in fp32, round again
                const float n = bf16_bits_to_f32(f32_to_bf16_bits(v[i][j] * r));
                out[j] = f32_to_bf16_bits(w[c0 + j] * n);
            }
            *reinterpret_cast<u16x8*>(yr + c0) = out;
        }
    }
}

// y[m,n] = act_out(sum_k act_in(x[m,k]) W[n,k] + bias[n]); one wave per output column n, all m rows.
// Weight-streaming GEMV: W is read exactly once (bf16x8 per lane), x lives in L1/L2.
template <int MAXM>
__global__ __launch_bounds__(256) void linear_small_m_kernel(const float* __restrict__ x,
                                                             const unsigned short* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ y, int m, int n, int k,
                                                             int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= n) return;
    const unsigned short* wr = W + (int64_t)col * k;
    float acc[MAXM];
#pragma unroll
    for (int i = 0; i < MAXM; ++i) acc[i] = 0.f;
    for (int k0 = lane * 8; k0 < k; k0 += 512) {
        u16x8 raw = *reinterpret_cast<const u16x8*>(wr + k0);
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = bf16_bits_to_f32(raw[j]);
#pragma unroll
        for (int i = 0; i < MAXM; ++i) {
            if (i < m) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float xv = x[(int64_t)i * k + k0 + j];
                    if (act_in == 1) xv = silu_f(xv);
                    acc[i] += xv * wv[j];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXM; ++i) {
        if (i < m) {
            float r = wave_sum(acc[i]);
            if (lane == 0) {
                if (bias) r += bias[col];
                if (act_out == 1) r = silu_f(r);
                y[(int64_t)i * n + col] = r;
            }
        }
    }
}

// reference: diffusers get_timestep_embedding as called at transformer3d.py:1399,1519
__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int batch,
                                         int dim, int round_bf16) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * half) return;
    const int b = i / half, c = i % half;
    const float f = expf(-logf(10000.0f) * (float)c / (float)half);
    const float a = t[b] * f;
    float cs = cosf(a), sn = sinf(a);
    if (round_bf16) {
        cs = bf16_bits_to_f32(f32_to_bf16_bits(cs));
        sn = bf16_bits_to_f32(f32_to_bf16_bits(sn));
    }
    out[(int64_t)b * dim + c] = cs;          // flip_sin_to_cos=True -> [cos | sin]
    out[(int64_t)b * dim + half + c] = sn;
}

int g_ln_wgs = 768;    // ea_set_option("ln_wgs", n): workgroups per batch element of the sweeping LayerNorm kernel (0: the round-1 kernel, 32 rows per workgroup)
int g_ln_nt = 2;       // ea_set_option("ln_nt", bits): 1 = streaming loads, 2 = streaming stores in the sweeping kernel

template <int NV>
int launch_ln(const ea_bf16* x, ea_bf16* y, const float* gamma, const float* beta, const float* scale,
              const float* shift, int64_t mod_stride, int batch, int rows, int dim, int64_t xbs,
              int64_t ybs, float eps, hipStream_t st) {
    if (g_ln_wgs > 0) {
        const int chunks = (rows + 3) / 4;
        dim3 grid(chunks < g_ln_wgs ? chunks : g_ln_wgs, batch);
        const size_t lds = 2 * dim * sizeof(float);
#define EA_LN_SWEEP(NT_)                                                                                                                \
    hipLaunchKernelGGL((layernorm_modulate_sweep_kernel<NV, NT_>), grid, dim3(256), lds, st, x, y, gamma, beta, scale, shift, mod_stride, \
                       rows, dim, xbs, ybs, eps)
        switch (g_ln_nt & 3) {
            case 0: EA_LN_SWEEP(0); break;
            case 1: EA_LN_SWEEP(1); break;
            case 2: EA_LN_SWEEP(2); break;
            default: EA_LN_SWEEP(3); break;
        }
#undef EA_LN_SWEEP
        return ea_check_launch("ea_layernorm_modulate_bf16");
    }
    dim3 grid((rows + 4 * LN_ROWS - 1) / (4 * LN_ROWS), batch);
    hipLaunchKernelGGL(layernorm_modulate_kernel<NV>, grid, dim3(256), 2 * dim * sizeof(float), st, x, y, gamma, beta, scale, shift,
                       mod_stride, rows, dim, xbs, ybs, eps);
    return ea_check_launch("ea_layernorm_modulate_bf16");
}

}  // namespace

int ea_ln_wgs_set(int v) { if (v < 0 || v > 65535) return -1; g_ln_wgs = v; return 0; }
int ea_ln_wgs_get() { return g_ln_wgs; }
int ea_ln_nt_set(int v) { if (v < 0 || v > 3) return -1; g_ln_nt = v; return 0; }
int ea_ln_nt_get() { return g_ln_nt; }

extern "C" int ea_layernorm_modulate_bf16(const ea_bf16* x, ea_bf16* y, const float* gamma, const float* beta,
                                          const float* scale, const float* shift, int64_t mod_stride,
                                          int batch, int rows, int dim, int64_t x_batch_stride,
                                          int64_t y_batch_stride, float eps, void* stream) {
    EA_REQUIRE(x && y, "ea_layernorm_modulate_bf16: null tensor");
    EA_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= LN_MAXV * 512, "ea_layernorm_modulate_bf16: dim %d unsupported", dim);
    EA_REQUIRE((gamma == nullptr) == (beta == nullptr), "ea_layernorm_modulate_bf16: gamma/beta must come together");
    EA_REQUIRE((scale == nullptr) == (shift == nullptr), "ea_layernorm_modulate_bf16: scale/shift must come together");
    EA_REQUIRE(batch > 0 && rows >= 0 && batch <= 65535, "ea_layernorm_modulate_bf16: bad batch/rows");
    EA_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)x | (uintptr_t)y) & 15) == 0 &&
                   mod_stride % 4 == 0 && x_batch_stride % 8 == 0 && y_batch_stride % 8 == 0,
               "ea_layernorm_modulate_bf16: pointers must be 16-byte aligned (mod_stride %% 4 == 0)");
    if (rows == 0) return EA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nv = (dim / 8 + 63) / 64;
#define EA_LN_CASE(N)                                                                                          \
    if (nv <= N)                                                                                               \
        return launch_ln<N>(x, y, gamma, beta, scale, shift, mod_stride, batch, rows, dim, x_batch_stride, \
                            y_batch_stride, eps, st);
    EA_LN_CASE(1) EA_LN_CASE(2) EA_LN_CASE(4) EA_LN_CASE(6) EA_LN_CASE(8) EA_LN_CASE(16)
#undef EA_LN_CASE
    return EA_ERR_ARG;
}

extern "C" int ea_rmsnorm_bf16(const ea_bf16* x, ea_bf16* y, const float* w, int rows, int dim, float eps,
                               void* stream) {
    EA_REQUIRE(x && y && w, "ea_rmsnorm_bf16: null tensor");
    EA_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= LN_MAXV * 512, "ea_rmsnorm_bf16: dim %d unsupported", dim);
    if (rows <= 0) return EA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nv = (dim / 8 + 63) / 64;
    dim3 grid((rows + 3) / 4);
    if (nv <= 2)
        hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, dim3(256), 0, st, x, y, w, rows, dim, eps);
    else if (nv <= 8)
        hipLaunchKernelGGL(rmsnorm_kernel<8>, grid, dim3(256), 0, st, x, y, w, rows, dim, eps);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<16>, grid, dim3(256), 0, st, x, y, w, rows, dim, eps);
    return ea_check_launch("ea_rmsnorm_bf16");
}

extern "C" int ea_linear_small_m(const float* x, const ea_bf16* W, const float* bias, float* y, int m, int n,
                                 int k, int act_in, int act_out, void* stream) {
    EA_REQUIRE(x && W && y, "ea_linear_small_m: null tensor");
    EA_REQUIRE(m > 0 && m <= 8, "ea_linear_small_m: m=%d must be in 1..8", m);
    EA_REQUIRE(k > 0 && k % 8 == 0 && n > 0, "ea_linear_small_m: k %% 8 != 0 or bad n");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + 3) / 4);
    if (m <= 2)
        hipLaunchKernelGGL(linear_small_m_kernel<2>, grid, dim3(256), 0, st, x, W, bias, y, m, n, k, act_in, act_out);
    else
        hipLaunchKernelGGL(linear_small_m_kernel<8>, grid, dim3(256), 0, st, x, W, bias, y, m, n, k, act_in, act_out);
    return ea_check_launch("ea_linear_small_m");
}

extern "C" int ea_timestep_sinusoid(const float* t, float* out, int batch, int dim, int round_bf16,
                                    void* stream) {
    EA_REQUIRE(t && out && batch > 0 && dim > 0 && dim % 2 == 0, "ea_timestep_sinusoid: bad arguments");
    const int n = batch * (dim / 2);
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out,
                       batch, dim, round_bf16);
    return ea_check_launch("ea_timestep_sinusoid");
}
