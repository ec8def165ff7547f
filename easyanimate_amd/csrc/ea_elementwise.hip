// Latent-space elementwise kernels: patchify gather, un-patchify scatter, CFG + Euler update.
#include "ea_common.h"

namespace {

template <bool BF16>
__device__ __forceinline__ float load_lat(const void* p, int64_t i) {
    if (BF16) return bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p)[i]);
    return reinterpret_cast<const float*>(p)[i];
}

// reference: easyanimate/models/transformer3d.py:1523-1531 (channel concat + Conv2d k=s=2 as im2col row)
// cols[b, (f,i,j), c*4 + di*2 + dj] ; one thread per (token, channel): reads 2 x 8-byte-ish pairs.
template <bool BF16>
__global__ void patchify_kernel(const void* __restrict__ lat, const void* __restrict__ extra,
                                unsigned short* __restrict__ cols, int c_lat, int c_extra, int F, int H, int W,
                                int k_pad) {
    const int h = H >> 1, w = W >> 1;
    const int64_t ntok = (int64_t)F * h * w;
    const int ctot = c_lat + c_extra;
    const int kq = k_pad >> 2;  // channel slots incl. zero padding
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= ntok * kq) return;
    const int c = (int)(idx % kq);
    const int64_t tok = idx / kq;
    const int j = (int)(tok % w);
    const int i = (int)((tok / w) % h);
    const int f = (int)(tok / ((int64_t)w * h));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < ctot) {
        const void* src = c < c_lat ? lat : extra;
        const int cc = c < c_lat ? c : c - c_lat;
        const int cn = c < c_lat ? c_lat : c_extra;
        const int64_t base = ((((int64_t)b * cn + cc) * F + f) * H + 2 * i) * W + 2 * j;
        v[0] = load_lat<BF16>(src, base);
        v[1] = load_lat<BF16>(src, base + 1);
        v[2] = load_lat<BF16>(src, base + W);
        v[3] = load_lat<BF16>(src, base + W + 1);
    }
    unsigned short* dst = cols + ((int64_t)b * ntok + tok) * k_pad + c * 4;
    ushort4 o;
    o.x = f32_to_bf16_bits(v[0]);
    o.y = f32_to_bf16_bits(v[1]);
    o.z = f32_to_bf16_bits(v[2]);
    o.w = f32_to_bf16_bits(v[3]);
    *reinterpret_cast<ushort4*>(dst) = o;
}

// reference: easyanimate/models/transformer3d.py:1683-1685
// tokens [b, (f,i,j), (c,di,dj)] -> out [b, c, f, 2i+di, 2j+dj]; one thread per output pixel pair (dj=0,1)
template <bool BF16>
__global__ void unpatchify_kernel(const unsigned short* __restrict__ tok, void* __restrict__ out, int C, int F,
                                  int h, int w) {
    const int H = 2 * h, W = 2 * w;
    const int64_t n = (int64_t)C * F * H * w;  // pairs
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (idx >= n) return;
    const int j = (int)(idx % w);
    const int y = (int)((idx / w) % H);
    const int f = (int)((idx / ((int64_t)w * H)) % F);
    const int c = (int)(idx / ((int64_t)w * H * F));
    const int i = y >> 1, di = y & 1;
    const int64_t t = ((int64_t)f * h + i) * w + j;
    const unsigned short* src = tok + (((int64_t)b * F * h * w + t) * (C * 4)) + c * 4 + di * 2;
    const int64_t o = ((((int64_t)b * C + c) * F + f) * H + y) * W + 2 * j;
    if (BF16) {
        unsigned short* d = reinterpret_cast<unsigned short*>(out);
        d[o] = src[0];
        d[o + 1] = src[1];
    } else {
        float* d = reinterpret_cast<float*>(out);
        d[o] = bf16_bits_to_f32(src[0]);
        d[o + 1] = bf16_bits_to_f32(src[1]);
    }
}

// reference: easyanimate/pipeline/pipeline_easyanimate.py:1102-1104 (CFG) and :1111 (scheduler.step)
template <bool BF16>
__global__ void cfg_euler_kernel(const void* __restrict__ v, void* __restrict__ x, int64_t n, float g, float dsigma,
                                 int do_cfg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float vu = load_lat<BF16>(v, i);
    float vv = vu;
    if (do_cfg) {
        const float vt = load_lat<BF16>(v, n + i);
        vv = vu + g * (vt - vu);
        // the reference combines in the model dtype: round like it does
        if (BF16) vv = bf16_bits_to_f32(f32_to_bf16_bits(vv));
    }
    const float xn = load_lat<BF16>(x, i) + dsigma * vv;
    if (BF16)
        reinterpret_cast<unsigned short*>(x)[i] = f32_to_bf16_bits(xn);
    else
        reinterpret_cast<float*>(x)[i] = xn;
}

// ---- guidance_rescale (rescale_noise_cfg, pipeline_easyanimate.py:100-112,1106-1108): the guided prediction is scaled
// by  r * std(v_text) / std(v_cfg) + (1 - r)  (unbiased std over all elements of the one sample) before the Euler step.
// Stage 1: per-block partial sums (sum t, sum t^2, sum c, sum c^2), fixed order; stage 2: one thread adds them in fp64;
// stage 3: the Euler kernel derives the factor from the four sums on the device (no host synchronisation).
template <bool BF16>
__global__ __launch_bounds__(256) void cfg_stats_partial_kernel(const void* __restrict__ v, int64_t n, float g,
                                                                float* __restrict__ partial) {
    __shared__ float red[4][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float vu = load_lat<BF16>(v, i), vt = load_lat<BF16>(v, n + i);
        float c = vu + g * (vt - vu);
        if (BF16) c = bf16_bits_to_f32(f32_to_bf16_bits(c));
        s[0] += vt;
        s[1] += vt * vt;
        s[2] += c;
        s[3] += c * c;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s[k] = wave_sum(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) partial[4 * blockIdx.x + threadIdx.x] =
        (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ void cfg_stats_final_kernel(const float* __restrict__ partial, int nblk, double* __restrict__ sums) {
    if (threadIdx.x < 4 && blockIdx.x == 0) {
        double a = 0.0;
        for (int i = 0; i < nblk; ++i) a += (double)partial[4 * i + threadIdx.x];
        sums[threadIdx.x] = a;
    }
}

template <bool BF16>
__global__ void cfg_rescale_euler_kernel(const void* __restrict__ v, void* __restrict__ x, int64_t n, float g, float dsigma,
                                         float rescale, const double* __restrict__ sums) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dn = (double)n;
    const double var_t = (sums[1] - sums[0] * sums[0] / dn) / (dn - 1.0);
    const double var_c = (sums[3] - sums[2] * sums[2] / dn) / (dn - 1.0);
    const float factor = rescale * (float)sqrt(var_t / var_c) + (1.0f - rescale);
    const float vu = load_lat<BF16>(v, i), vt = load_lat<BF16>(v, n + i);
    float c = vu + g * (vt - vu);
    if (BF16) c = bf16_bits_to_f32(f32_to_bf16_bits(c));
    const float xn = load_lat<BF16>(x, i) + dsigma * (factor * c);
    if (BF16)
        reinterpret_cast<unsigned short*>(x)[i] = f32_to_bf16_bits(xn);
    else
        reinterpret_cast<float*>(x)[i] = xn;
}

// ---- spatial tiling of the VAE wrapper (reference: autoencoder_magvit.py:319-337 blend_v / blend_h, :426-445 the lower-right corner)
// b[o, y, i] = a[o, a_n - extent + y, i] * (1 - y / extent) + b[o, y, i] * (y / extent)        y < extent, i < inner
// (blend_v: o = (b c t), y = row, i = column; blend_h: o = (b c t h), y = column, inner = 1).  fp32 arithmetic, one rounding.
template <bool BF16>
__global__ void tile_blend_kernel(const void* __restrict__ a, void* __restrict__ b, int64_t outer, int extent, int inner,
                                  int64_t a_os, int a_n, int64_t b_os) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= outer * extent * inner) return;
    const int i = (int)(idx % inner);
    const int y = (int)((idx / inner) % extent);
    const int64_t o = idx / ((int64_t)inner * extent);
    const int64_t ia = o * a_os + (int64_t)(a_n - extent + y) * inner + i, ib = o * b_os + (int64_t)y * inner + i;
    const float w = (float)y / (float)extent;
    const float r = load_lat<BF16>(a, ia) * (1.0f - w) + load_lat<BF16>(b, ib) * w;
    if (BF16) reinterpret_cast<unsigned short*>(b)[ib] = f32_to_bf16_bits(r);
    else reinterpret_cast<float*>(b)[ib] = r;
}

// dec[o, H - h + y, W - w + x] = wgt * q[o, y, x] + (1 - wgt) * dec[...],  wgt = min(x / (w - 1), y / (h - 1))   (linspace(0, 1, n))
template <bool BF16>
__global__ void tile_corner_kernel(const void* __restrict__ q, void* __restrict__ dec, int64_t outer, int h, int w, int H, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= outer * h * w) return;
    const int x = (int)(idx % w), y = (int)((idx / w) % h);
    const int64_t o = idx / ((int64_t)w * h);
    const int64_t id = (o * H + (H - h + y)) * W + (W - w + x);
    const float wx = w > 1 ? (float)x / (float)(w - 1) : 0.f, wy = h > 1 ? (float)y / (float)(h - 1) : 0.f;
    const float wg = fminf(wx, wy);
    const float r = wg * load_lat<BF16>(q, idx) + (1.0f - wg) * load_lat<BF16>(dec, id);
    if (BF16) reinterpret_cast<unsigned short*>(dec)[id] = f32_to_bf16_bits(r);
    else reinterpret_cast<float*>(dec)[id] = r;
}

}  // namespace

extern "C" int ea_patchify(const void* latents, const void* extra, ea_bf16* cols, int batch, int c_lat, int c_extra,
                           int frames, int height, int width, int k_pad, int lat_is_bf16, void* stream) {
    EA_REQUIRE(latents && cols, "ea_patchify: null tensor");
    EA_REQUIRE((extra != nullptr) == (c_extra > 0), "ea_patchify: extra and c_extra disagree");
    EA_REQUIRE(height % 2 == 0 && width % 2 == 0, "ea_patchify: height/width must be even");
    EA_REQUIRE(k_pad % 4 == 0 && k_pad >= 4 * (c_lat + c_extra), "ea_patchify: k_pad too small");
    const int64_t n = (int64_t)frames * (height / 2) * (width / 2) * (k_pad / 4);
    dim3 grid((unsigned)((n + 255) / 256), batch);
    hipStream_t st = (hipStream_t)stream;
    if (lat_is_bf16)
        hipLaunchKernelGGL(patchify_kernel<true>, grid, dim3(256), 0, st, latents, extra, cols, c_lat, c_extra, frames,
                           height, width, k_pad);
    else
        hipLaunchKernelGGL(patchify_kernel<false>, grid, dim3(256), 0, st, latents, extra, cols, c_lat, c_extra, frames,
                           height, width, k_pad);
    return ea_check_launch("ea_patchify");
}

extern "C" int ea_unpatchify(const ea_bf16* tokens, void* out, int batch, int channels, int frames, int h, int w,
                             int out_is_bf16, void* stream) {
    EA_REQUIRE(tokens && out, "ea_unpatchify: null tensor");
    const int64_t n = (int64_t)channels * frames * (2 * h) * w;
    dim3 grid((unsigned)((n + 255) / 256), batch);
    hipStream_t st = (hipStream_t)stream;
    if (out_is_bf16)
        hipLaunchKernelGGL(unpatchify_kernel<true>, grid, dim3(256), 0, st, tokens, out, channels, frames, h, w);
    else
        hipLaunchKernelGGL(unpatchify_kernel<false>, grid, dim3(256), 0, st, tokens, out, channels, frames, h, w);
    return ea_check_launch("ea_unpatchify");
}

extern "C" int ea_cfg_euler_step(const void* v, void* latents, int64_t n, float guidance, float dsigma, int do_cfg,
                                 int is_bf16, void* stream) {
    EA_REQUIRE(v && latents && n > 0, "ea_cfg_euler_step: bad arguments");
    dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (is_bf16)
        hipLaunchKernelGGL(cfg_euler_kernel<true>, grid, dim3(256), 0, st, v, latents, n, guidance, dsigma, do_cfg);
    else
        hipLaunchKernelGGL(cfg_euler_kernel<false>, grid, dim3(256), 0, st, v, latents, n, guidance, dsigma, do_cfg);
    return ea_check_launch("ea_cfg_euler_step");
}

extern "C" int ea_cfg_rescale_euler_step(const void* v, void* latents, int64_t n, float guidance, float dsigma,
                                         float guidance_rescale, float* partial, int nblk, double* sums, int is_bf16,
                                         void* stream) {
    EA_REQUIRE(v && latents && partial && sums && n > 1 && nblk > 0 && nblk <= 4096, "ea_cfg_rescale_euler_step: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((n + 255) / 256));
    if (is_bf16) {
        hipLaunchKernelGGL(cfg_stats_partial_kernel<true>, dim3(nblk), dim3(256), 0, st, v, n, guidance, partial);
        hipLaunchKernelGGL(cfg_stats_final_kernel, dim3(1), dim3(64), 0, st, partial, nblk, sums);
        hipLaunchKernelGGL(cfg_rescale_euler_kernel<true>, grid, dim3(256), 0, st, v, latents, n, guidance, dsigma, guidance_rescale, sums);
    } else {
        hipLaunchKernelGGL(cfg_stats_partial_kernel<false>, dim3(nblk), dim3(256), 0, st, v, n, guidance, partial);
        hipLaunchKernelGGL(cfg_stats_final_kernel, dim3(1), dim3(64), 0, st, partial, nblk, sums);
        hipLaunchKernelGGL(cfg_rescale_euler_kernel<false>, grid, dim3(256), 0, st, v, latents, n, guidance, dsigma, guidance_rescale, sums);
    }
    return ea_check_launch("ea_cfg_rescale_euler_step");
}

// ------------------------------------------------------------------------------------------------
// TeaCache on device (reference: easyanimate/models/transformer3d.py:90-121, 1564-1590, 1635).
// The reference moves two [B,N,d] tensors to the host every step and reduces them there; here the rel-L1 numerator
// and denominator are reduced on the device (8 bytes leave the GPU per step) and the residual stays resident.
// ------------------------------------------------------------------------------------------------
namespace {

// sums[0] += sum |bf16(cur - prev)| , sums[1] += sum |prev| over one contiguous chunk per block; the second stage adds
// the per-block partials in a fixed order (deterministic, bit-identical on every rank for identical data).
__global__ __launch_bounds__(256) void rel_l1_partial_kernel(const unsigned short* __restrict__ cur,
                                                             const unsigned short* __restrict__ prev, int64_t n,
                                                             float* __restrict__ partial) {
    __shared__ float red[2][4];
    const int64_t per = ((n / 8 + gridDim.x - 1) / gridDim.x) * 8;
    const int64_t lo = (int64_t)blockIdx.x * per;
    int64_t hi = lo + per;
    hi = hi < n ? hi : n;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = lo + (int64_t)threadIdx.x * 8; i + 8 <= hi; i += 256 * 8) {
        const u16x8 c = *reinterpret_cast<const u16x8*>(cur + i);
        const u16x8 p = *reinterpret_cast<const u16x8*>(prev + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float pf = bf16_bits_to_f32(p[e]);
            // torch: (cur - prev) is a bf16 tensor (rounded), abs() of it is exact
            s1 += fabsf(bf16_bits_to_f32(f32_to_bf16_bits(bf16_bits_to_f32(c[e]) - pf)));
            s2 += fabsf(pf);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void rel_l1_final_kernel(const float* __restrict__ partial, int nblk, double* __restrict__ sums) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < nblk; ++i) {
            a += (double)partial[2 * i];
            b += (double)partial[2 * i + 1];
        }
        sums[0] = a;
        sums[1] = b;
    }
}

template <int OP>  // 0: out = a - b    1: out = a + b      (bf16 in/out, fp32 arithmetic, one rounding: == torch bf16 ops)
__global__ __launch_bounds__(256) void bf16_binary_kernel(const unsigned short* __restrict__ a,
                                                          const unsigned short* __restrict__ b,
                                                          unsigned short* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const u16x8 x = reinterpret_cast<const u16x8*>(a)[i];
        const u16x8 y = reinterpret_cast<const u16x8*>(b)[i];
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xf = bf16_bits_to_f32(x[e]), yf = bf16_bits_to_f32(y[e]);
            o[e] = f32_to_bf16_bits(OP == 0 ? xf - yf : xf + yf);
        }
        reinterpret_cast<u16x8*>(out)[i] = o;
    }
}

// out[b, r, :] = res[b, r, :] + gate[b, :] * x[b, r, :]   (bf16 in/out, fp32 fma, one rounding)
__global__ __launch_bounds__(256) void gated_residual_kernel(const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ res,
                                                             const float* __restrict__ gate, unsigned short* __restrict__ out,
                                                             int64_t per_batch8, int dim8, int64_t gate_bs, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_batch8;
        const int c8 = (int)(i % dim8);
        const u16x8 xv = reinterpret_cast<const u16x8*>(x)[i];
        const u16x8 rv = reinterpret_cast<const u16x8*>(res)[i];
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + b * gate_bs + c8 * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gate + b * gate_bs + c8 * 8 + 4);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = f32_to_bf16_bits(__builtin_fmaf(g0[e], bf16_bits_to_f32(xv[e]), bf16_bits_to_f32(rv[e])));
            o[4 + e] = f32_to_bf16_bits(__builtin_fmaf(g1[e], bf16_bits_to_f32(xv[4 + e]), bf16_bits_to_f32(rv[4 + e])));
        }
        reinterpret_cast<u16x8*>(out)[i] = o;
    }
}

}  // namespace

extern "C" int ea_gated_residual_bf16(const ea_bf16* x, const ea_bf16* res, const float* gate, ea_bf16* out, int batch,
                                      int64_t rows, int dim, int64_t gate_batch_stride, void* stream) {
    EA_REQUIRE(x && res && gate && out, "ea_gated_residual_bf16: null tensor");
    EA_REQUIRE(batch > 0 && rows >= 0 && dim > 0 && dim % 8 == 0 && gate_batch_stride % 4 == 0, "ea_gated_residual_bf16: bad sizes");
    EA_REQUIRE((((uintptr_t)x | (uintptr_t)res | (uintptr_t)gate | (uintptr_t)out) & 15) == 0, "ea_gated_residual_bf16: pointers must be 16-byte aligned");
    const int64_t n8 = (int64_t)batch * rows * dim / 8;
    if (n8 == 0) return EA_OK;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gated_residual_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, res, gate, out,
                       rows * dim / 8, dim / 8, gate_batch_stride, n8);
    return ea_check_launch("ea_gated_residual_bf16");
}

extern "C" int ea_teacache_rel_l1_bf16(const ea_bf16* cur, const ea_bf16* prev, int64_t n, float* partial, int nblk,
                                       double* sums, void* stream) {
    EA_REQUIRE(cur && prev && partial && sums, "ea_teacache_rel_l1_bf16: null tensor");
    EA_REQUIRE(n > 0 && n % 8 == 0 && nblk > 0 && nblk <= 65535, "ea_teacache_rel_l1_bf16: n must be a positive multiple of 8");
    EA_REQUIRE(((uintptr_t)cur | (uintptr_t)prev) % 16 == 0, "ea_teacache_rel_l1_bf16: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rel_l1_partial_kernel, dim3(nblk), dim3(256), 0, st, cur, prev, n, partial);
    hipLaunchKernelGGL(rel_l1_final_kernel, dim3(1), dim3(64), 0, st, partial, nblk, sums);
    return ea_check_launch("ea_teacache_rel_l1_bf16");
}

// ---- SWA scan orders: V^T of a head re-ordered along its token axis by an AXIS PERMUTATION of the (f, h, w) token grid
// (processor.py:400-417) -- dst[c, p] = src[c, col_off + token(p)] for all 64 channels.  A key is a COLUMN of V^T, so this is a
// transposition-like move of 2-byte elements; a gather by token index would pull a 64-byte line per element.  It is done as
// tiled transposes through LDS instead: the tile spans 32 consecutive w (contiguous in the source) x 64 consecutive values of
// the destination's fastest axis y (contiguous in the destination) at a fixed third coordinate z, 8 channels at a time --
// 64-byte runs on the read side, 128-byte runs on the write side.  Orders whose fastest axis is w are row copies (same kernel,
// y := the middle axis, the tile is then written the way it was read).
struct PermuteArgs {
    const unsigned short* src;
    unsigned short* dst;
    const int* head_order;     // [heads]: index into the six axis orders (f h w), (f w h), (h f w), (h w f), (w f h), (w h f)
    int heads, F, Hh, Ww, src_pad, dst_pad, col_off;
};
__global__ __launch_bounds__(256) void permute_cols_kernel(PermuteArgs a) {
    __shared__ unsigned short tile[8][64][34];
    const int bh = blockIdx.z, cs = blockIdx.y;                 // 8-channel slab
    const int ord = a.head_order[bh % a.heads];
    const int o0 = ord >> 1;                                     // orders in lexicographic order of (o0, o1, o2)
    const int o1 = (ord & 1) ? (o0 == 2 ? 1 : 2) : (o0 == 0 ? 1 : 0);
    const int o2 = 3 - o0 - o1;
    const int dim[3] = {a.F, a.Hh, a.Ww};
    // y = the destination's fastest axis unless that is w itself (then the middle axis), z = the third one
    const int ya = o2 == 2 ? o1 : o2;
    const int za = 3 - 2 - ya;
    const int Dy = dim[ya], Dz = dim[za];
    const int ty = (Dy + 63) / 64, tw = (a.Ww + 31) / 32;
    int t = blockIdx.x;
    const int w0 = (t % tw) * 32; t /= tw;
    const int y0 = (t % ty) * 64; t /= ty;
    const int z = t;
    if (z >= Dz) return;
    const int64_t sbase = ((int64_t)bh * 64 + cs * 8) * a.src_pad + a.col_off;
    const int64_t dbase = ((int64_t)bh * 64 + cs * 8) * a.dst_pad;
    auto token = [&](int y, int w) {          // natural token index (f h w)
        int c[3];
        c[ya] = y; c[za] = z; c[2] = w;
        return (c[0] * a.Hh + c[1]) * a.Ww + c[2];
    };
    auto position = [&](int y, int w) {       // scan position in the order (o0 o1 o2)
        int c[3];
        c[ya] = y; c[za] = z; c[2] = w;
        return (c[o0] * dim[o1] + c[o1]) * dim[o2] + c[o2];
    };
    const int tid = threadIdx.x;
    {   // read: lane -> w (32 consecutive source elements), 8 rows of y per pass
        const int wl = tid & 31, yl = tid >> 5;
        const int w = w0 + wl;
#pragma unroll 2
        for (int yy = yl; yy < 64; yy += 8) {
            const int y = y0 + yy;
            if (y < Dy && w < a.Ww) {
                const int64_t n = token(y, w);
#pragma unroll
                for (int c = 0; c < 8; ++c) tile[c][yy][wl] = a.src[sbase + (int64_t)c * a.src_pad + n];
            }
        }
    }
    __syncthreads();
    if (o2 == 2) {   // w is the destination's fastest axis too: write the rows the way they were read
        const int wl = tid & 31, yl = tid >> 5;
        const int w = w0 + wl;
        for (int yy = yl; yy < 64; yy += 8) {
            const int y = y0 + yy;
            if (y < Dy && w < a.Ww) {
                const int64_t p = position(y, w);
#pragma unroll
                for (int c = 0; c < 8; ++c) a.dst[dbase + (int64_t)c * a.dst_pad + p] = tile[c][yy][wl];
            }
        }
    } else {         // lane -> y (64 consecutive destination elements), 4 columns of w per pass
        const int yl = tid & 63, wq = tid >> 6;
        const int y = y0 + yl;
        for (int ww = wq; ww < 32; ww += 4) {
            const int w = w0 + ww;
            if (y < Dy && w < a.Ww) {
                const int64_t p = position(y, w);
#pragma unroll
                for (int c = 0; c < 8; ++c) a.dst[dbase + (int64_t)c * a.dst_pad + p] = tile[c][yl][ww];
            }
        }
    }
}

extern "C" int ea_permute_cols_bf16(const ea_bf16* src, ea_bf16* dst, const int* head_order, int batch, int heads, int frames, int height,
                                    int width, int src_pad, int dst_pad, int col_off, void* stream) {
    EA_REQUIRE(src && dst && head_order, "ea_permute_cols_bf16: null tensor");
    EA_REQUIRE(batch > 0 && heads > 0 && frames > 0 && height > 0 && width > 0 && col_off >= 0, "ea_permute_cols_bf16: bad sizes");
    const int64_t n = (int64_t)frames * height * width;
    EA_REQUIRE(n < (1ll << 30) && col_off + n <= src_pad && n <= dst_pad, "ea_permute_cols_bf16: the token grid must fit the source / destination rows");
    EA_REQUIRE((int64_t)batch * heads <= 65535, "ea_permute_cols_bf16: grid too large");
    PermuteArgs a;
    a.src = src; a.dst = dst; a.head_order = head_order; a.heads = heads; a.F = frames; a.Hh = height; a.Ww = width;
    a.src_pad = src_pad; a.dst_pad = dst_pad; a.col_off = col_off;
    // tiles: z x ceil(Dy / 64) x ceil(W / 32) -- bounded by the largest case over the six orders (blocks beyond a head's own count return)
    const int dims[3] = {frames, height, width};
    int64_t tiles = 0;
    for (int ya = 0; ya < 2; ++ya) {
        const int za = 1 - ya;
        const int64_t t = (int64_t)dims[za] * ((dims[ya] + 63) / 64) * ((width + 31) / 32);
        tiles = t > tiles ? t : tiles;
    }
    EA_REQUIRE(tiles < (1ll << 31), "ea_permute_cols_bf16: grid too large");
    ea_count("permute_cols");
    hipLaunchKernelGGL(permute_cols_kernel, dim3((unsigned)tiles, 8, (unsigned)(batch * heads)), dim3(256), 0, (hipStream_t)stream, a);
    return ea_check_launch("ea_permute_cols_bf16");
}

extern "C" int ea_bf16_binary(const ea_bf16* a, const ea_bf16* b, ea_bf16* out, int64_t n, int op, void* stream) {
    EA_REQUIRE(a && b && out, "ea_bf16_binary: null tensor");
    EA_REQUIRE(n >= 0 && n % 8 == 0 && (op == 0 || op == 1), "ea_bf16_binary: n must be a multiple of 8, op 0 (sub) or 1 (add)");
    EA_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0, "ea_bf16_binary: pointers must be 16-byte aligned");
    if (n == 0) return EA_OK;
    const int64_t n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (op == 0)
        hipLaunchKernelGGL(bf16_binary_kernel<0>, dim3(blocks), dim3(256), 0, st, a, b, out, n8);
    else
        hipLaunchKernelGGL(bf16_binary_kernel<1>, dim3(blocks), dim3(256), 0, st, a, b, out, n8);
    return ea_check_launch("ea_bf16_binary");
}

extern "C" int ea_tile_blend(const void* a, void* b, int bf16, int64_t outer, int extent, int inner, int64_t a_outer_stride, int a_n,
                             int64_t b_outer_stride, void* stream) {
    EA_REQUIRE(a && b, "ea_tile_blend: null tensor");
    EA_REQUIRE(outer >= 0 && extent >= 0 && inner > 0 && a_n >= extent, "ea_tile_blend: bad geometry (outer %lld, extent %d, inner %d, a_n %d)",
               (long long)outer, extent, inner, a_n);
    const int64_t n = outer * extent * inner;
    if (n == 0) return EA_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (bf16) hipLaunchKernelGGL(tile_blend_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, b, outer, extent, inner, a_outer_stride, a_n, b_outer_stride);
    else hipLaunchKernelGGL(tile_blend_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, b, outer, extent, inner, a_outer_stride, a_n, b_outer_stride);
    return ea_check_launch("ea_tile_blend");
}

extern "C" int ea_tile_corner_blend(const void* q, void* dec, int bf16, int64_t outer, int h, int w, int H, int W, void* stream) {
    EA_REQUIRE(q && dec, "ea_tile_corner_blend: null tensor");
    EA_REQUIRE(outer >= 0 && h > 0 && w > 0 && h <= H && w <= W, "ea_tile_corner_blend: bad geometry (%d x %d inside %d x %d)", h, w, H, W);
    const int64_t n = outer * h * w;
    if (n == 0) return EA_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (bf16) hipLaunchKernelGGL(tile_corner_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q, dec, outer, h, w, H, W);
    else hipLaunchKernelGGL(tile_corner_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, q, dec, outer, h, w, H, W);
    return ea_check_launch("ea_tile_corner_blend");
}
