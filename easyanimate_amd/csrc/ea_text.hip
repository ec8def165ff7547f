// The text-encoder step in front of the sampling loop (reference call site pipeline_easyanimate.py:438-447:
// `text_encoder(input_ids, attention_mask, output_hidden_states=True).hidden_states[-2]` with Qwen2-VL-7B in the slot -- a
// decoder-only LLM run ONCE per call over 256 padded prompt tokens).  The linear layers are ea_gemm_bf16, the norms ea_rmsnorm_bf16;
// this file holds what a Qwen2 decoder layer needs beyond them:
//   * rotate-half rotary embedding + head-major scatter (transformers' apply_rotary_pos_emb form, not the interleaved one of the DiT);
//   * causal, key-padding-masked, grouped-query attention for head_dim 64 / 128 at SHORT sequences (operands straight from
//     global memory / L2, one wave per 16 queries: 4 S^2 D H = 0.9 GFLOP per layer at S = 256 -- nothing here is worth an LDS stage);
//   * SiLU(gate) * up.
#include "ea_common.h"

namespace {

// src [rows = B*S][src_ld] (one of the q / k column blocks of a projection output), heads x D columns; dst [B][heads][S][D].
// cos / sin fp32 [B*S][D] (null: plain scatter).  out[d] = x[d] cos[d] - x[d + D/2] sin[d]          (d <  D/2)
//                                                  out[d] = x[d] cos[d] + x[d - D/2] sin[d]          (d >= D/2)
__global__ void rope_half_scatter_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                         const float* __restrict__ cos, const float* __restrict__ sin, int batch, int seq, int heads,
                                         int D, int64_t src_ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (row, head, d < D/2)
    const int half = D >> 1;
    if (idx >= (int64_t)batch * seq * heads * half) return;
    const int d = (int)(idx % half);
    const int h = (int)((idx / half) % heads);
    const int64_t row = idx / ((int64_t)half * heads);
    const int b = (int)(row / seq), s = (int)(row % seq);
    const unsigned short* x = src + row * src_ld + h * D;
    const float a = bf16_bits_to_f32(x[d]), c = bf16_bits_to_f32(x[d + half]);
    float lo = a, hi = c;
    if (cos) {
        const float* cr = cos + row * D;
        const float* sr = sin + row * D;
        lo = a * cr[d] - c * sr[d];
        hi = c * cr[d + half] + a * sr[d + half];
    }
    unsigned short* y = dst + (((int64_t)b * heads + h) * seq + s) * D;
    y[d] = f32_to_bf16_bits(lo);
    y[d + half] = f32_to_bf16_bits(hi);
}

__global__ void silu_mul_kernel(const unsigned short* __restrict__ g, const unsigned short* __restrict__ u,
                                unsigned short* __restrict__ out, int64_t rows, int cols, int64_t g_ld, int64_t u_ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // 8 elements per thread
    const int cv = cols >> 3;
    if (idx >= rows * cv) return;
    const int64_t r = idx / cv;
    const int c = (int)(idx % cv) * 8;
    const u16x8 gv = *reinterpret_cast<const u16x8*>(g + r * g_ld + c);
    const u16x8 uv = *reinterpret_cast<const u16x8*>(u + r * u_ld + c);
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16_bits(silu_f(bf16_bits_to_f32(gv[j])) * bf16_bits_to_f32(uv[j]));
    *reinterpret_cast<u16x8*>(out + r * (int64_t)cols + c) = o;
}

typedef unsigned u32x4_t_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pack2_t(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// q [B][Hq][S][D], k [B][Hkv][S][D], vt [B][Hkv][D][s_pad] (columns >= S finite), out [B][S][Hq*D].  Transposed evaluation as in the
// DiT kernels (S^T = K Q^T, O^T = V^T P^T on v_mfma_f32_16x16x32_bf16: a lane owns one query column, its 8 scores of a 32-key
// block are 8 consecutive keys = its k-slice of P^T).  One wave = 16 queries of one (batch, head); classic online softmax.
// Masks: key >= valid[b] (right-padded prompts), key > query (causal), as transformers' create_causal_mask combines them.
template <int D>
__global__ __launch_bounds__(64) void attention_small_kernel(const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K,
                                                             const unsigned short* __restrict__ Vt, unsigned short* __restrict__ O,
                                                             const int* __restrict__ valid, int seq, int s_pad, int hq, int hkv,
                                                             int causal, float scale_log2e) {
    constexpr int NS = D / 32, NT = D / 16;
    const int lane = threadIdx.x;
    const int lr = lane & 15, lq = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / hq, h = bh % hq, hk = h / (hq / hkv);
    const unsigned short* Qh = Q + ((int64_t)(b * hq + h) * seq) * D;
    const unsigned short* Kh = K + ((int64_t)(b * hkv + hk) * seq) * D;
    const unsigned short* Vh = Vt + ((int64_t)(b * hkv + hk) * D) * s_pad;
    const int q0 = blockIdx.x * 16;
    const int qr = q0 + lr < seq ? q0 + lr : seq - 1;
    bf16x8 qf[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qh + (int64_t)qr * D + ks * 32 + lq * 8);
    const int n_valid = valid ? valid[b] : seq;
    const int key_end = causal ? (q0 + 16 < seq ? q0 + 16 : seq) : seq;
    const int nb = (key_end + 31) / 32;
    f32x4 o[NT];
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;
    for (int blk = 0; blk < nb; ++blk) {
        const int key0 = blk * 32;
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            int kr = key0 + 8 * (lr >> 2) + 4 * kt + (lr & 3);
            kr = kr < seq ? kr : seq - 1;
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kh + (int64_t)kr * D + ks * 32 + lq * 8);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        float t[8];
        float bm = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + 8 * lq + 4 * kt + r;
                float v = s[kt][r] * scale_log2e;
                v = (key < n_valid && key < seq && (!causal || key <= q0 + lr)) ? v : -3.0e38f;
                t[kt * 4 + r] = v;
                bm = fmaxf(bm, v);
            }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float m_new = fmaxf(m_run, bm);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) o[dt] *= alpha;
        float pv[8];
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pv[e] = t[e] > -1.0e38f ? __builtin_amdgcn_exp2f(t[e] - m_run) : 0.f;
            ps += pv[e];
        }
        l_run += ps;
        u32x4_t_ pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack2_t(pv[2 * e], pv[2 * e + 1]);
        const bf16x8 pb = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vh + (int64_t)(16 * dt + lr) * s_pad + key0 + 8 * lq);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb, o[dt], 0, 0, 0);
        }
    }
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q0 + lr < seq) {
        unsigned short* dst = O + ((int64_t)b * seq + q0 + lr) * ((int64_t)hq * D) + h * D + lq * 4;
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[dt][r] * inv);
            *reinterpret_cast<bf16x4*>(dst + dt * 16) = ov;
        }
    }
}

}  // namespace

extern "C" int ea_rope_half_scatter_bf16(const ea_bf16* src, ea_bf16* dst, const float* cos, const float* sin, int batch, int seq,
                                         int heads, int head_dim, int64_t src_ld, void* stream) {
    EA_REQUIRE(src && dst, "ea_rope_half_scatter_bf16: null tensor");
    EA_REQUIRE((cos == nullptr) == (sin == nullptr), "ea_rope_half_scatter_bf16: cos / sin must come together");
    EA_REQUIRE(batch > 0 && seq > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 && src_ld >= (int64_t)heads * head_dim,
               "ea_rope_half_scatter_bf16: bad geometry");
    const int64_t n = (int64_t)batch * seq * heads * (head_dim / 2);
    hipLaunchKernelGGL(rope_half_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, cos, sin, batch, seq, heads, head_dim, src_ld);
    return ea_check_launch("ea_rope_half_scatter_bf16");
}

extern "C" int ea_silu_mul_bf16(const ea_bf16* gate, const ea_bf16* up, ea_bf16* out, int64_t rows, int cols, int64_t gate_ld,
                                int64_t up_ld, void* stream) {
    EA_REQUIRE(gate && up && out, "ea_silu_mul_bf16: null tensor");
    EA_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && gate_ld % 8 == 0 && up_ld % 8 == 0 &&
                   (((uintptr_t)gate | (uintptr_t)up | (uintptr_t)out) & 15) == 0,
               "ea_silu_mul_bf16: cols and the row strides must be multiples of 8, pointers 16-byte aligned");
    const int64_t n = rows * (cols / 8);
    if (n == 0) return EA_OK;
    hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gate, up,
                       (unsigned short*)out, rows, cols, gate_ld, up_ld);
    return ea_check_launch("ea_silu_mul_bf16");
}

extern "C" int ea_attention_causal_gqa_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out, const int* valid,
                                            int batch, int q_heads, int kv_heads, int seq, int s_pad, int head_dim, int causal,
                                            float scale, void* stream) {
    EA_REQUIRE(q && k && vt && out, "ea_attention_causal_gqa_bf16: null tensor");
    EA_REQUIRE(head_dim == 64 || head_dim == 128, "ea_attention_causal_gqa_bf16: head_dim %d (64 and 128 are built)", head_dim);
    EA_REQUIRE(batch > 0 && q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0 && seq > 0 && s_pad >= seq && s_pad % 32 == 0 &&
                   (int64_t)batch * q_heads <= 65535,
               "ea_attention_causal_gqa_bf16: bad geometry (s_pad must be a multiple of 32 covering seq; q_heads %% kv_heads == 0)");
    EA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0, "ea_attention_causal_gqa_bf16: 16-byte alignment");
    ea_count("attention_causal_gqa");
    const dim3 grid((unsigned)((seq + 15) / 16), (unsigned)(batch * q_heads));
    const float sl = scale * 1.4426950408889634f;
    if (head_dim == 128)
        hipLaunchKernelGGL(attention_small_kernel<128>, grid, dim3(64), 0, (hipStream_t)stream, q, k, vt, (unsigned short*)out, valid, seq,
                           s_pad, q_heads, kv_heads, causal, sl);
    else
        hipLaunchKernelGGL(attention_small_kernel<64>, grid, dim3(64), 0, (hipStream_t)stream, q, k, vt, (unsigned short*)out, valid, seq,
                           s_pad, q_heads, kv_heads, causal, sl);
    return ea_check_launch("ea_attention_causal_gqa_bf16");
}
