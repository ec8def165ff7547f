// Single-head, head_dim 512 flash attention for the VAE mid block (vaemodules/attention.py:391-423 SpatialAttention +
// attention_processors.py:76-139: per latent frame, softmax(Q K^T / sqrt(512)) V over all H x W tokens -- n = 16 384 at 1024^2).
// Replaces Q K^T GEMM -> 1 GiB fp32 logits -> row softmax -> P V GEMM (three launches and ~3 GiB of HBM traffic per frame).
//
// Mapping (gfx950, wave = 64):
//   * workgroup = 8 waves x 16 queries = 128 queries of one frame, all 512 channels; keys in blocks of 32.
//   * everything transposed, as in the head_dim-64 kernel (ea_attention_v3.inc): S^T = K . Q^T and O^T = V^T . P^T on
//     v_mfma_f32_16x16x32_bf16, so a lane owns ONE query column: softmax state is lane-local, and the exponentiated scores are
//     the B operand of the PV MFMA without leaving registers.  K rows are fed in the order 8*(i>>2) + 4*kt + (i&3), which
//     makes the 8 scores a lane holds for its query (2 tiles x 4 rows) the 8 CONSECUTIVE keys 8*lq .. 8*lq+7 of the block --
//     exactly its k-slice of P^T.
//   * per wave and block: 32 QK MFMAs (2 key tiles x 16 k32 steps over the 512 dims) + 32 PV MFMAs (32 channel tiles);
//     registers: O^T 128, Q fragments 64 (read once), S 8.
//   * LDS (128 KiB): two stages of { K block [32 keys][512] = 32 KiB, V^T block [512][32 keys] = 32 KiB }, staged by LDS-DMA:
//     a K piece is one key row (1 KiB; 16-byte chunk c stored at c ^ f(row), f = 4*((row>>3)&3) + (row&3): the 16 rows of a
//     fragment read land on 16 different slots of the 256-byte bank row), a V^T piece is 16 channel rows x 64 B stored
//     chunk-major (position 16*chunk + row: a fragment read of 16 consecutive channels is 256 contiguous bytes).
//   * one barrier per block; block b+1 is requested right behind it and has a whole block time to land.
//   * online softmax with a LAZY shift: the running shift m of a query only moves (and O, l are only rescaled, a wave-uniform
//     branch) when a block maximum exceeds it by more than 8 (log2 units) -- P <= 2^8, l <= n * 2^8: exact in fp32, softmax
//     is shift-invariant -- so the hot loop has no per-block rescale of the 128 accumulator registers.
// Bound: every fragment (1 KiB) feeds ONE MFMA (16 queries per wave), i.e. 64 KiB of LDS reads per 64 MFMAs per wave, twice
// what the LDS port delivers under the MFMA rate: LDS-bound at <= 50 % of the MFMA peak (32 queries per wave would need the
// 256-register O^T in AGPRs).  That is still well ahead of the three-GEMM route it replaces.
#include "ea_common.h"

namespace {

constexpr int D5 = 512, KB5 = 32;
constexpr int K_TILE5 = KB5 * D5 * 2;      // 32 KiB
constexpr int V_TILE5 = D5 * KB5 * 2;      // 32 KiB
constexpr int STAGE5 = K_TILE5 + V_TILE5;  // 64 KiB
constexpr int LDS5 = 2 * STAGE5;           // 128 KiB

__device__ __forceinline__ void glds16_5(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned pack2_5(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
typedef unsigned u32x4_5 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void attention_d512_kernel(
    const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K, const unsigned short* __restrict__ Vt,
    unsigned short* __restrict__ O, int n_q, int n_keys, int n_kpad, int64_t q_fs, int64_t k_fs, int64_t v_fs, int64_t o_fs,
    float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int f = blockIdx.y;
    const unsigned short* Qf = Q + f * q_fs;
    const unsigned short* Kf = K + f * k_fs;
    const unsigned short* Vf = Vt + f * v_fs;
    unsigned short* Of = O + f * o_fs;
    const int q0 = blockIdx.x * 128 + wave * 16;

    // ---- Q fragments (B operand of S^T = K . Q^T): query q0 + lr, dims 32*ks + 8*lq ..
    bf16x8 qf[16];
    {
        int qr = q0 + lr;
        qr = qr < n_q ? qr : n_q - 1;
        const unsigned short* qrow = Qf + (int64_t)qr * D5 + lq * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + ks * 32);
    }

    // ---- DMA sources of this wave's pieces: K rows 4w .. 4w+3 of a block, V^T channel groups 4w .. 4w+3
    const unsigned short* ksrc[4];
    const unsigned short* vsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i;                                  // key row of the block / channel group
        const int fx = (((p >> 3) & 3) << 2) | (p & 3);
        ksrc[i] = Kf + (int64_t)p * D5 + ((lane ^ fx) << 3);        // + key0 * 512
        vsrc[i] = Vf + (int64_t)(p * 16 + (lane & 15)) * n_kpad + ((lane >> 4) << 3);   // + key0
    }
    auto issue = [&](int b, int stage) {
        char* ks_ = smem + stage * STAGE5 + wave * 4096;
        char* vs_ = smem + stage * STAGE5 + K_TILE5 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16_5(ksrc[i] + (int64_t)b * (KB5 * D5), ks_ + i * 1024);
            glds16_5(vsrc[i] + b * KB5, vs_ + i * 1024);
        }
    };

    // ---- fragment addresses inside a stage
    unsigned k_row[2];     // byte offset of this lane's K row for key tile kt
    {
        const int a = lr >> 2, bq = lr & 3;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) k_row[kt] = (unsigned)(8 * a + 4 * kt + bq) * 1024u;
    }
    const unsigned k_fx = (unsigned)(((lr >> 2) << 2) | (lr & 3));
    const unsigned v_off = K_TILE5 + (unsigned)(lq * 16 + lr) * 16u;   // + dt * 1024

    f32x4 o[32];
#pragma unroll
    for (int dt = 0; dt < 32; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;

    const int nb = (n_keys + KB5 - 1) / KB5;
    issue(0, 0);
    for (int b = 0; b < nb; ++b) {
        const int st = b & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // block b has landed everywhere; every wave is done with the other stage
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < nb) issue(b + 1, st ^ 1);
        const char* sb = smem + st * STAGE5;

        // ---- S^T (2 key tiles x 16 queries) = K . Q^T over the 512 dims
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const unsigned coff = (((unsigned)(4 * ks + lq)) ^ k_fx) << 4;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_row[kt] + coff);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        // ---- softmax of the lane's 8 scores (keys key0 + 8*lq + 4*kt + r of query lr), lazy shift
        float t[8];
        const int key_base = b * KB5 + 8 * lq;
        float bm = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][r] * scale_log2e;
                v = (key_base + 4 * kt + r) < n_keys ? v : -3.0e38f;
                t[kt * 4 + r] = v;
                bm = fmaxf(bm, v);
            }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        if (__any(bm > m_run + 8.0f)) {       // rare after the first blocks: move the shift, rescale O and l once
            const float m_new = fmaxf(m_run, bm);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 32; ++dt) o[dt] *= alpha;
            m_run = m_new;
        }
        float ps = 0.f;
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pv[e] = __builtin_amdgcn_exp2f(t[e] - m_run);
            ps += pv[e];
        }
        l_run += ps;
        u32x4_5 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack2_5(pv[2 * e], pv[2 * e + 1]);
        const bf16x8 pb = __builtin_bit_cast(bf16x8, pk);
        // ---- O^T (32 channel tiles x 16 queries) += V^T . P^T
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sb + v_off + dt * 1024);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb, o[dt], 0, 0, 0);
        }
    }

    // ---- normalise and store: lane -> query q0 + lr, channels 16*dt + 4*lq + r
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + lr;
    if (qr < n_q) {
        unsigned short* dst = Of + (int64_t)qr * D5 + lq * 4;
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[dt][r] * inv);
            *reinterpret_cast<bf16x4*>(dst + dt * 16) = ov;
        }
    }
}

}  // namespace

extern "C" int ea_attention_d512_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out, int frames, int n_q,
                                          int n_keys, int n_kpad, int64_t q_frame_stride, int64_t k_frame_stride,
                                          int64_t vt_frame_stride, int64_t out_frame_stride, float scale, void* stream) {
    EA_REQUIRE(q && k && vt && out, "ea_attention_d512_fwd_bf16: null tensor");
    EA_REQUIRE(frames > 0 && frames <= 65535 && n_q > 0 && n_keys > 0 && n_keys <= n_kpad && n_kpad % KB5 == 0,
               "ea_attention_d512_fwd_bf16: bad sizes (the padded key count must be a multiple of 32 and cover n_keys)");
    EA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0 && q_frame_stride % 8 == 0 &&
                   k_frame_stride % 8 == 0 && vt_frame_stride % 8 == 0 && out_frame_stride % 8 == 0,
               "ea_attention_d512_fwd_bf16: pointers / frame strides must be 16-byte aligned");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)attention_d512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS5);
        attr_done = true;
    }
    ea_count("attention_d512");
    const dim3 grid((unsigned)((n_q + 127) / 128), (unsigned)frames);
    hipLaunchKernelGGL(attention_d512_kernel, grid, dim3(512), LDS5, (hipStream_t)stream, q, k, vt, (unsigned short*)out, n_q, n_keys,
                       n_kpad, q_frame_stride, k_frame_stride, vt_frame_stride, out_frame_stride, scale * 1.4426950408889634f);
    return ea_check_launch("ea_attention_d512_fwd_bf16");
}
