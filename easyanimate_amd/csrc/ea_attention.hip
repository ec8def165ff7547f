// Joint text/video attention for the MMDiT block: qk-LayerNorm + RoPE + head-major scatter, and the
// flash-style forward  O = softmax(Q K^T * scale) V  for head_dim 64 (MFMA 32x32x16 bf16).
//
// ea_attention_fwd_bf16, mapping to CDNA4 (details + roofline in DESIGN.md):
//   * everything is computed TRANSPOSED: S^T = K.Q^T and O^T = V^T.P^T.  In the MFMA C layout a lane then
//     owns ONE query column, so the softmax running max / sum / rescale are lane-local (one cross-half
//     exchange per tile for the max), and the exponentiated S^T accumulator registers ARE the B operand of
//     the PV MFMA -- no LDS round trip, no permutes.  K rows (and V^T rows) are fed to the MFMA in
//     bit-2/bit-3-swapped order so that each lane's 8 consecutive accumulator registers are 8 consecutive
//     keys (resp. 8 consecutive output channels: 16-byte output stores).
//   * V is stored transposed in HBM ([B,H,64,S], written by ea_qknorm_rope_bf16), so both the K tile
//     ([64 keys][64 d]) and the V^T tile ([64 d][64 keys]) are plain 128-byte-row tiles, staged by LDS-DMA
//     (global_load_lds_dwordx4) with the 16-byte-chunk XOR swizzle ((row>>1)&7) on the source address:
//     every fragment is one conflict-free ds_read_b128.
//   * workgroup = 4 waves x 64 queries = 256 queries; KV tile = 64 keys, double-buffered (32 KiB LDS),
//     one s_barrier per tile, next tile's DMA in flight under the MFMAs; 2 workgroups per CU.
//   * blockIdx -> (XCD, head, q-block): all workgroups resident on one XCD work on the same (batch, head),
//     so its K / V^T stream (S*256 B) is served from that XCD's L2.
#include "ea_common.h"

namespace {

#ifndef EA_ATT3_BUFDMA
#define EA_ATT3_BUFDMA 1   // build-time A/B switch: K / V^T tiles by buffer-addressed LDS-DMA (0: per-lane 64-bit addresses)
#endif
// Buffer-addressed LDS-DMA: 16 bytes per lane from base + voff (per lane, fixed for the kernel) + soff (scalar: the key tile)
// to lds_wave_base + 16 * lane -- no per-piece vector address arithmetic beside the softmax's VALU work.
__device__ __forceinline__ void bdma16(const void* base, int extent, int voff, int soff, void* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, extent, 0x00020000);   // raw buffer, 32-bit format
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// ------------------------------------------------------------------------------------------------
// qk-LayerNorm + RoPE + scatter.  reference: easyanimate/models/processor.py:251-285
// block = 256 threads handles 64 tokens of one (batch, head): thread -> (token = t/8 (+32), 8 channels).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qknorm_rope_kernel(
    const unsigned short* __restrict__ qkv, int64_t qkv_bs, unsigned short* __restrict__ q_out,
    unsigned short* __restrict__ k_out, unsigned short* __restrict__ vt_out, const float* __restrict__ nq_w,
    const float* __restrict__ nq_b, const float* __restrict__ nk_w, const float* __restrict__ nk_b,
    const float* __restrict__ cosT, const float* __restrict__ sinT, int heads, int n_tok, int seq_off, int s_pad,
    int kv_off, int kv_rows, float eps, float q_scale) {
    __shared__ unsigned short vtile[64][72];  // [token][channel], padded
    const int tid = threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const int tok0 = blockIdx.x * 64;
    const int inner = heads * 64;
    const int sub = tid & 7;  // channels sub*8 .. sub*8+7
    const int64_t bh = (int64_t)b * heads + h;

#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int tl = it * 32 + (tid >> 3);
        const int tok = tok0 + tl;
        const bool valid = tok < n_tok;
        const int tokc = valid ? tok : n_tok - 1;
        const unsigned short* src = qkv + b * qkv_bs + (int64_t)tokc * 3 * inner + h * 64 + sub * 8;
        float cs[8], sn[8];
        if (cosT) {
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosT + (int64_t)tokc * 64 + sub * 8 + e);
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinT + (int64_t)tokc * 64 + sub * 8 + e);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cs[e + u] = c4[u];
                    sn[e + u] = s4[u];
                }
            }
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {  // 0 = q, 1 = k
            const u16x8 raw = *reinterpret_cast<const u16x8*>(src + which * inner);
            float v[8];
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = bf16_bits_to_f32(raw[e]);
                s += v[e];
            }
            // reduce over the 8 lanes that share a (token, head) row
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.0f / 64.0f);
            float qd = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[e] - mean;
                qd += d * d;
            }
            qd += __shfl_xor(qd, 1, 64);
            qd += __shfl_xor(qd, 2, 64);
            qd += __shfl_xor(qd, 4, 64);
            const float rstd = rsqrtf(qd * (1.0f / 64.0f) + eps);
            const float* gw = which ? nk_w : nq_w;
            const float* gb = which ? nk_b : nq_b;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (v[e] - mean) * rstd * gw[sub * 8 + e] + gb[sub * 8 + e];
                // nn.LayerNorm output is rounded to the model dtype before RoPE (processor.py:255-258)
                v[e] = bf16_bits_to_f32(f32_to_bf16_bits(t));
            }
            // q only: the softmax scale (x log2 e) may be folded in here, in fp32 ahead of the one bf16 rounding the
            // reference applies at this point, so that the attention kernel can exponentiate raw scores
            const float osc = which ? 1.0f : q_scale;
            u16x8 o;
            if (cosT) {
                // diffusers apply_rotary_emb, interleaved pairs: out[2i] = x[2i]c - x[2i+1]s ; out[2i+1] = x[2i+1]c + x[2i]s
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float x0 = v[e], x1 = v[e + 1];
                    o[e] = f32_to_bf16_bits((x0 * cs[e] - x1 * sn[e]) * osc);
                    o[e + 1] = f32_to_bf16_bits((x1 * cs[e + 1] + x0 * sn[e + 1]) * osc);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16_bits(v[e] * osc);
            }
            if (valid) {
                unsigned short* dst = which ? k_out + (bh * kv_rows + kv_off + tok) * 64 + sub * 8
                                            : q_out + (bh * s_pad + seq_off + tok) * 64 + sub * 8;
                *reinterpret_cast<u16x8*>(dst) = o;
            }
        }
        {
            u16x8 raw = *reinterpret_cast<const u16x8*>(src + 2 * inner);
            if (!valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) raw[e] = 0;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) vtile[tl][sub * 8 + e] = raw[e];
        }
    }
    __syncthreads();
    // transposed write: thread -> channel d = tid/4, 16 tokens (tid%4)*16.. ; destination columns seq_off+tok
    {
        const int d = tid >> 2;
        const int t0 = (tid & 3) * 16;
        unsigned short* dst = vt_out + (bh * 64 + d) * (int64_t)kv_rows + kv_off + tok0 + t0;
        const bool aligned = (((kv_off + tok0) & 7) == 0);
        if (aligned && tok0 + t0 + 16 <= n_tok) {
            u16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0[e] = vtile[t0 + e][d];
                o1[e] = vtile[t0 + 8 + e][d];
            }
            *reinterpret_cast<u16x8*>(dst) = o0;
            *reinterpret_cast<u16x8*>(dst + 8) = o1;
        } else {
            for (int e = 0; e < 16; ++e)
                if (tok0 + t0 + e < n_tok) dst[e] = vtile[t0 + e][d];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// flash attention forward, head_dim 64.  reference: easyanimate/models/processor.py:287-291
// ------------------------------------------------------------------------------------------------
constexpr int ATT_QB = 256;                 // queries per workgroup
constexpr int ATT_KV = 64;                  // keys per tile
constexpr int ATT_TILE = ATT_KV * 64 * 2;   // 8 KiB (K tile; V^T tile has the same size)
constexpr int ATT_STAGE = 2 * ATT_TILE;     // 16 KiB
constexpr int ATT_LDS = 2 * ATT_STAGE;      // 32 KiB

// WINDOW: sliding-window (band) attention of the SWA processor (processor.py:420, flash_attn_func(window_size=(w, w))):
// query i sees key j iff |i - j| <= window.  Only the key tiles that intersect the band of the workgroup's 256 queries
// are visited; inside them the band is masked with -inf.  A query whose visited tiles so far held none of its keys has a
// running maximum of -inf: the exponent shift is then taken as 0 (every p = exp2(-inf) = 0, alpha = exp2(-inf) = 0 on
// all-zero state) instead of forming -inf - (-inf); every query sees at least itself, so the final row sum is > 0.
// MAPPED (round 4, the SWA processor's six scan orders without index copies): the queries / keys of head h are visited in the
// scan order map[h][p] (scan position p -> token), but q / k stay where the projection wrote them -- row `row_off + token` of
// the natural-order workspace -- and are ADDRESSED through the map: the Q fragments of position p come from row
// row_off + map[h][p], and the K tile's LDS-DMA rows point at the mapped tokens (the per-lane row index of tile t + 1 is loaded
// one tile ahead).  V^T cannot be gathered by row DMA (a key is a COLUMN of it): it comes from a buffer permuted beforehand
// (ea_gather_cols_bf16; rows of vt_pad elements).  The result row of position p goes back to token order in the store,
// with the cross pass (token order, like out) added (processor.py:435): out[r] = bf16(bf16(window result) + cross[r]), r = row_off + map[h][p].
struct AttWindowMap {
    const int* map;                    // [heads, seq] int32: scan position -> token
    const unsigned short* cross;       // [batch, row_off + seq, heads*64] (out's strides), token order
    int row_off, vt_pad;
};
template <bool WINDOW, bool MAPPED = false>
__global__ __launch_bounds__(256, 2) void attention_fwd_kernel(
    const unsigned short* __restrict__ Q, const unsigned short* __restrict__ K, const unsigned short* __restrict__ Vt,
    unsigned short* __restrict__ O, int64_t o_bs, int heads, int bh_total, int seq, int s_pad, int q_begin,
    int q_end, int nqb, float scale_log2e, int window, AttWindowMap wm = AttWindowMap()) {
    static_assert(!MAPPED || WINDOW, "the mapped form serves the window pass only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    // block -> (xcd, slot) -> (bh, q-block): all blocks of an XCD walk the same head together
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bh = (slot / nqb) * 8 + xcd;
    const int qb = slot % nqb;
    if (bh >= bh_total) return;
    const int b = bh / heads, h = bh % heads;
    const int q0 = q_begin + qb * ATT_QB + wave * 64;  // first query row of this wave

    const int v_pad = MAPPED ? wm.vt_pad : s_pad;
    const unsigned short* Qh = Q + (int64_t)bh * s_pad * 64;
    const unsigned short* Kh = K + (int64_t)bh * s_pad * 64;
    const unsigned short* Vh = Vt + (int64_t)bh * 64 * v_pad;
    const int* const hm = MAPPED ? wm.map + (int64_t)h * seq : nullptr;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (n = query l31, k-half hi)
    bf16x8 qf[2][4];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        int qr = q0 + qi * 32 + l31;
        if (MAPPED) qr = wm.row_off + hm[qr < seq ? qr : seq - 1];
        else qr = qr < s_pad ? qr : s_pad - 1;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
            qf[qi][ds] = *reinterpret_cast<const bf16x8*>(Qh + (int64_t)qr * 64 + ds * 16 + hi * 8);
    }

    // ---- DMA source pointers: per wave 2 pieces of K tile + 2 pieces of V^T tile (1 KiB each)
    const unsigned short* ksrc[2];
    const unsigned short* vsrc[2];
    int krel[2], kcs[2], mrow[2] = {0, 0};    // MAPPED: tile-relative key row of the lane's two pieces, its source chunk, the mapped token
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (wave * 2 + i) * 64 + lane;
        const int r = L >> 3, c = L & 7;
        const int cs = c ^ ((r >> 1) & 7);
        krel[i] = r; kcs[i] = cs * 8;
        ksrc[i] = Kh + (int64_t)r * 64 + cs * 8;       // + kv0*64 per tile
        vsrc[i] = Vh + (int64_t)r * v_pad + cs * 8;    // + kv0 per tile
    }
    auto load_map = [&](int t) {      // the tokens behind the lane's two key rows of tile t (masked rows: any valid token)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kp = t * ATT_KV + krel[i];
            mrow[i] = hm[kp < seq ? kp : seq - 1];
        }
    };
    auto issue = [&](int t, int stage) {
        char* sk = smem + stage * ATT_STAGE + wave * 2048;
        char* sv = sk + ATT_TILE;
        const int kv0 = t * ATT_KV;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (MAPPED) glds16(Kh + (int64_t)(wm.row_off + mrow[i]) * 64 + kcs[i], sk + i * 1024);
            else glds16(ksrc[i] + (int64_t)kv0 * 64, sk + i * 1024);
            glds16(vsrc[i] + kv0, sv + i * 1024);
        }
    };

    // fragment read offsets: K row for MFMA row index l31 is swap23(l31) (+32 per key block);
    // V^T row likewise (+32 per output-channel tile).
    const int frow = swap23(l31);
    int f_off[2], f_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * 32 + frow;
        f_off[i] = r * 128;
        f_sw[i] = (r >> 1) & 7;
    }

    f32x16 o[2][2];  // [d tile][q tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][j][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};

    int nt = (seq + ATT_KV - 1) / ATT_KV;
    int t_lo = 0;
    const float MASKED = -INFINITY;
    if (WINDOW) {
        const int qblk0 = q_begin + qb * ATT_QB;          // the four waves share the K / V^T tiles: workgroup-wide band
        const int lo = qblk0 - window, hi_key = qblk0 + ATT_QB - 1 + window;
        t_lo = lo > 0 ? lo / ATT_KV : 0;
        const int t_hi = hi_key / ATT_KV + 1;
        nt = t_hi < nt ? t_hi : nt;
    }
    if (MAPPED) {
        load_map(t_lo);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mrow[0]), "+v"(mrow[1]) :: "memory");
    }
    issue(t_lo, t_lo & 1);
    if (MAPPED && t_lo + 1 < nt) load_map(t_lo + 1);
    for (int t = t_lo; t < nt; ++t) {
        // (MAPPED: the wait also covers the map rows of tile t + 1, requested a tile ago; the asm's operands keep their uses behind it)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(mrow[0]), "+v"(mrow[1]) :: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < nt) {
            issue(t + 1, (t + 1) & 1);
            if (MAPPED && t + 2 < nt) load_map(t + 2);
        }
        const char* sk = smem + (t & 1) * ATT_STAGE;
        const char* sv = sk + ATT_TILE;
        const int kv0 = t * ATT_KV;
        const bool tail = kv0 + ATT_KV > seq;

#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // ---- S^T[kb] = K[kb] . Q^T : 32 keys x (2 x 32 queries)
            f32x16 s[2];
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qi][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const int ch = ds * 2 + hi;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + f_off[kb] + ((ch ^ f_sw[kb]) << 4));
#pragma unroll
                for (int qi = 0; qi < 2; ++qi)
                    s[qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qi][ds], s[qi], 0, 0, 0);
            }
            // lane (query l31 of tile qi, half hi): s[qi][r] is key  kv0 + kb*32 + 16*(r>>3) + 8*hi + (r&7)
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= seq) {
                        s[0][r] = MASKED;
                        s[1][r] = MASKED;
                    }
                }
            }
            if (WINDOW) {
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const int qrow = q0 + qi * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                        const int d = key - qrow;
                        if (d > window || d < -window) s[qi][r] = MASKED;
                    }
                }
            }
            // ---- online softmax (exp2 domain), per query = per lane
            bf16x8 pf[2][2];  // [q tile][key slab of 16]
#pragma unroll
            for (int qi = 0; qi < 2; ++qi) {
                float mx = s[qi][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qi][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[qi], mx * scale_log2e);
                const float m_use = (WINDOW && m_new == -INFINITY) ? 0.f : m_new;   // no key of this query seen yet
                const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_use);
                m_run[qi] = m_new;
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qi][r], scale_log2e, -m_use));
                    psum += p;
                    pf[qi][r >> 3][r & 7] = (bf16_t)p;
                }
                l_run[qi] = l_run[qi] * alpha + psum;
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[dt][qi][r] *= alpha;
                }
            }
            // ---- O^T += V^T[:, keys of kb] . P^T
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = kb * 4 + ks * 2 + hi;  // 16-byte chunk = 8 keys
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sv + f_off[dt] + ((ch ^ f_sw[dt]) << 4));
#pragma unroll
                    for (int qi = 0; qi < 2; ++qi)
                        o[dt][qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qi][ks], o[dt][qi], 0, 0, 0);
                }
            }
        }
    }

    // ---- normalise and store: lane (query, hi) holds channels dt*32 + 16*g + 8*hi + 0..7
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const float l = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
        const float inv = 1.0f / l;
        const int qr = q0 + qi * 32 + l31;
        if (qr < q_end) {
            const int orow = MAPPED ? wm.row_off + hm[qr] : qr;          // back to token order in the store
            unsigned short* dst = O + b * o_bs + (int64_t)orow * heads * 64 + h * 64;
            const unsigned short* cr = MAPPED ? wm.cross + b * o_bs + (int64_t)orow * heads * 64 + h * 64 : nullptr;   // cross: token order, out's strides
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    u16x8 ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16_bits(o[dt][qi][g * 8 + e] * inv);
                    if (MAPPED) {   // + the cross pass: two bf16 results added in fp32, rounded once (the reference adds two bf16 tensors)
                        const u16x8 cv = *reinterpret_cast<const u16x8*>(cr + dt * 32 + g * 16 + hi * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = f32_to_bf16_bits(bf16_bits_to_f32(ov[e]) + bf16_bits_to_f32(cv[e]));
                    }
                    *reinterpret_cast<u16x8*>(dst + dt * 32 + g * 16 + hi * 8) = ov;
                }
        }
    }
}

// RAW mode (P = exp2 of the raw score, no shift) is exact while a query's block row sums stay inside [2^-60, 2^60)
constexpr float ATT_BIG = 1.152921504606846976e18f;   // 2^60
constexpr float ATT_TINY = 8.673617379884035e-19f;    // 2^-60
int g_attn_variant = 3;  // ea_set_option("attn_variant", 3): v3 (16x16x32 MFMA); EA_BUILD_VARIANTS=1 libraries also carry 2 = v2 (32x32x16) and 1 = the first, un-pipelined kernel
#if EA_BUILD_VARIANTS
#include "ea_attention_v2.inc"   // the 32x32x16 generation (and the only one that takes an un-folded softmax scale): cross-check builds only
#endif
#include "ea_attention_v3.inc"
#if EA_BUILD_VARIANTS
#include "ea_attention_v5.inc"   // the ping-pong measurement kernel (round 6; 15 % slower than v3): cross-check builds only
#endif

}  // namespace

int g_attn_stages = 2;   // ea_set_option("attn_stages", 2 | 3): LDS stages per operand of attention_fwd_v3_kernel (3: tiles requested one tile earlier; four-wave shape only)
int ea_attn_stages_get() { return g_attn_stages; }
int ea_attn_stages_set(int v) {
    if (v != 2 && v != 3) return -1;
    g_attn_stages = v;
    return 0;
}
int g_attn_nw = 4;   // ea_set_option("attn_nw", 4 | 8): waves per workgroup of attention_fwd_v3_kernel (8: one 512-query workgroup per CU shares ONE K / V^T stream)
int ea_attn_nw_get() { return g_attn_nw; }
int ea_attn_nw_set(int v) {
    if (v != 4 && v != 8 && !(v == 16 && EA_BUILD_VARIANTS)) return -1;     // 16: the ping-pong measurement kernel (attention_fwd_v5_kernel; EA_BUILD_VARIANTS=1 libraries, plain launches)
    g_attn_nw = v;
    return 0;
}
int ea_attn_variant_get() { return g_attn_variant; }
int ea_attn_variant_set(int v) {
    if (v != 3 && !((v == 1 || v == 2) && EA_BUILD_VARIANTS)) return -1;   // v1 / v2 (the cross-check generations): EA_BUILD_VARIANTS=1 libraries only
    g_attn_variant = v;
    return 0;
}

extern "C" int ea_qknorm_rope_bf16(const ea_bf16* qkv, int64_t qkv_batch_stride, ea_bf16* q_out, ea_bf16* k_out,
                                   ea_bf16* vt_out, const float* nq_w, const float* nq_b, const float* nk_w,
                                   const float* nk_b, const float* cos, const float* sin, int batch, int heads,
                                   int n_tok, int seq_off, int s_pad, int kv_off, int kv_rows, float ln_eps, float q_scale,
                                   void* stream) {
    EA_REQUIRE(qkv && q_out && k_out && vt_out && nq_w && nq_b && nk_w && nk_b, "ea_qknorm_rope_bf16: null tensor");
    if (kv_rows <= 0) { kv_rows = s_pad; kv_off = seq_off; }   // K / V^T share the geometry of Q
    EA_REQUIRE(kv_off >= 0 && kv_rows % 8 == 0 && kv_off + n_tok <= kv_rows, "ea_qknorm_rope_bf16: K / V^T rows [kv_off, kv_off + n_tok) must fit kv_rows");
    EA_REQUIRE((cos == nullptr) == (sin == nullptr), "ea_qknorm_rope_bf16: cos/sin must come together");
    EA_REQUIRE(batch > 0 && heads > 0 && n_tok >= 0 && seq_off >= 0, "ea_qknorm_rope_bf16: bad sizes");
    EA_REQUIRE(s_pad % 64 == 0 && seq_off + n_tok <= s_pad, "ea_qknorm_rope_bf16: s_pad must be a multiple of 64 and cover the rows");
    EA_REQUIRE(heads <= 65535 && batch <= 65535, "ea_qknorm_rope_bf16: grid too large");
    if (n_tok == 0) return EA_OK;
    dim3 grid((n_tok + 63) / 64, heads, batch);
    hipLaunchKernelGGL(qknorm_rope_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, qkv_batch_stride, q_out,
                       k_out, vt_out, nq_w, nq_b, nk_w, nk_b, cos, sin, heads, n_tok, seq_off, s_pad, kv_off, kv_rows, ln_eps, q_scale);
    return ea_check_launch("ea_qknorm_rope_bf16");
}

static int attention_launch(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                            int64_t out_batch_stride, int batch, int heads, int s_pad, int q_begin, int q_end,
                            int kv_begin, int kv_end, float scale, float* state, int flags, void* stream,
                            int q_head0 = 0, int q_heads = 0, int64_t kv_bstride = 0) {
    EA_REQUIRE(q_head0 >= 0 && (q_heads == 0 ? q_head0 == 0 : q_head0 + heads <= q_heads) && kv_bstride >= 0 && kv_bstride % 8 == 0,
               "ea_attention_fwd: the head window [q_head0, q_head0 + heads) must lie inside q_heads");
    EA_REQUIRE(q && k && vt && (out || (flags & 2)), "ea_attention_fwd: null tensor");
    EA_REQUIRE(batch > 0 && heads > 0 && kv_end > kv_begin && kv_begin >= 0, "ea_attention_fwd: bad sizes");
    EA_REQUIRE(s_pad % ATT_QB == 0 && s_pad >= kv_end, "ea_attention_fwd: s_pad must be a multiple of 256 and >= the key range");
    EA_REQUIRE(q_begin >= 0 && q_begin <= q_end && q_end <= s_pad, "ea_attention_fwd: bad query range");
    EA_REQUIRE(kv_begin % ATT_KV == 0, "ea_attention_fwd: kv_begin must be a multiple of 64");
    EA_REQUIRE((flags & ~3) == 0 && (flags == 0 || state), "ea_attention_fwd: bad flags / missing state buffer");
    if (q_end == q_begin) return EA_OK;
    // scale * log2(e) == 1: the caller folded the softmax scale into Q (q_scale of ea_qkv_gemm_norm_rope_bf16 / ea_qknorm_rope_bf16)
    const bool folded = fabsf(scale * 1.4426950408889634f - 1.0f) < 1e-6f;
#if !EA_BUILD_VARIANTS
    EA_REQUIRE(folded, "ea_attention_fwd: the softmax scale must be folded into Q (q_scale = scale * log2(e) at the projection, scale = "
                       "1 / log2(e) here); the kernel generation that multiplies every score is built with EA_BUILD_VARIANTS=1 only");
#endif
    const bool plain = flags == 0 && kv_begin == 0;
    const int nqb = (q_end - q_begin + ATT_QB - 1) / ATT_QB;
    const int bh = batch * heads;
    const int64_t blocks = (int64_t)((bh + 7) / 8) * nqb * 8;
    EA_REQUIRE(blocks < (1ll << 31), "ea_attention_fwd: grid too large");
    const float scale_log2e = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    unsigned short* o16 = (unsigned short*)out;
    f32x4* st4 = reinterpret_cast<f32x4*>(state);
#if EA_BUILD_VARIANTS
    EA_REQUIRE(q_heads == 0 || g_attn_variant >= 3, "ea_attention_fwd: a head window needs the v3 kernel");
    const int variant = plain ? g_attn_variant : (g_attn_variant >= 3 ? 3 : 2);   // key ranges / resumable state: v2 / v3 only
    if (variant == 1) {
        ea_count("attention_v1");
        hipLaunchKernelGGL(attention_fwd_kernel<false>, grid, blk, ATT_LDS, st, q, k, vt, o16, out_batch_stride, heads, bh, kv_end,
                           s_pad, q_begin, q_end, nqb, scale_log2e, 0);
        return ea_check_launch("ea_attention_fwd");
    }
    if (variant != 3 || !folded) {
        ea_count("attention_v2");
#define EA_ATT_LAUNCH2(MODE, FOLDED)                                                                                      \
        hipLaunchKernelGGL((attention_fwd_v2_kernel<MODE, FOLDED>), grid, blk, ATT_LDS, st, q, k, vt, o16,                 \
                           out_batch_stride, heads, bh, kv_begin, kv_end, s_pad, q_begin, q_end, nqb, scale_log2e, st4)
        switch (flags * 2 + (folded ? 1 : 0)) {
            case 0: EA_ATT_LAUNCH2(0, false); break;
            case 1: EA_ATT_LAUNCH2(0, true); break;
            case 2: EA_ATT_LAUNCH2(1, false); break;
            case 3: EA_ATT_LAUNCH2(1, true); break;
            case 4: EA_ATT_LAUNCH2(2, false); break;
            case 5: EA_ATT_LAUNCH2(2, true); break;
            case 6: EA_ATT_LAUNCH2(3, false); break;
            default: EA_ATT_LAUNCH2(3, true); break;
        }
#undef EA_ATT_LAUNCH2
        return ea_check_launch("ea_attention_fwd");
    }
#endif
    (void)plain;
    ea_count("attention_v3");
    AttSegments hw = AttSegments();          // not a segment launch: only the head window fields are read
    hw.q_head0 = q_head0; hw.q_heads = q_heads; hw.kv_bstride = kv_bstride;
#if EA_BUILD_VARIANTS
    if (g_attn_nw == 16 && flags == 0 && kv_begin == 0 && q_heads == 0 && kv_bstride == 0) {
        const int nqb5 = (q_end - q_begin + 511) / 512;
        hipLaunchKernelGGL(attention_fwd_v5_kernel, dim3((unsigned)att3_grid_blocks(bh, nqb5)), dim3(512), 3 * ATT_STAGE, st, q, k, vt, o16,
                           out_batch_stride, heads, bh, kv_end, s_pad, q_begin, q_end, nqb5);
        return ea_check_launch("ea_attention_fwd");
    }
#endif
    const int nw = g_attn_nw == 16 ? 4 : g_attn_nw;
    const int nqb3 = (q_end - q_begin + nw * 64 - 1) / (nw * 64);     // query blocks of nw * 64 rows
    const dim3 grid3((unsigned)att3_grid_blocks(bh, nqb3)), blk3(nw * 64);
#define EA_ATT_LAUNCH(MODE, NW_)                                                                                          \
    hipLaunchKernelGGL((attention_fwd_v3_kernel<MODE, false, NW_>), grid3, blk3, ATT_LDS, st, q, k, vt, o16,               \
                       out_batch_stride, heads, bh, kv_begin, kv_end, s_pad, q_begin, q_end, nqb3, scale_log2e, st4, hw)
#define EA_ATT_LAUNCH3(MODE)                                                                                              \
    hipLaunchKernelGGL((attention_fwd_v3_kernel<MODE, false, 4, 3>), grid3, blk3, 3 * ATT_STAGE, st, q, k, vt, o16,         \
                       out_batch_stride, heads, bh, kv_begin, kv_end, s_pad, q_begin, q_end, nqb3, scale_log2e, st4, hw)
    if (nw == 4 && g_attn_stages == 3) {
        switch (flags) {
            case 0: EA_ATT_LAUNCH3(0); break;
            case 1: EA_ATT_LAUNCH3(1); break;
            case 2: EA_ATT_LAUNCH3(2); break;
            default: EA_ATT_LAUNCH3(3); break;
        }
        return ea_check_launch("ea_attention_fwd");
    }
    switch (flags + (nw == 8 ? 4 : 0)) {
        case 0: EA_ATT_LAUNCH(0, 4); break;
        case 1: EA_ATT_LAUNCH(1, 4); break;
        case 2: EA_ATT_LAUNCH(2, 4); break;
        case 3: EA_ATT_LAUNCH(3, 4); break;
        case 4: EA_ATT_LAUNCH(0, 8); break;
        case 5: EA_ATT_LAUNCH(1, 8); break;
        case 6: EA_ATT_LAUNCH(2, 8); break;
        default: EA_ATT_LAUNCH(3, 8); break;
    }
#undef EA_ATT_LAUNCH
#undef EA_ATT_LAUNCH3
    return ea_check_launch("ea_attention_fwd");
}

extern "C" int ea_attention_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                     int64_t out_batch_stride, int batch, int heads, int seq, int s_pad, int q_begin,
                                     int q_end, float scale, void* stream) {
    EA_REQUIRE(q_end <= seq, "ea_attention_fwd_bf16: bad query range");
    return attention_launch(q, k, vt, out, out_batch_stride, batch, heads, s_pad, q_begin, q_end, 0, seq, scale, nullptr, 0,
                            stream);
}

static int attention_segments_launch(const ea_bf16* q, const ea_bf16* k_seg0, const ea_bf16* vt_seg0, ea_bf16* out,
                                     int64_t out_batch_stride, int batch, int heads, int q_pad, int q_begin,
                                     int q_end, int seg_rows, int n_seg, int skip_seg, int64_t seg_stride,
                                     int seg_first_row, int seg_used_rows, int kv_valid, float scale, float* state,
                                     int flags, void* stream, int q_head0, int q_heads, int64_t kv_bstride) {
    EA_REQUIRE(q && k_seg0 && vt_seg0 && (out || (flags & 2)), "ea_attention_fwd_segments_bf16: null tensor");
    EA_REQUIRE(q_head0 >= 0 && (q_heads == 0 ? q_head0 == 0 : q_head0 + heads <= q_heads) && kv_bstride >= 0 && kv_bstride % 8 == 0,
               "ea_attention_fwd_segments_bf16: the head window [q_head0, q_head0 + heads) must lie inside q_heads");
    EA_REQUIRE(batch > 0 && heads > 0 && q_pad % ATT_QB == 0 && q_begin >= 0 && q_begin <= q_end && q_end <= q_pad,
               "ea_attention_fwd_segments_bf16: bad query range");
    EA_REQUIRE(seg_rows > 0 && seg_rows % ATT_KV == 0 && n_seg >= 1 && seg_stride >= 0,
               "ea_attention_fwd_segments_bf16: segment rows must be a positive multiple of 64");
    EA_REQUIRE(seg_first_row >= 0 && seg_first_row % ATT_KV == 0 && seg_used_rows > 0 && seg_used_rows % ATT_KV == 0 &&
               seg_first_row + seg_used_rows <= seg_rows,
               "ea_attention_fwd_segments_bf16: the used row range of a segment must be 64-aligned and inside the segment");
    const int used = n_seg - ((skip_seg >= 0 && skip_seg < n_seg) ? 1 : 0);
    EA_REQUIRE(used >= 1 && kv_valid > 0 && (int64_t)kv_valid <= (int64_t)used * seg_used_rows &&
               kv_valid > (int64_t)(used - 1) * seg_used_rows,
               "ea_attention_fwd_segments_bf16: kv_valid must end inside the last used segment");
    EA_REQUIRE((flags & ~3) == 0 && (flags == 0 || state), "ea_attention_fwd_segments_bf16: bad flags / missing state buffer");
    EA_REQUIRE(fabsf(scale * 1.4426950408889634f - 1.0f) < 1e-6f,
               "ea_attention_fwd_segments_bf16: the softmax scale must be folded into Q (scale = ln 2)");
    EA_REQUIRE((((uintptr_t)q | (uintptr_t)k_seg0 | (uintptr_t)vt_seg0) & 15) == 0 && seg_stride % 8 == 0,
               "ea_attention_fwd_segments_bf16: pointers / segment stride must be 16-byte aligned");
    if (q_end == q_begin) return EA_OK;
    const int nqb = (q_end - q_begin + ATT_QB - 1) / ATT_QB;
    const int bh = batch * heads;
    const int64_t blocks = (int64_t)((bh + 7) / 8) * nqb * 8;
    EA_REQUIRE(blocks < (1ll << 31), "ea_attention_fwd_segments_bf16: grid too large");
    AttSegments sg;
    sg.rows = seg_rows; sg.tiles = seg_used_rows / ATT_KV; sg.skip = (skip_seg >= 0 && skip_seg < n_seg) ? skip_seg : n_seg;
    sg.total_tiles = used * sg.tiles; sg.stride = seg_stride; sg.first = seg_first_row / ATT_KV;
    sg.q_head0 = q_head0; sg.q_heads = q_heads; sg.kv_bstride = kv_bstride;
    const int nw = g_attn_nw == 16 ? 4 : g_attn_nw;
    const int nqb3 = (q_end - q_begin + nw * 64 - 1) / (nw * 64);
    const dim3 grid((unsigned)att3_grid_blocks(bh, nqb3)), blk(nw * 64);
    hipStream_t st = (hipStream_t)stream;
    unsigned short* o16 = (unsigned short*)out;
    f32x4* st4 = reinterpret_cast<f32x4*>(state);
    ea_count("attention_v3_segments");
#define EA_ATT_SEG(MODE, NW_) hipLaunchKernelGGL((attention_fwd_v3_kernel<MODE, true, NW_>), grid, blk, ATT_LDS, st, q, k_seg0, vt_seg0, o16, \
                                                 out_batch_stride, heads, bh, 0, kv_valid, q_pad, q_begin, q_end, nqb3, 1.0f, st4, sg)
#define EA_ATT_SEG3(MODE) hipLaunchKernelGGL((attention_fwd_v3_kernel<MODE, true, 4, 3>), grid, blk, 3 * ATT_STAGE, st, q, k_seg0, vt_seg0, o16, \
                                             out_batch_stride, heads, bh, 0, kv_valid, q_pad, q_begin, q_end, nqb3, 1.0f, st4, sg)
    if (nw == 4 && g_attn_stages == 3) {
        switch (flags) {
            case 0: EA_ATT_SEG3(0); break;
            case 1: EA_ATT_SEG3(1); break;
            case 2: EA_ATT_SEG3(2); break;
            default: EA_ATT_SEG3(3); break;
        }
        return ea_check_launch("ea_attention_fwd_segments_bf16");
    }
#undef EA_ATT_SEG3
    switch (flags + (nw == 8 ? 4 : 0)) {
        case 0: EA_ATT_SEG(0, 4); break;
        case 1: EA_ATT_SEG(1, 4); break;
        case 2: EA_ATT_SEG(2, 4); break;
        case 3: EA_ATT_SEG(3, 4); break;
        case 4: EA_ATT_SEG(0, 8); break;
        case 5: EA_ATT_SEG(1, 8); break;
        case 6: EA_ATT_SEG(2, 8); break;
        default: EA_ATT_SEG(3, 8); break;
    }
#undef EA_ATT_SEG
    return ea_check_launch("ea_attention_fwd_segments_bf16");
}

extern "C" int ea_attention_fwd_segments_bf16(const ea_bf16* q, const ea_bf16* k_seg0, const ea_bf16* vt_seg0, ea_bf16* out,
                                              int64_t out_batch_stride, int batch, int heads, int q_pad, int q_begin,
                                              int q_end, int seg_rows, int n_seg, int skip_seg, int64_t seg_stride,
                                              int seg_first_row, int seg_used_rows, int kv_valid, float scale, float* state,
                                              int flags, void* stream) {
    return attention_segments_launch(q, k_seg0, vt_seg0, out, out_batch_stride, batch, heads, q_pad, q_begin, q_end, seg_rows, n_seg,
                                     skip_seg, seg_stride, seg_first_row, seg_used_rows, kv_valid, scale, state, flags, stream, 0, 0, 0);
}

extern "C" int ea_attention_fwd_segments_heads_bf16(const ea_bf16* q, const ea_bf16* k_seg0, const ea_bf16* vt_seg0, ea_bf16* out,
                                                    int64_t out_batch_stride, int batch, int heads, int q_pad, int q_begin,
                                                    int q_end, int seg_rows, int n_seg, int skip_seg, int64_t seg_stride,
                                                    int seg_first_row, int seg_used_rows, int kv_valid, float scale, float* state,
                                                    int flags, int q_head0, int q_heads, int64_t kv_batch_stride, void* stream) {
    return attention_segments_launch(q, k_seg0, vt_seg0, out, out_batch_stride, batch, heads, q_pad, q_begin, q_end, seg_rows, n_seg,
                                     skip_seg, seg_stride, seg_first_row, seg_used_rows, kv_valid, scale, state, flags, stream, q_head0,
                                     q_heads, kv_batch_stride);
}

extern "C" int ea_attention_window_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                            int64_t out_batch_stride, int batch, int heads, int seq, int s_pad, int window,
                                            float scale, void* stream) {
    EA_REQUIRE(q && k && vt && out, "ea_attention_window_fwd_bf16: null tensor");
    EA_REQUIRE(batch > 0 && heads > 0 && seq > 0 && window >= 0, "ea_attention_window_fwd_bf16: bad sizes");
    EA_REQUIRE(s_pad % ATT_QB == 0 && s_pad >= seq, "ea_attention_window_fwd_bf16: s_pad must be a multiple of 256 and >= seq");
    const int nqb = (seq + ATT_QB - 1) / ATT_QB;
    const int bh = batch * heads;
    const int64_t blocks = (int64_t)((bh + 7) / 8) * nqb * 8;
    EA_REQUIRE(blocks < (1ll << 31), "ea_attention_window_fwd_bf16: grid too large");
    ea_count("attention_window");
    hipLaunchKernelGGL(attention_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), ATT_LDS, (hipStream_t)stream, q, k, vt,
                       (unsigned short*)out, out_batch_stride, heads, bh, seq, s_pad, 0, seq, nqb,
                       scale * 1.4426950408889634f, window);
    return ea_check_launch("ea_attention_window_fwd_bf16");
}

extern "C" int ea_attention_window_mapped_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt_perm, const ea_bf16* cross,
                                                   ea_bf16* out, int64_t out_batch_stride, int batch, int heads, int seq, int s_pad,
                                                   int vt_pad, int row_off, const int* map, int window, float scale, void* stream) {
    EA_REQUIRE(q && k && vt_perm && cross && out && map, "ea_attention_window_mapped_fwd_bf16: null tensor");
    EA_REQUIRE(batch > 0 && heads > 0 && seq > 0 && window >= 0 && row_off >= 0, "ea_attention_window_mapped_fwd_bf16: bad sizes");
    EA_REQUIRE(row_off + seq <= s_pad && vt_pad % ATT_KV == 0 && vt_pad >= (seq + ATT_KV - 1) / ATT_KV * ATT_KV,
               "ea_attention_window_mapped_fwd_bf16: q / k hold rows [row_off, row_off + seq) of s_pad; vt_perm rows of vt_pad >= seq rounded up to 64");
    EA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt_perm | (uintptr_t)cross | (uintptr_t)out) & 15) == 0 && ((uintptr_t)map & 3) == 0 &&
               out_batch_stride % 8 == 0, "ea_attention_window_mapped_fwd_bf16: pointers must be 16-byte aligned");
    const int nqb = (seq + ATT_QB - 1) / ATT_QB;
    const int bh = batch * heads;
    const int64_t blocks = (int64_t)((bh + 7) / 8) * nqb * 8;
    EA_REQUIRE(blocks < (1ll << 31), "ea_attention_window_mapped_fwd_bf16: grid too large");
    AttWindowMap wm;
    wm.map = map; wm.cross = cross; wm.row_off = row_off; wm.vt_pad = vt_pad;
    ea_count("attention_window_mapped");
    hipLaunchKernelGGL((attention_fwd_kernel<true, true>), dim3((unsigned)blocks), dim3(256), ATT_LDS, (hipStream_t)stream, q, k, vt_perm,
                       (unsigned short*)out, out_batch_stride, heads, bh, seq, s_pad, 0, seq, nqb, scale * 1.4426950408889634f, window, wm);
    return ea_check_launch("ea_attention_window_mapped_fwd_bf16");
}

extern "C" int64_t ea_attention_state_bytes(int batch, int heads, int q_begin, int q_end) {
    const int64_t nqb = (q_end - q_begin + ATT_QB - 1) / ATT_QB;
    return (int64_t)batch * heads * nqb * ATT3_STATE_F4 * 256 * 16;   // v3 layout (18 float4 / thread) >= v2 (17)
}

extern "C" int ea_attention_fwd_range_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                           int64_t out_batch_stride, int batch, int heads, int s_pad, int q_begin,
                                           int q_end, int kv_begin, int kv_end, float scale, float* state, int flags,
                                           void* stream) {
    return attention_launch(q, k, vt, out, out_batch_stride, batch, heads, s_pad, q_begin, q_end, kv_begin, kv_end, scale,
                            state, flags, stream);
}

extern "C" int ea_attention_fwd_range_heads_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                                 int64_t out_batch_stride, int batch, int heads, int s_pad, int q_begin,
                                                 int q_end, int kv_begin, int kv_end, float scale, float* state, int flags,
                                                 int q_head0, int q_heads, int64_t kv_batch_stride, void* stream) {
    return attention_launch(q, k, vt, out, out_batch_stride, batch, heads, s_pad, q_begin, q_end, kv_begin, kv_end, scale,
                            state, flags, stream, q_head0, q_heads, kv_batch_stride);
}
