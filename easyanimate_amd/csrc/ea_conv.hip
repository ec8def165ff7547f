// Causal 3-D convolution of the MAGVIT VAE as an im2col-free implicit GEMM on channels-last activations.
// reference: easyanimate/vae/ldm/modules/vaemodules/common.py:84-179 (CausalConv3d), downsamplers.py:24-94,
// upsamplers.py:21-37,123-153, ResidualBlock3D common.py:298-323.
//
//   y[to,ho,wo,:] = bias + sum_{dt,dh,dw} W[:, (dt,dh,dw), :] . x[clamp(to*st+dt-(kt-1), 0), ho*ss+dh-pad, wo*ss+dw-pad, :]
//
// GEMM view: M = T_out*H_out*W_out output voxels, N = C_out, K = taps*C_in (tap-major, channel-minor).  With
// NDHWC activations one A-tile row (one voxel, 64 channels of one tap) is 128 contiguous bytes -- exactly one
// LDS-DMA row -- so the A tile is *gathered by address*: each lane points its global_load_lds at the voxel the
// tap selects, or at a page of zeros for the spatial zero padding.  Temporal causality = clamping the frame
// index at 0 (replicate padding of the first frame; the reference's chunk caches reproduce exactly this, SURVEY
// 8c property 1).  Nearest x2 spatial up-sampling of the input (upsamplers.py:35,143) is `>> 1` in the address.
// The epilogue can add a residual (ResidualBlock3D) and can write every output frame t >= 1 twice (frames
// 2t-1, 2t): the temporal nearest x2 of SpatialTemporalUpsampler3D (upsamplers.py:146-152) costs no extra pass.
//
// Tile/MFMA/LDS structure is that of ea_gemm.hip (128 x 128 x 64, 4 waves, LDS-DMA + source swizzle, C^T MFMA).
#include "ea_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int CONV_LDS = 2 * STAGE_BYTES;

struct ConvArgs {
    const unsigned short* x;
    const unsigned short* w;
    const float* bias;
    const unsigned short* res;
    unsigned short* y;
    const unsigned short* zeros;  // >= 128 bytes of zeros
    int T_in, H_in, W_in, C_in, C_out;
    int T_out, H_out, W_out;
    int kt, kh, kw, st, ss, pad, ups, tdup;
    // virtual temporal x2 (SpatialTemporalUpsampler3D's nearest duplication kept un-materialised): vin -- x holds the T
    // frames the up-sampler computed, the LOGICAL input has 2T-1, logical frame f lives in physical frame (f + 1) >> 1;
    // vres -- the same for the residual operand (the block's shortcut, computed on the physical frames)
    int vin, vres;
    // tmerge (with vin): the three temporal taps of a layer whose input is a virtually duplicated clip collapse to TWO physical
    // frames -- logical frames (t-2, t-1, t) are physical (p-1, p-1, p) for odd t and (p-1, p, p) for even t, p = (t+1)>>1 -- so
    // p.w holds two merged-weight classes [2][C_out, 18 * C_in] (even t: {W0, W1+W2}; odd t: {W0+W1, W2}) and the K loop runs
    // over 2 x 3 x 3 taps: 2/3 of the MFMA work (row-slab kernels only)
    int tmerge;
    int x_cb;            // x is channel-blocked [C_in / 32][T_in][H_in][W_in][32] (conv3d_cl_row16_w4a_kernel<.., .., true> only)
    int tiles_m, tiles_n;
    int64_t M;
    float* gn_partial;   // optional (row-slab 16x16x32 kernel): per-(frame, row tile, wave row, 4-channel bundle) (sum, sumsq)
    int gn_nblk;         // partial blocks per output frame = (H_out * W_out / 256) * (waves along M)
};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// Buffer-addressed LDS-DMA: 16 bytes per lane from base + voff (per lane) + soff (scalar) to lds_wave_base + 16 * lane.  The
// descriptor (base, extent: wave-uniform, SGPRs) range-checks every lane: an offset at or beyond `extent` writes zeros.
// (A plain function, not inlined text in the kernel templates: the host pass drops template kernels whose bodies name the
// device-only descriptor type.)
__device__ __forceinline__ void bdma16(const void* base, int extent, int voff, int soff, void* lds_wave_base) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, extent, 0x00020000);   // raw buffer, 32-bit format
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int swap23(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// C8: the 8-channel input layers (encoder conv_in: RGB padded to 8 channels = one 16-byte chunk per voxel).  A K tile is then
// eight TAPS, one chunk each: the lane that stages chunk c of a row gathers it from the voxel tap 8 t + c selects (taps
// beyond kt*kh*kw read the page of zeros; the packed weight has zero columns there).  27 taps = 4 K tiles instead of an
// im2col pass (a 13 GB round trip at 49 x 1024^2) followed by a GEMM.
template <bool C8>
__global__ __launch_bounds__(256, 2) void conv3d_cl_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;

    // per-XCD bands of M-tiles, all N-tiles of an M-tile adjacent: neighbouring voxels (shared halos) and the
    // whole weight set stay in one XCD's L2
    int tm, tn;
    {
        const int rpx = (p.tiles_m + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int m_lo = xcd * rpx;
        int rows = p.tiles_m - m_lo;
        rows = rows < rpx ? rows : rpx;
        if (rows <= 0 || idx >= rows * p.tiles_n) return;
        tm = m_lo + idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int64_t row0 = (int64_t)tm * BM;
    const int col0 = tn * BN;

    // ---- the 4 A-rows (output voxels) this lane fetches per K-tile, and its W-row pointers
    int vt[4], vh[4], vw[4];
    bool vvalid[4];
    int csw[4];
    const unsigned short* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = (wave * 4 + i) * 64 + lane;
        const int r = L >> 3, c = L & 7;
        csw[i] = (c ^ ((r >> 1) & 7)) * 8;
        int64_t m = row0 + r;
        vvalid[i] = m < p.M;
        m = vvalid[i] ? m : p.M - 1;
        if (C8) {   // (the host checks M < 2^31 for this path: 32-bit divisions -- a workgroup only runs four K tiles)
            const unsigned m32 = (unsigned)m, q32 = m32 / (unsigned)p.W_out;
            vw[i] = (int)(m32 - q32 * (unsigned)p.W_out);
            vt[i] = (int)(q32 / (unsigned)p.H_out);
            vh[i] = (int)(q32 - (unsigned)vt[i] * (unsigned)p.H_out);
        } else {
            vw[i] = (int)(m % p.W_out);
            const int64_t q = m / p.W_out;
            vh[i] = (int)(q % p.H_out);
            vt[i] = (int)(q / p.H_out);
        }
        int rw = col0 + r;
        rw = rw < p.C_out ? rw : p.C_out - 1;
        const int64_t wk = C8 ? (int64_t)((p.kt * p.kh * p.kw + 7) / 8) * BK : (int64_t)p.kt * p.kh * p.kw * p.C_in;
        wsrc[i] = p.w + (int64_t)rw * wk + csw[i];
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[2], w_off[2], a_sw[2], w_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31;
        const int rw = wn * 64 + i * 32 + swap23(l31);
        a_off[i] = ra * 128;
        a_sw[i] = (ra >> 1) & 7;
        w_off[i] = rw * 128;
        w_sw[i] = (rw >> 1) & 7;
    }

    const int cblocks = C8 ? 1 : p.C_in / BK;
    const int ntaps = p.kt * p.kh * p.kw;
    const int nk = C8 ? (ntaps + 7) / 8 : ntaps * cblocks;
    const int H_eff = p.ups ? p.H_in * 2 : p.H_in;
    const int W_eff = p.ups ? p.W_in * 2 : p.W_in;

    auto issue = [&](int t, int stage) {
        int tap = t / cblocks;
        const int cb = t - tap * cblocks;
        int dw = tap % p.kw;
        int dh = (tap / p.kw) % p.kh;
        int dt = tap / (p.kw * p.kh);
        char* sa = smem + stage * STAGE_BYTES + wave * 4096;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (C8) {   // per lane: the tap of this lane's chunk (3 x 3 x 3: constant divisors)
                tap = t * 8 + (csw[i] >> 3);
                dt = tap / 9;
                const int r9 = tap - dt * 9;
                dh = r9 / 3;
                dw = r9 - dh * 3;
            }
            int ti = vt[i] * p.st + dt - (p.kt - 1);
            ti = ti < 0 ? 0 : ti;  // causal replicate padding
            ti = p.vin ? (ti + 1) >> 1 : ti;
            int hh = vh[i] * p.ss + dh - p.pad;
            int ww = vw[i] * p.ss + dw - p.pad;
            const bool ok = (unsigned)hh < (unsigned)H_eff && (unsigned)ww < (unsigned)W_eff && (!C8 || tap < ntaps);
            hh >>= (p.ups ? 1 : 0);
            ww >>= (p.ups ? 1 : 0);
            const unsigned vox = (unsigned)((ti * p.H_in + hh) * p.W_in + ww);   // < 2^31 voxels per clip
            const unsigned short* src = C8 ? p.x + (uint64_t)vox * 8u : p.x + (uint64_t)vox * (unsigned)p.C_in + cb * BK + csw[i];
            src = ok ? src : p.zeros + csw[i];
            glds16(src, sa + i * 1024);
            glds16(wsrc[i] + (int64_t)t * BK, sw + i * 1024);
        }
    };

    issue(0, 0);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < nk) issue(t + 1, (t + 1) & 1);
        const char* sa = smem + (t & 1) * STAGE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ks * 2 + hi;
            bf16x8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off[i] + ((ch ^ a_sw[i]) << 4));
                wf[i] = *reinterpret_cast<const bf16x8*>(sw + w_off[i] + ((ch ^ w_sw[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns voxel m, channels n0..n0+7
    const int64_t frame = (int64_t)p.H_out * p.W_out;
    // GroupNorm partial sums of the output (see conv3d_cl_row16_kernel): the host enables them only when a 128-voxel tile
    // never straddles two frames and nothing is duplicated; [j][g][4-channel bundle]
    const bool has_gn = p.gn_partial != nullptr;
    float gs[2][2][2], gq[2][2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int b = 0; b < 2; ++b) gs[j][g][b] = gq[j][g][b] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = row0 + wm * 64 + i * 32 + l31;
        if (m >= p.M) continue;
        int64_t m_dst0 = m, m_dst1 = -1;
        if (p.tdup) {
            const int64_t to = m / frame, rem = m - to * frame;
            if (to >= 1) {
                m_dst0 = (2 * to - 1) * frame + rem;
                m_dst1 = (2 * to) * frame + rem;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int n0 = col0 + wn * 64 + j * 32 + g * 16 + hi * 8;
                if (n0 >= p.C_out) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][j][g * 8 + e];
                if (p.bias) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] += b0[e];
                        v[4 + e] += b1[e];
                    }
                }
                if (p.res) {
                    int64_t mr = m;
                    if (p.vres) {   // the residual holds the physical frames of a virtually duplicated clip
                        const int64_t tr = m / frame;
                        mr = m - (tr - ((tr + 1) >> 1)) * frame;
                    }
                    const u16x8 rr = *reinterpret_cast<const u16x8*>(p.res + mr * p.C_out + n0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf16_bits_to_f32(rr[e]);
                }
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16_bits(v[e]);
                *reinterpret_cast<u16x8*>(p.y + m_dst0 * p.C_out + n0) = o;
                if (m_dst1 >= 0) *reinterpret_cast<u16x8*>(p.y + m_dst1 * p.C_out + n0) = o;
                if (has_gn) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float r = bf16_bits_to_f32(o[e]);   // the value the next layer's GroupNorm reads
                        gs[j][g][e >> 2] += r;
                        gq[j][g][e >> 2] += r * r;
                    }
                }
            }
        }
    }
    if (has_gn) {
        // 32 lanes (l31) hold the 64 voxels of the wave tile, two each: fixed-order butterfly, then one (sum, sumsq) pair
        // per (frame, 128-voxel tile, wave row, 4-channel bundle) -- the layout ea_groupnorm_finalize_bf16 reads
        const int64_t t_out = row0 / frame;
        const int64_t blk = t_out * p.gn_nblk + ((row0 - t_out * frame) >> 7) * 2 + wm;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float s_ = gs[j][g][b], q_ = gq[j][g][b];
#pragma unroll
                    for (int o_ = 1; o_ < 32; o_ <<= 1) {
                        s_ += __shfl_xor(s_, o_, 64);
                        q_ += __shfl_xor(q_, o_, 64);
                    }
                    const int n0 = col0 + wn * 64 + j * 32 + g * 16 + hi * 8 + b * 4;
                    if (l31 == 0 && n0 < p.C_out) {
                        float* dst = p.gn_partial + (blk * (p.C_out >> 2) + (n0 >> 2)) * 2;
                        dst[0] = s_;
                        dst[1] = q_;
                    }
                }
    }
}


// =================================================================================================
// 256 x {128 | 256} x 64 "ping-pong" variant for the large convolutions (the structure of gemm256_bf16_kernel in
// ea_gemm.hip: 8 waves, two wave groups staggered by one barrier, four phases per K-tile, s_setprio around the MFMA
// clusters, one workgroup per CU).  K-tiles run tap-major / channel-block-minor; the per-lane gather pointers of the
// A rows are recomputed only when the tap changes (every C_in/64 tiles) -- in the L-part of phase 0, i.e. under the
// other wave group's MFMAs -- with incremental (dt, dh, dw) counters instead of divisions.
//   256 x 256: waves 2 (M) x 4 (N), wave tile 128 x 64.   C_out = 128: 512 x 128, waves 4 x 2, wave tile 128 x 64 (all
//   160 KiB of LDS: the same 6 fragment reads per 8 MFMAs as 256 x 256); 256 x 128 (wave tile 64 x 64) for short M.
template <int BM, int BN>
__global__ __launch_bounds__(512, 2) void conv3d_cl_pp_kernel(ConvArgs p) {
    constexpr int WN = BN / 64, WM = 8 / WN;
    constexpr int MI = BM / WM / 32;           // 32-row MFMA tiles per wave along M
    constexpr int AP = BM / 64;                // A-tile DMA pieces (8 rows x 128 B) per wave
    constexpr int WP = BN / 64;                // W-tile DMA pieces per wave
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // A[2 stages] | W[2 stages]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;                 // waves w and w+4 share a SIMD: one of each group per SIMD
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    {
        const int rpx = (p.tiles_m + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int m_lo = xcd * rpx;
        int rows = p.tiles_m - m_lo;
        rows = rows < rpx ? rows : rpx;
        if (rows <= 0 || idx >= rows * p.tiles_n) return;
        tm = m_lo + idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int64_t row0 = (int64_t)tm * BM;
    const int col0 = tn * BN;

    // ---- this lane's AP A rows (output voxels) and W rows
    int vt[AP], vh[AP], vw[AP], csw[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int r = (wave * AP + i) * 8 + (lane >> 3), c = lane & 7;
        csw[i] = (c ^ ((r >> 1) & 7)) * 8;
        int64_t m = row0 + r;
        m = m < p.M ? m : p.M - 1;
        vw[i] = (int)(m % p.W_out);
        const int64_t q = m / p.W_out;
        vh[i] = (int)(q % p.H_out) * p.ss - p.pad;     // input coordinates of tap (0,0,0)
        vt[i] = (int)(q / p.H_out) * p.st - (p.kt - 1);
        vw[i] = vw[i] * p.ss - p.pad;
    }
    const int64_t wk = (int64_t)p.kt * p.kh * p.kw * p.C_in;
    const unsigned short* wsrc[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave * WP + i) * 8 + (lane >> 3), c = lane & 7;
        int rw = col0 + r;
        rw = rw < p.C_out ? rw : p.C_out - 1;
        wsrc[i] = p.w + (int64_t)rw * wk + ((c ^ ((r >> 1) & 7)) * 8);
    }
    char* const dma_a = smem + wave * (AP * 1024);
    char* const dma_w = smem + 2 * A_BYTES + wave * (WP * 1024);

    const int H_eff = p.ups ? p.H_in * 2 : p.H_in;
    const int W_eff = p.ups ? p.W_in * 2 : p.W_in;
    const int ups_sh = p.ups ? 1 : 0;
    const unsigned short* a_tap[AP];
    auto set_tap = [&](int dt, int dh, int dw) {
        // 32-bit voxel index (a clip has < 2^31 voxels), one 32x32->64 multiply-add for the address, selects instead
        // of branches: ~18 VALU per row (the straightforward int64 form compiled to ~125 under exec-masked branches)
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            int ti = vt[i] + dt;
            ti = ti < 0 ? 0 : ti;                       // causal replicate padding
            ti = p.vin ? (ti + 1) >> 1 : ti;
            int hh = vh[i] + dh, ww = vw[i] + dw;
            const bool ok = (unsigned)hh < (unsigned)H_eff && (unsigned)ww < (unsigned)W_eff;
            hh >>= ups_sh;
            ww >>= ups_sh;
            const unsigned vox = (unsigned)((ti * p.H_in + hh) * p.W_in + ww);
            const unsigned short* src = p.x + (uint64_t)vox * (unsigned)p.C_in;
            src = ok ? src : p.zeros;
            a_tap[i] = src + csw[i];
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = wr * (MI * 32) + l31;
    const int w_row = wc * 64 + swap23(l31);
    const int a_sw = (l31 >> 1) & 7, w_sw = (swap23(l31) >> 1) & 7;
    const char* a_k[4];
    const char* w_k[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_k[ks] = smem + a_row * 128 + (((ks * 2 + hi) ^ a_sw) << 4);
        w_k[ks] = smem + 2 * A_BYTES + w_row * 128 + (((ks * 2 + hi) ^ w_sw) << 4);
    }

    const int cblocks = p.C_in / BK;
    const int nk = p.kt * p.kh * p.kw * cblocks;
    // position of the NEXT tile to stage: (dt, dh, dw, cb); tile t reads channels cb*64.. of tap (dt,dh,dw)
    int n_dt = 0, n_dh = 0, n_dw = 0, n_cb = 0;
    auto advance = [&]() {
        if (++n_cb == cblocks) {
            n_cb = 0;
            if (++n_dw == p.kw) {
                n_dw = 0;
                if (++n_dh == p.kh) {
                    n_dh = 0;
                    ++n_dt;
                }
            }
            set_tap(n_dt, n_dh, n_dw);
        }
    };

#define EA_C2_PHASE(S, KS, HAS_NEXT)                                                                        \
    {                                                                                                       \
        bf16x8 af[MI], wf[2];                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            wf[j] = *reinterpret_cast<const bf16x8*>(w_k[KS] + (S) * W_BYTES + j * 4096);                   \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                      \
            af[i] = *reinterpret_cast<const bf16x8*>(a_k[KS] + (S) * A_BYTES + i * 4096);                   \
        if ((KS) == 0 && (HAS_NEXT)) {                                                                      \
            advance();                                                                                      \
            _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                \
                const unsigned short* src = a_tap[i];                                                       \
                src = (src >= p.zeros && src < p.zeros + 64) ? src : src + n_cb * BK;                       \
                glds16(src, dma_a + ((S) ^ 1) * A_BYTES + i * 1024);                                        \
            }                                                                                               \
        }                                                                                                   \
        if ((KS) == 1 && (HAS_NEXT)) {                                                                      \
            _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                \
                wsrc[i] += BK;                                                                              \
                glds16(wsrc[i], dma_w + ((S) ^ 1) * W_BYTES + i * 1024);                                    \
            }                                                                                               \
        }                                                                                                   \
        if ((KS) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);      \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
#define EA_C2_TILE(S, HAS_NEXT) \
    EA_C2_PHASE(S, 0, HAS_NEXT) \
    EA_C2_PHASE(S, 1, HAS_NEXT) \
    EA_C2_PHASE(S, 2, HAS_NEXT) \
    EA_C2_PHASE(S, 3, HAS_NEXT)

    // ---- prologue: tile 0 (tap 0, channel block 0) -> stage 0
    set_tap(0, 0, 0);
#pragma unroll
    for (int i = 0; i < AP; ++i) glds16(a_tap[i], dma_a + i * 1024);
#pragma unroll
    for (int i = 0; i < WP; ++i) glds16(wsrc[i], dma_w + i * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // stagger: group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nk; t += 2) {
        const bool n0_ = t + 1 < nk;
        EA_C2_TILE(0, n0_)
        if (n0_) {
            const bool n1_ = t + 2 < nk;
            EA_C2_TILE(1, n1_)
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
#undef EA_C2_TILE
#undef EA_C2_PHASE

    // ---- epilogue: lane owns voxel m, channels n0..n0+7
    const int64_t frame = (int64_t)p.H_out * p.W_out;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = row0 + wr * (MI * 32) + i * 32 + l31;
        if (m >= p.M) continue;
        int64_t m_dst0 = m, m_dst1 = -1;
        if (p.tdup) {
            const int64_t to = m / frame, rem = m - to * frame;
            if (to >= 1) {
                m_dst0 = (2 * to - 1) * frame + rem;
                m_dst1 = (2 * to) * frame + rem;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int n0 = col0 + wc * 64 + j * 32 + g * 16 + hi * 8;
                if (n0 >= p.C_out) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][j][g * 8 + e];
                if (p.bias) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] += b0[e];
                        v[4 + e] += b1[e];
                    }
                }
                if (p.res) {
                    int64_t mr = m;
                    if (p.vres) {   // the residual holds the physical frames of a virtually duplicated clip
                        const int64_t tr = m / frame;
                        mr = m - (tr - ((tr + 1) >> 1)) * frame;
                    }
                    const u16x8 rr = *reinterpret_cast<const u16x8*>(p.res + mr * p.C_out + n0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf16_bits_to_f32(rr[e]);
                }
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16_bits(v[e]);
                *reinterpret_cast<u16x8*>(p.y + m_dst0 * p.C_out + n0) = o;
                if (m_dst1 >= 0) *reinterpret_cast<u16x8*>(p.y + m_dst1 * p.C_out + n0) = o;
            }
        }
    }
}


// =================================================================================================
// "Row-slab" variant of the ping-pong kernel for the plain 3x3x3 / stride 1 / pad 1 convolutions (no folded
// up-sampling) whose output rows are a multiple of 256 voxels wide -- i.e. the full-resolution layers that dominate the
// VAE.  An M-tile is 256 consecutive voxels of ONE output row, so the three taps dw = 0,1,2 of a (dt, dh) pair read
// the same input row shifted by one voxel: the A operand is staged ONCE per (dt, dh, channel block) as a 258-row slab
// (w0-1 .. w0+256; 33 LDS-DMA pieces) and consumed three times with fragment row offsets 0 / 1 / 2.  LDS-DMA traffic per
// MFMA drops 1.8x (BN = 128) / 1.5x (BN = 256) against the tile-per-tap kernels above.
//   K order: (dt, dh) -> channel block -> dw  (the accumulation order differs from the other kernels: results agree to
//   fp32 summation-order noise, not bit for bit).
//   LDS: A slab stages at 0 and 34 KiB (33 KiB used each), W stages behind them; BN = 128: 100 KiB, BN = 256: 132 KiB.
template <int BN, bool UPS>
__global__ __launch_bounds__(512, 2) void conv3d_cl_row_kernel(ConvArgs p) {
    // UPS: nearest x2 up-sampling folded into the addressing -- output voxel u reads input voxel u >> 1, so the slab holds
    // the 130 input voxels under the 258 up-sampled ones and two neighbouring lanes share a fragment row
    constexpr int NPIECE = UPS ? 17 : 33, PPW = UPS ? 3 : 5, NROW = UPS ? 130 : 258, ISTEP = UPS ? 2048 : 4096;
    constexpr int WN = BN / 64, WM = 8 / WN;
    constexpr int MI = 256 / WM / 32;
    constexpr int WP = BN / 64;
    constexpr int A_STAGE = 34 * 1024, W_BYTES = BN * 128, W_BASE = 2 * A_STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;
    const int hi = lane >> 5, l31 = lane & 31;

    int tm, tn;
    {
        const int rpx = (p.tiles_m + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int m_lo = xcd * rpx;
        int rows = p.tiles_m - m_lo;
        rows = rows < rpx ? rows : rpx;
        if (rows <= 0 || idx >= rows * p.tiles_n) return;
        tm = m_lo + idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int tiles_w = p.W_out / 256;
    const int w0 = (tm % tiles_w) * 256;
    const int orow = tm / tiles_w;                 // t_out * H_out + h_out
    const int h_out = orow % p.H_out, t_out = orow / p.H_out;
    const int col0 = tn * BN;

    // ---- this lane's slab rows: piece q = wave*5 + i (q < 33), LDS row r = 8q + lane/8  <->  input voxel w0 - 1 + r
    int a_woff[PPW];      // element offset of (voxel, source chunk) inside an input row, or -1: zero padding / unused row
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int r = q * 8 + (lane >> 3), c = lane & 7;
        const int w = (UPS ? (w0 >> 1) : w0) - 1 + r;
        a_woff[i] = (q < NPIECE && r < NROW && w >= 0 && w < p.W_in) ? w * p.C_in + (c ^ ((r >> 1) & 7)) * 8 : -1;
    }
    const int zoff = (lane & 7) * 8;   // any 16 bytes of the zero page will do
    const int64_t wk = (int64_t)27 * p.C_in;
    const unsigned short* wbase[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave * WP + i) * 8 + (lane >> 3), c = lane & 7;
        int rw = col0 + r;
        rw = rw < p.C_out ? rw : p.C_out - 1;
        wbase[i] = p.w + (int64_t)rw * wk + ((c ^ ((r >> 1) & 7)) * 8);
    }
    char* const dma_a = smem + wave * PPW * 1024;
    char* const dma_w = smem + W_BASE + wave * (WP * 1024);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: A row = wr*(MI*32) + i*32 + l31 + dw (the swizzle term follows the actual LDS row)
    unsigned a_k[3][4];   // LDS byte offsets
    unsigned w_k[4];
    const int w_row = wc * 64 + swap23(l31);
    const int w_sw = (swap23(l31) >> 1) & 7;
#pragma unroll
    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
        {
            const int row = UPS ? (wr * (MI * 16) + ((l31 + dw + 1) >> 1)) : (wr * (MI * 32) + l31 + dw);
            a_k[dw][ks] = row * 128 + (((ks * 2 + hi) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w_k[ks] = W_BASE + w_row * 128 + (((ks * 2 + hi) ^ w_sw) << 4);

    const int cblocks = p.C_in / BK;
    // the NEXT tile to stage: (dtdh, cb, dw)
    int n_dtdh = 0, n_cb = 0;
    const unsigned short* slab_row = p.zeros;   // input row (ti, hh) of the staged (dt, dh), or the zero page
    bool slab_ok = false;
    auto set_slab = [&](int dtdh) {
        const int dt = dtdh / 3, dh = dtdh - dt * 3;
        int ti = t_out + dt - 2;
        ti = ti < 0 ? 0 : ti;                                  // causal replicate padding
        ti = p.vin ? (ti + 1) >> 1 : ti;                       // virtual temporal x2
        const int hu = h_out + dh - 1;                         // row in the (up-sampled) padded input
        slab_ok = hu >= 0 && hu < p.H_out;                     // wave-uniform (stride 1, pad 1: H_out rows)
        const int hh = UPS ? hu >> 1 : hu;
        slab_row = p.x + ((int64_t)ti * p.H_in + (slab_ok ? hh : 0)) * p.W_in * p.C_in;
    };
    int a_dst = 0, w_dst = 0;                   // stage (0 / 1) the NEXT A slab / W tile is written to
    auto stage_a = [&](int sa, int cb) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (wave * PPW + i < NPIECE) {                     // wave-uniform
                const bool ok = slab_ok && a_woff[i] >= 0;
                const unsigned short* src = ok ? slab_row + a_woff[i] + cb * BK : p.zeros + zoff;
                glds16(src, dma_a + sa * A_STAGE + i * 1024);
            }
        }
    };
    auto stage_w = [&](int sw, int dtdh, int cb, int dw) {
        const int koff = (dtdh * 3 + dw) * p.C_in + cb * BK;
#pragma unroll
        for (int i = 0; i < WP; ++i) glds16(wbase[i] + koff, dma_w + sw * W_BYTES + i * 1024);
    };
    auto next_slab = [&]() {
        if (++n_cb == cblocks) {
            n_cb = 0;
            ++n_dtdh;
            set_slab(n_dtdh);
        }
    };

#define EA_C3_PHASE(DW, KS, HAS_NEXT)                                                                       \
    {                                                                                                       \
        bf16x8 af[MI], wf[2];                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            wf[j] = *reinterpret_cast<const bf16x8*>(smem + (w_k[KS] + j * 4096));                          \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                      \
            af[i] = *reinterpret_cast<const bf16x8*>(smem + (a_k[DW][KS] + i * ISTEP));                     \
        if ((KS) == 0 && (DW) == 2 && (HAS_NEXT)) {                                                         \
            next_slab();                                                                                    \
            stage_a(a_dst, n_cb);                                                                           \
        }                                                                                                   \
        if ((KS) == 1 && (HAS_NEXT)) stage_w(w_dst, n_dtdh, n_cb, ((DW) + 1) % 3);                          \
        if ((KS) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);      \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
    // one tile = one (slab, dw): four k-steps; afterwards the W stage toggles (in place, on the fragment offsets)
#define EA_C3_TILE(DW, HAS_NEXT)                                          \
    EA_C3_PHASE(DW, 0, HAS_NEXT)                                          \
    EA_C3_PHASE(DW, 1, HAS_NEXT)                                          \
    EA_C3_PHASE(DW, 2, HAS_NEXT)                                          \
    EA_C3_PHASE(DW, 3, HAS_NEXT)                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) w_k[ks] += w_step;   \
    w_step = -w_step;                                                     \
    w_dst ^= 1;

    // ---- prologue: slab (dt,dh) = 0, channel block 0 -> A stage 0; its dw = 0 weights -> W stage 0
    set_slab(0);
    stage_a(0, 0);
    stage_w(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // one trip = one slab = three tiles (dw = 0, 1, 2); the A stage toggles per slab, the W stage per tile
    const int nslabs = 9 * cblocks;
    int a_step = A_STAGE, w_step = W_BYTES;
    a_dst = 1;
    w_dst = 1;
    for (int sl = 0; sl < nslabs; ++sl) {
        EA_C3_TILE(0, true)
        EA_C3_TILE(1, true)
        EA_C3_TILE(2, sl + 1 < nslabs)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a_k[dw][ks] += a_step;
        a_step = -a_step;
        a_dst ^= 1;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef EA_C3_TILE
#undef EA_C3_PHASE

    // ---- epilogue: lane owns voxel m, channels n0..n0+7
    const int64_t frame = (int64_t)p.H_out * p.W_out;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = (int64_t)orow * p.W_out + w0 + wr * (MI * 32) + i * 32 + l31;
        int64_t m_dst0 = m, m_dst1 = -1;
        if (p.tdup && t_out >= 1) {
            const int64_t rem = m - (int64_t)t_out * frame;
            m_dst0 = (2 * (int64_t)t_out - 1) * frame + rem;
            m_dst1 = (2 * (int64_t)t_out) * frame + rem;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int n0 = col0 + wc * 64 + j * 32 + g * 16 + hi * 8;
                if (n0 >= p.C_out) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][j][g * 8 + e];
                if (p.bias) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] += b0[e];
                        v[4 + e] += b1[e];
                    }
                }
                if (p.res) {
                    int64_t mr = m;
                    if (p.vres) {   // the residual holds the physical frames of a virtually duplicated clip
                        const int64_t tr = m / frame;
                        mr = m - (tr - ((tr + 1) >> 1)) * frame;
                    }
                    const u16x8 rr = *reinterpret_cast<const u16x8*>(p.res + mr * p.C_out + n0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf16_bits_to_f32(rr[e]);
                }
                u16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16_bits(v[e]);
                *reinterpret_cast<u16x8*>(p.y + m_dst0 * p.C_out + n0) = o;
                if (m_dst1 >= 0) *reinterpret_cast<u16x8*>(p.y + m_dst1 * p.C_out + n0) = o;
            }
        }
    }
}

// The row-slab kernel on v_mfma_f32_16x16x32_bf16 (see gemm256_mi16_kernel in ea_gemm.hip: the shape sustains 14 % more under
// the power limit).  Wave tile (256 / WM) voxels x 64 channels = MT x 4 tiles of 16 x 16; a (slab, dw) K-tile is four
// phases (k32 step, M half); LDS rows swizzled with row & 7; 8-byte stores (4 channels per lane and tile).
#ifndef EA_CONV_WIDE_PHASE
#define EA_CONV_WIDE_PHASE 1   // build-time A/B switch (EA_HIPCC_EXTRA=-DEA_CONV_WIDE_PHASE=0)
#endif
// SUB (1 / 2): the SUB-PIXEL form of "nearest x2 up-sampling, then 3x3x3 convolution" (upsamplers.py:21-37,123-153).  Output
// pixel (2i + a, 2j + b) sees only 2 x 2 distinct source pixels -- rows {i-1, i} for a = 0 / {i, i+1} for a = 1, the same for
// columns -- so each of the four parity classes (a, b) is a 3 x 2 x 2 convolution ON THE SOURCE GRID whose weights are sums of
// the original taps (vae_modules._pack_subpixel_weight): 12 taps instead of 27, 44 % of the MFMA work.  A workgroup computes
// 256 source voxels of one source row for ONE class: a = blockIdx.y, b = SUB - 1 (compile time: it selects the two shifted
// fragment-row sets dw = b, b + 1 of the 258-row slab); the slabs are the (dt, dh') pairs with source row h + a - 1 + dh';
// the outputs land at voxel (2h + a, 2w + b) of the up-sampled clip.  p.w holds the four classes' packed weights
// [4][C_out, 12 * C_in] (taps ordered dt, dh', dw'), p.H_out / p.W_out are the up-sampled sizes.
template <int BN, bool UPS, int SUB = 0>
__global__ __launch_bounds__(512, 2) void conv3d_cl_row16_kernel(ConvArgs p) {
    // UPS: nearest x2 up-sampling folded into the addressing -- output voxel u reads input voxel u >> 1, so the slab holds
    // the 130 input voxels under the 258 up-sampled ones and two neighbouring lanes share a fragment row
    static_assert(!(UPS && SUB), "the sub-pixel form works on the source grid");
    constexpr int NDH = SUB ? 2 : 3, NDW = SUB ? 2 : 3, DW0 = SUB == 2 ? 1 : 0, DWL = DW0 + NDW - 1;   // taps per axis, first / last dw
    constexpr int NPIECE = UPS ? 17 : 33, PPW = UPS ? 3 : 5, NROW = UPS ? 130 : 258, ISTEP = UPS ? 1024 : 2048;
    constexpr int WN = BN / 64, WM = 8 / WN;
    constexpr int MT = 256 / WM / 16, MH = MT / 2;   // 16-voxel MFMA tiles per wave, per phase
    constexpr bool WIDE_PHASE = (BN == 128) && (EA_CONV_WIDE_PHASE != 0);
    constexpr int WP = BN / 64;
    constexpr int A_STAGE = 34 * 1024, W_BYTES = BN * 128, W_BASE = 2 * A_STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;
    const int lr = lane & 15, lq = lane >> 4;

    int tm, tn;
    {
        const int rpx = (p.tiles_m + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int m_lo = xcd * rpx;
        int rows = p.tiles_m - m_lo;
        rows = rows < rpx ? rows : rpx;
        if (rows <= 0 || idx >= rows * p.tiles_n) return;
        tm = m_lo + idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int grid_h = SUB ? p.H_in : p.H_out;     // rows / row width of the grid the tiles walk (SUB: the source grid)
    const int tiles_w = (SUB ? p.W_in : p.W_out) / 256;
    const int w0 = (tm % tiles_w) * 256;
    const int orow = tm / tiles_w;                 // t_out * grid_h + h_out
    const int h_out = orow % grid_h, t_out = orow / grid_h;
    const int col0 = tn * BN;
    const int sub_a = SUB ? (int)blockIdx.y : 0;   // row parity class

    // ---- this lane's slab rows: piece q = wave*5 + i (q < 33), LDS row r = 8q + lane/8  <->  input voxel w0 - 1 + r.
    // The DMA is buffer-addressed (see the 512-voxel kernel below): descriptor = the slab's input row, a_voff = byte offset
    // of (voxel, source chunk) inside it; padding / unused rows point far beyond the row and read zeros
    int a_voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int r = q * 8 + (lane >> 3), c = lane & 7;
        const int w = (UPS ? (w0 >> 1) : w0) - 1 + r;
        a_voff[i] = (q < NPIECE && r < NROW && w >= 0 && w < p.W_in) ? (w * p.C_in + (c ^ (r & 7)) * 8) * 2 : 0x40000000;
    }
    const int wk = (SUB ? 12 : (p.tmerge ? 18 : 27)) * p.C_in;
    int w_voff[WP];       // byte offset of (weight row, source chunk) in this N tile's rows of the packed weights
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave * WP + i) * 8 + (lane >> 3), c = lane & 7;
        w_voff[i] = (r * wk + (c ^ (r & 7)) * 8) * 2;
    }
    // SUB: class (a, b)'s packed weights [C_out, 12 * C_in] are block 2 * a + b of p.w
    const unsigned short* const w_tile = p.w + (SUB ? (int64_t)(2 * sub_a + (SUB - 1)) * p.C_out * wk : 0) +
                                         (p.tmerge ? (int64_t)(t_out & 1) * p.C_out * wk : 0) + (int64_t)col0 * wk;
    const int w_bytes = BN * wk * 2, row_bytes = p.W_in * p.C_in * 2;
    char* const dma_a = smem + wave * PPW * 1024;
    char* const dma_w = smem + W_BASE + wave * (WP * 1024);

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment addresses (16x16x32: row = lane & 15, 16-byte chunk 4 * ks2 + lane / 16, swizzle row & 7):
    // A row = wr*(MT*16) + i*16 + lr + dw (the swizzle term follows the actual LDS row); W row = wc*64 + j*16 + lr
    unsigned a_k[3][2];   // LDS byte offsets
    unsigned w_k[2];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const int row = UPS ? (wr * (MT * 8) + ((lr + dw + 1) >> 1)) : (wr * (MT * 16) + lr + dw);
            a_k[dw][ks2] = row * 128 + (((ks2 * 4 + lq) ^ (row & 7)) << 4);
        }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) w_k[ks2] = W_BASE + (wc * 64 + lr) * 128 + (((ks2 * 4 + lq) ^ (lr & 7)) << 4);
    bf16x8 wf[4];

    const int cblocks = p.C_in / BK;
    // the NEXT tile to stage: (dtdh, cb, dw)
    int n_dtdh = 0, n_cb = 0;
    const unsigned short* slab_row = p.x;       // input row (ti, hh) of the staged (dt, dh)
    bool slab_ok = false;
    auto set_slab = [&](int dtdh) {
        const int dt = dtdh / NDH, dh = dtdh - dt * NDH;
        int ti = t_out + dt - 2;
        ti = ti < 0 ? 0 : ti;                                  // causal replicate padding
        ti = p.vin ? (ti + 1) >> 1 : ti;                       // virtual temporal x2
        if (p.tmerge) {                                        // two physical frames (p - 1, p), p = (t_out + 1) >> 1
            ti = ((t_out + 1) >> 1) - 1 + dt;
            ti = ti < 0 ? 0 : ti;
        }
        const int hu = SUB ? h_out + sub_a - 1 + dh : h_out + dh - 1;   // row in the (up-sampled) padded input / SUB: source row
        slab_ok = hu >= 0 && hu < grid_h;                      // wave-uniform (stride 1, pad 1)
        const int hh = UPS ? hu >> 1 : hu;
        slab_row = p.x + ((int64_t)ti * p.H_in + (slab_ok ? hh : 0)) * p.W_in * p.C_in;
    };
    int a_dst = 0, w_dst = 0;                   // stage (0 / 1) the NEXT A slab / W tile is written to
    auto stage_a = [&](int sa, int cb) {
        const int extent = slab_ok ? row_bytes : 0;            // a row of padding: every lane reads zeros
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (wave * PPW + i < NPIECE)                       // wave-uniform
                bdma16(slab_row, extent, a_voff[i], cb * (BK * 2), dma_a + sa * A_STAGE + i * 1024);
    };
    auto stage_w = [&](int sw, int dtdh, int cb, int dw) {
        const int koff = (dtdh * NDW + (dw - DW0)) * p.C_in + cb * BK;
#pragma unroll
        for (int i = 0; i < WP; ++i) bdma16(w_tile, w_bytes, w_voff[i], koff * 2, dma_w + sw * W_BYTES + i * 1024);
    };
    auto next_slab = [&]() {
        if (++n_cb == cblocks) {
            n_cb = 0;
            ++n_dtdh;
            set_slab(n_dtdh);
        }
    };

#define EA_C3_PHASE(DW, KS, HAS_NEXT)   /* KS = phase: k32 step KS >> 1, M half KS & 1 */                    \
    {                                                                                                       \
        bf16x8 af[MH];                                                                                      \
        if (((KS) & 1) == 0) {                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
                wf[j] = *reinterpret_cast<const bf16x8*>(smem + (w_k[(KS) >> 1] + j * 2048));               \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                      \
            af[i] = *reinterpret_cast<const bf16x8*>(smem + (a_k[DW][(KS) >> 1] + (((KS) & 1) * MH + i) * ISTEP)); \
        if ((KS) == 0 && (DW) == DWL && (HAS_NEXT)) {                                                       \
            next_slab();                                                                                    \
            stage_a(a_dst, n_cb);                                                                           \
        }                                                                                                   \
        if ((KS) == 1 && (HAS_NEXT)) stage_w(w_dst, n_dtdh, n_cb, (DW) == DWL ? DW0 : (DW) + 1);            \
        if ((KS) == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                  \
                acc[((KS) & 1) * MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
                    wf[j], af[i], acc[((KS) & 1) * MH + i][j], 0, 0, 0);                                    \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
    // BN = 128 (wave tile 64 voxels x 64 channels): a phase of 8 MFMAs (128 cycles) is too short next to the fixed cost of a
    // phase (two barriers, the fragment-read latency): the whole k32 step -- both M halves, 16 MFMAs -- is one phase, two
    // phases per tile.  The last phase retires its fragment reads before its first barrier (that frees the stage for the
    // DMA the other wave group issues right after that barrier) and waits for the next tile's DMA.
#define EA_C3_WPHASE(DW, KS, HAS_NEXT)   /* KS = k32 step */                                                \
    {                                                                                                       \
        bf16x8 af[MT];                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            wf[j] = *reinterpret_cast<const bf16x8*>(smem + (w_k[KS] + j * 2048));                          \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                      \
            af[i] = *reinterpret_cast<const bf16x8*>(smem + (a_k[DW][KS] + i * ISTEP));                     \
        if ((KS) == 0 && (DW) == DWL && (HAS_NEXT)) {                                                       \
            next_slab();                                                                                    \
            stage_a(a_dst, n_cb);                                                                           \
        }                                                                                                   \
        if ((KS) == 0 && (HAS_NEXT)) stage_w(w_dst, n_dtdh, n_cb, (DW) == DWL ? DW0 : (DW) + 1);            \
        if ((KS) == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);      \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
    // one tile = one (slab, dw): four k-steps; afterwards the W stage toggles (in place, on the fragment offsets)
#define EA_C3_TILE(DW, HAS_NEXT)                                          \
    if (WIDE_PHASE) {                                                     \
        EA_C3_WPHASE(DW, 0, HAS_NEXT)                                     \
        EA_C3_WPHASE(DW, 1, HAS_NEXT)                                     \
    } else {                                                              \
        EA_C3_PHASE(DW, 0, HAS_NEXT)                                      \
        EA_C3_PHASE(DW, 1, HAS_NEXT)                                      \
        EA_C3_PHASE(DW, 2, HAS_NEXT)                                      \
        EA_C3_PHASE(DW, 3, HAS_NEXT)                                      \
    }                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) w_k[ks] += w_step;   \
    w_step = -w_step;                                                     \
    w_dst ^= 1;

    // ---- prologue: slab (dt,dh) = 0, channel block 0 -> A stage 0; its dw = 0 weights -> W stage 0
    set_slab(0);
    stage_a(0, 0);
    stage_w(0, 0, 0, DW0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // one trip = one slab = three tiles (dw = 0, 1, 2); the A stage toggles per slab, the W stage per tile
    const int nslabs = (p.tmerge ? 2 : 3) * NDH * cblocks;
    int a_step = A_STAGE, w_step = W_BYTES;
    a_dst = 1;
    w_dst = 1;
    for (int sl = 0; sl < nslabs; ++sl) {
        if (SUB) {      // two dw' taps: fragment-row sets dw = b, b + 1
            EA_C3_TILE(DW0, true)
            EA_C3_TILE(DWL, sl + 1 < nslabs)
        } else {
            EA_C3_TILE(0, true)
            EA_C3_TILE(1, true)
            EA_C3_TILE(2, sl + 1 < nslabs)
        }
#pragma unroll
        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a_k[dw][ks] += a_step;
        a_step = -a_step;
        a_dst ^= 1;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef EA_C3_TILE
#undef EA_C3_PHASE
#undef EA_C3_WPHASE

    // ---- epilogue: lane owns voxel m (one per M tile), channels n0 .. n0+3 of every N tile.  The residual of M tile i + 1 is
    // requested before tile i is rounded and stored (uniform branches around each load would serialise the round trips)
    const int64_t frame = (int64_t)p.H_out * p.W_out;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_res = p.res != nullptr, has_gn = p.gn_partial != nullptr;
    const bool dup_t = p.tdup && t_out >= 1;
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (p.bias) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + col0 + wc * 64 + j * 16 + lq * 4);
        else b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // SUB: source voxel (h, w) of class (a, b) is output voxel (2h + a, 2w + b): consecutive M tiles are 32 output voxels apart
    const int64_t m_base = SUB ? ((int64_t)t_out * p.H_out + 2 * h_out + sub_a) * p.W_out + 2 * (w0 + wr * (MT * 16) + lr) + (SUB - 1)
                               : (int64_t)orow * p.W_out + w0 + wr * (MT * 16) + lr;
    const int ch0 = col0 + wc * 64 + lq * 4;
    const int64_t e_res = m_base * p.C_out + ch0;              // + i * e_step + j * 16
    const int64_t e_step = (int64_t)(SUB ? 32 : 16) * p.C_out;
    // virtual residual: logical frame t_out lives in physical frame (t_out + 1) >> 1 of p.res
    const unsigned short* const resp = p.res ? p.res - (p.vres ? (int64_t)(t_out - ((t_out + 1) >> 1)) * frame * p.C_out : 0) : nullptr;
    // with tdup the frame t_out >= 1 is stored twice: frames 2t - 1 and 2t of y
    const int64_t e_dst0 = dup_t ? e_res + ((int64_t)t_out - 1) * frame * p.C_out : e_res;
    const int64_t e_dup = frame * p.C_out;
    bf16x4 rr[2][4];
    if (has_res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[0][j] = *reinterpret_cast<const bf16x4*>(resp + e_res + j * 16);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (has_res && i + 1 < MT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[(i + 1) & 1][j] = *reinterpret_cast<const bf16x4*>(resp + e_res + (i + 1) * e_step + j * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[j][e];
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rr[i & 1][j][e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(p.y + e_dst0 + i * e_step + j * 16) = o;
            if (dup_t) *reinterpret_cast<bf16x4*>(p.y + e_dst0 + e_dup + i * e_step + j * 16) = o;
            if (has_gn) {   // GroupNorm statistics of the NEXT layer, over the values it will read (the rounded ones)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = (float)o[e];
                    gs[j] += r;
                    gq[j] += r * r;
                }
            }
        }
    }
    if (p.gn_partial) {
        // lanes lr = 0..15 hold the 16 voxels of every M tile: fixed-order butterfly over them, then one (sum, sumsq) pair
        // per 4-channel bundle and wave -- the layout ea_groupnorm_finalize_bf16 reads (deterministic, no atomics)
        // with tdup the frame t_out >= 1 is stored twice (frames 2t-1 and 2t of y): both get the same partial sums
        const bool dup = p.tdup && t_out >= 1;
        const int64_t f0 = dup ? 2 * (int64_t)t_out - 1 : t_out;
        // SUB: the four classes of a source row tile write four consecutive blocks (p.gn_nblk counts all of them)
        const int64_t in_frame = SUB ? ((((int64_t)h_out * tiles_w + (tm % tiles_w)) * 4 + 2 * sub_a + (SUB - 1)) * WM + wr)
                                     : ((int64_t)h_out * tiles_w + (tm % tiles_w)) * WM + wr;
        const int64_t blk = f0 * p.gn_nblk + in_frame;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s_ = gs[j], q_ = gq[j];
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) {
                s_ += __shfl_xor(s_, o_, 64);
                q_ += __shfl_xor(q_, o_, 64);
            }
            const int n0 = col0 + wc * 64 + j * 16 + lq * 4;
            if (lr == 0) {
                float* dst = p.gn_partial + (blk * (p.C_out >> 2) + (n0 >> 2)) * 2;
                dst[0] = s_;
                dst[1] = q_;
                if (dup) {
                    dst += (int64_t)p.gn_nblk * (p.C_out >> 2) * 2;
                    dst[0] = s_;
                    dst[1] = q_;
                }
            }
        }
    }
}

// =================================================================================================================
// The row-slab kernel for the C_out = 128 layers at full resolution (42 % of a 49 x 1024^2 decode): 512 voxels x 128
// channels per workgroup.  The 256-voxel version above gives each wave a 64 x 64 tile: 8 fragment reads (8 KiB) per 16
// MFMAs, and with eight waves that is 64 KiB of LDS reads per phase = 512 LDS cycles against 512 MFMA cycles per SIMD -- the
// LDS read port is exactly saturated and the MFMA pipe ends at 55 % (profiles/history/r02k_conv_row16_sq_counters.txt: it is not
// power-limited, 1.96 GHz).  A 128 x 64 wave tile needs 12 reads per 32 MFMAs (what the 256-channel tiles get), but its
// 514-row slab does not fit beside the weights at 64 channels per stage (164 KiB).  So this kernel stages HALF the
// channels: LDS rows are 64 bytes = 32 channels = exactly one k32 step,
//     A slab 514 rows x 64 B (33 KiB) x 2 stages + W tile 128 rows x 64 B (8 KiB) x 3 stages = 92 KiB,
// waves 4 (M) x 2 (N).  64-byte rows: the 16-byte chunk is XOR-swizzled with (row >> 1) & 3, so each ds_read_b128 lane group
// (MI355X_MICROARCH.md, LDS) covers the 16 slots of a bank row once, for all three dw row offsets.
//
// Schedule: a (slab, dw) tile is ONE phase -- 12 fragment reads, 32 MFMAs (512 cycles) per wave, two barriers -- for the
// two wave groups in turn.  A phase cannot wait for DMA it issued itself, so the weights run three stages deep (tile t + 2
// is issued while tile t is read; the W stage of a tile is its dw) and the next slab is issued at dw = 1; every phase opens
// with vmcnt(0) (the pieces issued one phase ago have had a whole MFMA phase to land) and retires its fragment reads before
// its first barrier (which frees the stage for the DMA the other wave group issues right after it).
//
// DMA addressing: buffer loads (`buffer_load_dwordx4 ... offen lds`), descriptor = the slab's input row (wave-uniform, SGPRs),
// per-lane byte offset fixed for the whole kernel, channel block in the scalar offset: no per-piece vector arithmetic at all.
// Zero padding is the descriptor's range check: lanes on padding voxels carry an offset beyond the row, a row of padding
// has extent 0 -- out-of-range buffer loads write zeros to the LDS.
//
// Measured (13 x 1024^2, in-session A/B, profiles/history/r02l_conv_m512_ab.txt): 128 -> 128 with residual + GroupNorm partials
// 11.81 -> 9.61 ms (1021 -> 1255 TFLOP/s), 256 -> 128 19.8 -> 17.5 ms (1217 -> 1376); of which 512-voxel tiles +4 %, one
// phase per tile +9 %, buffer addressing +1.5 %, the residual prefetch in the epilogue +5 % on residual layers.
//
// S2: the spatially strided down-sampler (3x3x3, spatial stride 2, pad 0 with one zero row / column on the high side;
// downsamplers.py:24-94): output voxel w reads input voxels 2w + dw.  The slab is staged de-interleaved -- LDS rows 0 .. TM
// hold the even input voxels 2 (w0 + r), rows TM + 1 .. 2 TM the odd ones -- so the three dw taps are again three shifted
// reads of one staged row: dw = 0 the even rows, dw = 1 the odd rows, dw = 2 the even rows + 1.  (The tile-per-tap kernel
// ran this layer at 690 TFLOP/s.)
template <int BN, int TM, bool S2>   // 128 x 512 or 256 x 256: (channels, voxels) per workgroup
__global__ __launch_bounds__(512, 2) void conv3d_cl_row16_k32_kernel(ConvArgs p) {
    static_assert(BN * TM == 65536, "wave tile 128 voxels x 64 channels");
    constexpr int RB = 64, KC = 32;                                 // LDS row bytes, channels per stage
    constexpr int NROW = S2 ? 2 * TM + 1 : TM + 2, NPIECE = (NROW + 15) / 16, PPW = (NPIECE + 7) / 8, ISTEP = 16 * RB;   // 1 KiB DMA pieces = 16 rows
    constexpr int WN = BN / 64, WM = 8 / WN, MT = TM / WM / 16;     // wave tile 128 voxels x 64 channels, MT = 8
    constexpr int WP = BN / 128;                                    // W pieces (16 weight rows each) per wave
    constexpr int A_STAGE = (NPIECE + 1) * 1024, W_BYTES = BN * RB, W_BASE = 2 * A_STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;
    const int lr = lane & 15, lq = lane >> 4;

    int tm, tn;
    {
        const int rpx = (p.tiles_m + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int m_lo = xcd * rpx;
        int rows = p.tiles_m - m_lo;
        rows = rows < rpx ? rows : rpx;
        if (rows <= 0 || idx >= rows * p.tiles_n) return;
        tm = m_lo + idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int col0 = tn * BN;
    const int tiles_w = p.W_out / TM;
    const int w0 = (tm % tiles_w) * TM;
    const int orow = tm / tiles_w;                 // t_out * H_out + h_out
    const int h_out = orow % p.H_out, t_out = orow / p.H_out;

    // ---- this lane's slab rows: piece q = wave*PPW + i (q < NPIECE), LDS row r = 16q + lane/4  <->  input voxel w0 - 1 + r.
    // Byte offset of (voxel, source chunk) inside the input row; padding / unused rows point far beyond the row
    int a_voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int r = q * 16 + (lane >> 2), c = lane & 3;
        const int w = S2 ? (r <= TM ? 2 * (w0 + r) : 2 * (w0 + r - TM - 1) + 1) : w0 - 1 + r;
        a_voff[i] = (q < NPIECE && r < NROW && w >= 0 && w < p.W_in) ? (w * p.C_in + (c ^ ((r >> 1) & 3)) * 8) * 2 : 0x40000000;
    }
    const int wk = (p.tmerge ? 18 : 27) * p.C_in;
    int w_voff[WP];                  // 1 KiB pieces of 16 weight rows, WP per wave
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave * WP + i) * 16 + (lane >> 2), c = lane & 3;
        w_voff[i] = (r * wk + (c ^ ((r >> 1) & 3)) * 8) * 2;
    }
    const unsigned short* const w_tile = p.w + ((!S2 && p.tmerge) ? (int64_t)(t_out & 1) * p.C_out * wk : 0) + (int64_t)col0 * wk;
    const int w_bytes = BN * wk * 2, row_bytes = p.W_in * p.C_in * 2;
    char* const dma_a = smem + wave * PPW * 1024;
    char* const dma_w = smem + W_BASE + wave * (WP * 1024);

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: row * 64 + ((lq ^ ((row >> 1) & 3)) << 4); A row = wr*128 + i*16 + lr + dw, W row = wc*64 + j*16 + lr
    unsigned a_k[3];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int row = wr * (MT * 16) + lr + (S2 ? (dw == 1 ? TM + 1 : dw >> 1) : dw);
        a_k[dw] = row * RB + ((lq ^ ((row >> 1) & 3)) << 4);
    }
    const unsigned w_k = W_BASE + (wc * 64 + lr) * RB + ((lq ^ ((lr >> 1) & 3)) << 4);
    bf16x8 wf[4];

    const int cblocks = p.C_in / KC;
    int n_dtdh = 0, n_cb = 0;                      // the slab being staged: (dt, dh), channel block
    const unsigned short* slab_row = p.x;
    bool slab_ok = false;
    auto set_slab = [&](int dtdh) {
        const int dt = dtdh / 3, dh = dtdh - dt * 3;
        int ti = (S2 ? t_out * p.st : t_out) + dt - 2;
        ti = ti < 0 ? 0 : ti;                                  // causal replicate padding
        ti = p.vin ? (ti + 1) >> 1 : ti;                       // virtual temporal x2
        if (!S2 && p.tmerge) {                                 // two physical frames (p - 1, p), p = (t_out + 1) >> 1
            ti = ((t_out + 1) >> 1) - 1 + dt;
            ti = ti < 0 ? 0 : ti;
        }
        const int hu = S2 ? 2 * h_out + dh : h_out + dh - 1;   // S2: pad 0, one zero row below the last one
        slab_ok = hu >= 0 && hu < p.H_in;
        slab_row = p.x + ((int64_t)ti * p.H_in + (slab_ok ? hu : 0)) * p.W_in * p.C_in;
    };
    int a_dst = 0;
    auto stage_a = [&](int sa, int cb) {
        // one descriptor per slab row (wave-uniform): base = the row, extent = the row (0 for a row of padding)
        const int extent = slab_ok ? row_bytes : 0;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (wave * PPW + i < NPIECE)                       // wave-uniform
                bdma16(slab_row, extent, a_voff[i], cb * (KC * 2), dma_a + sa * A_STAGE + i * 1024);
    };
    auto stage_w = [&](int sw, int dtdh, int cb, int dw) {
        const int koff = (dtdh * 3 + dw) * p.C_in + cb * KC;
#pragma unroll
        for (int i = 0; i < WP; ++i) bdma16(w_tile, w_bytes, w_voff[i], koff * 2, dma_w + sw * W_BYTES + i * 1024);
    };
    auto next_slab = [&]() {
        if (++n_cb == cblocks) {
            n_cb = 0;
            ++n_dtdh;
            set_slab(n_dtdh);
        }
    };

    // tile (slab, DW) reads W stage DW and issues tile t + 2: dw = (DW + 2) % 3 of this slab (DW = 0) or of the next one
#define EA_C5_TILE(DW, HAS_NEXT2)                                                                           \
    {                                                                                                       \
        bf16x8 af[MT];                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            wf[j] = *reinterpret_cast<const bf16x8*>(smem + (w_k + (DW) * W_BYTES + j * ISTEP));            \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                      \
            af[i] = *reinterpret_cast<const bf16x8*>(smem + (a_k[DW] + i * ISTEP));                         \
        if ((DW) == 1 && (HAS_NEXT2)) {                                                                     \
            next_slab();                                                                                    \
            stage_a(a_dst, n_cb);                                                                           \
        }                                                                                                   \
        if (HAS_NEXT2) stage_w(((DW) + 2) % 3, n_dtdh, n_cb, ((DW) + 2) % 3);                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);      \
        __builtin_amdgcn_s_setprio(0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        __builtin_amdgcn_s_barrier();                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }

    // ---- prologue: slab 0 -> A stage 0, its dw = 0 / 1 weights -> W stages 0 / 1; wave group 1 starts one phase late
    set_slab(0);
    stage_a(0, 0);
    stage_w(0, 0, 0, 0);
    stage_w(1, 0, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // one trip = one slab = three tiles (dw = 0, 1, 2); the A stage toggles per slab
    const int nslabs = (p.tmerge ? 6 : 9) * cblocks;
    int a_step = A_STAGE;
    a_dst = 1;
    for (int sl = 0; sl < nslabs; ++sl) {
        EA_C5_TILE(0, true)
        EA_C5_TILE(1, sl + 1 < nslabs)
        EA_C5_TILE(2, sl + 1 < nslabs)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) a_k[dw] += a_step;
        a_step = -a_step;
        a_dst ^= 1;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef EA_C5_TILE

    // ---- epilogue: lane owns voxel m (one per M tile), channels n0 .. n0+3 of every N tile.  The residual of M tile i + 1 is
    // requested before tile i is rounded and stored (uniform branches around each load would serialise 32 round trips)
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_res = p.res != nullptr, has_gn = p.gn_partial != nullptr;
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n0 = col0 + wc * 64 + j * 16 + lq * 4;
        if (p.bias) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + n0);
        else b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t m_base = (int64_t)orow * p.W_out + w0 + wr * (MT * 16) + lr;
    const int64_t e_base = m_base * p.C_out + col0 + wc * 64 + lq * 4;   // + i * 16 * C_out + j * 16
    const int64_t e_step = (int64_t)16 * p.C_out;
    // virtual residual: logical frame t_out lives in physical frame (t_out + 1) >> 1 of p.res
    const unsigned short* const resp =
        p.res ? p.res - (p.vres ? (int64_t)(t_out - ((t_out + 1) >> 1)) * p.H_out * p.W_out * p.C_out : 0) : nullptr;
    bf16x4 rr[2][4];
    if (has_res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[0][j] = *reinterpret_cast<const bf16x4*>(resp + e_base + j * 16);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (has_res && i + 1 < MT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[(i + 1) & 1][j] = *reinterpret_cast<const bf16x4*>(resp + e_base + (i + 1) * e_step + j * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[j][e];
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rr[i & 1][j][e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(p.y + e_base + i * e_step + j * 16) = o;
            if (has_gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = (float)o[e];
                    gs[j] += r;
                    gq[j] += r * r;
                }
            }
        }
    }
    if (p.gn_partial) {
        const int64_t in_frame = ((int64_t)h_out * tiles_w + (tm % tiles_w)) * WM + wr;
        const int64_t blk = (int64_t)t_out * p.gn_nblk + in_frame;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s_ = gs[j], q_ = gq[j];
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) {
                s_ += __shfl_xor(s_, o_, 64);
                q_ += __shfl_xor(q_, o_, 64);
            }
            const int n0 = col0 + wc * 64 + j * 16 + lq * 4;
            if (lr == 0) {
                float* dst = p.gn_partial + (blk * (p.C_out >> 2) + (n0 >> 2)) * 2;
                dst[0] = s_;
                dst[1] = q_;
            }
        }
    }
}

// =================================================================================================================
// The same row-slab implicit GEMM over 32-channel stages on FOUR waves, one per SIMD, 128 voxels x 128 channels per wave, with the
// main loop placed by hand as one inline-asm block (round 5; tools/gen_conv_w4_asm.py -> ea_conv_w4_loop.inc, where the schedule
// is documented; the GEMM got +7..16 % from the same move, ea_gemm.hip gemm256_w4a_kernel).  Same products in the same order per
// accumulator as conv3d_cl_row16_k32_kernel (slabs in (dt, dh, channel block) order, dw inside): bit-identical results, and the
// same GroupNorm partial-sum layout (waves along M: 4 at 512 x 128, 2 at 256 x 256).
//   LDS: A stage = PPW * 4 pieces of 1 KiB (16 rows of 64 B; rows >= TM + 2 are zero-filled by out-of-range requests), stage 1 at
//   stage 0 ^ A_XOR; three W stages of BN x 64 B.  512 x 128: A0 [0, 36 K) | W [36 K, 60 K) | A1 [64 K, 100 K);  256 x 256: A0 [0, 20 K)
//   | A1 [32 K, 52 K) | W [64 K, 112 K).
#include "ea_conv_w4_loop.inc"
#include "ea_gemm_w4_loop.inc"     // EA_W4A_READ_HALF0 / 1: the accumulator read-out (acc(i, j) = a[4 * (8 j + i)] in both loops)

// epilogue of one 128-voxel x 64-channel accumulator half (conv3d_cl_row16_k32_kernel's, per half): bias, residual (prefetched one
// M tile ahead), one bf16 rounding, store, GroupNorm partial sums of the rounded values
template <int WM>
__device__ __forceinline__ void conv_w4a_epilogue(const ConvArgs& p, f32x4 (&acc)[8][4], const int col_w, const int wr, const int orow,
                                                  const int w0, const int h_out, const int t_out, const int tiles_w, const int tm,
                                                  const int lane) {
    const int lr = lane & 15, lq = lane >> 4;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_res = p.res != nullptr, has_gn = p.gn_partial != nullptr;
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n0 = col_w + j * 16 + lq * 4;
        if (p.bias) b4[j] = *reinterpret_cast<const f32x4*>(p.bias + n0);
        else b4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t m_base = (int64_t)orow * p.W_out + w0 + wr * 128 + lr;
    const int64_t e_base = m_base * p.C_out + col_w + lq * 4;   // + i * 16 * C_out + j * 16
    const int64_t e_step = (int64_t)16 * p.C_out;
    const unsigned short* const resp =
        p.res ? p.res - (p.vres ? (int64_t)(t_out - ((t_out + 1) >> 1)) * p.H_out * p.W_out * p.C_out : 0) : nullptr;
    bf16x4 rr[2][4];
    if (has_res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[0][j] = *reinterpret_cast<const bf16x4*>(resp + e_base + j * 16);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (has_res && i + 1 < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[(i + 1) & 1][j] = *reinterpret_cast<const bf16x4*>(resp + e_base + (i + 1) * e_step + j * 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[j][e];
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rr[i & 1][j][e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(p.y + e_base + i * e_step + j * 16) = o;
            if (has_gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = (float)o[e];
                    gs[j] += r;
                    gq[j] += r * r;
                }
            }
        }
    }
    if (p.gn_partial) {
        const int64_t in_frame = ((int64_t)h_out * tiles_w + (tm % tiles_w)) * WM + wr;
        const int64_t blk = (int64_t)t_out * p.gn_nblk + in_frame;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s_ = gs[j], q_ = gq[j];
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) {
                s_ += __shfl_xor(s_, o_, 64);
                q_ += __shfl_xor(q_, o_, 64);
            }
            const int n0 = col_w + j * 16 + lq * 4;
            if (lr == 0) {
                float* dst = p.gn_partial + (blk * (p.C_out >> 2) + (n0 >> 2)) * 2;
                dst[0] = s_;
                dst[1] = q_;
            }
        }
    }
}

// ---- the same epilogue through a wave-private 16 KiB LDS image (128 voxels x 64 channels, 16-byte chunks XOR-swizzled with
// (row >> 1) & 7: ea_gemm.hip's gemm_wave_epilogue), so that global memory sees whole 128-byte runs: the accumulator layout's own
// stores are 16 rows x 32 bytes per instruction, partial lines that the store path takes at about a third of the rate (measured on
// the GEMM: profiles/r05q_gemm_anatomy_direct_stores_dropped.jsonl).  The residual rows come in by LDS-DMA (requested for both
// halves before the first accumulator is read out), the stores are streaming (the activations are far larger than the L2 and not
// read again by this kernel; only with whole-line stores: on the 8- and 16-byte-per-lane stores of the other convolution kernels the
// same hint made the sub-pixel up-sampler 12 % and the eight-wave row-slab kernels 6 % slower, profiles/r05t_bench_c3_kernel_stats.csv
// against r05m).  Same arithmetic in the same order as conv_w4a_epilogue: bit-identical.
__device__ __forceinline__ void conv_w4a_residual_request(const ConvArgs& p, const unsigned short* const resp, char* const img, const int64_t m0,
                                                          const int col_w, const int lane) {
    const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = q * 8 + r8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(resp + (m0 + r) * p.C_out + col_w + ((c8 ^ ((r >> 1) & 7)) << 3)),
                                         (__attribute__((address_space(3))) void*)(img + q * 1024), 16, 0, 0);
    }
}

template <int WM>
__device__ __forceinline__ void conv_w4a_epilogue_img(const ConvArgs& p, f32x4 (&acc)[8][4], char* const img, const f32x4* const b4, const int col_w,
                                                      const int wr, const int64_t m0, const int h_out, const int t_out, const int tiles_w, const int tm,
                                                      const int lane) {
    const int lr = lane & 15, lq = lane >> 4;
    const int r8 = lane >> 3, c8 = lane & 7;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_res = p.res != nullptr, has_gn = p.gn_partial != nullptr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            char* const cell = img + r * 128 + (((j * 2 + (lq >> 1)) ^ ((r >> 1) & 7)) << 4) + (lq & 1) * 8;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b4[j][e];
            if (has_res) {
                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(cell);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(cell) = o;
            if (has_gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r_ = (float)o[e];
                    gs[j] += r_;
                    gq[j] += r_ * r_;
                }
            }
        }
    }
    u16x8 o[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = q * 8 + r8;
        o[q] = *reinterpret_cast<const u16x8*>(img + r * 128 + ((c8 ^ ((r >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        __builtin_nontemporal_store(o[q], reinterpret_cast<u16x8*>(p.y + (m0 + q * 8 + r8) * p.C_out + col_w + c8 * 8));
    if (p.gn_partial) {
        const int64_t in_frame = ((int64_t)h_out * tiles_w + (tm % tiles_w)) * WM + wr;
        const int64_t blk = (int64_t)t_out * p.gn_nblk + in_frame;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s_ = gs[j], q_ = gq[j];
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) {
                s_ += __shfl_xor(s_, o_, 64);
                q_ += __shfl_xor(q_, o_, 64);
            }
            const int n0 = col_w + j * 16 + lq * 4;
            if (lr == 0) {
                float* dst = p.gn_partial + (blk * (p.C_out >> 2) + (n0 >> 2)) * 2;
                dst[0] = s_;
                dst[1] = q_;
            }
        }
    }
}

// CB (round 6): x is CHANNEL-BLOCKED, [C_in / 32][T_in][H_in][W_in][32] (what ea_groupnorm_apply_bf16 writes on request): a slab's
// 16-voxel LDS-DMA piece is then 1 KiB of consecutive memory instead of sixteen 64-byte runs C_in * 2 bytes apart (the slab fill the
// main loop waits on: 55 -> 124 GB/s per CU in profiles/r04m_conv_slab_dma_pattern.jsonl).  Same values into the same LDS rows:
// bit-identical to the voxel-major input.
template <int BN, int TM, bool CB = false>   // 128 x 512 or 256 x 256: (channels, voxels) per workgroup
__global__ __launch_bounds__(256) void conv3d_cl_row16_w4a_kernel(ConvArgs p) {
    static_assert((BN == 128 && TM == 512) || (BN == 256 && TM == 256), "wave tile 128 voxels x 128 channels, four waves");
    constexpr int RB = 64, KC = 32;
    constexpr int NROW = TM + 2;
    constexpr int PPW = TM == 512 ? EA_CONV_W4A_PPW_M512 : EA_CONV_W4A_PPW_N256;
    constexpr int WP = TM == 512 ? EA_CONV_W4A_WP_M512 : EA_CONV_W4A_WP_N256;
    constexpr int A_XOR = TM == 512 ? EA_CONV_W4A_AXOR_M512 : EA_CONV_W4A_AXOR_N256;
    constexpr int W_BASE = TM == 512 ? 36 * 1024 : 64 * 1024;
    constexpr int WN = BN / 128, WM = 4 / WN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;

    int tm, tn;
    {
        const int rpx = (p.tiles_m + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int m_lo = xcd * rpx;
        int rows = p.tiles_m - m_lo;
        rows = rows < rpx ? rows : rpx;
        if (rows <= 0 || idx >= rows * p.tiles_n) return;
        tm = m_lo + idx / p.tiles_n;
        tn = idx % p.tiles_n;
    }
    const int col0 = tn * BN;
    const int tiles_w = p.W_out / TM;
    const int w0 = (tm % tiles_w) * TM;
    const int orow = tm / tiles_w;                 // t_out * H_out + h_out
    const int h_out = orow % p.H_out, t_out = orow / p.H_out;

    // this lane's slab rows: piece q = wave * PPW + i, LDS row r = 16 q + lane / 4  <->  input voxel w0 - 1 + r; padding and the rows
    // behind the slab point far beyond the input row (the request writes zeros)
    int a_voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int r = q * 16 + (lane >> 2), c = lane & 3;
        const int w = w0 - 1 + r;
        a_voff[i] = (r < NROW && w >= 0 && w < p.W_in) ? (w * (CB ? 32 : p.C_in) + (c ^ ((r >> 1) & 3)) * 8) * 2 : 0x40000000;
    }
    const int wk = (p.tmerge ? 18 : 27) * p.C_in;
    int w_voff[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave * WP + i) * 16 + (lane >> 2), c = lane & 3;
        w_voff[i] = (r * wk + (c ^ ((r >> 1) & 3)) * 8) * 2;
    }
    const unsigned short* const w_tile = p.w + (p.tmerge ? (int64_t)(t_out & 1) * p.C_out * wk : 0) + (int64_t)col0 * wk;
    const unsigned w_bytes = (unsigned)(BN * wk * 2);
    const int row_bytes = p.W_in * (CB ? 32 : p.C_in) * 2;              // (blocked: one row of ONE channel block)

    // the (dt, dh) -> input row table: lane l < ntab holds the row of dtdh = l (base address, extent; extent 0 = zero padding)
    const int ntab = p.tmerge ? 6 : 9;
    unsigned t_lo = 0, t_hi = 0, t_ext = 0;
    {
        const int dtdh = lane < ntab ? lane : 0;
        const int dt = dtdh / 3, dh = dtdh - dt * 3;
        int ti = t_out + dt - 2;
        ti = ti < 0 ? 0 : ti;                                  // causal replicate padding
        ti = p.vin ? (ti + 1) >> 1 : ti;                       // virtual temporal x2
        if (p.tmerge) {                                        // two physical frames (p - 1, p), p = (t_out + 1) >> 1
            ti = ((t_out + 1) >> 1) - 1 + dt;
            ti = ti < 0 ? 0 : ti;
        }
        const int hu = h_out + dh - 1;
        const bool ok = hu >= 0 && hu < p.H_in;
        const unsigned short* row = p.x + ((int64_t)ti * p.H_in + (ok ? hu : 0)) * p.W_in * (CB ? 32 : p.C_in);   // (blocked: channel block 0)
        t_lo = (unsigned)(uintptr_t)row;
        t_hi = (unsigned)((uintptr_t)row >> 32);
        t_ext = ok ? (unsigned)row_bytes : 0u;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // fragment addresses: row * 64 + ((lq ^ ((row >> 1) & 3)) << 4); A row = wr * 128 + i * 16 + lr + dw, W row = wc * 128 + j * 16 + lr
    unsigned ak[3];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int row = wr * 128 + lr + dw;
        ak[dw] = lds0 + row * RB + ((lq ^ ((row >> 1) & 3)) << 4);
    }
    const unsigned wkf = lds0 + W_BASE + (wc * 128 + lr) * RB + ((lq ^ ((lr >> 1) & 3)) << 4);
    const unsigned lds_a = __builtin_amdgcn_readfirstlane(lds0 + wave * (PPW * 1024));
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(lds0 + W_BASE + wave * (WP * 1024));
    const unsigned w_lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)w_tile);
    const unsigned w_hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)w_tile >> 32));
    const unsigned w_ext = __builtin_amdgcn_readfirstlane(w_bytes);
    const unsigned cblocks = __builtin_amdgcn_readfirstlane((unsigned)(p.C_in / KC));
    const unsigned nslabs = __builtin_amdgcn_readfirstlane((unsigned)(ntab * (p.C_in / KC)));
    const unsigned cin2 = __builtin_amdgcn_readfirstlane((unsigned)(p.C_in * 2));
    const uint64_t cb_bytes = (uint64_t)(p.vin ? (p.T_in + 1) >> 1 : p.T_in) * p.H_in * p.W_in * 64;      // one channel block of a blocked input (PHYSICAL frames)
    const unsigned cb_lo = __builtin_amdgcn_readfirstlane((unsigned)cb_bytes), cb_hi = __builtin_amdgcn_readfirstlane((unsigned)(cb_bytes >> 32));
#define EA_CW4_COMMON                                                                                                             \
    [t_lo] "v"(t_lo), [t_hi] "v"(t_hi), [t_ext] "v"(t_ext), [ak0] "v"(ak[0]), [ak1] "v"(ak[1]), [ak2] "v"(ak[2]), [wk] "v"(wkf),   \
        [w_lo] "s"(w_lo), [w_hi] "s"(w_hi), [w_ext] "s"(w_ext), [cblocks] "s"(cblocks), [nslabs] "s"(nslabs), [cin2] "s"(cin2),     \
        [lds_w] "s"(lds_w), [lds_a] "s"(lds_a)
    f32x4 accq[64];        // the 256 accumulators a0..a255, as the main asm's outputs: live until the read-outs consume them
#define EA_CW4_M512 [aoff0] "v"(a_voff[0]), [aoff1] "v"(a_voff[1]), [aoff2] "v"(a_voff[2]), [aoff3] "v"(a_voff[3]), [aoff4] "v"(a_voff[4]),      \
                    [aoff5] "v"(a_voff[5]), [aoff6] "v"(a_voff[6]), [aoff7] "v"(a_voff[7]), [aoff8] "v"(a_voff[8]), [woff0] "v"(w_voff[0]),       \
                    [woff1] "v"(w_voff[1])
#define EA_CW4_N256 [aoff0] "v"(a_voff[0]), [aoff1] "v"(a_voff[1]), [aoff2] "v"(a_voff[2]), [aoff3] "v"(a_voff[3]), [aoff4] "v"(a_voff[4]),      \
                    [woff0] "v"(w_voff[0]), [woff1] "v"(w_voff[1]), [woff2] "v"(w_voff[2]), [woff3] "v"(w_voff[3])
    if constexpr (TM == 512 && !CB) {
        asm volatile(EA_CONV_W4A_ASM_M512 : EA_W4A_ACC_OUTPUTS(accq) : EA_CW4_COMMON, EA_CW4_M512 : EA_CONV_W4A_CLOBBERS);
    } else if constexpr (TM == 512) {
        asm volatile(EA_CONV_W4A_ASM_M512_CB : EA_W4A_ACC_OUTPUTS(accq) : EA_CW4_COMMON, EA_CW4_M512, [cb_lo] "s"(cb_lo), [cb_hi] "s"(cb_hi)
                     : EA_CONV_W4A_CLOBBERS);
    } else if constexpr (!CB) {
        asm volatile(EA_CONV_W4A_ASM_N256 : EA_W4A_ACC_OUTPUTS(accq) : EA_CW4_COMMON, EA_CW4_N256 : EA_CONV_W4A_CLOBBERS);
    } else {
        asm volatile(EA_CONV_W4A_ASM_N256_CB : EA_W4A_ACC_OUTPUTS(accq) : EA_CW4_COMMON, EA_CW4_N256, [cb_lo] "s"(cb_lo), [cb_hi] "s"(cb_hi)
                     : EA_CONV_W4A_CLOBBERS);
    }
#undef EA_CW4_M512
#undef EA_CW4_N256
#undef EA_CW4_COMMON

    // ---- epilogue: the wave's 128 channels as two halves of 64 (accumulator column blocks 0..3, then 4..7)
    int lane_e;                                       // the lane id again: nothing lane-derived has to stay live across the main loop
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    f32x4 acc[8][4];
#ifdef EA_CONV_DIRECT_EPILOGUE     // (diagnostic builds: the accumulator layout's own 8-byte stores)
    EA_W4A_READ_HALF0(acc, accq)
    conv_w4a_epilogue<WM>(p, acc, col0 + wc * 128, wr, orow, w0, h_out, t_out, tiles_w, tm, lane_e);
    EA_W4A_READ_HALF1(acc, accq)
    conv_w4a_epilogue<WM>(p, acc, col0 + wc * 128 + 64, wr, orow, w0, h_out, t_out, tiles_w, tm, lane_e);
#else
    __builtin_amdgcn_s_barrier();      // every wave is past its last fragment read: the stages become the epilogue images (2 x 16 KiB per wave)
    char* const img0 = smem + wave * 32768;
    char* const img1 = img0 + 16384;
    const int col_w = col0 + wc * 128;
    const int64_t m0 = (int64_t)orow * p.W_out + w0 + wr * 128;
    if (p.res) {
        const unsigned short* const resp = p.res - (p.vres ? (int64_t)(t_out - ((t_out + 1) >> 1)) * p.H_out * p.W_out * p.C_out : 0);
        conv_w4a_residual_request(p, resp, img0, m0, col_w, lane_e);
        conv_w4a_residual_request(p, resp, img1, m0, col_w + 64, lane_e);
    }
    f32x4 b4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col_w + j * 16 + (lane_e >> 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    EA_W4A_READ_HALF0(acc, accq)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as the builtin (the compiler's bookkeeping sees it): bias vectors and residual rows, in flight under the read-out
    __builtin_amdgcn_sched_barrier(0);
    conv_w4a_epilogue_img<WM>(p, acc, img0, b4, col_w, wr, m0, h_out, t_out, tiles_w, tm, lane_e);
    EA_W4A_READ_HALF1(acc, accq)
    conv_w4a_epilogue_img<WM>(p, acc, img1, b4 + 4, col_w + 64, wr, m0, h_out, t_out, tiles_w, tm, lane_e);
#endif
}

// ea_set_option("conv_m512", bit 0: the 512-voxel x 128-channel kernel, bit 1: the 256 x 256 kernel over 32-channel stages).
// Bit 1 is off by default: for the 256-channel tiles the four-phase kernel over 64-channel stages is 1.5-3 % faster
// (profiles/history/r02p_conv_k32_256_ab.txt) -- their W tile is 16 pieces of half cache lines per 32-MFMA phase.
int g_conv_w4a = 3;    // ea_set_option("conv_w4a", bits): 1 = the 512 x 128 tiles, 2 = the 256 x 256 tiles on conv3d_cl_row16_w4a_kernel (default 3; 0 = the eight-wave kernels)
int g_conv_m512 = 1;
int g_conv_mfma = 16;  // ea_set_option("conv_mfma", 16 | 32): MFMA shape of the row-slab kernel
int g_conv_tile = 0;   // 0 = auto; 128: force the 128^2 kernel; 256 / 512: force the ping-pong kernels with 256- / 512-row tiles;
                       // 1024: force the row-slab kernel wherever it applies

// Explicit im2col for the few convolutions whose C_in is not a multiple of 64 (conv_in 3->128, decoder conv_in
// 16->512, 1x1x1 quant convs): cols[m, tap*C_in + c], zero-padded to k_pad; the product is then ea_gemm_bf16.
__global__ void im2col3d_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ cols, int T_in,
                                int H_in, int W_in, int C_in, int H_out, int W_out, int kt, int kh, int kw, int st,
                                int ss, int pad, int k_pad, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % k_pad);
    const int64_t m = idx / k_pad;
    unsigned short v = 0;
    if (k < kt * kh * kw * C_in) {
        const int c = k % C_in, tap = k / C_in;
        const int dw = tap % kw, dh = (tap / kw) % kh, dt = tap / (kw * kh);
        const int wo = (int)(m % W_out);
        const int64_t q = m / W_out;
        const int ho = (int)(q % H_out), to = (int)(q / H_out);
        int ti = to * st + dt - (kt - 1);
        ti = ti < 0 ? 0 : ti;
        const int hh = ho * ss + dh - pad, ww = wo * ss + dw - pad;
        if (hh >= 0 && hh < H_in && ww >= 0 && ww < W_in) v = x[(((int64_t)ti * H_in + hh) * W_in + ww) * C_in + c];
    }
    cols[idx] = v;
}

}  // namespace

int ea_conv_mfma_get() { return g_conv_mfma; }
int ea_conv_mfma_set(int v) {
    if (v != 16 && !(v == 32 && EA_BUILD_VARIANTS)) return -1;   // the 32x32x16 row-slab kernel: EA_BUILD_VARIANTS=1 libraries only
    g_conv_mfma = v;
    return 0;
}
int ea_conv_w4a_get() { return g_conv_w4a; }
int ea_conv_w4a_set(int v) {
    if (v < 0 || v > 3) return -1;
    g_conv_w4a = v;
    return 0;
}
int ea_conv_m512_get() { return g_conv_m512; }
int ea_conv_m512_set(int v) {
    if (v < 0 || v > 3) return -1;
    g_conv_m512 = v;
    return 0;
}
int ea_conv_tile_get() { return g_conv_tile; }
int ea_conv_tile_set(int v) {
    if (v != 0 && v != 128 && v != 256 && v != 512 && v != 1024) return -1;
    g_conv_tile = v;
    return 0;
}

static int conv_out_dim(int in, int k, int s, int pad_lo, int pad_hi) { return (in + pad_lo + pad_hi - k) / s + 1; }

static int conv3d_cl_impl(const ea_bf16* x, const ea_bf16* w, const float* bias, const ea_bf16* res, ea_bf16* y,
                          const ea_bf16* zeros, int T_in, int H_in, int W_in, int C_in, int C_out, int kt, int kh,
                          int kw, int st, int ss, int pad, int ups, int tdup, float* gn_partial, int64_t gn_capacity,
                          int* gn_nblk_out, void* stream) {
    if (gn_nblk_out) *gn_nblk_out = 0;
    EA_REQUIRE(x && w && y && zeros, "ea_conv3d_cl_bf16: null tensor");
    EA_REQUIRE(C_in == 8 || (C_in > 0 && C_in % BK == 0),
               "ea_conv3d_cl_bf16: C_in=%d must be 8 or a multiple of 64 (otherwise ea_im2col3d_bf16 + ea_gemm_bf16)", C_in);
    EA_REQUIRE(!(C_in == 8 && (kt != 3 || ups || tdup)), "ea_conv3d_cl_bf16: the 8-channel layers are plain 3x3x3 convolutions");
    EA_REQUIRE(C_out > 0 && C_out % 8 == 0, "ea_conv3d_cl_bf16: C_out must be a multiple of 8");
    EA_REQUIRE((kt == 3 && kh == 3 && kw == 3) || (kt == 1 && kh == 1 && kw == 1), "ea_conv3d_cl_bf16: kernel must be 3x3x3 or 1x1x1");
    EA_REQUIRE((st == 1 || st == 2) && (ss == 1 || ss == 2) && (pad == 0 || pad == 1), "ea_conv3d_cl_bf16: bad stride/pad");
    EA_REQUIRE(!(ups && ss != 1), "ea_conv3d_cl_bf16: upsample addressing needs spatial stride 1");
    EA_REQUIRE(!(kh == 1 && pad != 0), "ea_conv3d_cl_bf16: 1x1x1 kernels take pad 0");
    EA_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)res | (uintptr_t)bias | (uintptr_t)zeros) & 15) == 0,
               "ea_conv3d_cl_bf16: pointers must be 16-byte aligned");
    ConvArgs p;
    p.gn_partial = nullptr; p.gn_nblk = 0;
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.y = y; p.zeros = zeros;
    p.T_in = T_in; p.H_in = H_in; p.W_in = W_in; p.C_in = C_in; p.C_out = C_out;
    p.kt = kt; p.kh = kh; p.kw = kw; p.st = st; p.ss = ss; p.pad = pad; p.ups = ups;
    // tdup flags: 1 = store every output frame but the first twice; 2 = the input's frames are virtually duplicated (x holds
    // T_in physical frames, the convolution sees 2 T_in - 1); 4 = the residual's frames are virtually duplicated
    EA_REQUIRE((tdup & ~31) == 0, "ea_conv3d_cl_bf16: tdup is a bit set of 1 (duplicate store), 2 (virtual input), 4 (virtual residual), 8 (merged temporal taps), "
                                  "16 (channel-blocked input)");
    p.tdup = tdup & 1; p.vin = (tdup >> 1) & 1; p.vres = (tdup >> 2) & 1; p.tmerge = (tdup >> 3) & 1; p.x_cb = (tdup >> 4) & 1;
    tdup = p.tdup;   // from here on `tdup` is the duplicate-store flag alone (the kernel choice below tests it)
    EA_REQUIRE(!(p.vin && (kt != 3 || st != 1 || C_in == 8)), "ea_conv3d_cl_bf16: virtual input frames need a 3x3x3 temporal-stride-1 layer");
    EA_REQUIRE(!(p.vres && !res), "ea_conv3d_cl_bf16: virtual residual without a residual");
    EA_REQUIRE(!(p.tmerge && !p.vin), "ea_conv3d_cl_bf16: merged temporal taps are the taps of a virtually duplicated input (bit 2)");
    if (p.vin) T_in = T_in > 1 ? 2 * T_in - 1 : T_in;      // logical frames from here on
    p.T_in = T_in;
    const int He = ups ? 2 * H_in : H_in, We = ups ? 2 * W_in : W_in;
    // temporal: kt-1 replicated leading frames; spatial: `pad` zeros low, and for the strided (pad 0) convs one
    // zero row/column high (downsamplers.py:44-46 F.pad(x,(0,1,0,1)))
    p.T_out = conv_out_dim(T_in, kt, st, kt - 1, 0);
    const int pad_hi = (kh == 1) ? 0 : (pad ? pad : 1);
    p.H_out = conv_out_dim(He, kh, ss, pad, pad_hi);
    p.W_out = conv_out_dim(We, kw, ss, pad, pad_hi);
    p.M = (int64_t)p.T_out * p.H_out * p.W_out;
    EA_REQUIRE(p.M > 0 && p.M < (1ll << 40), "ea_conv3d_cl_bf16: bad output size");
    EA_REQUIRE(!p.x_cb || (kt == 3 && st == 1 && ss == 1 && pad == 1 && !ups && !tdup && ea_conv3d_cl_blocked_ok(p.T_out, p.H_out, p.W_out, C_in, C_out)),
               "ea_conv3d_cl_bf16: a channel-blocked input (tdup bit 16) is read by the four-wave row-slab kernels only: ask ea_conv3d_cl_blocked_ok first");
    if (C_in == 8) {   // one 16-byte chunk per voxel: eight taps per K tile (conv3d_cl_kernel<true>)
        EA_REQUIRE(p.M < (1ll << 31), "ea_conv3d_cl_bf16: too many output voxels for the 8-channel kernel");
        p.tiles_m = (int)((p.M + BM - 1) / BM);
        p.tiles_n = (C_out + BN - 1) / BN;
        const int64_t grid8 = (int64_t)8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
        EA_REQUIRE(grid8 < (1ll << 31), "ea_conv3d_cl_bf16: grid too large");
        static bool attr8_done = false;
        if (!attr8_done) {
            (void)hipFuncSetAttribute((const void*)conv3d_cl_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS);
            attr8_done = true;
        }
        // GroupNorm partial sums from the epilogue: one block per (128-voxel tile, wave row), tiles inside one frame
        const int64_t frame8 = (int64_t)p.H_out * p.W_out;
        if (gn_partial && frame8 % BM == 0 && C_out % 4 == 0) {
            const int64_t nblk = frame8 / BM * 2;
            const int64_t need = (int64_t)p.T_out * nblk * (C_out / 4) * 2;
            if (need <= gn_capacity && nblk < (1 << 30) && ((uintptr_t)gn_partial & 7) == 0) {
                p.gn_partial = gn_partial;
                p.gn_nblk = (int)nblk;
                if (gn_nblk_out) *gn_nblk_out = (int)nblk;
            }
        }
        ea_count("conv_c8_128x128");
        hipLaunchKernelGGL(conv3d_cl_kernel<true>, dim3((unsigned)grid8), dim3(256), CONV_LDS, (hipStream_t)stream, p);
        return ea_check_launch("ea_conv3d_cl_bf16");
    }
    // ping-pong kernels: C_out a multiple of 128 and enough tiles to fill the chip
    const int bn = C_out % 256 == 0 ? 256 : (C_out % 128 == 0 ? 128 : 0);
    const int64_t tiles256 = (p.M + 255) / 256;
    bool pp = bn != 0 && tiles256 * (C_out / (bn ? bn : 1)) >= 256;
    if (g_conv_tile == 128) pp = false;
    if (g_conv_tile >= 256 && bn != 0) pp = true;
    // strided down-sampler with C_out == 128 and output rows a multiple of 512 voxels (encoder level 1 at 1024^2): the
    // de-interleaved row-slab kernel
    if (g_conv_mfma == 16 && (g_conv_m512 & 1) && (g_conv_tile == 0 || g_conv_tile == 1024) && kt == 3 && ss == 2 && pad == 0 &&
        !ups && !tdup && C_out == 128 && p.W_out % 512 == 0 && C_in % 32 == 0) {
        p.tiles_m = (int)(p.M / 512);
        p.tiles_n = 1;
        const int64_t grid6 = (int64_t)8 * ((p.tiles_m + 7) / 8);
        EA_REQUIRE(grid6 < (1ll << 31), "ea_conv3d_cl_bf16: grid too large");
        if (gn_partial) {
            const int64_t nblk = (int64_t)p.H_out * (p.W_out / 512) * 4;
            const int64_t need = (int64_t)p.T_out * nblk * (C_out / 4) * 2;
            if (need <= gn_capacity && nblk < (1 << 30) && ((uintptr_t)gn_partial & 7) == 0) {
                p.gn_partial = gn_partial;
                p.gn_nblk = (int)nblk;
                if (gn_nblk_out) *gn_nblk_out = (int)nblk;
            }
        }
        const int lds6 = 2 * 66 * 1024 + 3 * 128 * 64;      // two A stages of 65 pieces + 1 KiB, three W stages
        static bool attr6_done = false;
        if (!attr6_done) {
            (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_k32_kernel<128, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds6);
            attr6_done = true;
        }
        ea_count("conv_row16_m512_s2");
        hipLaunchKernelGGL((conv3d_cl_row16_k32_kernel<128, 512, true>), dim3((unsigned)grid6), dim3(512), lds6, (hipStream_t)stream, p);
        return ea_check_launch("ea_conv3d_cl_bf16");
    }
    // row-slab kernel: 3x3x3, stride 1, pad 1 (with or without the folded x2 up-sampling), output rows a multiple of 256
    // voxels wide
    const bool row_ok = bn != 0 && kt == 3 && st == 1 && ss == 1 && pad == 1 && p.W_out % 256 == 0 && C_in % 64 == 0;
    const bool row_use = row_ok && (g_conv_tile == 1024 || (g_conv_tile == 0 && tiles256 * (C_out / bn) >= 512));
    EA_REQUIRE(!p.tmerge || (row_use && !ups && g_conv_mfma == 16),
               "ea_conv3d_cl_bf16: merged temporal taps are served by the 16x16x32 row-slab kernels only (ea_conv3d_cl_tmerge_ok)");
    if (p.tmerge) ea_count("conv_tmerge_18_taps");   // (counted beside the kernel's own name, which follows and stays ea_last_dispatch)
    if (row_use) {
        p.tiles_m = (int)(p.M / 256);
        p.tiles_n = C_out / bn;
        const int64_t grid3 = (int64_t)8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
        EA_REQUIRE(grid3 < (1ll << 31), "ea_conv3d_cl_bf16: grid too large");
        const int lds3 = 2 * 34 * 1024 + 2 * bn * 128;
        static bool attr3_done = false;
        if (!attr3_done) {
#if EA_BUILD_VARIANTS
            (void)hipFuncSetAttribute((const void*)conv3d_cl_row_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 128 * 128);
            (void)hipFuncSetAttribute((const void*)conv3d_cl_row_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 256 * 128);
            (void)hipFuncSetAttribute((const void*)conv3d_cl_row_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 128 * 128);
            (void)hipFuncSetAttribute((const void*)conv3d_cl_row_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 256 * 128);
#endif
            attr3_done = true;
        }
        const dim3 g3((unsigned)grid3), b3(512);
        // no folded up-sampling / temporal duplication: the one-phase-per-tile kernels over 32-channel stages -- 512 voxels x
        // 128 channels for C_out == 128 with rows a multiple of 512 voxels, 256 x 256 for the 256-channel tiles
        const bool k32_128 = C_out == 128 && p.W_out % 512 == 0 && (g_conv_m512 & 1);
        const bool k32_256 = bn == 256 && ((g_conv_m512 & 2) || (g_conv_w4a & 2));
        if (g_conv_mfma == 16 && !ups && !tdup && (k32_128 || k32_256)) {
            const int tmv = k32_128 ? 512 : 256, wm = k32_128 ? 4 : 2;
            const int tiles5 = (int)(p.M / tmv);
            p.tiles_m = tiles5;
            p.tiles_n = k32_128 ? 1 : C_out / 256;
            const int64_t grid5 = (int64_t)8 * ((tiles5 + 7) / 8) * p.tiles_n;
            EA_REQUIRE(grid5 < (1ll << 31), "ea_conv3d_cl_bf16: grid too large");
            if (gn_partial) {
                const int64_t nblk = (int64_t)p.H_out * (p.W_out / tmv) * wm;
                const int64_t need = (int64_t)p.T_out * nblk * (C_out / 4) * 2;
                if (need <= gn_capacity && nblk < (1 << 30) && ((uintptr_t)gn_partial & 7) == 0) {
                    p.gn_partial = gn_partial;
                    p.gn_nblk = (int)nblk;
                    if (gn_nblk_out) *gn_nblk_out = (int)nblk;
                }
            }
            // LDS: two A stages of (pieces + 1) KiB, three W stages of BN x 64 B
            const int lds5 = k32_128 ? 2 * 34 * 1024 + 3 * 128 * 64 : 2 * 18 * 1024 + 3 * 256 * 64;
            static bool attr5_done = false;
            if (!attr5_done) {
                (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_k32_kernel<128, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 3 * 128 * 64);
                (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_k32_kernel<256, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 18 * 1024 + 3 * 256 * 64);
                attr5_done = true;
            }
            // the four-wave kernels with the hand-placed main loop (ea_set_option("conv_w4a", bit 0: 512 x 128, bit 1: 256 x 256)
            if ((k32_128 && (g_conv_w4a & 1)) || (!k32_128 && (g_conv_w4a & 2))) {
                static bool attr6_done = false;
                if (!attr6_done) {
                    (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_w4a_kernel<128, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                    (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_w4a_kernel<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                    attr6_done = true;
                }
                if (p.x_cb) {
                    static bool attr7_done = false;
                    if (!attr7_done) {
                        (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_w4a_kernel<128, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                        (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_w4a_kernel<256, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                        attr7_done = true;
                    }
                    ea_count("conv_w4a");
                    ea_count("conv_blocked_input");
                    ea_count(k32_128 ? "conv_row16_m512" : "conv_row16_256_k32");
                    if (k32_128)
                        hipLaunchKernelGGL((conv3d_cl_row16_w4a_kernel<128, 512, true>), dim3((unsigned)grid5), dim3(256), 128 * 1024, (hipStream_t)stream, p);
                    else
                        hipLaunchKernelGGL((conv3d_cl_row16_w4a_kernel<256, 256, true>), dim3((unsigned)grid5), dim3(256), 128 * 1024, (hipStream_t)stream, p);
                    return ea_check_launch("ea_conv3d_cl_bf16");
                }
                if (k32_128) {
                    ea_count("conv_w4a");            // (marker first: ea_last_dispatch() names the kernel family, as before)
                    ea_count("conv_row16_m512");
                    hipLaunchKernelGGL((conv3d_cl_row16_w4a_kernel<128, 512>), dim3((unsigned)grid5), dim3(256), 128 * 1024, (hipStream_t)stream, p);
                } else {
                    ea_count("conv_w4a");
                    ea_count("conv_row16_256_k32");
                    hipLaunchKernelGGL((conv3d_cl_row16_w4a_kernel<256, 256>), dim3((unsigned)grid5), dim3(256), 128 * 1024, (hipStream_t)stream, p);
                }
                return ea_check_launch("ea_conv3d_cl_bf16");
            }
            EA_REQUIRE(!p.x_cb, "ea_conv3d_cl_bf16: a channel-blocked input is served by the four-wave row-slab kernels only (ea_conv3d_cl_blocked_ok)");
            if (k32_128) {
                ea_count("conv_row16_m512");
                hipLaunchKernelGGL((conv3d_cl_row16_k32_kernel<128, 512, false>), dim3((unsigned)grid5), b3, lds5, (hipStream_t)stream, p);
            } else {
                ea_count("conv_row16_256_k32");
                hipLaunchKernelGGL((conv3d_cl_row16_k32_kernel<256, 256, false>), dim3((unsigned)grid5), b3, lds5, (hipStream_t)stream, p);
            }
            return ea_check_launch("ea_conv3d_cl_bf16");
        }
        if (g_conv_mfma == 16 && gn_partial) {
            // fused GroupNorm statistics: one (sum, sumsq) pair per (frame, row tile, wave row, 4-channel bundle)
            const int wm = 8 / (bn / 64);
            const int64_t nblk = (int64_t)p.H_out * (p.W_out / 256) * wm;
            const int64_t frames_y = (tdup && p.T_out > 1) ? 2 * (int64_t)p.T_out - 1 : p.T_out;
            const int64_t need = frames_y * nblk * (C_out / 4) * 2;
            if (need <= gn_capacity && nblk < (1 << 30) && ((uintptr_t)gn_partial & 7) == 0) {
                p.gn_partial = gn_partial;
                p.gn_nblk = (int)nblk;
                if (gn_nblk_out) *gn_nblk_out = (int)nblk;
            }
        }
        if (g_conv_mfma == 16) {
            static bool attr4_done = false;
            if (!attr4_done) {
                (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 128 * 128);
                (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 256 * 128);
                (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 128 * 128);
                (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 34 * 1024 + 2 * 256 * 128);
                attr4_done = true;
            }
            ea_count(bn == 256 ? (ups ? "conv_row16_256_ups" : "conv_row16_256") : (ups ? "conv_row16_128_ups" : "conv_row16_128"));
            if (bn == 256 && ups)
                hipLaunchKernelGGL((conv3d_cl_row16_kernel<256, true>), g3, b3, lds3, (hipStream_t)stream, p);
            else if (bn == 256)
                hipLaunchKernelGGL((conv3d_cl_row16_kernel<256, false>), g3, b3, lds3, (hipStream_t)stream, p);
            else if (ups)
                hipLaunchKernelGGL((conv3d_cl_row16_kernel<128, true>), g3, b3, lds3, (hipStream_t)stream, p);
            else
                hipLaunchKernelGGL((conv3d_cl_row16_kernel<128, false>), g3, b3, lds3, (hipStream_t)stream, p);
            return ea_check_launch("ea_conv3d_cl_bf16");
        }
#if EA_BUILD_VARIANTS
        ea_count(bn == 256 ? (ups ? "conv_row32_256_ups" : "conv_row32_256") : (ups ? "conv_row32_128_ups" : "conv_row32_128"));
        if (bn == 256 && ups)
            hipLaunchKernelGGL((conv3d_cl_row_kernel<256, true>), g3, b3, lds3, (hipStream_t)stream, p);
        else if (bn == 256)
            hipLaunchKernelGGL((conv3d_cl_row_kernel<256, false>), g3, b3, lds3, (hipStream_t)stream, p);
        else if (ups)
            hipLaunchKernelGGL((conv3d_cl_row_kernel<128, true>), g3, b3, lds3, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((conv3d_cl_row_kernel<128, false>), g3, b3, lds3, (hipStream_t)stream, p);
        return ea_check_launch("ea_conv3d_cl_bf16");
#else
        ea_set_error("ea_conv3d_cl_bf16: the 32x32x16 row-slab kernel is built with EA_BUILD_VARIANTS=1 only");
        return EA_ERR_ARG;
#endif
    }
    if (pp) {
        // 512 x 128 when there are enough 512-voxel tiles (or when forced: g_conv_tile == 256 tests every variant by size)
        const bool big_m = bn == 128 && (g_conv_tile == 512 || (g_conv_tile == 0 && (p.M + 511) / 512 * (C_out / 128) >= 512));
        const int bm = big_m ? 512 : 256;
        p.tiles_m = (int)((p.M + bm - 1) / bm);
        p.tiles_n = C_out / bn;
        const int64_t grid2 = (int64_t)8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
        EA_REQUIRE(grid2 < (1ll << 31), "ea_conv3d_cl_bf16: grid too large");
        const int lds = 2 * bm * 128 + 2 * bn * 128;
        static bool attr2_done = false;
        if (!attr2_done) {
            (void)hipFuncSetAttribute((const void*)conv3d_cl_pp_kernel<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            (void)hipFuncSetAttribute((const void*)conv3d_cl_pp_kernel<256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
            (void)hipFuncSetAttribute((const void*)conv3d_cl_pp_kernel<512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            attr2_done = true;
        }
        hipStream_t st2 = (hipStream_t)stream;
        ea_count(bn == 256 ? "conv_pp_256x256" : (bm == 512 ? "conv_pp_512x128" : "conv_pp_256x128"));
        if (bn == 256)
            hipLaunchKernelGGL((conv3d_cl_pp_kernel<256, 256>), dim3((unsigned)grid2), dim3(512), lds, st2, p);
        else if (bm == 512)
            hipLaunchKernelGGL((conv3d_cl_pp_kernel<512, 128>), dim3((unsigned)grid2), dim3(512), lds, st2, p);
        else
            hipLaunchKernelGGL((conv3d_cl_pp_kernel<256, 128>), dim3((unsigned)grid2), dim3(512), lds, st2, p);
        return ea_check_launch("ea_conv3d_cl_bf16");
    }
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = (C_out + BN - 1) / BN;
    const int64_t grid = (int64_t)8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
    EA_REQUIRE(grid < (1ll << 31), "ea_conv3d_cl_bf16: grid too large");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3d_cl_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS);
        attr_done = true;
    }
    ea_count("conv_128x128");
    hipLaunchKernelGGL(conv3d_cl_kernel<false>, dim3((unsigned)grid), dim3(256), CONV_LDS, (hipStream_t)stream, p);
    return ea_check_launch("ea_conv3d_cl_bf16");
}

extern "C" int ea_conv3d_cl_bf16(const ea_bf16* x, const ea_bf16* w, const float* bias, const ea_bf16* res, ea_bf16* y,
                                 const ea_bf16* zeros, int T_in, int H_in, int W_in, int C_in, int C_out, int kt, int kh,
                                 int kw, int st, int ss, int pad, int ups, int tdup, void* stream) {
    return conv3d_cl_impl(x, w, bias, res, y, zeros, T_in, H_in, W_in, C_in, C_out, kt, kh, kw, st, ss, pad, ups, tdup, nullptr, 0,
                          nullptr, stream);
}

extern "C" int ea_conv3d_cl_stats_bf16(const ea_bf16* x, const ea_bf16* w, const float* bias, const ea_bf16* res, ea_bf16* y,
                                       const ea_bf16* zeros, int T_in, int H_in, int W_in, int C_in, int C_out, int kt,
                                       int kh, int kw, int st, int ss, int pad, int ups, int tdup, float* gn_partial,
                                       int64_t gn_capacity_floats, int* gn_nblk_out, void* stream) {
    return conv3d_cl_impl(x, w, bias, res, y, zeros, T_in, H_in, W_in, C_in, C_out, kt, kh, kw, st, ss, pad, ups, tdup, gn_partial,
                          gn_capacity_floats, gn_nblk_out, stream);
}

// Would ea_conv3d_cl_bf16 serve a 3x3x3 / stride 1 / pad 1 layer of this shape with a kernel that supports tdup bit 3 (merged
// temporal taps)?  Mirrors the kernel choice of conv3d_cl_impl: T_logical = the layer's logical (= output) frame count.
extern "C" int ea_conv3d_cl_tmerge_ok(int T_logical, int H, int W, int C_in, int C_out) {
    const int bn = C_out % 256 == 0 ? 256 : (C_out % 128 == 0 ? 128 : 0);
    if (bn == 0 || C_in % 64 != 0 || W % 256 != 0 || g_conv_mfma != 16 || T_logical < 2) return 0;
    const int64_t tiles256 = ((int64_t)T_logical * H * W + 255) / 256;
    return (g_conv_tile == 1024 || (g_conv_tile == 0 && tiles256 * (C_out / bn) >= 512)) ? 1 : 0;
}

// Would ea_conv3d_cl_bf16 serve a 3x3x3 / stride 1 / pad 1 layer of this shape (no folded up-sampling, no duplicate store) with the
// kernels that read a channel-blocked input (tdup bit 16)?  Mirrors the kernel choice of conv3d_cl_impl; T = the layer's output
// (= logical input) frame count.
extern "C" int ea_conv3d_cl_blocked_ok(int T, int H, int W, int C_in, int C_out) {
    const int bn = C_out % 256 == 0 ? 256 : (C_out % 128 == 0 ? 128 : 0);
    if (bn == 0 || C_in % 64 != 0 || W % 256 != 0 || g_conv_mfma != 16 || T < 1) return 0;
    const int64_t tiles256 = ((int64_t)T * H * W + 255) / 256;
    if (!(g_conv_tile == 1024 || (g_conv_tile == 0 && tiles256 * (C_out / bn) >= 512))) return 0;
    const bool k32_128 = C_out == 128 && W % 512 == 0 && (g_conv_m512 & 1);
    const bool k32_256 = bn == 256 && ((g_conv_m512 & 2) || (g_conv_w4a & 2));
    if (!(k32_128 || k32_256)) return 0;
    return ((k32_128 && (g_conv_w4a & 1)) || (!k32_128 && (g_conv_w4a & 2))) ? 1 : 0;
}

extern "C" int ea_conv3d_cl_subpixel_bf16(const ea_bf16* x, const ea_bf16* w4, const float* bias, ea_bf16* y, int T_in, int H_in,
                                          int W_in, int C_in, int C_out, int tdup, float* gn_partial, int64_t gn_capacity_floats,
                                          int* gn_nblk_out, void* stream) {
    if (gn_nblk_out) *gn_nblk_out = 0;
    EA_REQUIRE(x && w4 && y, "ea_conv3d_cl_subpixel_bf16: null tensor");
    EA_REQUIRE(T_in > 0 && H_in > 0 && W_in > 0 && W_in % 256 == 0 && C_in > 0 && C_in % BK == 0 && C_out > 0 && C_out % 256 == 0,
               "ea_conv3d_cl_subpixel_bf16: needs source rows a multiple of 256 voxels wide, C_in % 64 == 0, C_out % 256 == 0 "
               "(other shapes: ea_conv3d_cl_bf16 with ups = 1)");
    EA_REQUIRE((tdup & ~1) == 0, "ea_conv3d_cl_subpixel_bf16: tdup is 0 or 1 (duplicate store)");
    EA_REQUIRE((((uintptr_t)x | (uintptr_t)w4 | (uintptr_t)y | (uintptr_t)bias) & 15) == 0, "ea_conv3d_cl_subpixel_bf16: pointers must be 16-byte aligned");
    ConvArgs p;
    p.x = x; p.w = w4; p.bias = bias; p.res = nullptr; p.y = y; p.zeros = nullptr;
    p.T_in = T_in; p.H_in = H_in; p.W_in = W_in; p.C_in = C_in; p.C_out = C_out;
    p.T_out = T_in; p.H_out = 2 * H_in; p.W_out = 2 * W_in;
    p.kt = p.kh = p.kw = 3; p.st = p.ss = 1; p.pad = 1; p.ups = 0; p.tdup = tdup; p.vin = 0; p.vres = 0; p.tmerge = 0; p.x_cb = 0;
    p.M = (int64_t)p.T_out * H_in * W_in;          // source voxels: the M axis of ONE parity class
    EA_REQUIRE(p.M < (1ll << 31) && (int64_t)p.T_out * p.H_out * p.W_out < (1ll << 40), "ea_conv3d_cl_subpixel_bf16: clip too large");
    p.tiles_m = (int)(p.M / 256);
    p.tiles_n = C_out / 256;
    const int64_t grid = (int64_t)8 * ((p.tiles_m + 7) / 8) * p.tiles_n;
    EA_REQUIRE(grid < (1ll << 31), "ea_conv3d_cl_subpixel_bf16: grid too large");
    p.gn_partial = nullptr; p.gn_nblk = 0;
    if (gn_partial) {   // one (sum, sumsq) pair per (frame, source row tile, class, wave row, 4-channel bundle)
        const int64_t nblk = (int64_t)H_in * (W_in / 256) * 4 * 2;
        const int64_t frames_y = (tdup && p.T_out > 1) ? 2 * (int64_t)p.T_out - 1 : p.T_out;
        const int64_t need = frames_y * nblk * (C_out / 4) * 2;
        if (need <= gn_capacity_floats && nblk < (1 << 30) && ((uintptr_t)gn_partial & 7) == 0) {
            p.gn_partial = gn_partial;
            p.gn_nblk = (int)nblk;
            if (gn_nblk_out) *gn_nblk_out = (int)nblk;
        }
    }
    const int lds = 2 * 34 * 1024 + 2 * 256 * 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_kernel<256, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv3d_cl_row16_kernel<256, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    ea_count("conv_row16_256_subpixel");
    const dim3 g((unsigned)grid, 2), b(512);       // blockIdx.y = row parity a; the column parity b is the template instance
    hipLaunchKernelGGL((conv3d_cl_row16_kernel<256, false, 1>), g, b, lds, (hipStream_t)stream, p);
    hipLaunchKernelGGL((conv3d_cl_row16_kernel<256, false, 2>), g, b, lds, (hipStream_t)stream, p);
    return ea_check_launch("ea_conv3d_cl_subpixel_bf16");
}

extern "C" int ea_im2col3d_bf16(const ea_bf16* x, ea_bf16* cols, int T_in, int H_in, int W_in, int C_in, int kt, int kh,
                                int kw, int st, int ss, int pad, int k_pad, void* stream) {
    EA_REQUIRE(x && cols, "ea_im2col3d_bf16: null tensor");
    EA_REQUIRE(k_pad >= kt * kh * kw * C_in && k_pad % 8 == 0, "ea_im2col3d_bf16: k_pad too small");
    const int T_out = conv_out_dim(T_in, kt, st, kt - 1, 0);
    const int pad_hi = (kh == 1) ? 0 : (pad ? pad : 1);
    const int H_out = conv_out_dim(H_in, kh, ss, pad, pad_hi), W_out = conv_out_dim(W_in, kw, ss, pad, pad_hi);
    const int64_t total = (int64_t)T_out * H_out * W_out * k_pad;
    EA_REQUIRE(total > 0 && (total + 255) / 256 < (1ll << 31), "ea_im2col3d_bf16: bad size");
    hipLaunchKernelGGL(im2col3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, cols,
                       T_in, H_in, W_in, C_in, H_out, W_out, kt, kh, kw, st, ss, pad, k_pad, total);
    return ea_check_launch("ea_im2col3d_bf16");
}
