/*
 * ea_mi355x.h -- C ABI of libea_mi355x.so: the MI355X (gfx950) kernels behind EasyAnimate's
 * diffusion sampling hot path.
 *
 * The reference is 100% Python and has no FFI of its own; every entry point
 * below therefore replaces a *library call site* on the path, cited as
 * /root/reference/<file>:<line>.  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add at each site.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library never allocates,
 *     frees or synchronises.  All launches are asynchronous on `stream` (a hipStream_t passed as
 *     void*; NULL = the legacy default stream).
 *   - bf16 tensors are raw uint16 storage (`ea_bf16`); fp32 tensors are float.
 *   - return value: 0 = success; EA_ERR_ARG = invalid argument / unsupported shape; any other value
 *     is the hipError_t of the failed launch.  ea_last_error_string() describes the last non-zero
 *     return on the calling thread.  Nothing aborts or throws across this boundary.
 *   - the compute entry points keep no state between calls and own no memory.  Two pieces of PROCESS-WIDE mutable
 *     state exist, both host-side and neither affecting results: the tuning switches of ea_set_option() (they select
 *     WHICH kernel variant serves a call) and the dispatch counters read by ea_get_counter().  Neither is
 *     synchronised: call from one thread per process (one process per GPU), as the reference's single Python thread does.
 */
#ifndef EA_MI355X_H
#define EA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t ea_bf16;

#define EA_OK 0
#define EA_ERR_ARG (-1)

/* epilogues of ea_gemm_bf16 */
#define EA_EPI_BIAS 0           /* C = A.W^T + bias                                              */
#define EA_EPI_BIAS_GELU_TANH 1 /* C = gelu_tanh(A.W^T + bias)       (diffusers FeedForward.net.0) */
#define EA_EPI_BIAS_GATE_RES 2  /* C = res + gate[b,:] * (A.W^T + bias)  (attention.py:1140,1161)  */
#define EA_EPI_F32_OUT 3        /* C (fp32, ldc/c_batch_stride in floats) = A.W^T + bias: attention logits */

const char* ea_last_error_string(void);
int ea_version(void);
/* Tuning / benchmarking switches (process-wide; results never depend on them):
 *   "gemm_tile":    0 = automatic (default), 128 / 256 = force that block tile in ea_gemm_bf16;
 *   "gemm_mfma":    MFMA shape of the 256^2 GEMM kernel: 16 = v_mfma_f32_16x16x32_bf16 (default), 32 = 32x32x16;
 *   "conv_tile":    0 = automatic, 128 = the 128^2 kernel, 256 / 512 = the ping-pong kernels with 256- / 512-voxel tiles,
 *                   1024 = the row-slab kernel wherever it applies (3x3x3, stride 1, rows a multiple of 256 voxels wide);
 *   "conv_mfma":    MFMA shape of the row-slab convolution kernel: 16 = 16x16x32 (default), 32 = 32x32x16;
 *   "conv_m512":    row-slab layers without folded up-sampling, bit 0 (1): C_out == 128 with rows a multiple of 512 voxels use the
 *                   512-voxel x 128-channel kernel (stride 1, and the spatially strided down-sampler in its de-interleaved form), bit 1 (2): the 256-channel tiles use the 256 x 256 kernel -- both one phase
 *                   per tile over 32-channel stages; 1 = default (bit 1 measured 1.5-3 % slower), 0 = the four-phase kernels over 64-channel stages;
 *   "attn_variant": 3 = the pipelined kernel on 16x16x32 MFMAs (the only generation in the default library: the softmax
 *                   scale must be folded into Q, see ea_attention_fwd_bf16); EA_BUILD_VARIANTS=1 libraries also carry 2 = the
 *                   pipelined kernel on 32x32x16 (takes any scale) and 1 = the first, un-pipelined kernel. */
int ea_set_option(const char* name, int value);
/* Current value of a switch (tests restore what they found). */
int ea_get_option(const char* name, int* value);
/* Dispatch counters (host-side bookkeeping): how many launches each kernel variant has served since the last reset,
 * e.g. "conv_row16_128", "conv_row16_256_ups", "conv_pp_256x256", "conv_128x128", "gemm_256_mi16", "gemm_128",
 * "attention_v3".  Tests use them to assert that the kernels a parity case is meant to cover actually ran.
 * ea_get_counter: count (0 for a name never hit, -1 for NULL); ea_counter_name: name of the index-th counter seen so
 * far (EA_ERR_ARG past the end). */
long long ea_get_counter(const char* name);
int ea_counter_name(int index, char* buf, int buf_len);
void ea_reset_counters(void);
/* name of the kernel variant that served the most recent counted launch ("" before the first) */
const char* ea_last_dispatch(void);

/* ---- normalisation ------------------------------------------------------------------------- */

/* FP32 LayerNorm + affine + adaLN modulation, one pass:
 *   y[b,r,:] = (LN_fp32(x[b,r,:]; gamma, beta, eps)) * (1 + scale[b,:]) + shift[b,:]
 * Replaces easyanimate/models/norm.py:164-165 (EasyAnimateLayerNormZero.forward, FP32LayerNorm
 * norm.py:16-26) and, with the same operand order, diffusers AdaLayerNorm as used at
 * easyanimate/models/transformer3d.py:1678.  gamma/beta may be NULL (no affine); scale/shift may be
 * NULL (plain LayerNorm, transformer3d.py:1674).  scale/shift are fp32 rows of a [batch, mod_stride]
 * table (the chunked output of the adaLN Linear), so no chunk copies are needed.
 * x/y: bf16 [batch, rows, dim], row-contiguous, batch strides in elements.  dim % 8 == 0, dim <= 8192. */
int ea_layernorm_modulate_bf16(const ea_bf16* x, ea_bf16* y, const float* gamma, const float* beta,
                               const float* scale, const float* shift, int64_t mod_stride,
                               int batch, int rows, int dim, int64_t x_batch_stride,
                               int64_t y_batch_stride, float eps, void* stream);

/* EasyAnimateRMSNorm (easyanimate/models/norm.py:28-42): y = w * bf16(x * rsqrt(mean(x^2)+eps)).
 * x,y bf16 [rows, dim]; w fp32 [dim]. */
int ea_rmsnorm_bf16(const ea_bf16* x, ea_bf16* y, const float* w, int rows, int dim, float eps,
                    void* stream);

/* ---- small-M linear (adaLN tables, timestep MLP) -------------------------------------------- */

/* y[m,n] = act_out( sum_k act_in(x[m,k]) * W[n,k] + bias[n] ),  m < 16.
 * act_in: 0 none, 1 SiLU (norm.py:163 `linear(silu(temb))`).  act_out: 0 none, 1 SiLU
 * (diffusers TimestepEmbedding).  x fp32 [m,k]; W bf16 [n,k]; bias fp32 [n] or NULL; y fp32 [m,n]. */
int ea_linear_small_m(const float* x, const ea_bf16* W, const float* bias, float* y, int m, int n,
                      int k, int act_in, int act_out, void* stream);

/* diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0) (transformer3d.py:1399,1519):
 * out[b, 0:half] = cos(t_b * f_i), out[b, half:] = sin(t_b * f_i), f_i = exp(-ln(1e4) * i / half).
 * The result is rounded to bf16 and widened again (the reference casts to the latent dtype). */
int ea_timestep_sinusoid(const float* t, float* out, int batch, int dim, int round_bf16, void* stream);

/* ---- GEMM ---------------------------------------------------------------------------------- */

/* Batched-over-rows GEMM with shared weight and fused epilogue (MFMA bf16, fp32 accumulate):
 *   for b in [0,batch): C_b[M,N] = epi( A_b[M,K] . W[N,K]^T + bias[N] )
 * A_b = A + b*a_batch_stride (row stride lda), C_b likewise (ldc), res_b likewise (ldres; may alias C),
 * gate row = gate + b*gate_batch_stride (fp32 [N]).  W is nn.Linear.weight layout [N,K] row-major.
 * Replaces nn.Linear at processor.py:244-246,261-263,303-311, diffusers FeedForward (attention.py:
 * 1156-1160), transformer3d.py:1531 (patch embed as GEMM), :1533, :1680.
 * Requirements: K % 64 == 0, lda/ldc/ldres % 8 == 0, N % 8 == 0, all base pointers 16-byte aligned. */
int ea_gemm_bf16(const ea_bf16* A, const ea_bf16* W, const float* bias, ea_bf16* C,
                 const ea_bf16* res, const float* gate, int batch, int M, int N, int K, int64_t lda,
                 int64_t a_batch_stride, int64_t ldc, int64_t c_batch_stride, int64_t ldres,
                 int64_t res_batch_stride, int64_t gate_batch_stride, int epilogue, void* stream);

/* ea_gemm_bf16 on K-BLOCKED operands (round 4; the feed-forward pair attention.py:1156-1162 / diffusers FeedForward): an operand
 * laid out [K / 64][rows][64] makes a 64-deep K tile of 256 rows ONE contiguous 32 KiB block for the kernel's LDS-DMA instead of
 * 256 pieces a row stride apart -- with K = 12 288 (24 KiB between rows) the strided form fills the LDS at 53 GB/s per CU against
 * 85 GB/s (tools/ubench/lds_dma_rate.hip), and the second FFN GEMM is bound by exactly that.  layout bit 0: A is
 * [batch][K / 64][M][64]; bit 1: W is [K / 64][N][64]; bit 2: C is written as [batch][N / 64][M][64] (the A operand of the next
 * call: the first FFN GEMM writes what the second one reads).  Operands without their bit are row-major with contiguous rows
 * (lda = K, ldc = N); res / gate / bias / epilogue as ea_gemm_bf16.  Same products in the same order: results identical to
 * ea_gemm_bf16 on the row-major form of the same values.  Requires bf16 weights, N % 256 == 0 (served by the 256 x 256 kernel). */
int ea_gemm_bf16_kblocked(const ea_bf16* A, const ea_bf16* W, const float* bias, ea_bf16* C, const ea_bf16* res,
                          const float* gate, int batch, int M, int N, int K, int64_t a_batch_stride, int64_t c_batch_stride,
                          int64_t ldres, int64_t res_batch_stride, int64_t gate_batch_stride, int epilogue, int layout,
                          void* stream);

/* The same GEMM with the weight stored as fp8: W_fp8 = torch.float8_e4m3fn (OCP E4M3) bytes, [N,K] row-major, the storage
 * mode of the reference's `model_cpu_offload_and_qfloat8` (utils/fp8_optimization.py:17-35 keeps every Linear weight in fp8
 * and up-casts it to bf16 for each call).  The kernel reads the fp8 bytes, widens them to bf16 on the way into the LDS
 * (exact) and runs the same MFMAs: results are bit-identical to ea_gemm_bf16 on the up-cast weight; the weight costs one
 * byte per element in HBM and no bf16 copy exists.  EA_EPI_F32_OUT is not offered. */
int ea_gemm_bf16_w8(const ea_bf16* A, const uint8_t* W_fp8, const float* bias, ea_bf16* C,
                    const ea_bf16* res, const float* gate, int batch, int M, int N, int K, int64_t lda,
                    int64_t a_batch_stride, int64_t ldc, int64_t c_batch_stride, int64_t ldres,
                    int64_t res_batch_stride, int64_t gate_batch_stride, int epilogue, void* stream);

/* ---- attention ----------------------------------------------------------------------------- */

/* qk-LayerNorm(head_dim=64) + interleaved RoPE + head-major scatter, one pass over a QKV buffer.
 * Replaces processor.py:251-258,268-285 (view/transpose, norm_q/norm_k, torch.cat, apply_rotary_emb).
 *   qkv   : bf16 [batch, n_tok, 3*heads*64]  (q | k | v along the last axis), batch stride given
 *   q_out : bf16 [batch, heads, s_pad, 64]   rows [seq_off, seq_off+n_tok) are written
 *   k_out : same
 *   vt_out: bf16 [batch, heads, 64, s_pad]   (V transposed: the PV operand layout of ea_attention_fwd)
 *   nq_w,nq_b,nk_w,nk_b: fp32 [64] LayerNorm affine of norm_q / norm_k (eps = ln_eps)
 *   cos,sin: fp32 [n_tok, 64] or NULL (text tokens: no RoPE), row r applies to token r
 *   q_scale: multiplies q (not k) in fp32 ahead of its bf16 rounding; 1.0 reproduces the reference's q.  Passing
 *            scale*log2(e) here and scale = ln(2) to ea_attention_fwd_* folds the softmax scale into Q: the
 *            attention kernel then exponentiates the raw MFMA scores (one VALU instruction per score less).
 *   kv_off, kv_rows: geometry of k_out [batch, heads, kv_rows, 64] / vt_out [batch, heads, 64, kv_rows], rows / columns
 *            [kv_off, kv_off + n_tok) are written; kv_rows <= 0 means "the same as q_out" (seq_off, s_pad).  Sequence
 *            parallelism lets K / V^T land directly in the rank's slot of the exchange buffer (no pack copy).
 * s_pad % 64 == 0. */
int ea_qknorm_rope_bf16(const ea_bf16* qkv, int64_t qkv_batch_stride, ea_bf16* q_out, ea_bf16* k_out,
                        ea_bf16* vt_out, const float* nq_w, const float* nq_b, const float* nk_w,
                        const float* nk_b, const float* cos, const float* sin, int batch, int heads,
                        int n_tok, int seq_off, int s_pad, int kv_off, int kv_rows, float ln_eps, float q_scale,
                        void* stream);

/* The three QKV projections of one token stream and everything up to the attention operands in ONE launch:
 *   q, k, v = A.Wq^T + bq, A.Wk^T + bk, A.Wv^T + bv   (processor.py:244-246 / 261-263)
 * followed, in the GEMM epilogue, by exactly what ea_qknorm_rope_bf16 does to the stored QKV values (qk-LayerNorm(64),
 * RoPE, q_scale, head-major scatter of q / k, transposed scatter of v; processor.py:251-285).  The [batch, M, 3*heads*64]
 * QKV buffer never exists and A is read once.  Same roundings at the same points as ea_gemm_bf16 x 3 +
 * ea_qknorm_rope_bf16: V^T is bit-identical, q / k agree up to the fp32 summation order of the LayerNorm statistics (<= 1 bf16 ulp).
 *   A: bf16 [batch, M, K] (row stride lda);  Wq/Wk/Wv: bf16 [heads*64, K] (three separate nn.Linear weights);
 *   bq/bk/bv: fp32 [heads*64] or NULL;  q_out/k_out: bf16 [batch, heads, s_pad, 64];  vt_out: bf16 [batch, heads, 64, s_pad];
 *   rows / columns [seq_off, seq_off + M) are written;  cos/sin: fp32 [M, 64] or NULL.
 *   kv_off, kv_rows: as in ea_qknorm_rope_bf16 (k_out / vt_out may have their own rows-per-head and first row).
 *   parts: which thirds of the q | k | v output axis this launch computes: 7 (or 0) = all, 6 = k | v, 1 = q.  The
 *          sequence-parallel step projects K | V first, starts the K / V^T exchange, and projects Q under it.
 * Requirements: (heads*64) % 256 == 0, K % 64 == 0, seq_off % 8 == 0, kv_off % 8 == 0; other shapes (test-size models)
 * use ea_gemm_bf16 + ea_qknorm_rope_bf16.  M is arbitrary: a ragged last 256-row tile re-reads row M - 1 and stores rows
 * < M only (exactly rows / columns [off, off + M) are written). */
int ea_qkv_gemm_norm_rope_bf16(const ea_bf16* A, const ea_bf16* Wq, const ea_bf16* Wk, const ea_bf16* Wv,
                               const float* bq, const float* bk, const float* bv, ea_bf16* q_out, ea_bf16* k_out,
                               ea_bf16* vt_out, const float* nq_w, const float* nq_b, const float* nk_w,
                               const float* nk_b, const float* cos, const float* sin, int batch, int M, int heads,
                               int K, int64_t lda, int64_t a_batch_stride, int seq_off, int s_pad, int kv_off,
                               int kv_rows, int parts, float ln_eps, float q_scale, void* stream);

/* The same launch with K / V^T written in HEAD GROUPS (round 5: the sequence-parallel exchange pipelined by head groups,
 * easyanimate_amd/sequence_parallel.py): head h of batch element b lands in group g = h / kv_group_heads at
 * k_out + g * kv_group_stride + ((b * kv_group_heads + h % kv_group_heads) * kv_rows + kv_off + row) * 64 (vt_out likewise with
 * [64, kv_rows] per head), i.e. every group is its own [batch, kv_group_heads, kv_rows, 64] buffer kv_group_stride elements
 * after the previous one -- one contiguous all-gather operand per group.  q_out is unchanged ([batch, heads, s_pad, 64]).
 * kv_group_heads must divide heads; kv_group_heads = heads, kv_group_stride = 0 is ea_qkv_gemm_norm_rope_bf16. */
int ea_qkv_gemm_norm_rope_grouped_bf16(const ea_bf16* A, const ea_bf16* Wq, const ea_bf16* Wk, const ea_bf16* Wv,
                                       const float* bq, const float* bk, const float* bv, ea_bf16* q_out,
                                       ea_bf16* k_out, ea_bf16* vt_out, const float* nq_w, const float* nq_b,
                                       const float* nk_w, const float* nk_b, const float* cos, const float* sin,
                                       int batch, int M, int heads, int K, int64_t lda, int64_t a_batch_stride,
                                       int seq_off, int s_pad, int kv_off, int kv_rows, int parts, int kv_group_heads,
                                       int64_t kv_group_stride, float ln_eps, float q_scale, void* stream);
/* ... with the three weights stored as fp8 (see ea_gemm_bf16_w8): bit-identical to ea_qkv_gemm_norm_rope_bf16 on the up-cast weights. */
int ea_qkv_gemm_norm_rope_bf16_w8(const ea_bf16* A, const uint8_t* Wq_fp8, const uint8_t* Wk_fp8, const uint8_t* Wv_fp8,
                                  const float* bq, const float* bk, const float* bv, ea_bf16* q_out, ea_bf16* k_out,
                                  ea_bf16* vt_out, const float* nq_w, const float* nq_b, const float* nk_w,
                                  const float* nk_b, const float* cos, const float* sin, int batch, int M, int heads,
                                  int K, int64_t lda, int64_t a_batch_stride, int seq_off, int s_pad, int kv_off,
                                  int kv_rows, int parts, float ln_eps, float q_scale, void* stream);

/* Non-causal, unmasked softmax(Q K^T * scale) V, head_dim 64, bf16 in/out, fp32 softmax state.
 * Replaces F.scaled_dot_product_attention at processor.py:287-289 plus the transpose/reshape at :291.
 *   q,k : bf16 [batch, heads, s_pad, 64];  vt: bf16 [batch, heads, 64, s_pad]; rows/cols >= seq are
 *         never read as valid keys (they are masked) but must be finite-or-zero in vt.
 *   out : bf16 [batch, seq, heads*64] (out_batch_stride elements between batches)
 * Only query rows [q_begin, q_end) are computed (sequence parallelism: a rank owns a query range but
 * sees all keys).  s_pad % 256 == 0 and s_pad >= seq.
 * scale: the softmax scale must live in Q -- project with q_scale = scale * log2(e) (ea_qkv_gemm_norm_rope_bf16 /
 * ea_qknorm_rope_bf16) and pass scale = ln 2 here, so that scale * log2(e) == 1 and the kernel exponentiates the raw MFMA
 * scores; any other value is EA_ERR_ARG (same rule for ea_attention_fwd_range_bf16; EA_BUILD_VARIANTS=1 libraries carry the
 * 32x32x16 generation that multiplies every score and accept any scale). */
int ea_attention_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                          int64_t out_batch_stride, int batch, int heads, int seq, int s_pad,
                          int q_begin, int q_end, float scale, void* stream);

/* The resumable attention over keys that live in SEGMENTS: the ranks' K / V^T slots exactly where the (in-place)
 * all_gather_into_tensor left them (sequence parallelism; no pack / unpack copies).  Segment g (g = 0 .. n_seg-1, rank
 * order) holds K as [batch*heads, seg_rows, 64] at k_seg0 + g*seg_stride and V^T as [batch*heads, 64, seg_rows] at
 * vt_seg0 + g*seg_stride (elements); segment skip_seg (pass -1 to use every segment) is left out.  Of every used segment
 * the rows [seg_first_row, seg_first_row + seg_used_rows) are keys (a slot is [text rows | shard rows]: the remote pass
 * skips the replicated text rows, the own-slot pass -- n_seg = 1, k_seg0 / vt_seg0 at the own slot -- takes both).
 * kv_valid = number of valid keys over the used segments in order (all full except possibly the last).  q / out / state /
 * flags / q_begin / q_end as ea_attention_fwd_range_bf16 (q: [batch, heads, q_pad, 64]); the softmax scale must be folded
 * into Q (scale = ln 2).  seg_rows, seg_first_row, seg_used_rows % 64 == 0. */
int ea_attention_fwd_segments_bf16(const ea_bf16* q, const ea_bf16* k_seg0, const ea_bf16* vt_seg0, ea_bf16* out,
                                   int64_t out_batch_stride, int batch, int heads, int q_pad, int q_begin, int q_end,
                                   int seg_rows, int n_seg, int skip_seg, int64_t seg_stride, int seg_first_row,
                                   int seg_used_rows, int kv_valid, float scale, float* state, int flags, void* stream);

/* ea_attention_fwd_segments_bf16 over a HEAD WINDOW: the launch covers heads [q_head0, q_head0 + heads) of q / out / state, which
 * are laid out over q_heads heads per batch element (q: [batch, q_heads, q_pad, 64], out rows of q_heads * 64 columns, the state
 * of ea_attention_state_bytes(batch, q_heads, ..)), while the segments hold K / V^T of those `heads` heads only
 * ([batch, heads, seg_rows, 64] per segment, kv_batch_stride elements between batch elements; 0 = heads * seg_rows * 64).  One
 * call per head group of the pipelined sequence-parallel exchange; q_heads = 0 (then q_head0 = 0) is the plain call. */
int ea_attention_fwd_segments_heads_bf16(const ea_bf16* q, const ea_bf16* k_seg0, const ea_bf16* vt_seg0, ea_bf16* out,
                                         int64_t out_batch_stride, int batch, int heads, int q_pad, int q_begin, int q_end,
                                         int seg_rows, int n_seg, int skip_seg, int64_t seg_stride, int seg_first_row,
                                         int seg_used_rows, int kv_valid, float scale, float* state, int flags,
                                         int q_head0, int q_heads, int64_t kv_batch_stride, void* stream);

/* Single-head, head_dim 512 flash attention of the VAE mid block (vaemodules/attention.py:391-423 SpatialAttention with
 * attention_processors.py:76-139: per latent frame softmax(Q K^T * scale) V over all H x W tokens of the frame, one head of
 * 512 channels).  One launch for all frames; no logits buffer.
 *   q  : bf16 [frames, n_q, 512]      (frame stride q_frame_stride elements; rows are queries)
 *   k  : bf16 [frames, n_kpad, 512]   (rows >= n_keys are never used as keys: masked)
 *   vt : bf16 [frames, 512, n_kpad]   (V transposed; columns >= n_keys must be finite)
 *   out: bf16 [frames, n_q, 512]
 * n_kpad % 32 == 0, n_keys <= n_kpad.  n_q may differ from n_keys (a spatially split VAE attends its rows' queries over
 * the whole frame's keys). */
int ea_attention_d512_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out, int frames, int n_q,
                               int n_keys, int n_kpad, int64_t q_frame_stride, int64_t k_frame_stride,
                               int64_t vt_frame_stride, int64_t out_frame_stride, float scale, void* stream);

/* Sliding-window (band) attention of EasyAnimateSWAttnProcessor2_0 (processor.py:420: flash_attn_func(q, k, v,
 * window_size=(w, w)) on the six re-ordered head groups): query row i attends key rows j with |i - j| <= window, rows
 * [0, seq) of q / k / vt (same layouts as ea_attention_fwd_bf16), softmax(QK^T * scale) V.  Only key tiles intersecting
 * the band are visited. */
int ea_attention_window_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                 int64_t out_batch_stride, int batch, int heads, int seq, int s_pad, int window,
                                 float scale, void* stream);

/* The window pass of the SWA processor WITHOUT index copies (processor.py:400-435: six head groups, each visiting the video tokens
 * in its own scan order; results brought back to (f h w) order and added to the cross pass).  map: int32 [heads, seq], scan
 * position -> token of the (f h w) grid.  q / k stay where the projection wrote them -- [batch, heads, s_pad, 64], video token n
 * at row row_off + n -- and are addressed through the map (Q fragments by row, the K tiles' LDS-DMA rows through a per-lane row
 * index loaded one tile ahead).  vt_perm: V^T already in scan order, [batch, heads, 64, vt_pad] (ea_permute_cols_bf16: a key is a
 * column of V^T and cannot be gathered by row DMA; columns >= seq must be finite).  The result row of scan position p is stored
 * at token order with the cross pass added: out[b, r, h*64..] = bf16(bf16(window result) + cross[b, r, h*64..]), r = row_off +
 * map[h][p]; cross has out's strides.  Rows < row_off of out are not written.  Same arithmetic as ea_attention_window_fwd_bf16
 * on gathered operands followed by a bf16 add: bit-identical (tested). */
int ea_attention_window_mapped_fwd_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt_perm, const ea_bf16* cross,
                                        ea_bf16* out, int64_t out_batch_stride, int batch, int heads, int seq, int s_pad,
                                        int vt_pad, int row_off, const int* map, int window, float scale, void* stream);

/* dst[bh, c, p] = src[bh, c, col_off + token(p)] for the 64 channels of every (batch, head): V^T re-ordered along its token axis
 * by an axis permutation of the (frames, height, width) token grid -- head_order[h] in 0..5 picks the scan order of head h among
 * (f h w), (f w h), (h f w), (h w f), (w f h), (w h f) (processor.py:400-417).  src rows of src_pad elements, dst rows of dst_pad
 * (columns >= frames*height*width of dst are not written).  Tiled transposes through LDS: 64-byte runs read, 128-byte runs written. */
int ea_permute_cols_bf16(const ea_bf16* src, ea_bf16* dst, const int* head_order, int batch, int heads, int frames, int height,
                         int width, int src_pad, int dst_pad, int col_off, void* stream);

/* The same attention, resumable over key ranges: keys [kv_begin, kv_end) only (kv_begin % 64 == 0, kv_end
 * arbitrary), with the online-softmax state (un-normalised O, running max, partial row sums; fp32) carried in a
 * caller-owned `state` buffer of ea_attention_state_bytes(batch, heads, q_begin, q_end) bytes:
 *   flags bit 0: start from `state` instead of the empty state;  bit 1: write `state` instead of `out`.
 * A sequence-parallel rank runs {its local keys, flags 2} while the all-gather of the remote K / V^T shards is in
 * flight and then {remote keys, flags 1}: softmax is invariant to the key order, so the result is the full
 * attention (new capability: the reference has no multi-GPU inference).  Both calls must use
 * the same batch / heads / q_begin / q_end.  q_end <= s_pad. */
int64_t ea_attention_state_bytes(int batch, int heads, int q_begin, int q_end);
int ea_attention_fwd_range_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                int64_t out_batch_stride, int batch, int heads, int s_pad, int q_begin,
                                int q_end, int kv_begin, int kv_end, float scale, float* state, int flags,
                                void* stream);

/* ea_attention_fwd_range_bf16 over a head window (see ea_attention_fwd_segments_heads_bf16): k / vt are
 * [batch, heads, s_pad, 64] / [batch, heads, 64, s_pad] of the window's heads (kv_batch_stride elements between batch elements). */
int ea_attention_fwd_range_heads_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out,
                                      int64_t out_batch_stride, int batch, int heads, int s_pad, int q_begin, int q_end,
                                      int kv_begin, int kv_end, float scale, float* state, int flags, int q_head0, int q_heads,
                                      int64_t kv_batch_stride, void* stream);

/* ---- latent-space elementwise ---------------------------------------------------------------- */

/* Patchify gather for the 2x2/stride-2 Conv2d patch embedding (transformer3d.py:1523-1531):
 *   cols[b, (f,i,j), (c,di,dj)] = cat(latents, extra)[b, c, f, 2i+di, 2j+dj], zero-padded to k_pad.
 * latents fp32-or-bf16 selected by lat_is_bf16; extra may be NULL (c_extra = 0). */
int ea_patchify(const void* latents, const void* extra, ea_bf16* cols, int batch, int c_lat,
                int c_extra, int frames, int height, int width, int k_pad, int lat_is_bf16,
                void* stream);

/* Un-patchify (transformer3d.py:1683-1685): tokens bf16 [batch, F*h*w, C*2*2] -> out [batch,C,F,2h,2w]
 * (out fp32 or bf16). */
int ea_unpatchify(const ea_bf16* tokens, void* out, int batch, int channels, int frames, int h, int w,
                  int out_is_bf16, void* stream);

/* CFG combine + Flow-matching Euler step, fp32 math (pipeline_easyanimate.py:1102-1104,1111;
 * diffusers FlowMatchEulerDiscreteScheduler.step):
 *   v = v_uncond + g*(v_text - v_uncond);  x <- bf16/float( float(x) + dsigma * v )
 * v: [2, n] (uncond, text) when do_cfg else [1, n]; model-dtype in/out selected by is_bf16. */
int ea_cfg_euler_step(const void* v, void* latents, int64_t n, float guidance, float dsigma, int do_cfg,
                      int is_bf16, void* stream);

/* The same step with guidance_rescale > 0 (rescale_noise_cfg, pipeline_easyanimate.py:100-112,1106-1108):
 *   v_cfg = v_uncond + g*(v_text - v_uncond);  v = v_cfg * (r * std(v_text)/std(v_cfg) + 1 - r);  x <- x + dsigma * v
 * with the unbiased standard deviations over the n elements of the (single) sample, reduced on the device in a fixed
 * order (partial: fp32 workspace of 4*nblk floats; sums: 4 doubles; nothing is read back to the host). */
int ea_cfg_rescale_euler_step(const void* v, void* latents, int64_t n, float guidance, float dsigma,
                              float guidance_rescale, float* partial, int nblk, double* sums, int is_bf16, void* stream);

/* ---- VAE (AutoencoderKLMagvit), channels-last (NDHWC) activations of ONE sample -------------------- */

/* ---- TeaCache (transformer3d.py:90-121, 1564-1590, 1635) -------------------------------------- */

/* Numerator and denominator of the reference's rel-L1 distance between consecutive modulated inputs
 * (TeaCache.compute_rel_l1_distance, transformer3d.py:113-117), reduced on the device:
 *   sums[0] = sum_i | bf16(cur_i - prev_i) |      sums[1] = sum_i | prev_i |        (fp64, deterministic order)
 * cur/prev: bf16 [n], n % 8 == 0; partial: fp32 workspace of 2*nblk floats.  The host forms
 * bf16(bf16(sums[0]/n) / bf16(sums[1]/n)) -- the roundings torch applies to bf16 tensors -- after an 16-byte copy;
 * under sequence parallelism the two sums are all-reduced first. */
int ea_teacache_rel_l1_bf16(const ea_bf16* cur, const ea_bf16* prev, int64_t n, float* partial, int nblk,
                            double* sums, void* stream);

/* out = a - b (op 0) or a + b (op 1), bf16, fp32 arithmetic with one rounding (== the torch bf16 tensor op):
 * the cached residual `previous_residual = hidden_states - ori_hidden_states` (:1635) and its re-application
 * `hidden_states += previous_residual` (:1590).  out may alias a or b.  n % 8 == 0. */
int ea_bf16_binary(const ea_bf16* a, const ea_bf16* b, ea_bf16* out, int64_t n, int op, void* stream);

/* ---- the text-encoder step (pipeline_easyanimate.py:438-447: Qwen2-VL-7B's penultimate hidden state of 256 padded prompt tokens,
 * once per call; the encoder is a `transformers` dependency, its decoder layer restated from transformers' Qwen2-VL / Qwen2 model:
 * RMSNorm -> q/k/v Linear (bias) -> rotate-half rotary embedding -> causal grouped-query attention -> o Linear -> residual ->
 * RMSNorm -> down(SiLU(gate(x)) * up(x)) -> residual).  Linear layers: ea_gemm_bf16 (the residual adds ride in its gate + residual
 * epilogue with a gate of ones); norms: ea_rmsnorm_bf16. */

/* Rotary embedding in the rotate-half form (transformers apply_rotary_pos_emb: x * cos + rotate_half(x) * sin, rotate_half(x) =
 * [-x2 | x1]) fused with the head-major scatter: src [batch*seq, src_ld] holds heads x head_dim columns of a projection output,
 * dst [batch, heads, seq, head_dim]; cos / sin fp32 [batch*seq, head_dim] (both null: scatter only).  fp32 arithmetic, one rounding. */
int ea_rope_half_scatter_bf16(const ea_bf16* src, ea_bf16* dst, const float* cos, const float* sin, int batch, int seq, int heads,
                              int head_dim, int64_t src_ld, void* stream);

/* out[r, c] = SiLU(gate[r, c]) * up[r, c]  (Qwen2MLP act_fn(gate_proj(x)) * up_proj(x)); gate / up rows gate_ld / up_ld apart (two
 * column blocks of one projection output), out [rows, cols] contiguous; cols % 8 == 0. */
int ea_silu_mul_bf16(const ea_bf16* gate, const ea_bf16* up, ea_bf16* out, int64_t rows, int cols, int64_t gate_ld, int64_t up_ld,
                     void* stream);

/* softmax(q k^T * scale + mask) v for SHORT sequences with grouped-query heads (transformers sdpa_attention_forward with the mask of
 * create_causal_mask): q [batch, q_heads, seq, D], k [batch, kv_heads, seq, D], vt = v^T [batch, kv_heads, D, s_pad] (s_pad % 32 == 0,
 * columns >= seq finite), out [batch, seq, q_heads * D]; head h reads kv head h / (q_heads / kv_heads).  Masked: key > query when
 * causal != 0; key >= valid[b] when valid != NULL (int32 per batch element: the count of real tokens of a right-padded prompt).
 * D = 64 or 128.  One wave per 16 queries, operands straight from global memory: for prompt-sized sequences only. */
int ea_attention_causal_gqa_bf16(const ea_bf16* q, const ea_bf16* k, const ea_bf16* vt, ea_bf16* out, const int* valid, int batch,
                                 int q_heads, int kv_heads, int seq, int s_pad, int head_dim, int causal, float scale, void* stream);

/* Spatial tiling of the VAE wrapper (use_tiling / use_tiling_encoder / use_tiling_decoder, autoencoder_magvit.py:249-254,276-279):
 * the seam blends of tiled_encode / tiled_decode, blend_v (:319-327) and blend_h (:329-337), in place on tile b:
 *     b[o, y, i] = a[o, a_n - extent + y, i] * (1 - y / extent) + b[o, y, i] * (y / extent)      y < extent, i < inner, o < outer
 * blend_v on [B,C,T,H,W] tiles: outer = B*C*T, a_n = H_a, inner = W, outer strides = H*W of each tile; blend_h: outer = B*C*T*H,
 * a_n = W_a, inner = 1, outer strides = W of each tile.  bf16 != 0: bf16 elements, else fp32; fp32 arithmetic, one rounding (the
 * reference rounds each product and the sum to the tensor dtype). */
int ea_tile_blend(const void* a, void* b, int bf16, int64_t outer, int extent, int inner, int64_t a_outer_stride, int a_n,
                  int64_t b_outer_stride, void* stream);

/* tiled_decode's lower-right corner (autoencoder_magvit.py:426-445): the last tile_latent_min_size^2 latents are decoded once more
 * (q [outer, h, w]) and mixed into the last h x w pixels of dec [outer, H, W], in place:
 *     dec = wgt * q + (1 - wgt) * dec,   wgt[y][x] = min(linspace(0,1,w)[x], linspace(0,1,h)[y]). */
int ea_tile_corner_blend(const void* q, void* dec, int bf16, int64_t outer, int h, int w, int H, int W, void* stream);

/* out[b,r,:] = res[b,r,:] + gate[b,:] * x[b,r,:]  (bf16 in/out, fp32 fma, one rounding): the gated residual of
 * attention.py:1161-1162 as a stand-alone pass, used only by after_norm=True blocks (norm3 sits between the FFN GEMM and
 * the residual, :1150-1155, so the add cannot ride in the GEMM epilogue).  x/res/out: [batch, rows, dim] contiguous,
 * gate: fp32 rows of a [batch, gate_batch_stride] table.  dim % 8 == 0.  out may alias res or x. */
int ea_gated_residual_bf16(const ea_bf16* x, const ea_bf16* res, const float* gate, ea_bf16* out, int batch,
                           int64_t rows, int dim, int64_t gate_batch_stride, void* stream);

/* Causal 3-D convolution as an im2col-free implicit GEMM (vaemodules/common.py:84-179 CausalConv3d; the
 * strided down-samplers downsamplers.py:24-94; the up-samplers upsamplers.py:21-37,123-153; the residual add of
 * ResidualBlock3D common.py:322).
 *   x : bf16 [T_in, H_in, W_in, C_in]      C_in % 64 == 0, or C_in == 8 (below)
 *   w : bf16 [C_out, kt*kh*kw*C_in]        (the [C_out,C_in,kt,kh,kw] parameter permuted to tap-major)
 * C_in == 8 (3x3x3 only; the encoder's conv_in on RGB padded to one 16-byte chunk per voxel, omnigen_enc_dec.py:100-107):
 *   w : bf16 [C_out, 256] = 32 tap slots x 8 channels, zero in slots 27..31 and in the padded channels; a K tile is
 *   eight taps gathered by address (no im2col buffer).
 *   y : bf16 [T_out, H_out, W_out, C_out]  C_out % 8 == 0
 * Temporal padding = kt-1 replicated leading frames (identical to the reference's chunk caches, SURVEY 8c
 * property 1); spatial padding `pad` zeros on the low side and, for the pad-0 strided convs, one zero
 * row/column on the high side (F.pad(x,(0,1,0,1))).  st / ss = temporal / spatial stride in {1,2}.
 * ups = 1: the input is read through a nearest x2 spatial up-sampling (never materialised).
 * tdup is a bit set:
 *   1: output frame t >= 1 is written to frames 2t-1 and 2t of y (y then has 2*T_out-1 frames): the temporal nearest x2
 *      of SpatialTemporalUpsampler3D under spatial_group_norm (upsamplers.py:146-152), materialised;
 *   2: the same duplication kept VIRTUAL on the input side: x holds the T_in frames an up-sampler computed, the layer
 *      convolves the 2*T_in-1 logical frames (logical frame f = physical frame (f+1)>>1) -- the duplicated frames are
 *      never written, normalised or read twice from HBM (3x3x3, temporal stride 1 layers);
 *   4: the residual operand is such a virtual clip ((T_out+1)/2 physical frames);
 *   8: (with 2) MERGED temporal taps: the three temporal taps of a layer that reads a virtually duplicated clip touch only two
 *      physical frames -- logical (t-2, t-1, t) = physical (p-1, p-1, p) for odd t, (p-1, p, p) for even t, p = (t+1)>>1 -- so
 *      w is [2, C_out, 18 * C_in]: class 0 (even t) = {W_dt0, W_dt1 + W_dt2}, class 1 (odd t) = {W_dt0 + W_dt1, W_dt2}, summed in
 *      fp32 and rounded to bf16 once, taps ordered (dt', kh, kw); 2/3 of the MFMA work.  Row-slab shapes only
 *      (ea_conv3d_cl_tmerge_ok).
 *  16: x is CHANNEL-BLOCKED, bf16 [C_in/32][T_in][H_in][W_in][32] (ea_groupnorm_apply_bf16 with act bit 1 writes it): the four-wave
 *      row-slab kernels then stage a slab's 16-voxel pieces as 1 KiB of consecutive memory.  Only where ea_conv3d_cl_blocked_ok says
 *      so (3x3x3, stride 1, pad 1, ups = 0, no bit 1); combines with bits 2 / 4 / 8.  Same values, same result.
 * res (optional, same shape as the un-duplicated output -- or its physical frames with tdup & 4): y = conv + bias + res.
 * zeros: any device buffer holding >= 128 zero bytes (source of the spatial zero padding). */
int ea_conv3d_cl_bf16(const ea_bf16* x, const ea_bf16* w, const float* bias, const ea_bf16* res, ea_bf16* y,
                      const ea_bf16* zeros, int T_in, int H_in, int W_in, int C_in, int C_out, int kt, int kh,
                      int kw, int st, int ss, int pad, int ups, int tdup, void* stream);

/* 1 if ea_conv3d_cl_bf16 serves a 3x3x3 / stride 1 / pad 1 layer of this shape (T_logical output frames) with a kernel that
 * accepts tdup bit 8 (merged temporal taps), else 0. */
int ea_conv3d_cl_tmerge_ok(int T_logical, int H, int W, int C_in, int C_out);

/* Would ea_conv3d_cl_bf16 read a CHANNEL-BLOCKED input for this layer (3x3x3, stride 1, pad 1, no folded up-sampling / duplicate
 * store; T = output frames)?  tdup bit 16 of ea_conv3d_cl_bf16 / ea_conv3d_cl_stats_bf16 declares x as [C_in/32][T_in][H_in][W_in][32]
 * (T_in PHYSICAL frames) -- the layout ea_groupnorm_apply_bf16 writes with act bit 1 (2) -- which only the four-wave row-slab
 * kernels read: a slab's 16-voxel LDS-DMA piece is then 1 KiB of consecutive memory.  Same values, same results. */
int ea_conv3d_cl_blocked_ok(int T, int H, int W, int C_in, int C_out);

/* "Nearest x2 spatial up-sampling, then 3x3x3 causal convolution" (SpatialUpsampler3D / SpatialTemporalUpsampler3D,
 * upsamplers.py:21-37,123-153) in SUB-PIXEL form: output pixel (2i + a, 2j + b) sees only 2 x 2 distinct source pixels, so each
 * of the four parity classes (a, b) is a 3 x 2 x 2 convolution on the SOURCE grid whose weights are sums of the original
 * taps -- 12 taps instead of 27 (44 % of the MFMA work of ea_conv3d_cl_bf16 with ups = 1).  The sums are formed in fp32 and
 * rounded to bf16 once (a rounding point the reference does not have: results agree to bf16 weight-rounding noise, not
 * bit for bit).
 *   x  : bf16 [T, H, W, C_in]            W % 256 == 0, C_in % 64 == 0
 *   w4 : bf16 [4, C_out, 12 * C_in]      class 2a + b; taps ordered (dt, row tap, column tap), channel-minor; row tap 0 / 1 of
 *                                        class a = 0 is kh {0} / {1,2}, of a = 1 is kh {0,1} / {2}; columns likewise with b
 *   y  : bf16 [T (or 2T-1 with tdup = 1), 2H, 2W, C_out]   C_out % 256 == 0
 * gn_partial / gn_capacity_floats / gn_nblk_out: as ea_conv3d_cl_stats_bf16 (NULL / 0 / NULL: no statistics). */
int ea_conv3d_cl_subpixel_bf16(const ea_bf16* x, const ea_bf16* w4, const float* bias, ea_bf16* y, int T_in, int H_in, int W_in,
                               int C_in, int C_out, int tdup, float* gn_partial, int64_t gn_capacity_floats, int* gn_nblk_out,
                               void* stream);

/* ea_conv3d_cl_bf16 that also emits the per-frame GroupNorm statistics of ITS OUTPUT -- what the next layer's
 * GroupNorm (common.py:301-305,318) needs -- from the convolution epilogue, instead of a separate pass over the
 * activation (3.5 % of a 49 x 1024^2 decode).  Deterministic two-level form: the epilogue writes one (sum, sumsq) pair per
 * (frame, 256-voxel row tile, wave row, 4-channel bundle) into gn_partial (capacity in floats); ea_groupnorm_finalize_bf16
 * reduces them in a fixed order.  Only the row-slab kernel does this (3x3x3 / stride 1 / pad 1 layers whose output rows
 * are a multiple of 256 voxels wide; with tdup both copies of a frame get the sums): *gn_nblk_out (HOST int) is the number of partial blocks per frame
 * that were written, or 0 -- then the caller runs ea_groupnorm_stats_bf16 as before. */
int ea_conv3d_cl_stats_bf16(const ea_bf16* x, const ea_bf16* w, const float* bias, const ea_bf16* res, ea_bf16* y,
                            const ea_bf16* zeros, int T_in, int H_in, int W_in, int C_in, int C_out, int kt, int kh,
                            int kw, int st, int ss, int pad, int ups, int tdup, float* gn_partial,
                            int64_t gn_capacity_floats, int* gn_nblk_out, void* stream);
/* stats fp32 [T, groups, 2] = (mean, rstd) from partial [T, nblk, C/4, 2] (sum, sumsq per 4-channel bundle), fp64,
 * fixed order; hw = voxels per frame. */
int ea_groupnorm_finalize_bf16(const float* partial, float* stats, int T, int64_t hw, int C, int groups, int nblk,
                               float eps, void* stream);

/* Second half of a narrow-N (C_out <= 4) 3x3x3 / stride 1 / pad 1 causal convolution -- the decoder's conv_out 128 -> 3
 * (omnigen_enc_dec.py:611): the first half is ONE ea_gemm_bf16 (EA_EPI_F32_OUT) with the weight, re-packed to
 * [27*C_out, C_in] (row tap*C_out + co, tap = (dt*3+dh)*3+dw), as the ROW operand and the activation voxels as the column
 * operand: z[tap*C_out + co, v] = sum_ci w[co,ci,tap] * x[v,ci] -- one plane of voxels per (tap, co).  This kernel sums
 * the 27 taps:  y[v, co] = bias[co] + sum_tap z[tap*C_out + co, v + offset(tap)], frame index clamped at 0 (causal
 * replicate padding), zero padding in space.  27x fewer MFMA flops than running the 3-column problem through a 128-wide
 * implicit-GEMM tile.  z: fp32 [27*C_out, ld], ld >= T*H*W; y: bf16 [T,H,W,C_pad] (channels >= C_out zero). */
int ea_conv3d_tap_gather_f32(const float* z, const float* bias, ea_bf16* y, int T, int H, int W, int64_t ld, int C_out,
                             int C_pad, void* stream);

/* Explicit im2col for the few convolutions with C_in % 64 != 0 (conv_in 3->128, decoder conv_in 16->512, the
 * 1x1x1 quant convs autoencoder_magvit.py:181-182): cols bf16 [T_out*H_out*W_out, k_pad],
 * cols[m, tap*C_in + c], zero padded; the product is ea_gemm_bf16. */
int ea_im2col3d_bf16(const ea_bf16* x, ea_bf16* cols, int T_in, int H_in, int W_in, int C_in, int kt, int kh,
                     int kw, int st, int ss, int pad, int k_pad, void* stream);

/* Per-frame GroupNorm statistics (common.py:301-305 under spatial_group_norm; omnigen_enc_dec.py:258-262):
 * x bf16 [T, hw, C] -> stats fp32 [T, groups, 2] = (mean, rstd).  Deterministic two-level reduction:
 * partial is an fp32 workspace of T*nblk*(C/4)*2 floats.  (C/groups) % 4 == 0. */
int ea_groupnorm_stats_bf16(const ea_bf16* x, float* partial, float* stats, int T, int64_t hw, int C, int groups,
                            int nblk, float eps, void* stream);

/* y = act((x - mean) * rstd * gamma + beta); act: 0 none, 1 SiLU (common.py:306,318); + 2: y is written channel-blocked,
 * [C/32][T][hw][32], for a convolution that reads it with tdup bit 16 (ea_conv3d_cl_blocked_ok); y must not alias x then. */
int ea_groupnorm_apply_bf16(const ea_bf16* x, ea_bf16* y, const float* stats, const float* gamma,
                            const float* beta, int T, int64_t hw, int C, int groups, int act, void* stream);

/* y = softmax(x * scale) along rows; bf16 [rows, cols], cols % 8 == 0, cols <= 32768.  Used with two
 * ea_gemm_bf16 calls for the single-head (head_dim 512) spatial attention of the VAE mid block
 * (vaemodules/attention.py:391-423, attention_processors.py:76-139). */
int ea_softmax_rows_bf16(const ea_bf16* x, ea_bf16* y, int64_t rows, int cols, float scale, void* stream);
/* same with fp32 logits in (written by ea_gemm_bf16 with EA_EPI_F32_OUT), bf16 probabilities out */
int ea_softmax_rows_f32in(const float* x, ea_bf16* y, int64_t rows, int cols, float scale, void* stream);

/* Layout changes between the reference's NCDHW tensors and the kernels' NDHWC (one sample).
 * to NDHWC: src [C, voxels] fp32/bf16 -> dst bf16 [voxels, C_pad] (extra channels zero).
 * to NCDHW: src bf16 [voxels, C_src] -> dst [C, voxels] fp32/bf16 (first C channels); post = 1 applies
 * clamp(-1,1) -> x/2+0.5 -> clamp(0,1) (pipeline_easyanimate.py:731,739). */
int ea_ncdhw_to_ndhwc(const void* src, ea_bf16* dst, int C, int C_pad, int64_t voxels, int src_is_bf16,
                      void* stream);
int ea_ndhwc_to_ncdhw(const ea_bf16* src, void* dst, int C, int C_src, int64_t voxels, int dst_is_bf16, int post,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EA_MI355X_H */
