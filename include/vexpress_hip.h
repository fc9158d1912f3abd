/* libvexpress_hip.so — C ABI of the MI355X (gfx950) V-Express denoising hot path.
 *
 * The reference (tencent-ailab/V-Express) has no FFI/plugin interface: its hot path is Python calling
 * PyTorch aten ops.  Each entry point below replaces the aten call sites named in
 * its comment (paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates all buffers, incl. workspaces);
 *   - activations are 16-bit, channels-last tokens:  [(b f), H*W, C]  ==  NHWC per frame.  The element type is a
 *     property of the LIBRARY (vx_element_type()): libvexpress_hip.so computes on bfloat16, libvexpress_hip_f16.so - the same
 *     sources compiled with -DVX_ELEM_F16, the same ABI - on IEEE half, the reference's default `--dtype fp16`
 *     (inference.py:44,150-151); both accumulate in float32.  "bf16" in the comments below = "the library's element";
 *   - norm affine parameters, biases and statistics are float32; weights are 16-bit elements [N, K], K contiguous,
 *     conv weights pre-laid-out as [Cout][ky][kx][Cin];
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*); no host threads, no
 *     device-wide synchronisation, no global mutable state except the last-error string;
 *   - return value: 0 = ok, <0 = error (VX_ERR_*), message via vx_last_error_string(); nothing throws or exits.
 */
#ifndef VEXPRESS_HIP_H
#define VEXPRESS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_ABI_VERSION 15

const char* vx_last_error_string(void);
int vx_abi_version(void);
/* Identity of this BINARY: "<hash of the kernel sources it was compiled from>|<extra -D flags of the build>" as stamped by
 * csrc/Makefile ("unstamped|" for a library built any other way).  The Python binding compares the hash with the sources
 * on disk (v_express_amd.lib.source_id): measurements under profiles/ are keyed by it. */
const char* vx_build_id(void);
/* "bf16" or "f16": the 16-bit element type this binary stores activations / weights in and feeds the MFMAs with (ABI 14) */
const char* vx_element_type(void);
/* device properties snapshot: out[0]=CU count, out[1]=LDS bytes per block, out[2]=wavefront size, out[3]=clock kHz */
int vx_device_info(int device, int* out4);

/* ---- GEMM / implicit-GEMM convolution on MFMA -------------------------------------------------------------
 * out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),  m = (frame, oy, ox), k = (ky, kx, ci).
 * Replaces: F.conv2d via InflatedConv3d (modules/resnet.py:9-17: resnet conv1/conv2 :223,:244, conv_shortcut :247,
 * Downsample3D :106-118, Upsample3D :59-88 with `upsample`=1 fusing the nearest-2x gather), Transformer3DModel
 * proj_in/proj_out 1x1 (modules/transformer_3d.py:126,154), every F.linear of diffusers Attention/FeedForward reached
 * from modules/mutual_self_attention.py:177-247 and modules/motion_module.py:158,177,243-256, TimestepEmbedding
 * (modules/unet_3d.py:470) and ResnetBlock3D.time_emb_proj (modules/resnet.py:226), torch.cat skip concat
 * (modules/unet_3d_blocks.py:694,831) as the dual-source A operand (a/a2).
 */
enum { VX_EPI_STORE = 0, VX_EPI_GEGLU = 1, VX_EPI_SPLIT = 2 };
enum { VX_PART_ROWS = 0, VX_PART_VT = 1 };
enum { VX_ACT_NONE = 0, VX_ACT_SILU = 1, VX_ACT_GELU = 2 };   /* GELU = erf form (nn.GELU default) */

typedef struct {
  /* A operand: up to two channel-concatenated NHWC sources (a2 may be NULL) */
  const void* a;
  const void* a2;
  int32_t c1, c2;            /* channels taken from a / a2 (multiples of 8)                       */
  int32_t lda1, lda2;        /* pixel (row) stride of a / a2 in elements                          */
  int32_t nb, h_in, w_in;    /* frames, stored input height/width (before the optional upsample)   */
  int32_t kh, kw, stride, pad;
  int32_t upsample;          /* 1: input is nearest-2x upsampled on the fly                        */
  int32_t h_out, w_out;
  /* B operand */
  const void* w;             /* bf16 [n, k], k = kh*kw*(c1+c2)                                    */
  int32_t n, k, m;           /* m = nb*h_out*w_out                                                */
  /* epilogue */
  int32_t epi;               /* VX_EPI_*                                                          */
  int32_t act;               /* VX_ACT_* (STORE only)                                             */
  float alpha;               /* STORE: out = residual + alpha * act(acc + bias + rowbias)         */
  const float* bias;         /* [n] or NULL                                                       */
  const float* rowbias;      /* [m / rows_per_group][rowbias_ld] or NULL (time-embedding add)     */
  int32_t rowbias_ld, rows_per_group;
  const void* residual;      /* bf16 [m, ldr] or NULL (may alias out)                             */
  int32_t ldr;
  void* out;                 /* STORE/GEGLU destination                                           */
  int32_t ldc;
  int32_t out_f32;           /* STORE: 1 = write float32 instead of bf16                          */
  /* SPLIT: columns [p*part_cols, (p+1)*part_cols) go to part p */
  int32_t part_cols, n_parts;
  void* part_out[3];
  int32_t part_kind[3];      /* VX_PART_ROWS: [m, part_ld] ; VX_PART_VT: [m/seq_len, heads, head_dim, vt_pitch] */
  int32_t part_ld[3];
  int32_t seq_len, head_dim, vt_pitch;
  /* split-K (STORE only): splitk >= 2 cuts the K loop into that many contiguous slices, each computed by its own
   * blocks into the float32 workspace splitk_ws[splitk][m][n] (caller-owned, vx_gemm_splitk_ws_bytes()); a second
   * launch adds the slices in slice order (deterministic) and applies the epilogue.  For tall-K, few-row problems
   * (the 8x8 level: M = 2048) that would otherwise fill half of the 256 CUs.  0 / 1 = off. */
  int32_t splitk;
  void* splitk_ws;
  /* Kernel-selection hint for the persistent ring-staged kernel (its conv K order differs from the classic tiles', so
   * callers that need bit-identical results for any sub-batch must make the choice from batch-independent facts):
   * 0 = automatic (ring when eligible and the launch has >= 192 tiles), 1 = ring whenever structurally eligible,
   * -1 = never.  The choice is a PER-CALL fact (ABI 14: the process-wide vx_gemm_set_ring_mode / vx_gemm_set_fp8_ring
   * switches of ABI <= 13 are gone - two pipelines in one process raced on them; what is left in the library are A/B
   * environment knobs read once at first use, none of them settable through the ABI).
   * 3 (ABI 14) = fp8 operands (a_fp8) on the persistent kernel: an explicit opt-in per call - fp8 launches with any other
   * hint run on the classic fp8 tiles, which measure faster (tests/test_gpu_kernels.py::test_gemm_fp8_ring_vs_classic_tiles).
   * 2 (round 5, ABI 13) = the persistent kernel with a COOPERATIVE two-way K split, for launches with fewer 256 x 320
   * tiles than CUs (the 16x16 level: 8192 x 1280 = 128 tiles): set splitk = 2 and splitk_ws = a workspace of
   * vx_gemm_splitk_ws_bytes(m, n, 2) bytes that the caller ZEROED once, and coop_epoch (below) = that workspace's launch
   * counter; launches that share a workspace must be stream-ordered.  Each (tile, K half) is one work item; the two
   * halves meet inside the launch (the first to finish parks its fp32 accumulators in the workspace, the second adds
   * them and runs the epilogue: one launch, a fixed summation order, no reduce pass).  STORE epilogue (bias, row bias,
   * SiLU, residual, GroupNorm partial sums), bf16 operands, (c1 + c2) / 64 even.  Ask vx_gemm_ring_coop_ok() - a function
   * of the problem's shape only; callers that need batch-invariant bits must ALSO make the choice itself from per-item
   * facts.  VX_ERR_UNSUPPORTED if the launch cannot run this way. */
  int32_t ring_hint;
  /* FP8 operands (BASELINE.json configs[4]: "fp8 MFMA QKV/out-proj"): a_fp8 = 1 -> a and w hold OCP e4m3 bytes
   * (v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, fp32 accumulation), row-scaled: the true operands are
   * a[m, :] * a_scale[m] and w[n, :] * w_scale[n], so out = epilogue(acc * a_scale[m] * w_scale[n]).  Plain linears
   * only (kh = kw = 1, one source, no padding / upsampling / split-K); k, lda1 count fp8 elements and must be multiples
   * of 128 / 16 (vx_layernorm_fp8 pads K with zeros); STORE and SPLIT epilogues. */
  int32_t a_fp8;
  const float* a_scale;      /* [m] */
  const float* w_scale;      /* [n] */
  /* LayerNorm folded into the GEMM (round 2): with W' = gamma (.) W (columns scaled), s[n] = sum_k W'[n, k] and
   * b' = b + W beta prepared at load time, LN(x) W^T + b == rstd[m] * (x W'^T - mean[m] * s[n]) + b'.  ln_stats != NULL:
   * the raw accumulators are transformed acc <- rstd[m] * (acc - mean[m] * ln_colsum[n]) before the epilogue (any
   * epilogue; not with split-K), so the LayerNorm output is never written or re-read: the caller passes the UN-normalised
   * rows as A, the folded weight as w, b' as bias and the row statistics of vx_row_stats.  Replaces the F.layer_norm +
   * F.linear pairs of modules/mutual_self_attention.py:177-247 and modules/motion_module.py:243-256. */
  const float* ln_stats;     /* [m][2] = (mean, rstd) per row, or NULL */
  const float* ln_colsum;    /* [n] */
  /* Row statistics of the OUTPUT (round 3; STORE epilogue into bf16): row_stats_out != NULL -> row_stats_out[m] = (mean,
   * 1 / sqrt(var + row_stats_eps)) of the stored bf16 row out[m, 0:n], i.e. exactly what the next GEMM needs as its
   * ln_stats when it folds the LayerNorm of this output (modules/mutual_self_attention.py:177-247: every norm reads the
   * residual stream a projection has just written).  When one 256 x 320 tile of the persistent kernel holds whole rows
   * (n == 320: the 64x64 level) the epilogue produces them from its registers and the read-only vx_row_stats pass over the
   * tensor disappears; otherwise vx_gemm runs vx_row_stats on `out` itself after the launch - same result contract
   * (variance of the stored values), NOT the same low bits: vx_row_stats is two-pass (mean, then squared deviations),
   * the fused form is one-pass: var = max(0, E[x^2] - mean^2) from float32 sums of x and x^2 taken in a fixed order.
   * Precision of the one-pass form over 320 values: relative error of the variance about 1e-7 * (1 + mean^2 / var) *
   * a few, i.e. rstd within 1e-4 of float64 for |mean| / std <= 2.5 and within 3e-3 at |mean| / std = 30 (both tested,
   * tests/test_gpu_kernels.py::test_gemm_row_stats_out).  Which form a row gets is a function of n only (never of m), so
   * a frame's statistics do not depend on what else is in the launch. */
  float* row_stats_out;      /* [m][2] or NULL */
  float row_stats_eps;
  /* Per-row-group weights (round 3): w_group_rows > 0 -> output rows [g * w_group_rows, (g + 1) * w_group_rows) multiply
   * the weight matrix w + g * n * k (w holds m / w_group_rows matrices back to back).  This is how a GroupNorm WITHOUT
   * activation in front of a 1x1 / linear layer is folded into that layer (vx_groupnorm_fold_linear: one scaled weight
   * copy per frame, modules/transformer_3d.py:124-126, modules/motion_module.py:156-158).  Only the persistent 256 x 320
   * kernel implements it: w_group_rows % 256 == 0 and every other ring condition must hold, else VX_ERR_UNSUPPORTED. */
  int32_t w_group_rows;
  /* GroupNorm partial sums of the OUTPUT (round 4; STORE epilogue into bf16): gn_ws != NULL -> the epilogue adds up the
   * stored bf16 values per (frame, row slab, group) and writes (sum, sum of squares) into gn_ws in the workspace layout
   * vx_groupnorm's own statistics pass produces, [frame][slab][group][2] float32 with vx_gemm_gn_slabs(p) slabs per frame
   * (one slab = the rows one wave row of the launch's tile owns: 128 for the persistent 256 x 320 kernel, 64 for the
   * classic tiles).  The GroupNorm that reads this tensor next (modules/resnet.py:220-221,235-241: norm2 reads conv1's
   * output; modules/transformer_3d.py:124 and modules/motion_module.py:156 read the previous block's) then runs its apply
   * pass only (vx_groupnorm_apply) or folds into its linear layer (vx_groupnorm_fold_linear) without ever re-reading the
   * tensor for statistics.  gn_groups = group count of that GroupNorm over this tensor's n channels, gn_hw = output rows
   * per frame.  A launch that cannot produce them (vx_gemm_gn_slabs(p) == 0: split-K, fp32 output, groups straddling
   * wave columns, ...) is rejected with VX_ERR_UNSUPPORTED - ask first.  Deterministic and batch-invariant (fixed
   * summation order per slab). */
  float* gn_ws;
  int32_t gn_groups, gn_hw;
  /* Row statistics in TWO parts (round 4, n == 640 / k == 640: the 32x32 level, where one 256 x 320 tile of the persistent
   * kernel holds half a row).  row_stats_parts == 2 (requires n == 640): row_stats_out is [m][2][2] = (sum, sum of squares)
   * of the stored bf16 values of columns [0, 320) and [320, 640) of every row - each half comes out of one tile's epilogue,
   * no pass re-reads the tensor (launches that do not run on the persistent kernel fill the same format with
   * vx_row_stats_parts).  The two producers sum in different fp32 orders (epilogue: 20 columns per lane, then the 4 wave
   * columns; vx_row_stats_parts: 32 lanes, stride 8), so the low bits of a row's sums depend on WHICH one ran: they are a
   * function of per-item facts (batch-invariant) only when the caller pins the kernel choice with ring_hint != 0, as the
   * Python layer does under ops.frame_rows(items=...); with ring_hint == 0 the choice follows the launch size.
   * ln_stats_parts == 2 (requires k == 640 and a launch on the persistent kernel - ask
   * vx_gemm_config_name; else VX_ERR_UNSUPPORTED: convert with vx_row_stats_finalize): ln_stats holds that format; the
   * epilogue adds the two halves and takes mean = s / k, rstd = 1 / sqrt(max(0, q / k - mean^2) + ln_eps) itself (one-pass
   * variance, see row_stats_out).  0 / 1 = the (mean, rstd) format. */
  int32_t row_stats_parts, ln_stats_parts;
  float ln_eps;
  /* Epoch of a cooperative-split launch (ring_hint == 2; ABI 14): a counter the caller keeps PER WORKSPACE, starting at 1
   * after zeroing it and incremented for every launch on it (1 <= coop_epoch < 2^27; re-zero the workspace before wrapping).
   * The rendezvous words of the workspace carry the epoch of the launch that wrote them and are never reset, so a launch
   * that gave up on a lost partner (the poll is bounded) or was aborted cannot poison later launches: they wait for THEIR
   * epoch.  VX_ERR_INVALID for a value outside the range. */
  int32_t coop_epoch;
} vx_gemm_params;

int vx_gemm(const vx_gemm_params* p, void* stream);
/* slabs per frame that vx_gemm(p) writes into p->gn_ws (p->gn_groups / gn_hw set), or 0 when the launch p would get
 * cannot produce GroupNorm partial sums (the caller then leaves gn_ws NULL and runs vx_groupnorm as before) */
int vx_gemm_gn_slabs(const vx_gemm_params* p);
int64_t vx_gemm_splitk_ws_bytes(int m, int n, int splitk);
/* 1 if vx_gemm(p) could run with ring_hint = 2 (cooperative two-way K split on the persistent kernel; p's own ring_hint /
 * splitk / splitk_ws are ignored), else 0 */
int vx_gemm_ring_coop_ok(const vx_gemm_params* p);
/* name of the tile configuration vx_gemm would launch for p (profiling / roofline reports); thread-local storage */
const char* vx_gemm_config_name(const vx_gemm_params* p);
/* the kernel instantiation the LAST vx_gemm call of this thread launched, spelled as rocprofv3 prints it without the
 * namespace ("gemm_ring_kernel<0, true, false, false, false, true>"): lets bench.py pair its HIP-event figures with the
 * rows of a committed kernel trace one to one.  "" before the first launch. */
const char* vx_gemm_last_kernel(void);
/* the same for EVERY MFMA entry point of the library (vx_gemm, vx_attention, vx_attention_bounded, vx_temporal_attention,
 * vx_ff_fused, vx_tblock_fused): the instantiation the last such call of this thread launched
 * ("attn3_kernel<2, false, true, false>", "ff_fused_kernel<2>", "tblock_kernel", ...). */
const char* vx_last_kernel(void);

/* ---- Fused GEGLU feed-forward of the 64x64 level (round 4) -------------------------------------------------------
 * out = residual + (value * gelu(gate)) W2^T + bias2,  [value | gate] = LN(x) W1^T + b1   in ONE launch: the [m, 4C]
 * intermediate of diffusers FeedForward(activation_fn="geglu") (modules/mutual_self_attention.py:247,
 * modules/motion_module.py:256) never leaves the CU.  C = 320, hidden = 1280 only (x fragments of a 128-row tile live in
 * registers).  The LayerNorm in front is folded in exactly as vx_gemm_params.ln_stats does it: w1 = the folded,
 * value/gate-interleaved weight, bias1 / ln_colsum its bias and column sums (float32 [2 hidden], interleaved order),
 * ln_stats = (mean, rstd) per row.  w1t / w2t = the two weights re-tiled by vx_ff_pack_weights (MFMA-fragment-major:
 * a 32-channel hidden chunk is one contiguous 40 KB / 20 KB block).  m % 128 == 0. */
typedef struct {
  const void* x;             /* bf16 [m, ldx]: the un-normalised rows */
  int32_t ldx, m, c, hidden;
  const void* w1t;           /* bf16, vx_ff_pack_weights layout of the [2 hidden, c] interleaved weight */
  const void* w2t;           /* bf16, vx_ff_pack_weights layout of the [c, hidden] weight */
  const float* bias1;        /* [2 hidden] or NULL */
  const float* ln_colsum;    /* [2 hidden] */
  const float* ln_stats;     /* [m][2] */
  const float* bias2;        /* [c] or NULL */
  const void* residual;      /* bf16 [m, ldr] or NULL (may alias out) */
  int32_t ldr;
  void* out;                 /* bf16 [m, ldo] */
  int32_t ldo;
} vx_ff_params;
int vx_ff_pack_weights(const void* w1_interleaved, const void* w2, void* w1t, void* w2t, int c, int hidden, void* stream);
int vx_ff_fused(const vx_ff_params* p, void* stream);

/* ---- Fused temporal self-attention block of the 64x64 level (round 4) ----------------------------------------------
 * x <- x + to_out(softmax_over_frames(q k^T * scale) v),  [q | k | v] = (LN(x) + pe[frame]) Wqkv^T + b   in ONE launch:
 * VersatileAttention.forward inside TemporalTransformerBlock.forward (modules/motion_module.py:243-256, :351-388: the two
 * rearranges, pos_encoder, to_q / to_k / to_v, attention over the frame axis per (pixel, head), to_out, residual) - the
 * [m, 3C] projections and the [m, C] attention output never reach memory.  C = 320, 8 heads; f = 16 frames (a tile = 8
 * pixels x their 16 frames, hw % 8 == 0) or - round 5 - f = 24, the reference's default window (inference.py:67: a tile = 4
 * pixels x two 16-row blocks, the second one frames 16-23 + 8 masked padding rows; hw % 4 == 0); the x rows of a tile live
 * in registers; rows are [(b f), hw] frame-major as everywhere (row = (item * f + frame) * hw + pixel).  The LayerNorm is folded in as in vx_gemm_params.ln_stats: wqkv =
 * the folded weight, bias / colsum its bias and column sums, pe_rows = the positional table pushed through the weight
 * (float32 [f][3C]: row `frame` is added to every row of that frame).  ln_stats = (mean, rstd) per row, or NULL: the
 * kernel then takes them from the rows it holds (two-pass, float32).  vx_tblock_pack re-tiles the weights and tables once
 * per layer AND window length f:  wqkv_t vx_tblock_packed_bytes(f) = 720896 / 786432 B (weights and the bias / positional
 * tables of each 16-column block together), wo_t 204800 B, colsum_p 4096 B. */
typedef struct {
  void* x;                   /* bf16 [b f hw, ldx], updated in place */
  int32_t ldx, b, f, hw, c, heads;
  const void* wqkv_t;        /* vx_tblock_pack outputs */
  const void* wo_t;
  const float* colsum_p;
  const float* bias_o;       /* [c] or NULL */
  const float* ln_stats;     /* [m][2] or NULL */
  float* stats_out;          /* [m][2] or NULL: (mean, rstd) of the rows written, for the next LayerNorm fold (one-pass
                                variance like vx_gemm_params.row_stats_out; may alias ln_stats: a tile reads its own rows'
                                entries before it writes them) */
  float ln_eps, scale;       /* LayerNorm eps (ln_stats == NULL, stats_out); softmax scale (head_dim^-0.5) */
} vx_tblock_params;
int vx_tblock_pack(const void* wqkv, const float* bias, const float* colsum, const float* pe_rows, int pe_ld,
                   const void* wo, void* wqkv_t, void* wo_t, float* colsum_p, int c, int heads, int f, void* stream);
int vx_tblock_fused(const vx_tblock_params* p, void* stream);
int64_t vx_tblock_packed_bytes(int f);

/* ---- Audio cross-attention of a spatial transformer block in ONE launch (round 6, ABI 15) ------------------------------
 * h <- h + alpha * to_out(softmax(q k^T / sqrt d) v),  q = to_q(LayerNorm(h)),  k | v = to_k | to_v of the frame's audio tokens:
 * BasicTransformerBlock.attn2 as patched by modules/mutual_self_attention.py:227-244 (diffusers Attention + AttnProcessor2_0,
 * hidden_states = norm2(h), encoder_hidden_states = audio tokens, the result weighted by audio_attention_weight and added to
 * h).  A frame has FIVE audio tokens (AudioProjection num_queries, inference.py / modules/audio_projection.py), so per
 * frame the block is two skinny products with frame-dependent operands that do not change over the DDIM steps:
 *   S[m, (head, t)] = rstd_m (sum_c x[m, c] Kq_f[(head, t), c] - mean_m colsum_f[(head, t)]) + sbias_f[(head, t)],
 *       Kq_f = log2(e) / sqrt(d) * K_f,head to_q_head  (to_q LayerNorm-folded like vx_gemm_params.ln_stats),
 *   P = softmax over the 5 tokens of each head,   y[m, :] = x[m, :] + alpha (P VO_f^T + bias_o),   VO_f = to_out_head V_f,head^T.
 * vx_audio_xattn_pack builds Kq_f / colsum / sbias / VO_f once per clip (kv: [frames * 5, ldkv] elements, K | V columns = the
 * output of the to_k | to_v GEMM; wq: folded to_q weight [c, c] with bq its folded bias or NULL; wo: to_out weight [c, c]) into
 * MFMA-fragment-major buffers of vx_audio_xattn_packed_bytes(c, frames) / 2 bytes each (+ float32 [frames][48] x 2);
 * vx_audio_xattn streams the rows once: no [rows, c] query or attention-output tensor exists, FLOPs drop by c / 48 against the
 * q / to_out projections it replaces.  8 heads, 5 tokens, c % 320 == 0, frames of rows_per_frame % 16 == 0 rows
 * (vx_audio_xattn_supported); anything else keeps vx_gemm + vx_small_kv_attention + vx_gemm. */
typedef struct {
  const void* x;             /* elements [rows, ldx]: the residual stream h (read as the query source AND the residual) */
  int32_t ldx;
  void* out;                 /* elements [rows, ldo]; may be x (in place: a wave reads its rows before it writes them) */
  int32_t ldo;
  int32_t rows, c, rows_per_frame;   /* row r belongs to frame r / rows_per_frame of the packed operands */
  const float* ln_stats;     /* LayerNorm statistics of x's rows: [rows][2] (mean, rstd) or [rows][4] two-part sums */
  int32_t ln_stats_parts;    /* 0 / 2, as vx_gemm_params.ln_stats_parts */
  float ln_eps;
  const void* kq;            /* vx_audio_xattn_pack outputs */
  const float* kq_colsum;
  const float* kq_bias;
  const void* vo;
  const float* bias_o;       /* to_out bias [c] */
  float alpha;               /* audio_attention_weight */
  float* row_stats_out;      /* NULL, or the statistics of the STORED rows for the next LayerNorm fold (may alias ln_stats) */
  int32_t row_stats_parts;   /* 0: (mean, rstd) [rows][2]; 2: two-part sums [rows][4] (vx_gemm_params.row_stats_parts) */
  float row_stats_eps;
} vx_axattn_params;
int64_t vx_audio_xattn_packed_bytes(int c, int frames);
int vx_audio_xattn_supported(int c, int heads, int n_ctx, int rows_per_frame);
int vx_audio_xattn_pack(const void* kv, int ldkv, const void* wq, const float* bq, const void* wo, int c, int heads, int n_ctx,
                        int frames, void* kq, float* kq_colsum, float* kq_bias, void* vo, void* stream);
int vx_audio_xattn(const vx_axattn_params* p, void* stream);

/* ---- GroupNorm (+SiLU), per-frame statistics, NHWC, optional dual (concat) source --------------------------
 * Replaces F.group_norm via InflatedGroupNorm (modules/resnet.py:20-28; :220-221,:235,:241), Transformer3DModel.norm
 * (modules/transformer_3d.py:124), motion-module norm (modules/motion_module.py:156), conv_norm_out + SiLU
 * (modules/unet_3d.py:571-572).  ws: float32 workspace of vx_groupnorm_ws_floats() elements.
 * out_pad = 0: out is [frames, hw, C].  out_pad = p > 0: out is the interior of a zero-bordered
 * [frames, H + 2p, W + 2p, C] image (W = width, H = hw / width) whose border the caller keeps zero, so that the
 * following 3x3 convolution runs as a pad-0 ("valid") conv without bounds checks (vx_gemm fast addressing).
 * silu: activation code applied after the affine (VX_ACT_*: 0 none, 1 SiLU, 2 erf-GELU - the wav2vec2 feature
 * encoder's GroupNorm(C groups) + GELU over the time axis: frames = 1, hw = time steps, groups = C). */
int64_t vx_groupnorm_ws_floats(int frames, int slices, int groups);
int vx_groupnorm(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups, float eps,
                 const float* gamma, const float* beta, int silu, void* out, float* ws, int slices, int width,
                 int out_pad, void* stream);

/* GroupNorm WITHOUT activation folded into the 1x1 / linear layer that consumes it (round 3; Transformer3DModel.norm ->
 * proj_in, modules/transformer_3d.py:124-126, and the motion module's norm -> proj_in, modules/motion_module.py:156-158):
 * vx_groupnorm_stats is the statistics pass of vx_groupnorm alone (same workspace, same bits); vx_groupnorm_fold_linear
 * turns them into one scaled weight copy and one bias row per frame,
 *   w_out[f][n][c] = bf16(w[n][c] * gamma[c] * rstd[f][group(c)]),
 *   bias_out[f][n] = bias_beta[n] - sum_c float(w_out[f][n][c]) * mean[f][group(c)],   bias_beta = bias + w beta,
 * so that GN(x) w^T + bias == x w_out[f]^T + bias_out[f] for the pixels of frame f: the GEMM reads the raw tensor with
 * vx_gemm_params.w_group_rows = hw and rowbias = bias_out, and the normalised tensor is never written or re-read. */
int vx_groupnorm_stats(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups, float* ws,
                       int slices, void* stream);
/* the apply pass of vx_groupnorm alone, over statistics that already sit in ws as `stat_slices` partial sums per frame
 * (written by vx_groupnorm_stats, or by the GEMM that produced x1: vx_gemm_params.gn_ws / vx_gemm_gn_slabs).  `slices` =
 * pieces per frame of THIS pass (element-wise: any value gives the same bits). */
int vx_groupnorm_apply(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups, float eps,
                       const float* gamma, const float* beta, int silu, void* out, const float* ws, int stat_slices,
                       int slices, int width, int out_pad, void* stream);
int vx_groupnorm_fold_linear(const float* ws, int frames, int hw, int slices, int groups, float eps, const float* gamma,
                             int c, const void* w, const float* bias_beta, int n, void* w_out, float* bias_out,
                             void* stream);

/* ---- LayerNorm over the channel axis (+ optional additive table: motion-module positional encoding) -------
 * Replaces F.layer_norm (modules/attention.py:329-376 norms via mutual_self_attention.py:176-247;
 * modules/motion_module.py:228,234) and `x + pe[:, :f]` (modules/motion_module.py:262-277,365-366):
 * out[r, :] = LN(x[r, :]) * gamma + beta + (add ? add[(r / add_rows_per_entry) % add_entries, :] : 0). */
int vx_layernorm(const void* x, int ldx, int rows, int c, float eps, const float* gamma, const float* beta,
                 const float* add, int add_rows_per_entry, int add_entries, void* out, int ldo, void* stream);

/* Row statistics of a LayerNorm that is folded into the consuming GEMM (vx_gemm_params.ln_stats): stats[r] = (mean,
 * 1 / sqrt(var + eps)) of x[r, 0:c], two-pass variance in registers exactly as vx_layernorm computes it. */
int vx_row_stats(const void* x, int ldx, int rows, int c, float eps, float* stats, void* stream);
/* the two-part format of vx_gemm_params.row_stats_parts: stats[r][p] = (sum, sum of squares) of x[r, p * c / 2 : (p + 1) * c / 2] */
int vx_row_stats_parts(const void* x, int ldx, int rows, int c, float* stats, void* stream);
/* two-part sums [rows][2][2] -> (mean, rstd) [rows][2] with the arithmetic of the ln_stats_parts == 2 epilogue: for a consumer
 * launch that does not run on the persistent kernel (only that kernel finishes the two-part format itself) */
int vx_row_stats_finalize(const float* parts, int rows, int c, float eps, float* stats, void* stream);

/* LayerNorm (or, with gamma == NULL, no normalisation) whose result is quantised per row to OCP e4m3 for an fp8
 * projection GEMM: out8[r, 0:c] = e4m3(y[r, :] / scale[r]) with y as vx_layernorm computes it (incl. the additive
 * table), scale[r] = max|y[r, :]| / 448 (1 when the row is all zero); columns [c, ldo8) are written as zeros (the K
 * padding of the fp8 GEMM: ldo8 = c rounded up to 128).  The V-Express reference has no fp8 path (dtypes fp16 / bf16 /
 * fp32, inference.py:150-157); this is the quantiser in front of attn*.to_q / to_k / to_v / to_out when
 * UNet3DConditionModel.fp8_projections is on. */
int vx_layernorm_fp8(const void* x, int ldx, int rows, int c, float eps, const float* gamma, const float* beta,
                     const float* add, int add_rows_per_entry, int add_entries, void* out8, int ldo8, float* scale,
                     void* stream);

/* ---- Fused (flash-style) attention, softmax(q k^T * scale) v, per (batch, head) ----------------------------
 * Replaces F.scaled_dot_product_attention via diffusers AttnProcessor2_0 for attn1 / attn1_5
 * (modules/mutual_self_attention.py:177-224) and the sd-vae-ft-mse mid-block attention.
 * q: bf16 rows [batch*n_q] with row stride ldq, head h at column h*head_dim; k likewise (kv batch = batch / q_per_kv);
 * vt: bf16 [kv_batches, heads, head_dim, vt_pitch] (keys contiguous); out: bf16 rows, stride ldo.
 * scale = 0: K already carries scale * log2(e) (the model folds it into the to_k weights at load time, so neither
 * operand is rounded twice): softmax_j 2^(q . k_j), no multiply in the kernel. */
int vx_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out, int ldo,
                 int batch, int heads, int n_q, int n_kv, int head_dim, int q_per_kv, float scale, void* stream);

/* vx_attention with the bounded softmax shift (the 64x64 level's d = 40 kernel; other head dims ignore the table and
 * run vx_attention): m_i = scale |q_i| key_norm_max[kv batch, head] >= max_j s_ij replaces the running row max, which
 * removes the per-tile max / rescale work from a VALU-bound kernel.  Same result up to fp32 rounding; rows whose
 * probabilities would underflow under the looser shift are detected in the kernel and recomputed exactly.
 * key_norm_max: float32 [batch / q_per_kv, heads] from vx_key_norm_max on the same k. */
int vx_key_norm_max(const void* k, int ldk, int kv_batches, int heads, int n_kv, int head_dim, float* out,
                    void* stream);
int vx_attention_bounded(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out,
                         int ldo, int batch, int heads, int n_q, int n_kv, int head_dim, int q_per_kv, float scale,
                         const float* key_norm_max, void* stream);

/* ---- Temporal self-attention over the frame axis (one sequence per (batch row, pixel, head)) ---------------
 * Replaces VersatileAttention.forward (modules/motion_module.py:351-388) incl. both einops transposes:
 * tokens stay [(b f), hw, 3C] / [(b f), hw, C]; the kernel strides over f. */
int vx_temporal_attention(const void* qkv, int ldqkv, void* out, int ldo, int b, int f, int hw, int heads,
                          int head_dim, float scale, void* stream);

/* ---- Cross-attention against a short key/value list (audio tokens: 5 keys) --------------------------------
 * Replaces attn2 SDPA (modules/mutual_self_attention.py:227-244).  q: [batch*n_q, ldq]; kv: [batch, n_kv, ldkv] with
 * K at column 0 and V at column v_off; n_kv <= 16. */
int vx_small_kv_attention(const void* q, int ldq, const void* kv, int ldkv, int v_off, void* out, int ldo,
                          int batch, int n_q, int n_kv, int heads, int head_dim, float scale, void* stream);

/* ---- elementwise / layout ---------------------------------------------------------------------------------- */
/* x[r, :] += alpha * bias[:]  (uncond half of reference attention == to_out bias) */
int vx_add_row_bias(void* x, int ldx, int rows, int c, const float* bias, float alpha, void* stream);
/* out[r, :] = element(x[r, :] + y[r, :]) with y float32 (ABI 15): the last two steps of vx_gemm's STORE epilogue (add the
 * residual, round once) for an accumulator row that was written with out_f32 and moved between GPUs before the residual
 * it belongs to is at hand - the frame-sharded motion module's folded feed-forward output / proj_out GEMM runs in the
 * pixel-shard layout, its float32 rows go through the all-to-all, and the residual add happens in the frame-shard layout
 * with the bits of the unsharded launch. */
int vx_add_residual_f32(const void* x, int ldx, const float* y, int ldy, int rows, int c, void* out, int ldo, void* stream);
/* Nearest-2x upsampling + 3x3 convolution (Upsample3D, modules/resnet.py:53-90; diffusers Upsample2D of the VAE decoder) as
 * FOUR 2x2 convolutions over the ORIGINAL image (round 6, ABI 15): output pixel (2y + a, 2x + b) sees only a 2x2 neighbourhood
 * of the input, with the 3x3 taps that land on the same input pixel added up (weights.fold_upsample_phases: float32 sums, one
 * rounding) - 16 instead of 36 tap products per input pixel, 2.25x fewer FLOPs than the convolution over the upsampled image.
 * vx_pad_image copies x [frames, h, w, c] into the interior of a zero-bordered [frames, h + 2, w + 2, c] image (border kept zero
 * by the caller), the four phases run as pad-0 2x2 vx_gemm launches over it (A pointer advanced by (a (w + 2) + b) pixels,
 * h_out = h, w_out = w), vx_pixel_shuffle2x interleaves their outputs (phases[a * 2 + b] at phases + (a * 2 + b) * phase_stride
 * elements, each [frames, h, w, c]) into out [frames, 2h, 2w, c]. */
int vx_pad_image(const void* x, int frames, int h, int w, int c, void* out, void* stream);
int vx_pixel_shuffle2x(const void* phases, int64_t phase_stride, int frames, int h, int w, int c, void* out, void* stream);
/* gather window frames of the fp32 latent clip [1,C,F,h,w] into NHWC bf16 [reps*f, h*w, c_pad] (channels >= C zero).
 * Replaces latents[:, :, context].repeat(2,...) + rearrange (pipelines/v_express_pipeline.py:538-539, resnet.py:13). */
int vx_gather_latents(const float* latents, int c, int total_frames, int hw, const int32_t* frame_ids, int f,
                      int reps, int c_pad, void* out, void* stream);
/* CFG combine of one window: pred[w_slot, c, li, hw] = u + s * (c - u) from the conv_out result
 * [2f, hw, ld] float32 (rows: uncond frames then cond frames).  pipelines/v_express_pipeline.py:548-550. */
int vx_cfg_combine(const float* unet_out, int ld, int c, int f, int hw, float guidance, float* pred_slot,
                   void* stream);
/* Multi-GPU exchange of a timestep's predictions (one process per GPU).  vx_pack_rows: the first c
 * channels of the conv_out result float32 [rows, ld] densely packed into this rank's send slots [rows, c].
 * vx_combine_units: the all-gathered buffer float32 [units_total, (f / shards) * hw, c] -> CFG-combined predictions
 * float32 [n_windows, c, f, hw] = u + s (c - u) for every window in one launch; unit_index: int32
 * [n_windows][halves][shards] = which unit buffer holds frame shard j of (window, CFG half); halves == 1 (no
 * classifier-free guidance) copies the prediction.  pipelines/v_express_pipeline.py:548-550. */
int vx_pack_rows(const float* src, int ld, int64_t rows, int c, float* dst, void* stream);
int vx_combine_units(const float* gathered, const int32_t* unit_index, int n_windows, int halves, int shards, int c,
                     int f, int hw, float guidance, float* preds, void* stream);
/* per-frame mean-overlap + DDIM v-prediction step (eta=0):  v = sum_t (pred[term_slot[t]] / count) ;
 * latents[:, :, frame] = step(v).  terms: int32 [n_frames][max_terms][2] = (window slot, latent idx) or -1.
 * pipelines/v_express_pipeline.py:552-572 + diffusers DDIMScheduler.step.  */
int vx_overlap_ddim_step(float* latents, int c, int total_frames, int hw, const float* preds, int f_window,
                         const int32_t* terms, int max_terms, const int32_t* frame_ids, const float* count,
                         int n_frames, float sqrt_a, float sqrt_1ma, float sqrt_ap, float sqrt_1map, void* stream);
/* NCHW-ish float32 [b, C, f, h, w] -> NHWC bf16 [(b f), h*w, c_pad]  (API-boundary layout change) */
int vx_ncfhw_to_nhwc(const float* x, int b, int c, int f, int hw, int c_pad, void* out, void* stream);
/* NHWC float32 [(b f), hw, ld] -> [b, C, f, h*w] float32 */
int vx_nhwc_to_ncfhw(const float* x, int ld, int b, int c, int f, int hw, float* out, void* stream);
/* VAE post-process: NHWC float32 [n, hw, ld] -> clamp(x/2+0.5, 0, 1) as [n, 3, hw] float32
 * (pipelines/v_express_pipeline.py:160). */
int vx_vae_postprocess(const float* x, int ld, int n, int c, int hw, float* out, void* stream);
/* 3x3x3 median over (frame, y, x) with reflect padding + optional uint8 packing of the result.
 * Replaces pipelines/utils.py:46-61 (median_filter_3d, kernel_size 3) and the (video*255).astype(uint8) + HWC
 * permute of save_video (:70-73).  video: float32 [c, f, h, w]; out_f32: float32 [c, f, h, w] or NULL;
 * out_u8: uint8 [f, h, w, c] or NULL. */
int vx_median3d(const float* video, int c, int f, int h, int w, float* out_f32, void* out_u8, void* stream);
/* First wav2vec2 feature-encoder convolution on the raw waveform: out[t, co] = sum_j wt[j, co] * wave[t*stride + j]
 * (1 input channel, no padding, no bias; t < (samples - taps) / stride + 1).  Replaces the first nn.Conv1d of
 * transformers' Wav2Vec2FeatureEncoder reached from pipelines/v_express_pipeline.py:377 (self.audio_encoder(...)).
 * wave: float32 [samples]; wt: float32 [taps, c] (the Conv1d weight [c, 1, taps] transposed), taps <= 16, c % 8 == 0;
 * out: bf16 [t_out, c].  The remaining conv1d layers run on vx_gemm: in the time-major layout a k-tap stride-s window
 * is k*C contiguous elements, i.e. a plain GEMM over overlapping rows (lda1 = s*C < c1 = k*C). */
int vx_wave_conv1d(const float* wave, int samples, const float* wt, int c, int taps, int stride, void* out,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif
