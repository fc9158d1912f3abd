/* C interface of tools/conv3/libvx_conv3.so - round 5's one-pass GroupNorm + SiLU + 3x3 convolution (vx_conv3.hip).
 * It was part of the shipped ABI (13) in round 5; it is correct and 0.8 % SLOWER than the two launches it replaces
 * (LABNOTES.md 12.3), so since round 6 it lives here as a tool - with its emulator (conv3_emulate.py), schedule checker
 * (conv3_schedule_check.py) and bench (conv3_bench.py) - and the product library no longer carries it.
 * Build: tools/conv3/build_conv3_variants.sh (links against v-express_amd/libvexpress_hip.so for vx_set_error & co). */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- 3x3 convolution with the GroupNorm + SiLU in front of it applied in its A-staging path (round 5) ------------
 * out[f, oy, ox, n] = residual + bias[n] + rowbias[m / rows_per_group][n]
 *                     + sum_{ky, kx, c} act(x[f, oy + ky - 1, ox + kx - 1, c]) w[n][ky][kx][c],
 * act(v) = silu(v * scale[f][c] + shift[f][c]) inside the image (silu = 0: no activation), 0 outside (padding 1).
 * Replaces InflatedGroupNorm -> SiLU -> InflatedConv3d of ResnetBlock3D (modules/resnet.py:220-223 norm1 / conv1,
 * :235-244 norm2 / conv2, incl. the skip concat of the up blocks as the dual source x1 | x2): the normalised, zero-bordered
 * copy of the tensor that vx_groupnorm_apply + vx_gemm went through is never written; a tile's activations cross the
 * CU's L1 once per 32-channel chunk instead of nine times (haloed tile resident in LDS, normalised in place).
 *   x1 / x2   RAW (un-normalised) bf16 NHWC inputs [frames, h, w, c1 | c2], pixel strides ldx1 / ldx2 (elements)
 *   ab        float32 [frames][ab_ld][2] = (scale, shift) per frame and (concatenated) channel: vx_groupnorm_scale_shift
 *             (ab_ld must be 1024: the kernel stages one 8 KiB row per tile)
 *   w_perm    bf16 [n][9 (c1 + c2)], the conv weight [n][ky][kx][c] with K re-ordered to (32-channel chunk, tap, 32):
 *             w_perm[n][(chunk * 9 + tap) * 32 + i] = w[n][tap][chunk * 32 + i]
 *   gn_ws     NULL, or GroupNorm partial sums of the STORED output for the next GroupNorm, exactly as
 *             vx_gemm_params.gn_ws ([frame][gn_hw / 128 slabs][gn_groups] (sum, sum of squares))
 * Supported (vx_conv3x3_gn_supported; VX_ERR_UNSUPPORTED otherwise): w = 64 or 32, h * w a multiple of 256, c1, c2
 * multiples of 32 with c1 + c2 a multiple of 64 and <= 1024, n a multiple of 320, bf16 output.  Deterministic; every
 * output element's summation order is a function of the per-frame geometry only (batch-invariant). */
typedef struct {
  const void* x1;
  const void* x2;            /* or NULL (c2 = 0) */
  int32_t c1, c2, ldx1, ldx2;
  int32_t frames, h, w;
  const void* w_perm;
  int32_t n;
  const float* ab;
  int32_t ab_ld;
  int32_t silu;
  const float* bias;         /* [n] or NULL */
  const float* rowbias;      /* [m / rows_per_group][rowbias_ld] or NULL (time-embedding row of the frame's batch item) */
  int32_t rowbias_ld, rows_per_group;
  const void* residual;      /* bf16 [m, ldr] or NULL */
  int32_t ldr;
  void* out;                 /* bf16 [m, ldc], m = frames * h * w */
  int32_t ldc;
  float* gn_ws;
  int32_t gn_groups, gn_hw;
} vx_conv3_params;
int vx_conv3x3_gn_supported(const vx_conv3_params* p);   /* 1 / 0 */
int vx_conv3x3_gn(const vx_conv3_params* p, void* stream);
/* (scale, shift) = (gamma rstd, beta - mean gamma rstd) per frame and channel from GroupNorm partial sums (`stat_slices`
 * per frame, as written by vx_groupnorm_stats or a producer's gn_ws): what the apply pass of vx_groupnorm builds in LDS,
 * same re-reduction order (float64).  ab: float32 [frames][ab_ld][2], entries c .. ab_ld - 1 zero. */
int vx_groupnorm_scale_shift(const float* ws, int stat_slices, int frames, int hw, int groups, float eps,
                             const float* gamma, const float* beta, int c, float* ab, int ab_ld, void* stream);


#ifdef __cplusplus
}
#endif
