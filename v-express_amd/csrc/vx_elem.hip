// Elementwise / layout kernels of the denoising loop (see include/vexpress_hip.h).  All are tiny next to the
// UNet; they exist so that latents, predictions and the DDIM state never leave HBM (the reference round-trips
// them through the host every window: pipelines/v_express_pipeline.py:521,538,572).
#include "vx_common.h"
#include "../../include/vexpress_hip.h"

namespace {

__global__ void add_row_bias_kernel(bf16_t* x, int ldx, int rows, int c, const float* bias, float alpha) {
  const int cch = c >> 3;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cch) return;
  int row = (int)(idx / cch), ch = (int)(idx % cch) * 8;
  bf16_t* p = x + (size_t)row * ldx + ch;
  float f[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(p), f);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] += alpha * bias[ch + e];
  *reinterpret_cast<uint4*>(p) = pack_bf16x8(f);
}

// out[r, :] = element(x[r, :] + y[r, :]), y float32: the last two steps of vx_gemm's STORE epilogue (v += residual; round)
// on an accumulator row that travelled as float32 (frame-sharded motion module: blocks._motion_module)
__global__ void add_residual_f32_kernel(const bf16_t* x, int ldx, const float* y, int ldy, int rows, int c, bf16_t* out, int ldo) {
  const int cch = c >> 3;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cch) return;
  int row = (int)(idx / cch), ch = (int)(idx % cch) * 8;
  float f[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (size_t)row * ldx + ch), f);
  const float4 a = *reinterpret_cast<const float4*>(y + (size_t)row * ldy + ch);
  const float4 b = *reinterpret_cast<const float4*>(y + (size_t)row * ldy + ch + 4);
  f[0] = a.x + f[0]; f[1] = a.y + f[1]; f[2] = a.z + f[2]; f[3] = a.w + f[3];
  f[4] = b.x + f[4]; f[5] = b.y + f[5]; f[6] = b.z + f[6]; f[7] = b.w + f[7];
  *reinterpret_cast<uint4*>(out + (size_t)row * ldo + ch) = pack_bf16x8(f);
}

// x [frames, H, W, C] -> the interior of the zero-bordered image out [frames, H + 2, W + 2, C] (the border is the caller's to
// keep zero); one 16-byte chunk per thread
__global__ void pad_image_kernel(const bf16_t* __restrict__ x, int H, int W, int c8, bf16_t* __restrict__ out, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % c8);
  long px = idx / c8;
  const int xx = (int)(px % W);
  px /= W;
  const int yy = (int)(px % H);
  const long f = px / H;
  const long dst = ((f * (H + 2) + yy + 1) * (W + 2) + xx + 1) * c8 + ch;
  reinterpret_cast<uint4*>(out)[dst] = reinterpret_cast<const uint4*>(x)[idx];
}

// the four phase images ph[a][b] [frames, H, W, C] (phase_stride 16-byte chunks apart) -> out [frames, 2H, 2W, C] with
// out[f, 2y + a, 2x + b] = ph[a][b][f, y, x]; indexed by the OUTPUT chunk (coalesced stores; a pixel's C channels are one
// contiguous run on both sides)
__global__ void pixel_shuffle2_kernel(const bf16_t* __restrict__ ph, long phase_stride, int H, int W, int c8,
                                      bf16_t* __restrict__ out, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ch = (int)(idx % c8);
  long px = idx / c8;
  const int ox = (int)(px % (2 * W));
  px /= 2 * W;
  const int oy = (int)(px % (2 * H));
  const long f = px / (2 * H);
  const long src = (long)((oy & 1) * 2 + (ox & 1)) * phase_stride + ((f * H + (oy >> 1)) * W + (ox >> 1)) * c8 + ch;
  reinterpret_cast<uint4*>(out)[idx] = reinterpret_cast<const uint4*>(ph)[src];
}

// latents fp32 [1, C, F, hw] -> out bf16 [reps*f, hw, c_pad]
__global__ void gather_latents_kernel(const float* latents, int c, int total_frames, int hw, const int32_t* frame_ids,
                                      int f, int reps, int c_pad, bf16_t* out) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (rep, li, pixel)
  long total = (long)reps * f * hw;
  if (idx >= total) return;
  int px = (int)(idx % hw);
  int li = (int)((idx / hw) % f);
  int fr = frame_ids[li];
  bf16_t* o = out + idx * c_pad;
  for (int ch = 0; ch < c_pad; ++ch) {
    float v = ch < c ? latents[((size_t)ch * total_frames + fr) * hw + px] : 0.f;
    o[ch] = f32_to_bf16(v);
  }
}

// unet_out fp32 [2f, hw, ld] -> pred_slot fp32 [c, f, hw]
__global__ void cfg_combine_kernel(const float* unet_out, int ld, int c, int f, int hw, float guidance,
                                   float* pred_slot) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (li, pixel)
  if (idx >= (long)f * hw) return;
  const float* u = unet_out + idx * ld;
  const float* cnd = unet_out + ((long)f * hw + idx) * ld;
  for (int ch = 0; ch < c; ++ch) pred_slot[(size_t)ch * f * hw + idx] = u[ch] + guidance * (cnd[ch] - u[ch]);
}

// conv_out result fp32 [rows, ld] -> the first c channels densely packed [rows, c]: what one rank contributes to the
// per-timestep exchange (the GEMM pads conv_out's 4 output channels to 8; only 4 travel)
__global__ void pack_rows_kernel(const float* __restrict__ src, int ld, long rows, int c, float* __restrict__ dst) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (row, ch)
  if (idx >= rows * c) return;
  long r = idx / c;
  int ch = (int)(idx - r * c);
  dst[idx] = src[r * ld + ch];
}

// The all-gathered unit predictions [units_total, f_loc*hw, c] -> CFG-combined window predictions [nW, c, f, hw] in ONE
// launch (replaces the per-slot copies + one cfg_combine per window).  unit_index: int32 [nW][halves][S] = index of the
// unit buffer holding frame shard j of (window, CFG half).  halves == 1: the prediction itself (no guidance, :548-550
// skipped: u + 1 * (u - u) is exactly u).
__global__ void combine_units_kernel(const float* __restrict__ gathered, const int32_t* __restrict__ unit_index,
                                     int n_windows, int halves, int shards, int c, int f, int f_loc, int hw,
                                     float guidance, float* __restrict__ preds) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (window, li, pixel)
  long total = (long)n_windows * f * hw;
  if (idx >= total) return;
  int px = (int)(idx % hw);
  int li = (int)((idx / hw) % f);
  int wi = (int)(idx / ((long)hw * f));
  int j = li / f_loc;
  long row = (long)(li - j * f_loc) * hw + px;
  long unit_sz = (long)f_loc * hw * c;
  const float* u = gathered + unit_index[(wi * halves + 0) * shards + j] * unit_sz + row * c;
  const float* cnd = halves > 1 ? gathered + unit_index[(wi * halves + 1) * shards + j] * unit_sz + row * c : u;
  for (int ch = 0; ch < c; ++ch)
    preds[(((size_t)wi * c + ch) * f + li) * hw + px] = u[ch] + guidance * (cnd[ch] - u[ch]);
}

__global__ void overlap_ddim_kernel(float* latents, int c, int total_frames, int hw, const float* preds, int f_window,
                                    const int32_t* terms, int max_terms, const int32_t* frame_ids,
                                    const float* count, int n_frames, float sqrt_a, float sqrt_1ma,
                                    float sqrt_ap, float sqrt_1map) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (frame slot, channel, pixel)
  long total = (long)n_frames * c * hw;
  if (idx >= total) return;
  int px = (int)(idx % hw);
  int ch = (int)((idx / hw) % c);
  int fs = (int)(idx / ((long)hw * c));
  int fr = frame_ids[fs];
  float ic = count[fs];
  float v = 0.f;
  bool first = true;
  for (int t = 0; t < max_terms; ++t) {
    int slot = terms[(fs * max_terms + t) * 2 + 0];
    int li = terms[(fs * max_terms + t) * 2 + 1];
    if (slot < 0) continue;
    // the reference divides each window's prediction by the coverage count BEFORE summing (:553, :556-564)
    float term = preds[(((size_t)slot * c + ch) * f_window + li) * hw + px] / ic;
    v = first ? term : v + term;
    first = false;
  }
  float* lp = latents + ((size_t)ch * total_frames + fr) * hw + px;
  float x = *lp;
  float x0 = sqrt_a * x - sqrt_1ma * v;
  float eps = sqrt_a * v + sqrt_1ma * x;
  *lp = sqrt_ap * x0 + sqrt_1map * eps;
}

// x fp32 [b, c, f, hw] -> out bf16 [(b f), hw, c_pad]; LDS transpose so both sides are coalesced for wide c
__global__ void ncfhw_to_nhwc_kernel(const float* x, int b, int c, int f, int hw, int c_pad, bf16_t* out) {
  __shared__ float tile[32][33];
  // grid: x = pixel tiles of 32, y = channel tiles of 32, z = (b f)
  int bf = blockIdx.z;
  int bb = bf / f, fr = bf % f;
  int px0 = blockIdx.x * 32, ch0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    int ch = ch0 + r, px = px0 + tx;
    float v = 0.f;
    if (ch < c && px < hw) v = x[(((size_t)bb * c + ch) * f + fr) * hw + px];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int px = px0 + r, ch = ch0 + tx;
    if (px < hw && ch < c_pad) out[((size_t)bf * hw + px) * c_pad + ch] = f32_to_bf16(tile[tx][r]);
  }
}

__global__ void nhwc_to_ncfhw_kernel(const float* x, int ld, int b, int c, int f, int hw, float* out) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, c, f, px)
  long total = (long)b * c * f * hw;
  if (idx >= total) return;
  int px = (int)(idx % hw);
  int fr = (int)((idx / hw) % f);
  int ch = (int)((idx / ((long)hw * f)) % c);
  int bb = (int)(idx / ((long)hw * f * c));
  out[idx] = x[((size_t)(bb * f + fr) * hw + px) * ld + ch];
}

__global__ void vae_post_kernel(const float* x, int ld, int n, int c, int hw, float* out) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (n, px)
  if (idx >= (long)n * hw) return;
  int px = (int)(idx % hw);
  int fr = (int)(idx / hw);
  const float* src = x + idx * ld;
  for (int ch = 0; ch < c; ++ch) {
    float v = src[ch] * 0.5f + 0.5f;
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[((size_t)fr * c + ch) * hw + px] = v;
  }
}

inline dim3 grid1d(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

// 3x3x3 median over (frame, y, x) with reflect padding in all three axes (pipelines/utils.py:46-61: func.pad(...,
// mode='reflect') + unfold + torch.median over the 27 window values).  One thread per output pixel and channel; the 27
// values are partially selection-sorted in registers with min/max exchanges (the median is the 14th smallest: a pure
// selection, so the result is bit-identical to the reference for any input without NaNs).
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void median3d_kernel(const float* __restrict__ x, int c, int f, int h, int w,
                                                       float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
  const long total = (long)c * f * h * w;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xx = (int)(idx % w);
  const int yy = (int)((idx / w) % h);
  const int ff = (int)((idx / ((long)w * h)) % f);
  const int cc = (int)(idx / ((long)w * h * f));
  const float* xc = x + (size_t)cc * f * h * w;
  float v[27];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
    const float* xf = xc + (size_t)reflect_idx(ff + dt - 1, f) * h * w;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float* xr = xf + (size_t)reflect_idx(yy + dy - 1, h) * w;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) v[(dt * 3 + dy) * 3 + dx] = xr[reflect_idx(xx + dx - 1, w)];
    }
  }
  // selection: after pass i, v[i] is the (i+1)-th smallest
#pragma unroll
  for (int i = 0; i < 14; ++i) {
#pragma unroll
    for (int j = i + 1; j < 27; ++j) {
      const float lo = fminf(v[i], v[j]), hi = fmaxf(v[i], v[j]);
      v[i] = lo;
      v[j] = hi;
    }
  }
  const float med = v[13];
  if (out_f32 != nullptr) out_f32[idx] = med;
  if (out_u8 != nullptr)   // (video * 255).astype(uint8): fp32 product, truncation (pipelines/utils.py:72-73)
    out_u8[(((size_t)ff * h + yy) * w + xx) * c + cc] = (uint8_t)(unsigned)(med * 255.0f);
}

// First layer of the wav2vec2 feature encoder: conv1d over the raw float32 waveform (1 input channel, `taps` taps,
// stride `stride`, no padding, no bias) -> bf16 time-major tokens [t_out, c].  One thread = one time step x 8 output
// channels: the <= 16 waveform samples of the step are held in registers (every thread of a step reads the same ones:
// a broadcast from L1), the 8 x taps weights come from the transposed weight table wt[taps][c] so that a wave's loads
// are contiguous, fp32 FMA chain in tap order, one 16-B store.
constexpr int WAVE_CONV_MAX_TAPS = 16;
__global__ __launch_bounds__(256) void wave_conv1d_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                          bf16_t* __restrict__ out, int t_out, int c, int taps,
                                                          int stride) {
  const int chunks = c >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)t_out * chunks) return;
  const int t = (int)(idx / chunks), ch = (int)(idx % chunks) * 8;
  const float* xs = x + (size_t)t * stride;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int j = 0; j < taps; ++j) {
    const float xv = xs[j];
    const float4 w0 = *reinterpret_cast<const float4*>(wt + (size_t)j * c + ch);
    const float4 w1 = *reinterpret_cast<const float4*>(wt + (size_t)j * c + ch + 4);
    acc[0] = fmaf(w0.x, xv, acc[0]); acc[1] = fmaf(w0.y, xv, acc[1]);
    acc[2] = fmaf(w0.z, xv, acc[2]); acc[3] = fmaf(w0.w, xv, acc[3]);
    acc[4] = fmaf(w1.x, xv, acc[4]); acc[5] = fmaf(w1.y, xv, acc[5]);
    acc[6] = fmaf(w1.z, xv, acc[6]); acc[7] = fmaf(w1.w, xv, acc[7]);
  }
  *reinterpret_cast<uint4*>(out + (size_t)t * c + ch) = pack_bf16x8(acc);
}

extern "C" int vx_add_row_bias(void* x, int ldx, int rows, int c, const float* bias, float alpha, void* stream) {
  VX_REQUIRE(x && bias && rows > 0 && (c % 8) == 0 && (ldx % 8) == 0, "vx_add_row_bias: bad arguments");
  hipLaunchKernelGGL(add_row_bias_kernel, grid1d((long)rows * (c / 8)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)x, ldx, rows, c, bias, alpha);
  return vx_check_launch("vx_add_row_bias");
}

extern "C" int vx_add_residual_f32(const void* x, int ldx, const float* y, int ldy, int rows, int c, void* out, int ldo,
                                   void* stream) {
  VX_REQUIRE(x && y && out && rows > 0 && c > 0 && (c % 8) == 0 && (ldx % 8) == 0 && (ldy % 4) == 0 && (ldo % 8) == 0,
             "vx_add_residual_f32: bad arguments");
  hipLaunchKernelGGL(add_residual_f32_kernel, grid1d((long)rows * (c / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, y, ldy, rows, c, (bf16_t*)out, ldo);
  return vx_check_launch("vx_add_residual_f32");
}

extern "C" int vx_pad_image(const void* x, int frames, int h, int w, int c, void* out, void* stream) {
  VX_REQUIRE(x && out && frames > 0 && h > 0 && w > 0 && c > 0 && (c % 8) == 0, "vx_pad_image: bad arguments");
  const long total = (long)frames * h * w * (c / 8);
  hipLaunchKernelGGL(pad_image_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, h, w, c / 8,
                     (bf16_t*)out, total);
  return vx_check_launch("vx_pad_image");
}

extern "C" int vx_pixel_shuffle2x(const void* phases, int64_t phase_stride, int frames, int h, int w, int c, void* out,
                                  void* stream) {
  VX_REQUIRE(phases && out && frames > 0 && h > 0 && w > 0 && c > 0 && (c % 8) == 0 && (phase_stride % 8) == 0 &&
                 phase_stride >= (int64_t)frames * h * w * c, "vx_pixel_shuffle2x: bad arguments");
  const long total = (long)frames * 4 * h * w * (c / 8);
  hipLaunchKernelGGL(pixel_shuffle2_kernel, grid1d(total), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)phases,
                     (long)(phase_stride / 8), h, w, c / 8, (bf16_t*)out, total);
  return vx_check_launch("vx_pixel_shuffle2x");
}

extern "C" int vx_gather_latents(const float* latents, int c, int total_frames, int hw, const int32_t* frame_ids,
                                 int f, int reps, int c_pad, void* out, void* stream) {
  VX_REQUIRE(latents && frame_ids && out && c_pad >= c && (c_pad % 8) == 0 && f > 0 && reps > 0,
             "vx_gather_latents: bad arguments");
  hipLaunchKernelGGL(gather_latents_kernel, grid1d((long)reps * f * hw), dim3(256), 0, (hipStream_t)stream, latents,
                     c, total_frames, hw, frame_ids, f, reps, c_pad, (bf16_t*)out);
  return vx_check_launch("vx_gather_latents");
}

extern "C" int vx_cfg_combine(const float* unet_out, int ld, int c, int f, int hw, float guidance, float* pred_slot,
                              void* stream) {
  VX_REQUIRE(unet_out && pred_slot && ld >= c, "vx_cfg_combine: bad arguments");
  hipLaunchKernelGGL(cfg_combine_kernel, grid1d((long)f * hw), dim3(256), 0, (hipStream_t)stream, unet_out, ld, c, f,
                     hw, guidance, pred_slot);
  return vx_check_launch("vx_cfg_combine");
}

extern "C" int vx_pack_rows(const float* src, int ld, int64_t rows, int c, float* dst, void* stream) {
  VX_REQUIRE(src && dst && rows > 0 && c > 0 && ld >= c, "vx_pack_rows: bad arguments");
  hipLaunchKernelGGL(pack_rows_kernel, grid1d((long)rows * c), dim3(256), 0, (hipStream_t)stream, src, ld, (long)rows,
                     c, dst);
  return vx_check_launch("vx_pack_rows");
}

extern "C" int vx_combine_units(const float* gathered, const int32_t* unit_index, int n_windows, int halves,
                                int shards, int c, int f, int hw, float guidance, float* preds, void* stream) {
  VX_REQUIRE(gathered && unit_index && preds && n_windows > 0 && (halves == 1 || halves == 2) && shards > 0 && c > 0 &&
                 f > 0 && hw > 0 && f % shards == 0,
             "vx_combine_units: bad arguments");
  hipLaunchKernelGGL(combine_units_kernel, grid1d((long)n_windows * f * hw), dim3(256), 0, (hipStream_t)stream,
                     gathered, unit_index, n_windows, halves, shards, c, f, f / shards, hw, guidance, preds);
  return vx_check_launch("vx_combine_units");
}

extern "C" int vx_overlap_ddim_step(float* latents, int c, int total_frames, int hw, const float* preds, int f_window,
                                    const int32_t* terms, int max_terms, const int32_t* frame_ids,
                                    const float* count, int n_frames, float sqrt_a, float sqrt_1ma, float sqrt_ap,
                                    float sqrt_1map, void* stream) {
  VX_REQUIRE(latents && preds && terms && frame_ids && count && n_frames > 0 && max_terms > 0,
             "vx_overlap_ddim_step: bad arguments");
  hipLaunchKernelGGL(overlap_ddim_kernel, grid1d((long)n_frames * c * hw), dim3(256), 0, (hipStream_t)stream, latents,
                     c, total_frames, hw, preds, f_window, terms, max_terms, frame_ids, count, n_frames, sqrt_a,
                     sqrt_1ma, sqrt_ap, sqrt_1map);
  return vx_check_launch("vx_overlap_ddim_step");
}

extern "C" int vx_ncfhw_to_nhwc(const float* x, int b, int c, int f, int hw, int c_pad, void* out, void* stream) {
  VX_REQUIRE(x && out && c_pad >= c && (long)b * f <= 65535, "vx_ncfhw_to_nhwc: bad arguments");
  dim3 grid((hw + 31) / 32, (c_pad + 31) / 32, b * f);
  hipLaunchKernelGGL(ncfhw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, b, c, f, hw, c_pad,
                     (bf16_t*)out);
  return vx_check_launch("vx_ncfhw_to_nhwc");
}

extern "C" int vx_nhwc_to_ncfhw(const float* x, int ld, int b, int c, int f, int hw, float* out, void* stream) {
  VX_REQUIRE(x && out && ld >= c, "vx_nhwc_to_ncfhw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_ncfhw_kernel, grid1d((long)b * c * f * hw), dim3(256), 0, (hipStream_t)stream, x, ld, b,
                     c, f, hw, out);
  return vx_check_launch("vx_nhwc_to_ncfhw");
}

extern "C" int vx_vae_postprocess(const float* x, int ld, int n, int c, int hw, float* out, void* stream) {
  VX_REQUIRE(x && out && ld >= c, "vx_vae_postprocess: bad arguments");
  hipLaunchKernelGGL(vae_post_kernel, grid1d((long)n * hw), dim3(256), 0, (hipStream_t)stream, x, ld, n, c, hw, out);
  return vx_check_launch("vx_vae_postprocess");
}

extern "C" int vx_median3d(const float* video, int c, int f, int h, int w, float* out_f32, void* out_u8, void* stream) {
  VX_REQUIRE(video && (out_f32 || out_u8), "vx_median3d: null pointer");
  VX_REQUIRE(c >= 1 && f >= 2 && h >= 2 && w >= 2, "vx_median3d: reflect padding needs f, h, w >= 2 (got %d %d %d)", f, h, w);
  hipLaunchKernelGGL(median3d_kernel, grid1d((long)c * f * h * w), dim3(256), 0, (hipStream_t)stream, video, c, f, h, w,
                     out_f32, (uint8_t*)out_u8);
  return vx_check_launch("vx_median3d");
}

extern "C" int vx_wave_conv1d(const float* wave, int samples, const float* wt, int c, int taps, int stride, void* out,
                              void* stream) {
  VX_REQUIRE(wave && wt && out, "vx_wave_conv1d: null pointer");
  VX_REQUIRE(c > 0 && (c % 8) == 0 && taps >= 1 && taps <= WAVE_CONV_MAX_TAPS && stride >= 1 && samples >= taps,
             "vx_wave_conv1d: bad geometry (c=%d taps=%d stride=%d samples=%d)", c, taps, stride, samples);
  const int t_out = (samples - taps) / stride + 1;
  hipLaunchKernelGGL(wave_conv1d_kernel, grid1d((long)t_out * (c / 8)), dim3(256), 0, (hipStream_t)stream, wave, wt,
                     (bf16_t*)out, t_out, c, taps, stride);
  return vx_check_launch("vx_wave_conv1d");
}
