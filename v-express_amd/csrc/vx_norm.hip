// GroupNorm(+SiLU) and LayerNorm(+additive table) for gfx950 — HBM-bound kernels (see include/vexpress_hip.h).
//
// GroupNorm works on channels-last frames [frames, hw, C] (optionally the channel concat of two sources, which
// is how the up-block skip `torch.cat` is consumed without ever being materialised).  Statistics are per frame
// and group.  Two launches, both deterministic (no atomics, fixed summation order):
//   1. gn_stats : grid (frames x slices).  Each thread owns fixed 16-B channel chunks and strides over the
//                 pixels of its slice with 16-B loads, keeping 8 sums + 8 sums-of-squares in registers; lanes that
//                 share a channel are reduced through LDS in fixed order, then channels -> groups; one
//                 (sum, sumsq) pair per (frame, slice, group) goes to the fp32 workspace.
//   2. gn_apply : every block re-reduces its frame's `slices` partials in fp64, builds per-channel
//                 scale/shift in LDS and streams its slice: y = x*scale + shift (+SiLU), 16-B loads/stores.
// Algorithmic traffic: stats read 2 B/elem, apply read 2 + write 2 B/elem (the second read mostly hits the
// 256 MB infinity cache for UNet-sized tensors).
//
// LayerNorm: one wave per row, the whole row lives in registers (C <= 64*8*MAXC), two-pass mean/variance via
// wave shuffles, fused affine and the motion module's additive sinusoid table.
#include "vx_common.h"
#include "vx_gemm_common.h"
#include "../../include/vexpress_hip.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SETS = 2;   // chunks per thread per pixel -> C <= 8*256*2 = 4096

// Per-thread plan shared by both kernels: thread -> (pixel lane pl, fixed 16-B channel chunks cc + u*GN_THREADS).
// For every owned chunk the source pointer (x1 or x2 of the concat) and its pixel stride are resolved once, so the
// pixel loops contain no division and no source select, and are unrolled GN_UNROLL deep to keep that many 16-B
// loads per thread in flight.
constexpr int GN_UNROLL = 4;
constexpr int GN_SUBS = 8;      // parallel sub-sums of the per-slice partials in gn_apply

struct GnPlan {
  const bf16_t* base[GN_MAX_SETS];   // first pixel of the frame, at this thread's chunk (nullptr: no chunk)
  int pstride[GN_MAX_SETS];          // elements between consecutive pixels
  int chunk[GN_MAX_SETS];
  int pl, pl_count;
};

__device__ __forceinline__ GnPlan gn_plan(const bf16_t* x1, int c1, const bf16_t* x2, int c2, int hw, int frame) {
  GnPlan P;
  const int nchunks = (c1 + c2) >> 3;
  const int tid = threadIdx.x;
  const int tp = nchunks < GN_THREADS ? nchunks : GN_THREADS;
  P.pl_count = GN_THREADS / tp;
  const int cc = tid % tp;
  P.pl = tid / tp;
  const bool active = P.pl < P.pl_count;
#pragma unroll
  for (int u = 0; u < GN_MAX_SETS; ++u) {
    const int chunk = cc + u * GN_THREADS;
    P.chunk[u] = chunk;
    const int ch = chunk * 8;
    if (active && chunk < nchunks) {
      if (ch < c1) {
        P.base[u] = x1 + (size_t)frame * hw * c1 + ch;
        P.pstride[u] = c1;
      } else {
        P.base[u] = x2 + (size_t)frame * hw * c2 + (ch - c1);
        P.pstride[u] = c2;
      }
    } else {
      P.base[u] = nullptr;
      P.pstride[u] = 0;
    }
  }
  return P;
}

__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const bf16_t* __restrict__ x1, int c1,
                                                              const bf16_t* __restrict__ x2, int c2, int hw,
                                                              int groups, int slices, int slice_pix,
                                                              float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = c1 + c2;
  // (frames, slices) in XCD-contiguous order: XCD x reads / writes the x-th eighth of the frames, as the GEMM and attention
  // kernels that produce and consume these tensors do (vx_gemm_ring.hip, VX_XCD_ROWS)
  const int bid = VX_XCD_ROWS ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int frame = bid / slices, slice = bid % slices;
  const int tid = threadIdx.x;
  const GnPlan P = gn_plan(x1, c1, x2, c2, hw, frame);
  const int p_begin = slice * slice_pix;
  const int p_end = min(hw, p_begin + slice_pix);

  float s[GN_MAX_SETS][8], q[GN_MAX_SETS][8];
#pragma unroll
  for (int u = 0; u < GN_MAX_SETS; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[u][e] = q[u][e] = 0.f;

#pragma unroll
  for (int u = 0; u < GN_MAX_SETS; ++u) {
    if (P.base[u] == nullptr) continue;
    const bf16_t* src = P.base[u];
    const int ps = P.pstride[u];
    int px = p_begin + P.pl;
    // fixed summation order per thread: pixels ascending (the unrolled body adds in the same order)
    for (; px + (GN_UNROLL - 1) * P.pl_count < p_end; px += GN_UNROLL * P.pl_count) {
      uint4 raw[GN_UNROLL];
#pragma unroll
      for (int k = 0; k < GN_UNROLL; ++k)
        raw[k] = *reinterpret_cast<const uint4*>(src + (size_t)(px + k * P.pl_count) * ps);
#pragma unroll
      for (int k = 0; k < GN_UNROLL; ++k) {
        float f[8];
        unpack_bf16x8(raw[k], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[u][e] += f[e];
          q[u][e] += f[e] * f[e];
        }
      }
    }
    for (; px < p_end; px += P.pl_count) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(src + (size_t)px * ps), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[u][e] += f[e];
        q[u][e] += f[e] * f[e];
      }
    }
  }
  // per-channel partials: [pl_count][C][2]
  float* part = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int u = 0; u < GN_MAX_SETS; ++u) {
    if (P.base[u] == nullptr) continue;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      part[((size_t)P.pl * C + P.chunk[u] * 8 + e) * 2 + 0] = s[u][e];
      part[((size_t)P.pl * C + P.chunk[u] * 8 + e) * 2 + 1] = q[u][e];
    }
  }
  __syncthreads();
  // channel totals (fixed order over pixel lanes), written back into lane 0's slot
  for (int ch = tid; ch < C; ch += GN_THREADS) {
    float a = 0.f, b = 0.f;
    for (int l = 0; l < P.pl_count; ++l) {
      a += part[((size_t)l * C + ch) * 2 + 0];
      b += part[((size_t)l * C + ch) * 2 + 1];
    }
    part[(size_t)ch * 2 + 0] = a;
    part[(size_t)ch * 2 + 1] = b;
  }
  __syncthreads();
  const int cg = C / groups;
  for (int g = tid; g < groups; g += GN_THREADS) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < cg; ++j) {
      a += part[(size_t)(g * cg + j) * 2 + 0];
      b += part[(size_t)(g * cg + j) * 2 + 1];
    }
    float* o = ws + (((size_t)frame * slices + slice) * groups + g) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// (mean, rstd) of every group of one frame from the per-slice partial sums of gn_stats_kernel: re-reduced in fp64, all
// threads helping: sub-sum `sub` takes slices sub, sub+8, ... (ascending), then the GN_SUBS sub-sums are added in fixed
// order -> same bits in every block that asks for the frame (gn_apply and gn_fold_linear).  Ends with a barrier.
__device__ __forceinline__ void gn_group_stats(const float* __restrict__ ws, int frame, int slices, int groups, int cg,
                                               int hw, float eps, float* gstat, double* dpart) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < groups * GN_SUBS; idx += GN_THREADS) {
    const int g = idx % groups, sub = idx / groups;
    double a = 0.0, b = 0.0;
    for (int sl = sub; sl < slices; sl += GN_SUBS) {
      const float2 o = *reinterpret_cast<const float2*>(ws + (((size_t)frame * slices + sl) * groups + g) * 2);
      a += (double)o.x;
      b += (double)o.y;
    }
    dpart[(sub * groups + g) * 2 + 0] = a;
    dpart[(sub * groups + g) * 2 + 1] = b;
  }
  __syncthreads();
  for (int g = tid; g < groups; g += GN_THREADS) {
    double a = 0.0, b = 0.0;
    for (int sub = 0; sub < GN_SUBS; ++sub) {
      a += dpart[(sub * groups + g) * 2 + 0];
      b += dpart[(sub * groups + g) * 2 + 1];
    }
    double cnt = (double)cg * (double)hw;
    double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    gstat[g * 2 + 0] = (float)mean;
    gstat[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
}

// GroupNorm (no activation) folded into the 1x1 / linear layer behind it: per frame f
//   w_out[f][n][c] = bf16( w[n][c] * gamma[c] * rstd[f][g(c)] )                      (ONE rounding)
//   bias_out[f][n] = bias_beta[n] - sum_c float(w_out[f][n][c]) * mean[f][g(c)]      (of the ROUNDED weights the MFMA sees)
// with bias_beta[n] = bias[n] + sum_c w[n][c] beta[c] (frame independent, prepared at load time), so that
//   GN(x)[m, :] W^T + bias == x[m, :] w_out[f]^T + bias_out[f]      for every pixel m of frame f.
// grid (n / GN_FOLD_ROWS, frames); a thread octet owns one output row.
constexpr int GN_FOLD_ROWS = GN_THREADS / 8;
__global__ __launch_bounds__(GN_THREADS) void gn_fold_linear_kernel(const float* __restrict__ ws, int hw, int slices,
                                                                    int groups, float eps,
                                                                    const float* __restrict__ gamma, int C,
                                                                    const bf16_t* __restrict__ w,
                                                                    const float* __restrict__ bias_beta, int n,
                                                                    bf16_t* __restrict__ w_out,
                                                                    float* __restrict__ bias_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int frame = blockIdx.y, tid = threadIdx.x;
  float* scale = reinterpret_cast<float*>(smem);   // [C] gamma * rstd
  float* meanc = scale + C;                        // [C] mean of the channel's group
  float* gstat = meanc + C;                        // [groups][2]
  double* dpart = reinterpret_cast<double*>(gstat + 2 * groups);
  const int cg = C / groups;
  gn_group_stats(ws, frame, slices, groups, cg, hw, eps, gstat, dpart);
  for (int ch = tid; ch < C; ch += GN_THREADS) {
    const int g = ch / cg;
    scale[ch] = gamma[ch] * gstat[g * 2 + 1];
    meanc[ch] = gstat[g * 2 + 0];
  }
  __syncthreads();
  const int row = blockIdx.x * GN_FOLD_ROWS + (tid >> 3), sub = tid & 7;
  float dot = 0.f;
  if (row < n) {
    const bf16_t* src = w + (size_t)row * C;
    bf16_t* dst = w_out + ((size_t)frame * n + row) * C;
    for (int ch = sub * 8; ch < C; ch += 64) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(src + ch), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= scale[ch + e];
      const uint4 pk = pack_bf16x8(f);
      *reinterpret_cast<uint4*>(dst + ch) = pk;
      unpack_bf16x8(pk, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot = fmaf(f[e], meanc[ch + e], dot);
    }
  }
  dot = wave_xor_sum(dot, 1);
  dot = wave_xor_sum(dot, 2);
  dot = wave_xor_sum(dot, 4);
  if (row < n && sub == 0) bias_out[(size_t)frame * n + row] = bias_beta[row] - dot;
}

__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const bf16_t* __restrict__ x1, int c1,
                                                              const bf16_t* __restrict__ x2, int c2, int hw,
                                                              int groups, float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int silu,
                                                              bf16_t* __restrict__ out, int slices, int slice_pix,
                                                              const float* __restrict__ ws, int width, int w_shift,
                                                              int out_pad, int stat_slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = c1 + c2;
  const int bid = VX_XCD_ROWS ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;      // XCD-contiguous (see gn_stats_kernel)
  const int frame = bid / slices, slice = bid % slices;
  const int tid = threadIdx.x;
  float* scale = reinterpret_cast<float*>(smem);   // [C]
  float* shift = scale + C;                        // [C]
  float* gstat = shift + C;                        // [groups][2] mean, rstd
  double* dpart = reinterpret_cast<double*>(gstat + 2 * groups);   // [GN_SUBS][groups][2]
  const int cg = C / groups;
  // `slices` partitions THIS kernel's work; the statistics were written as `stat_slices` partial sums per frame
  gn_group_stats(ws, frame, stat_slices, groups, cg, hw, eps, gstat, dpart);
  for (int ch = tid; ch < C; ch += GN_THREADS) {
    int g = ch / cg;
    float sc = gamma[ch] * gstat[g * 2 + 1];
    scale[ch] = sc;
    shift[ch] = beta[ch] - gstat[g * 2 + 0] * sc;
  }
  __syncthreads();
  const GnPlan P = gn_plan(x1, c1, x2, c2, hw, frame);
  const int p_begin = slice * slice_pix;
  const int p_end = min(hw, p_begin + slice_pix);
  // out_pad > 0: the destination is the interior of a [frames, H + 2*pad, W + 2*pad, C] image whose border the
  // caller keeps zero (so the following 3x3 conv needs no bounds checks: vx_gemm FAST path)
  const int wp = width + 2 * out_pad;
  const int hp = hw / width + 2 * out_pad;
  bf16_t* const oframe = out + (size_t)frame * hp * wp * C;
  auto opix = [&](int px) {
    if (out_pad == 0) return px;
    const int y = w_shift >= 0 ? px >> w_shift : px / width;
    return (y + out_pad) * wp + (px - y * width) + out_pad;
  };
#pragma unroll
  for (int u = 0; u < GN_MAX_SETS; ++u) {
    if (P.base[u] == nullptr) continue;
    const bf16_t* src = P.base[u];
    const int ps = P.pstride[u];
    const int ch = P.chunk[u] * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = scale[ch + e];
      sh[e] = shift[ch + e];
    }
    auto emit = [&](const uint4& raw, int px) {
      float f[8];
      unpack_bf16x8(raw, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = f[e] * sc[e] + sh[e];
      if (silu == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
      } else if (silu == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = gelu_f(f[e]);
      }
      *reinterpret_cast<uint4*>(oframe + (size_t)opix(px) * C + ch) = pack_bf16x8(f);
    };
    int px = p_begin + P.pl;
    for (; px + (GN_UNROLL - 1) * P.pl_count < p_end; px += GN_UNROLL * P.pl_count) {
      uint4 raw[GN_UNROLL];
#pragma unroll
      for (int k = 0; k < GN_UNROLL; ++k)
        raw[k] = *reinterpret_cast<const uint4*>(src + (size_t)(px + k * P.pl_count) * ps);
#pragma unroll
      for (int k = 0; k < GN_UNROLL; ++k) emit(raw[k], px + k * P.pl_count);
    }
    for (; px < p_end; px += P.pl_count) emit(*reinterpret_cast<const uint4*>(src + (size_t)px * ps), px);
  }
}

// ---------------------------------------------------------------------------------------------------- LayerNorm
// One wave per row, LN_R rows per pass: the 16-B loads of all LN_R rows are issued before the first reduction, and a
// wave keeps walking rows (grid-stride) with its gamma/beta chunks resident in registers, so the kernel is bound by
// HBM rather than by one-load-per-wave latency.
constexpr int LN_R = 4;

template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int ldx, int rows, int c,
                                                        float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ add, int add_rows_per_entry,
                                                        int add_entries, bf16_t* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nchunks = c >> 3;
  const float inv_c = 1.0f / (float)c;
  bool live[MAXC];
  float g[MAXC][8], b[MAXC][8];
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    const int chunk = lane + u * 64;
    live[u] = chunk < nchunks;
    const int cs = live[u] ? chunk : 0;
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + cs * 8);
    const float4 g1 = *reinterpret_cast<const float4*>(gamma + cs * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + cs * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(beta + cs * 8 + 4);
    g[u][0] = g0.x; g[u][1] = g0.y; g[u][2] = g0.z; g[u][3] = g0.w;
    g[u][4] = g1.x; g[u][5] = g1.y; g[u][6] = g1.z; g[u][7] = g1.w;
    b[u][0] = b0.x; b[u][1] = b0.y; b[u][2] = b0.z; b[u][3] = b0.w;
    b[u][4] = b1.x; b[u][5] = b1.y; b[u][6] = b1.z; b[u][7] = b1.w;
  }
  for (int row0 = wave * LN_R; row0 < rows; row0 += nwaves * LN_R) {
    uint4 raw[LN_R][MAXC];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const int row = min(row0 + r, rows - 1);
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        const int chunk = live[u] ? lane + u * 64 : 0;
        raw[r][u] = *reinterpret_cast<const uint4*>(x + (size_t)row * ldx + chunk * 8);
      }
    }
    float sum[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      sum[r] = 0.f;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        float v[8];
        unpack_bf16x8(raw[r][u], v);
        float t = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        sum[r] += live[u] ? t : 0.f;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int r = 0; r < LN_R; ++r) sum[r] = wave_xor_sum(sum[r], m);
    float sq[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const float mean = sum[r] * inv_c;
      sq[r] = 0.f;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        float v[8];
        unpack_bf16x8(raw[r][u], v);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[e] - mean;
          t += d * d;
        }
        sq[r] += live[u] ? t : 0.f;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int r = 0; r < LN_R; ++r) sq[r] = wave_xor_sum(sq[r], m);
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const int row = row0 + r;
      if (row >= rows) break;
      const float mean = sum[r] * inv_c;
      const float rstd = rsqrtf(sq[r] * inv_c + eps);
      const float* addrow = nullptr;
      if (add != nullptr) addrow = add + (size_t)((row / add_rows_per_entry) % add_entries) * c;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        if (!live[u]) continue;
        const int chunk = lane + u * 64;
        float v[8], o[8];
        unpack_bf16x8(raw[r][u], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[e] - mean) * rstd * g[u][e] + b[u][e];
        if (addrow != nullptr) {
          const float4 a0 = *reinterpret_cast<const float4*>(addrow + chunk * 8);
          const float4 a1 = *reinterpret_cast<const float4*>(addrow + chunk * 8 + 4);
          o[0] += a0.x; o[1] += a0.y; o[2] += a0.z; o[3] += a0.w;
          o[4] += a1.x; o[5] += a1.y; o[6] += a1.z; o[7] += a1.w;
        }
        *reinterpret_cast<uint4*>(out + (size_t)row * ldo + chunk * 8) = pack_bf16x8(o);
      }
    }
  }
}

// Row statistics for a LayerNorm folded into its consumer GEMM (vx_row_stats): the read half of layernorm_kernel -
// LN_R rows per wave pass, two-pass variance in registers - writing (mean, rstd) per row instead of the normalised row.
template <int MAXC>
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ x, int ldx, int rows, int c, float eps,
                                                        float2* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nchunks = c >> 3;
  const float inv_c = 1.0f / (float)c;
  for (int row0 = wave * LN_R; row0 < rows; row0 += nwaves * LN_R) {
    uint4 raw[LN_R][MAXC];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const int row = min(row0 + r, rows - 1);
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        const int chunk = lane + u * 64 < nchunks ? lane + u * 64 : 0;
        raw[r][u] = *reinterpret_cast<const uint4*>(x + (size_t)row * ldx + chunk * 8);
      }
    }
    float sum[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      sum[r] = 0.f;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        float v[8];
        unpack_bf16x8(raw[r][u], v);
        const float t = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        sum[r] += lane + u * 64 < nchunks ? t : 0.f;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int r = 0; r < LN_R; ++r) sum[r] = wave_xor_sum(sum[r], m);
    float sq[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
      const float mean = sum[r] * inv_c;
      sq[r] = 0.f;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        float v[8];
        unpack_bf16x8(raw[r][u], v);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[e] - mean;
          t += d * d;
        }
        sq[r] += lane + u * 64 < nchunks ? t : 0.f;
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int r = 0; r < LN_R; ++r) sq[r] = wave_xor_sum(sq[r], m);
    if (lane < LN_R && row0 + lane < rows) {
      float mean = 0.f, var = 0.f;
#pragma unroll
      for (int r = 0; r < LN_R; ++r)
        if (lane == r) {
          mean = sum[r] * inv_c;
          var = sq[r] * inv_c;
        }
      stats[row0 + lane] = make_float2(mean, rsqrtf(var + eps));
    }
  }
}

// LayerNorm (NORM) or identity whose output is quantised per row to OCP e4m3 (vx_layernorm_fp8): same row-in-registers
// structure as layernorm_kernel; the normalised row stays in registers, its max |y| is reduced over the wave,
// scale = max / 448 and the row is written as 8 bytes per 8-channel chunk; chunks between c and ldo8 are zero-filled
// (the K padding of the fp8 GEMM).
template <int MAXC, bool NORM>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const bf16_t* __restrict__ x, int ldx, int rows, int c,
                                                            float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ add, int add_rows_per_entry,
                                                            int add_entries, uint8_t* __restrict__ out8, int ldo8,
                                                            float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nchunks = c >> 3, ochunks = ldo8 >> 3;
  const float inv_c = 1.0f / (float)c;
  for (int row = wave; row < rows; row += nwaves) {
    float y[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int chunk = lane + u * 64;
      if (chunk < nchunks) {
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (size_t)row * ldx + chunk * 8), y[u]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[u][e] = 0.f;
      }
      sum += ((y[u][0] + y[u][1]) + (y[u][2] + y[u][3])) + ((y[u][4] + y[u][5]) + (y[u][6] + y[u][7]));
    }
    if (NORM) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) sum = wave_xor_sum(sum, m);
      const float mean = sum * inv_c;
      float sq = 0.f;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        if (lane + u * 64 < nchunks) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = y[u][e] - mean;
            sq += d * d;
          }
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) sq = wave_xor_sum(sq, m);
      const float rstd = rsqrtf(sq * inv_c + eps);
      const float* addrow = add != nullptr ? add + (size_t)((row / add_rows_per_entry) % add_entries) * c : nullptr;
#pragma unroll
      for (int u = 0; u < MAXC; ++u) {
        const int chunk = lane + u * 64;
        if (chunk < nchunks) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float o = (y[u][e] - mean) * rstd * gamma[chunk * 8 + e] + beta[chunk * 8 + e];
            if (addrow != nullptr) o += addrow[chunk * 8 + e];
            y[u][e] = o;
          }
        }
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(y[u][e]));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) amax = wave_xor_max(amax, m);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int chunk = lane + u * 64;
      if (chunk < ochunks) {
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[u][0] * inv, y[u][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[u][2] * inv, y[u][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[u][4] * inv, y[u][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[u][6] * inv, y[u][7] * inv, w1, true);
        *reinterpret_cast<uint2*>(out8 + (size_t)row * ldo8 + chunk * 8) = make_uint2((uint32_t)w0, (uint32_t)w1);
      }
    }
  }
}

}  // namespace

extern "C" int64_t vx_groupnorm_ws_floats(int frames, int slices, int groups) {
  return (int64_t)frames * slices * groups * 2;
}

extern "C" int vx_groupnorm(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups, float eps,
                            const float* gamma, const float* beta, int silu, void* out, float* ws, int slices,
                            int width, int out_pad, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int C = c1 + c2;
  VX_REQUIRE(x1 != nullptr && out != nullptr && ws != nullptr && gamma != nullptr && beta != nullptr,
             "vx_groupnorm: null pointer");
  VX_REQUIRE((c2 == 0) == (x2 == nullptr), "vx_groupnorm: x2/c2 mismatch");
  VX_REQUIRE(c1 > 0 && (c1 % 8) == 0 && (c2 % 8) == 0, "vx_groupnorm: channels must be multiples of 8");
  VX_REQUIRE(groups > 0 && (C % groups) == 0, "vx_groupnorm: C=%d not divisible by groups=%d", C, groups);
  VX_REQUIRE(C <= 8 * GN_THREADS * GN_MAX_SETS, "vx_groupnorm: C=%d too large", C);
  VX_REQUIRE(frames > 0 && hw > 0 && slices > 0 && slices <= hw, "vx_groupnorm: bad geometry");
  VX_REQUIRE(silu >= 0 && silu <= 2, "vx_groupnorm: activation code %d (0 none, 1 SiLU, 2 erf-GELU)", silu);
  VX_REQUIRE(out_pad >= 0 && (out_pad == 0 || (width > 0 && hw % width == 0)),
             "vx_groupnorm: padded output needs the image width (hw=%d width=%d)", hw, width);
  if (out_pad == 0) width = hw;
  int w_shift = -1;
  for (int sft = 0; sft < 31; ++sft)
    if ((1 << sft) == width) w_shift = sft;
  const int slice_pix = ceil_div(hw, slices);
  const int nchunks = C / 8;
  const int tp = nchunks < GN_THREADS ? nchunks : GN_THREADS;
  const int pl_count = GN_THREADS / tp;
  size_t smem_stats = (size_t)pl_count * C * 2 * sizeof(float);
  // scale[C] + shift[C] + gstat[2*groups] floats, then the fp64 sub-sums (8-byte aligned: C, groups even)
  size_t smem_apply = (size_t)(2 * C + 2 * groups) * sizeof(float) + (size_t)GN_SUBS * groups * 2 * sizeof(double);
  VX_REQUIRE(smem_stats <= 64 * 1024, "vx_groupnorm: stats LDS %zu too large", smem_stats);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(frames * slices), dim3(GN_THREADS), smem_stats, stream,
                     (const bf16_t*)x1, c1, (const bf16_t*)x2, c2, hw, groups, slices, slice_pix, ws);
  int rc = vx_check_launch("vx_groupnorm(stats)");
  if (rc) return rc;
  // The apply pass may cut a frame into fewer, larger pieces than the statistics pass: every block re-reduces the
  // frame's partial sums before it streams (a fixed ~2 us prologue), so 64 blocks of 64 pixels per frame pay it 64 times.
  // VX_GN_APPLY_SLICES (A/B knob): upper bound on the apply pass's pieces per frame; element-wise pass, same bits.
  static int amax = -1;
  if (amax < 0) {
    const char* e = getenv("VX_GN_APPLY_SLICES");
    amax = e ? atoi(e) : 64;
    if (amax < 1) amax = 1;
  }
  const int aslices = slices < amax ? slices : amax;
  const int apix = ceil_div(hw, aslices);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(frames * aslices), dim3(GN_THREADS), smem_apply, stream,
                     (const bf16_t*)x1, c1, (const bf16_t*)x2, c2, hw, groups, eps, gamma, beta, silu,
                     (bf16_t*)out, aslices, apix, (const float*)ws, width, w_shift, out_pad, slices);
  return vx_check_launch("vx_groupnorm(apply)");
}

extern "C" int vx_groupnorm_apply(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups,
                                  float eps, const float* gamma, const float* beta, int silu, void* out,
                                  const float* ws, int stat_slices, int slices, int width, int out_pad, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int C = c1 + c2;
  VX_REQUIRE(x1 != nullptr && out != nullptr && ws != nullptr && gamma != nullptr && beta != nullptr,
             "vx_groupnorm_apply: null pointer");
  VX_REQUIRE((c2 == 0) == (x2 == nullptr), "vx_groupnorm_apply: x2/c2 mismatch");
  VX_REQUIRE(c1 > 0 && (c1 % 8) == 0 && (c2 % 8) == 0, "vx_groupnorm_apply: channels must be multiples of 8");
  VX_REQUIRE(groups > 0 && (C % groups) == 0, "vx_groupnorm_apply: C=%d not divisible by groups=%d", C, groups);
  VX_REQUIRE(C <= 8 * GN_THREADS * GN_MAX_SETS, "vx_groupnorm_apply: C=%d too large", C);
  VX_REQUIRE(frames > 0 && hw > 0 && slices > 0 && slices <= hw && stat_slices > 0, "vx_groupnorm_apply: bad geometry");
  VX_REQUIRE(silu >= 0 && silu <= 2, "vx_groupnorm_apply: activation code %d (0 none, 1 SiLU, 2 erf-GELU)", silu);
  VX_REQUIRE(out_pad >= 0 && (out_pad == 0 || (width > 0 && hw % width == 0)),
             "vx_groupnorm_apply: padded output needs the image width (hw=%d width=%d)", hw, width);
  if (out_pad == 0) width = hw;
  int w_shift = -1;
  for (int sft = 0; sft < 31; ++sft)
    if ((1 << sft) == width) w_shift = sft;
  const size_t smem_apply = (size_t)(2 * C + 2 * groups) * sizeof(float) + (size_t)GN_SUBS * groups * 2 * sizeof(double);
  const int apix = ceil_div(hw, slices);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(frames * slices), dim3(GN_THREADS), smem_apply, stream, (const bf16_t*)x1, c1,
                     (const bf16_t*)x2, c2, hw, groups, eps, gamma, beta, silu, (bf16_t*)out, slices, apix, ws, width,
                     w_shift, out_pad, stat_slices);
  return vx_check_launch("vx_groupnorm_apply");
}

extern "C" int vx_groupnorm_stats(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups,
                                  float* ws, int slices, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int C = c1 + c2;
  VX_REQUIRE(x1 != nullptr && ws != nullptr, "vx_groupnorm_stats: null pointer");
  VX_REQUIRE((c2 == 0) == (x2 == nullptr), "vx_groupnorm_stats: x2/c2 mismatch");
  VX_REQUIRE(c1 > 0 && (c1 % 8) == 0 && (c2 % 8) == 0, "vx_groupnorm_stats: channels must be multiples of 8");
  VX_REQUIRE(groups > 0 && (C % groups) == 0, "vx_groupnorm_stats: C=%d not divisible by groups=%d", C, groups);
  VX_REQUIRE(C <= 8 * GN_THREADS * GN_MAX_SETS, "vx_groupnorm_stats: C=%d too large", C);
  VX_REQUIRE(frames > 0 && hw > 0 && slices > 0 && slices <= hw, "vx_groupnorm_stats: bad geometry");
  const int slice_pix = ceil_div(hw, slices);
  const int nchunks = C / 8;
  const int tp = nchunks < GN_THREADS ? nchunks : GN_THREADS;
  const size_t smem_stats = (size_t)(GN_THREADS / tp) * C * 2 * sizeof(float);
  VX_REQUIRE(smem_stats <= 64 * 1024, "vx_groupnorm_stats: stats LDS %zu too large", smem_stats);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(frames * slices), dim3(GN_THREADS), smem_stats, stream,
                     (const bf16_t*)x1, c1, (const bf16_t*)x2, c2, hw, groups, slices, slice_pix, ws);
  return vx_check_launch("vx_groupnorm_stats");
}

extern "C" int vx_groupnorm_fold_linear(const float* ws, int frames, int hw, int slices, int groups, float eps,
                                        const float* gamma, int c, const void* w, const float* bias_beta, int n,
                                        void* w_out, float* bias_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(ws != nullptr && gamma != nullptr && w != nullptr && bias_beta != nullptr && w_out != nullptr &&
                 bias_out != nullptr, "vx_groupnorm_fold_linear: null pointer");
  VX_REQUIRE(frames > 0 && hw > 0 && slices > 0 && n > 0 && c > 0 && (c % 8) == 0 && groups > 0 && (c % groups) == 0,
             "vx_groupnorm_fold_linear: bad geometry (c=%d groups=%d n=%d)", c, groups, n);
  const size_t smem = (size_t)(2 * c + 2 * groups) * sizeof(float) + (size_t)GN_SUBS * groups * 2 * sizeof(double);
  VX_REQUIRE(smem <= 64 * 1024, "vx_groupnorm_fold_linear: c=%d too large", c);
  hipLaunchKernelGGL(gn_fold_linear_kernel, dim3(ceil_div(n, GN_FOLD_ROWS), frames), dim3(GN_THREADS), smem, stream, ws,
                     hw, slices, groups, eps, gamma, c, (const bf16_t*)w, bias_beta, n, (bf16_t*)w_out, bias_out);
  return vx_check_launch("vx_groupnorm_fold_linear");
}

extern "C" int vx_layernorm(const void* x, int ldx, int rows, int c, float eps, const float* gamma,
                            const float* beta, const float* add, int add_rows_per_entry, int add_entries, void* out,
                            int ldo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(x != nullptr && out != nullptr && gamma != nullptr && beta != nullptr, "vx_layernorm: null pointer");
  VX_REQUIRE(rows > 0 && c > 0 && (c % 8) == 0 && (ldx % 8) == 0 && (ldo % 8) == 0, "vx_layernorm: bad shape");
  VX_REQUIRE(add == nullptr || (add_rows_per_entry > 0 && add_entries > 0), "vx_layernorm: bad add table");
  const int nchunks = c / 8;
  // 4 waves per block, LN_R rows per wave-pass; cap the grid so that long inputs amortise the gamma/beta loads
  int nblk = ceil_div(rows, 4 * LN_R);
  if (nblk > 4096) nblk = 4096;
  dim3 grid(nblk), block(256);
#define VX_LN(MAXC)                                                                                              \
  hipLaunchKernelGGL(layernorm_kernel<MAXC>, grid, block, 0, stream, (const bf16_t*)x, ldx, rows, c, eps, gamma, \
                     beta, add, add_rows_per_entry, add_entries, (bf16_t*)out, ldo)
  if (nchunks <= 64) VX_LN(1);
  else if (nchunks <= 128) VX_LN(2);
  else if (nchunks <= 192) VX_LN(3);
  else if (nchunks <= 256) VX_LN(4);
  else {
    vx_set_error("vx_layernorm: c=%d exceeds 2048", c);
    return VX_ERR_UNSUPPORTED;
  }
#undef VX_LN
  return vx_check_launch("vx_layernorm");
}

extern "C" int vx_layernorm_fp8(const void* x, int ldx, int rows, int c, float eps, const float* gamma,
                                const float* beta, const float* add, int add_rows_per_entry, int add_entries,
                                void* out8, int ldo8, float* scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(x != nullptr && out8 != nullptr && scale != nullptr, "vx_layernorm_fp8: null pointer");
  VX_REQUIRE((gamma == nullptr) == (beta == nullptr), "vx_layernorm_fp8: gamma and beta come together");
  VX_REQUIRE(rows > 0 && c > 0 && (c % 8) == 0 && (ldx % 8) == 0 && (ldo8 % 16) == 0 && ldo8 >= c,
             "vx_layernorm_fp8: bad shape (c=%d ldx=%d ldo8=%d)", c, ldx, ldo8);
  VX_REQUIRE(add == nullptr || (gamma != nullptr && add_rows_per_entry > 0 && add_entries > 0),
             "vx_layernorm_fp8: bad add table");
  const int ochunks = ldo8 / 8;
  int nblk = ceil_div(rows, 4);
  if (nblk > 8192) nblk = 8192;
  dim3 grid(nblk), block(256);
#define VX_LN8(MAXC)                                                                                                  \
  do {                                                                                                                \
    if (gamma != nullptr)                                                                                             \
      hipLaunchKernelGGL((layernorm_fp8_kernel<MAXC, true>), grid, block, 0, stream, (const bf16_t*)x, ldx, rows, c,  \
                         eps, gamma, beta, add, add_rows_per_entry, add_entries, (uint8_t*)out8, ldo8, scale);        \
    else                                                                                                              \
      hipLaunchKernelGGL((layernorm_fp8_kernel<MAXC, false>), grid, block, 0, stream, (const bf16_t*)x, ldx, rows, c, \
                         eps, gamma, beta, add, add_rows_per_entry, add_entries, (uint8_t*)out8, ldo8, scale);        \
  } while (0)
  if (ochunks <= 64) VX_LN8(1);
  else if (ochunks <= 128) VX_LN8(2);
  else if (ochunks <= 192) VX_LN8(3);
  else if (ochunks <= 256) VX_LN8(4);
  else {
    vx_set_error("vx_layernorm_fp8: c=%d exceeds 2048", c);
    return VX_ERR_UNSUPPORTED;
  }
#undef VX_LN8
  return vx_check_launch("vx_layernorm_fp8");
}

// Two-part row statistics (vx_gemm_params.row_stats_parts): one wave per row, lanes 0-31 the first half of the row, lanes
// 32-63 the second; (sum, sum of squares) per half, float32, fixed order
__global__ __launch_bounds__(256) void row_stats_parts_kernel(const bf16_t* __restrict__ x, int ldx, int rows, int c,
                                                              float4* __restrict__ stats) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const int hc = c >> 1;                                 // columns per half (a multiple of 8)
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const bf16_t* src = x + (size_t)row * ldx + half * hc;
    float sm = 0.f, sq = 0.f;
    for (int ch = l32 * 8; ch < hc; ch += 32 * 8) {
      float v[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(src + ch), v);
      sm += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
      for (int e = 0; e < 8; ++e) sq = fmaf(v[e], v[e], sq);
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      sm = wave_xor_sum(sm, m);
      sq = wave_xor_sum(sq, m);
    }
    const float s1 = __shfl(sm, 32, 64), q1 = __shfl(sq, 32, 64);
    if (lane == 0) stats[row] = make_float4(sm, sq, s1, q1);
  }
}

extern "C" int vx_row_stats_parts(const void* x, int ldx, int rows, int c, float* stats, void* stream_) {
  VX_REQUIRE(x != nullptr && stats != nullptr, "vx_row_stats_parts: null pointer");
  VX_REQUIRE(rows > 0 && c > 0 && (c % 16) == 0 && (ldx % 8) == 0, "vx_row_stats_parts: bad shape");
  int nblk = ceil_div(rows, 4);
  if (nblk > 8192) nblk = 8192;
  hipLaunchKernelGGL(row_stats_parts_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, ldx, rows, c,
                     reinterpret_cast<float4*>(stats));
  return vx_check_launch("vx_row_stats_parts");
}

// two-part sums -> (mean, rstd): for a consumer that only takes the finished format (a launch off the persistent kernel)
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float4* __restrict__ parts, int rows, float inv_c,
                                                                 float eps, float2* __restrict__ stats) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const float4 t = parts[r];
    const float mean = (t.x + t.z) * inv_c;
    float var = (t.y + t.w) * inv_c - mean * mean;
    var = var > 0.f ? var : 0.f;
    stats[r] = make_float2(mean, 1.0f / sqrtf(var + eps));
  }
}

extern "C" int vx_row_stats_finalize(const float* parts, int rows, int c, float eps, float* stats, void* stream_) {
  VX_REQUIRE(parts != nullptr && stats != nullptr && rows > 0 && c > 0 && eps > 0.f, "vx_row_stats_finalize: bad arguments");
  int nblk = ceil_div(rows, 256);
  if (nblk > 2048) nblk = 2048;
  hipLaunchKernelGGL(row_stats_finalize_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream_,
                     reinterpret_cast<const float4*>(parts), rows, 1.0f / (float)c, eps, reinterpret_cast<float2*>(stats));
  return vx_check_launch("vx_row_stats_finalize");
}

extern "C" int vx_row_stats(const void* x, int ldx, int rows, int c, float eps, float* stats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(x != nullptr && stats != nullptr, "vx_row_stats: null pointer");
  VX_REQUIRE(rows > 0 && c > 0 && (c % 8) == 0 && (ldx % 8) == 0, "vx_row_stats: bad shape");
  const int nchunks = c / 8;
  int nblk = ceil_div(rows, 4 * LN_R);
  if (nblk > 4096) nblk = 4096;
  dim3 grid(nblk), block(256);
#define VX_RS(MAXC)                                                                                        \
  hipLaunchKernelGGL(row_stats_kernel<MAXC>, grid, block, 0, stream, (const bf16_t*)x, ldx, rows, c, eps, \
                     reinterpret_cast<float2*>(stats))
  if (nchunks <= 64) VX_RS(1);
  else if (nchunks <= 128) VX_RS(2);
  else if (nchunks <= 192) VX_RS(3);
  else if (nchunks <= 256) VX_RS(4);
  else {
    vx_set_error("vx_row_stats: c=%d exceeds 2048", c);
    return VX_ERR_UNSUPPORTED;
  }
#undef VX_RS
  return vx_check_launch("vx_row_stats");
}
