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
#include "../../include/vexpress_hip.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SETS = 2;   // chunks per thread per pixel -> C <= 8*256*2 = 4096

__device__ __forceinline__ const bf16_t* gn_src(const bf16_t* x1, int c1, const bf16_t* x2, int c2, size_t pix,
                                                int ch) {
  return ch < c1 ? x1 + pix * (size_t)c1 + ch : x2 + pix * (size_t)c2 + (ch - c1);
}

__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const bf16_t* __restrict__ x1, int c1,
                                                              const bf16_t* __restrict__ x2, int c2, int hw,
                                                              int groups, int slices, int slice_pix,
                                                              float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = c1 + c2;
  const int nchunks = C >> 3;
  const int frame = blockIdx.x / slices, slice = blockIdx.x % slices;
  const int tid = threadIdx.x;
  const int tp = nchunks < GN_THREADS ? nchunks : GN_THREADS;   // threads per pixel
  const int pl_count = GN_THREADS / tp;                         // pixel lanes
  const int cc = tid % tp, pl = tid / tp;
  const bool active = pl < pl_count;
  const int p_begin = slice * slice_pix;
  const int p_end = min(hw, p_begin + slice_pix);

  float s[GN_MAX_SETS][8], q[GN_MAX_SETS][8];
#pragma unroll
  for (int u = 0; u < GN_MAX_SETS; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[u][e] = q[u][e] = 0.f;

  if (active) {
    for (int px = p_begin + pl; px < p_end; px += pl_count) {
      size_t pix = (size_t)frame * hw + px;
#pragma unroll
      for (int u = 0; u < GN_MAX_SETS; ++u) {
        int chunk = cc + u * GN_THREADS;
        if (chunk < nchunks) {
          float f[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(gn_src(x1, c1, x2, c2, pix, chunk * 8)), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s[u][e] += f[e];
            q[u][e] += f[e] * f[e];
          }
        }
      }
    }
  }
  // per-channel partials: [pl_count][C][2]
  float* part = reinterpret_cast<float*>(smem);
  if (active) {
#pragma unroll
    for (int u = 0; u < GN_MAX_SETS; ++u) {
      int chunk = cc + u * GN_THREADS;
      if (chunk < nchunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          part[((size_t)pl * C + chunk * 8 + e) * 2 + 0] = s[u][e];
          part[((size_t)pl * C + chunk * 8 + e) * 2 + 1] = q[u][e];
        }
      }
    }
  }
  __syncthreads();
  // channel totals (fixed order over pixel lanes), written back into lane 0's slot
  for (int ch = tid; ch < C; ch += GN_THREADS) {
    float a = 0.f, b = 0.f;
    for (int l = 0; l < pl_count; ++l) {
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

__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const bf16_t* __restrict__ x1, int c1,
                                                              const bf16_t* __restrict__ x2, int c2, int hw,
                                                              int groups, float eps, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int silu,
                                                              bf16_t* __restrict__ out, int slices, int slice_pix,
                                                              const float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int C = c1 + c2;
  const int nchunks = C >> 3;
  const int frame = blockIdx.x / slices, slice = blockIdx.x % slices;
  const int tid = threadIdx.x;
  float* scale = reinterpret_cast<float*>(smem);   // [C]
  float* shift = scale + C;                        // [C]
  float* gstat = shift + C;                        // [groups][2] mean, rstd
  const int cg = C / groups;
  for (int g = tid; g < groups; g += GN_THREADS) {
    double a = 0.0, b = 0.0;
    for (int sl = 0; sl < slices; ++sl) {
      const float* o = ws + (((size_t)frame * slices + sl) * groups + g) * 2;
      a += (double)o[0];
      b += (double)o[1];
    }
    double cnt = (double)cg * (double)hw;
    double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    gstat[g * 2 + 0] = (float)mean;
    gstat[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int ch = tid; ch < C; ch += GN_THREADS) {
    int g = ch / cg;
    float sc = gamma[ch] * gstat[g * 2 + 1];
    scale[ch] = sc;
    shift[ch] = beta[ch] - gstat[g * 2 + 0] * sc;
  }
  __syncthreads();
  const int p_begin = slice * slice_pix;
  const int p_end = min(hw, p_begin + slice_pix);
  const long total = (long)(p_end - p_begin) * nchunks;
  for (long idx = tid; idx < total; idx += GN_THREADS) {
    int px = p_begin + (int)(idx / nchunks);
    int chunk = (int)(idx % nchunks);
    size_t pix = (size_t)frame * hw + px;
    float f[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(gn_src(x1, c1, x2, c2, pix, chunk * 8)), f);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + chunk * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(scale + chunk * 8 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + chunk * 8);
    const float4 h1 = *reinterpret_cast<const float4*>(shift + chunk * 8 + 4);
    f[0] = f[0] * s0.x + h0.x; f[1] = f[1] * s0.y + h0.y; f[2] = f[2] * s0.z + h0.z; f[3] = f[3] * s0.w + h0.w;
    f[4] = f[4] * s1.x + h1.x; f[5] = f[5] * s1.y + h1.y; f[6] = f[6] * s1.z + h1.z; f[7] = f[7] * s1.w + h1.w;
    if (silu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
    }
    *reinterpret_cast<uint4*>(out + pix * (size_t)C + chunk * 8) = pack_bf16x8(f);
  }
}

// ---------------------------------------------------------------------------------------------------- LayerNorm
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int ldx, int rows, int c,
                                                        float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ add, int add_rows_per_entry,
                                                        int add_entries, bf16_t* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunks = c >> 3;
  float v[MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    int chunk = lane + u * 64;
    if (chunk < nchunks) {
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (size_t)row * ldx + chunk * 8), v[u]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[u][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sum = wave_xor_sum(sum, m);
  const float mean = sum / (float)c;
  float sq = 0.f;
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    int chunk = lane + u * 64;
    if (chunk < nchunks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float d = v[u][e] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) sq = wave_xor_sum(sq, m);
  const float rstd = rsqrtf(sq / (float)c + eps);
  const float* addrow = nullptr;
  if (add != nullptr) addrow = add + (size_t)((row / add_rows_per_entry) % add_entries) * c;
#pragma unroll
  for (int u = 0; u < MAXC; ++u) {
    int chunk = lane + u * 64;
    if (chunk < nchunks) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int ch = chunk * 8 + e;
        float y = (v[u][e] - mean) * rstd * gamma[ch] + beta[ch];
        if (addrow != nullptr) y += addrow[ch];
        o[e] = y;
      }
      *reinterpret_cast<uint4*>(out + (size_t)row * ldo + chunk * 8) = pack_bf16x8(o);
    }
  }
}

}  // namespace

extern "C" int64_t vx_groupnorm_ws_floats(int frames, int slices, int groups) {
  return (int64_t)frames * slices * groups * 2;
}

extern "C" int vx_groupnorm(const void* x1, int c1, const void* x2, int c2, int frames, int hw, int groups, float eps,
                            const float* gamma, const float* beta, int silu, void* out, float* ws, int slices,
                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int C = c1 + c2;
  VX_REQUIRE(x1 != nullptr && out != nullptr && ws != nullptr && gamma != nullptr && beta != nullptr,
             "vx_groupnorm: null pointer");
  VX_REQUIRE((c2 == 0) == (x2 == nullptr), "vx_groupnorm: x2/c2 mismatch");
  VX_REQUIRE(c1 > 0 && (c1 % 8) == 0 && (c2 % 8) == 0, "vx_groupnorm: channels must be multiples of 8");
  VX_REQUIRE(groups > 0 && (C % groups) == 0, "vx_groupnorm: C=%d not divisible by groups=%d", C, groups);
  VX_REQUIRE(C <= 8 * GN_THREADS * GN_MAX_SETS, "vx_groupnorm: C=%d too large", C);
  VX_REQUIRE(frames > 0 && hw > 0 && slices > 0 && slices <= hw, "vx_groupnorm: bad geometry");
  const int slice_pix = ceil_div(hw, slices);
  const int nchunks = C / 8;
  const int tp = nchunks < GN_THREADS ? nchunks : GN_THREADS;
  const int pl_count = GN_THREADS / tp;
  size_t smem_stats = (size_t)pl_count * C * 2 * sizeof(float);
  size_t smem_apply = (size_t)(2 * C + 2 * groups) * sizeof(float);
  VX_REQUIRE(smem_stats <= 64 * 1024, "vx_groupnorm: stats LDS %zu too large", smem_stats);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(frames * slices), dim3(GN_THREADS), smem_stats, stream,
                     (const bf16_t*)x1, c1, (const bf16_t*)x2, c2, hw, groups, slices, slice_pix, ws);
  int rc = vx_check_launch("vx_groupnorm(stats)");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(frames * slices), dim3(GN_THREADS), smem_apply, stream,
                     (const bf16_t*)x1, c1, (const bf16_t*)x2, c2, hw, groups, eps, gamma, beta, silu,
                     (bf16_t*)out, slices, slice_pix, (const float*)ws);
  return vx_check_launch("vx_groupnorm(apply)");
}

extern "C" int vx_layernorm(const void* x, int ldx, int rows, int c, float eps, const float* gamma,
                            const float* beta, const float* add, int add_rows_per_entry, int add_entries, void* out,
                            int ldo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(x != nullptr && out != nullptr && gamma != nullptr && beta != nullptr, "vx_layernorm: null pointer");
  VX_REQUIRE(rows > 0 && c > 0 && (c % 8) == 0 && (ldx % 8) == 0 && (ldo % 8) == 0, "vx_layernorm: bad shape");
  VX_REQUIRE(add == nullptr || (add_rows_per_entry > 0 && add_entries > 0), "vx_layernorm: bad add table");
  const int nchunks = c / 8;
  dim3 grid(ceil_div(rows, 4)), block(256);
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
