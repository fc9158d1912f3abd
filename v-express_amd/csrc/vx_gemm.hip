// MFMA GEMM / implicit-GEMM convolution for gfx950 (see include/vexpress_hip.h: vx_gemm).
//
//   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] ),   bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Block = 256 threads = 4 waves (WARPS_M x WARPS_N), tile BM x BN x 64.  Both operands are K-contiguous, so a
// K-tile row is 128 B = eight 16-B chunks.  Each thread stages (BM+BN)/32 chunks global -> VGPR -> LDS; the A
// gather computes (frame, iy, ix, ci) per chunk so 3x3/1x1 convs (stride 1/2, fused nearest-2x upsample, channel
// concat of two sources) and plain linears share one kernel.  LDS rows are XOR-swizzled at 16-B granularity
// (chunk ^= (row>>1)&7) so the ds_read_b128 fragment reads of 16 rows x 4 k-groups are bank-conflict-free.
// Software pipeline: the next K-tile's global loads are issued before the MFMAs of the current one and written to
// the other LDS stage afterwards (one barrier per K-tile).  The epilogue stages the fp32 tile through LDS in
// 64-row slabs so that bias / time-embedding / activation / residual / bf16 packing happen on 16-B coalesced rows.
#include "vx_common.h"
#include "../../include/vexpress_hip.h"

namespace {

constexpr int BK = 64;
constexpr int NTHREADS = 256;

struct RowInfo {
  int pix_base;  // frame * h_in * w_in
  int iy0, ix0;  // oy*stride - pad, ox*stride - pad   (very negative when the row is out of range)
};

template <typename T>
__device__ __forceinline__ T sel3(int i, T a, T b, T c) { return i == 0 ? a : (i == 1 ? b : c); }

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 16 zero bytes: the source of every out-of-range chunk (conv padding, M/N/K tails), so the staging loads are
// branch-free and can all be in flight at once
__device__ __attribute__((aligned(16))) const uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// async 16-B global -> LDS copy (global_load_lds_dwordx4): LDS address = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// XCD-aware block remap (8 XCDs, blocks are dealt round-robin): logical ids that are adjacent run on the same
// XCD, so the column tiles of one A row-tile share that XCD's L2.  Bijective for any block count.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const vx_gemm_params p) {
  constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert(WM == 64, "epilogue assumes 64-row wave slabs");
  static_assert(WARPS_M * WARPS_N == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int n_tiles = (p.n + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / n_tiles, tile_n = lid - tile_m * n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ A1 = (const bf16_t*)p.a;
  const bf16_t* __restrict__ A2 = (const bf16_t*)p.a2;
  const bf16_t* __restrict__ Wt = (const bf16_t*)p.w;
  const int cin = p.c1 + p.c2;
  const int c1 = p.c1;
  const int up = p.upsample;
  const int h_eff = p.h_in << up, w_eff = p.w_in << up;
  const int hw_out = p.h_out * p.w_out;
  const int w_in = p.w_in, kw = p.kw, kh = p.kh;
  const int lda1 = p.lda1, lda2 = p.lda2;
  const bf16_t* const zsrc = reinterpret_cast<const bf16_t*>(&g_zero16);

  // ---- per-thread staging coordinates: LDS slot `s` of row r0 + 32*i holds K-chunk s ^ ((row >> 1) & 7)
  const int r0 = tid >> 3;
  const int cc = (tid & 7) ^ ((r0 >> 1) & 7);   // K-chunk (8 elements) this thread fetches
  RowInfo ri[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < p.m) {
      int fr = m / hw_out;
      int rem = m - fr * hw_out;
      int oy = rem / p.w_out;
      int ox = rem - oy * p.w_out;
      ri[i].pix_base = fr * p.h_in * p.w_in;
      ri[i].iy0 = oy * p.stride - p.pad;
      ri[i].ix0 = ox * p.stride - p.pad;
    } else {
      ri[i].pix_base = 0;
      ri[i].iy0 = -(1 << 28);
      ri[i].ix0 = -(1 << 28);
    }
  }
  long wrow[B_IT];   // element offset of weight row n (or -1: beyond N)
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int n = n0 + r0 + 32 * i;
    wrow[i] = n < p.n ? (long)n * p.k : -1;
  }
  // K position of this thread's chunk, tracked incrementally: (ky, kx, ci) with k = ((ky*kw)+kx)*cin + ci
  int kg = cc * 8;
  int ky, kx, ci;
  {
    int tap = kg / cin;
    ci = kg - tap * cin;
    ky = tap / kw;
    kx = tap - ky * kw;
  }
  char* const lds_wave = smem + (wave * 8) * 128;   // this wave's 8-row (1 KiB) slab within each 32-row group

  auto issue_tile = [&](int stage) {
    char* sa = lds_wave + stage * STAGE_BYTES;
    char* sb = sa + BM * 128;
    const bool kval = ky < kh;
    const bool first = ci < c1;
    const bf16_t* src = first ? A1 + ci : A2 + (ci - c1);
    const int cs = first ? lda1 : lda2;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int iy = ri[i].iy0 + ky, ix = ri[i].ix0 + kx;
      bool ok = kval && (unsigned)iy < (unsigned)h_eff && (unsigned)ix < (unsigned)w_eff;
      int pix = ri[i].pix_base + (iy >> up) * w_in + (ix >> up);
      const bf16_t* g = ok ? src + (long)pix * (long)cs : zsrc;
      glds16(g, sa + i * 32 * 128);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      bool ok = kval && wrow[i] >= 0;
      const bf16_t* g = ok ? Wt + wrow[i] + kg : zsrc;
      glds16(g, sb + i * 32 * 128);
    }
    // advance to the next K-tile
    kg += BK;
    ci += BK;
    while (ci >= cin) {
      ci -= cin;
      if (++kx == kw) { kx = 0; ++ky; }
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.k + BK - 1) / BK;
  issue_tile(0);

  const int frow = lane & 15, fgrp = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    // tile kt has landed (own DMAs drained, then the barrier covers everyone's); the barrier also orders the
    // other stage's last reads (iteration kt-1) before it is refilled below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue_tile(stage ^ 1);
    const char* sa = smem + stage * STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const uint4*>(sa + lds_off(wm * WM + i * 16 + frow, kk * 4 + fgrp));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off(wn * WN + j * 16 + frow, kk * 4 + fgrp));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  }
  __syncthreads();   // all fragment reads done before the epilogue reuses the stages

  // ------------------------------------------------------------------ epilogue (fp32 tile through LDS, 64-row slabs)
  constexpr int OUTW = (EPI == VX_EPI_GEGLU) ? BN / 2 : BN;   // staged tile width
  constexpr int CT_LD = OUTW + 4;                             // floats; keeps 16-B alignment, spreads banks
  float* ct = reinterpret_cast<float*>(smem);
  const float* __restrict__ bias = p.bias;

  for (int pass = 0; pass < WARPS_M; ++pass) {
    if (wm == pass) {
      if constexpr (EPI == VX_EPI_GEGLU) {
        // fragment pairs (2q, 2q+1) hold the value / gate columns of the same 16 output channels
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          int ncol = n0 + wn * WN + j * 16 + frow;           // interleaved weight row of the value column
          float bh = 0.f, bg = 0.f;
          if (bias != nullptr) {
            if (ncol < p.n) bh = bias[ncol];
            if (ncol + 16 < p.n) bg = bias[ncol + 16];
          }
          int ocol = (wn * WN + j * 16) / 2 + frow;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float hval = acc[i][j][r] + bh;
              float gval = acc[i][j + 1][r] + bg;
              ct[(i * 16 + fgrp * 4 + r) * CT_LD + ocol] = hval * gelu_f(gval);
            }
        }
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              ct[(i * 16 + fgrp * 4 + r) * CT_LD + wn * WN + j * 16 + frow] = acc[i][j][r];
      }
    }
    __syncthreads();

    const int mbase = m0 + pass * 64;
    if constexpr (EPI == VX_EPI_STORE || EPI == VX_EPI_GEGLU) {
      const int nout = (EPI == VX_EPI_GEGLU) ? p.n / 2 : p.n;
      const int nbase = (EPI == VX_EPI_GEGLU) ? n0 / 2 : n0;
      constexpr int CPR = OUTW / 8;   // 8-column chunks per row
      for (int idx = tid; idx < 64 * CPR; idx += NTHREADS) {
        int row = idx / CPR, c8 = idx - row * CPR;
        int m = mbase + row, n = nbase + c8 * 8;
        if (m >= p.m || n >= nout) continue;
        float v[8];
        const float4 lo = *reinterpret_cast<const float4*>(ct + row * CT_LD + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ct + row * CT_LD + c8 * 8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        if constexpr (EPI == VX_EPI_STORE) {
          if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[n + e];
          }
          if (p.rowbias != nullptr) {
            const float* rbp = p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rbp[e];
          }
          if (p.act == VX_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
          }
          if (p.alpha != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
          }
          if (p.residual != nullptr) {
            float rr[8];
            unpack_bf16x8(*reinterpret_cast<const uint4*>((const bf16_t*)p.residual + (size_t)m * p.ldr + n), rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
          }
        }
        if (EPI == VX_EPI_STORE && p.out_f32) {
          float* o = (float*)p.out + (size_t)m * p.ldc + n;
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>((bf16_t*)p.out + (size_t)m * p.ldc + n) = pack_bf16x8(v);
        }
      }
    } else {  // VX_EPI_SPLIT
      constexpr int CPR = BN / 8;
      // (a) row-major parts: 16-B coalesced rows
      for (int idx = tid; idx < 64 * CPR; idx += NTHREADS) {
        int row = idx / CPR, c8 = idx - row * CPR;
        int m = mbase + row, n = n0 + c8 * 8;
        if (m >= p.m || n >= p.n) continue;
        int part = n / p.part_cols;
        if (sel3(part, p.part_kind[0], p.part_kind[1], p.part_kind[2]) != VX_PART_ROWS) continue;
        int nn = n - part * p.part_cols;
        float v[8];
        const float4 lo = *reinterpret_cast<const float4*>(ct + row * CT_LD + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(ct + row * CT_LD + c8 * 8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        if (bias != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias[n + e];
        }
        bf16_t* dst = (bf16_t*)sel3(part, p.part_out[0], p.part_out[1], p.part_out[2]);
        int ldp = sel3(part, p.part_ld[0], p.part_ld[1], p.part_ld[2]);
        *reinterpret_cast<uint4*>(dst + (size_t)m * ldp + nn) = pack_bf16x8(v);
      }
      // (b) transposed (V^T) parts: [seq, head, dim, key] with keys contiguous; 8 consecutive tokens per store
      for (int idx = tid; idx < 8 * BN; idx += NTHREADS) {
        int rg = idx / BN, col = idx - rg * BN;
        int n = n0 + col;
        int mfirst = mbase + rg * 8;
        if (n >= p.n || mfirst >= p.m) continue;
        int part = n / p.part_cols;
        if (sel3(part, p.part_kind[0], p.part_kind[1], p.part_kind[2]) != VX_PART_VT) continue;
        int nn = n - part * p.part_cols;
        int head = nn / p.head_dim, dd = nn - head * p.head_dim;
        int heads = p.part_cols / p.head_dim;
        float bv = bias != nullptr ? bias[n] : 0.f;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ct[(rg * 8 + e) * CT_LD + col] + bv;
        bf16_t* vt = (bf16_t*)sel3(part, p.part_out[0], p.part_out[1], p.part_out[2]);
        int seq = mfirst / p.seq_len, tok = mfirst - seq * p.seq_len;
        if ((p.seq_len & 7) == 0 && mfirst + 8 <= p.m) {
          size_t off = ((size_t)(seq * heads + head) * p.head_dim + dd) * p.vt_pitch + tok;
          *reinterpret_cast<uint4*>(vt + off) = pack_bf16x8(v);
        } else {
          for (int e = 0; e < 8; ++e) {
            int m = mfirst + e;
            if (m >= p.m) break;
            int s = m / p.seq_len, t = m - s * p.seq_len;
            vt[((size_t)(s * heads + head) * p.head_dim + dd) * p.vt_pitch + t] = f32_to_bf16(v[e]);
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int EPI>
int launch(const vx_gemm_params& p, hipStream_t stream) {
  constexpr int stage_bytes = 2 * (BM + BN) * 128;
  constexpr int outw = (EPI == VX_EPI_GEGLU) ? BN / 2 : BN;
  constexpr int epi_bytes = 64 * (outw + 4) * 4;
  constexpr int smem = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static bool attr_set = false;
  auto kern = gemm_kernel<BM, BN, WARPS_M, WARPS_N, EPI>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vx_set_error("vx_gemm: hipFuncSetAttribute(%d B LDS) failed: %s", smem, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    attr_set = true;
  }
  long tiles = (long)ceil_div(p.m, BM) * ceil_div(p.n, BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NTHREADS), smem, stream, p);
  return vx_check_launch("vx_gemm");
}

// column-tile width with the least padding (ties -> 160: fewer, fatter tiles)
bool prefer160(int n) {
  int w160 = ceil_div(n, 160) * 160 - n, w128 = ceil_div(n, 128) * 128 - n;
  return w160 * 128 <= w128 * 160;
}

}  // namespace

extern "C" int vx_gemm(const vx_gemm_params* pp, void* stream_) {
  const vx_gemm_params& p = *pp;
  hipStream_t stream = (hipStream_t)stream_;
  VX_REQUIRE(p.a != nullptr && p.w != nullptr, "vx_gemm: null operand");
  VX_REQUIRE(p.m > 0 && p.n > 0 && p.k > 0, "vx_gemm: empty problem m=%d n=%d k=%d", p.m, p.n, p.k);
  VX_REQUIRE((p.c1 % 8) == 0 && (p.c2 % 8) == 0 && p.c1 > 0, "vx_gemm: channels must be multiples of 8 (c1=%d c2=%d)",
             p.c1, p.c2);
  VX_REQUIRE((p.c2 == 0) == (p.a2 == nullptr), "vx_gemm: a2/c2 mismatch");
  VX_REQUIRE(p.k == p.kh * p.kw * (p.c1 + p.c2), "vx_gemm: k=%d != kh*kw*(c1+c2)=%d", p.k,
             p.kh * p.kw * (p.c1 + p.c2));
  VX_REQUIRE(p.m == p.nb * p.h_out * p.w_out, "vx_gemm: m=%d != nb*h_out*w_out", p.m);
  VX_REQUIRE((p.n % 8) == 0, "vx_gemm: n=%d must be a multiple of 8", p.n);
  VX_REQUIRE((p.lda1 % 8) == 0 && (p.c2 == 0 || (p.lda2 % 8) == 0), "vx_gemm: lda must be a multiple of 8");
  VX_REQUIRE(p.upsample == 0 || p.upsample == 1, "vx_gemm: upsample must be 0/1");
  VX_REQUIRE(p.stride >= 1 && p.kh >= 1 && p.kw >= 1, "vx_gemm: bad conv geometry");
  if (p.epi == VX_EPI_STORE) {
    VX_REQUIRE(p.out != nullptr && (p.ldc % 8) == 0, "vx_gemm: STORE needs out and ldc%%8==0");
    VX_REQUIRE(p.residual == nullptr || (p.ldr % 8) == 0, "vx_gemm: ldr%%8");
    VX_REQUIRE(p.rowbias == nullptr || p.rows_per_group > 0, "vx_gemm: rows_per_group");
    if (p.n <= 32) return launch<256, 32, 4, 1, VX_EPI_STORE>(p, stream);
    if (prefer160(p.n)) return launch<128, 160, 2, 2, VX_EPI_STORE>(p, stream);
    return launch<128, 128, 2, 2, VX_EPI_STORE>(p, stream);
  } else if (p.epi == VX_EPI_GEGLU) {
    VX_REQUIRE(p.out != nullptr && (p.n % 32) == 0 && (p.ldc % 8) == 0,
               "vx_gemm: GEGLU needs n%%32==0 (16-wide value/gate interleave)");
    return launch<128, 128, 2, 2, VX_EPI_GEGLU>(p, stream);
  } else if (p.epi == VX_EPI_SPLIT) {
    VX_REQUIRE(p.n_parts >= 1 && p.n_parts <= 3 && p.part_cols > 0 && p.n == p.n_parts * p.part_cols,
               "vx_gemm: SPLIT n=%d != n_parts*part_cols", p.n);
    VX_REQUIRE((p.part_cols % 8) == 0, "vx_gemm: part_cols%%8");
    for (int i = 0; i < p.n_parts; ++i) {
      VX_REQUIRE(p.part_out[i] != nullptr, "vx_gemm: SPLIT part %d has no destination", i);
      if (p.part_kind[i] == VX_PART_VT)
        VX_REQUIRE(p.seq_len > 0 && p.head_dim > 0 && (p.part_cols % p.head_dim) == 0 && (p.vt_pitch % 8) == 0 &&
                       p.vt_pitch >= p.seq_len && (p.m % p.seq_len) == 0,
                   "vx_gemm: bad V^T geometry seq_len=%d head_dim=%d pitch=%d", p.seq_len, p.head_dim, p.vt_pitch);
      else
        VX_REQUIRE((p.part_ld[i] % 8) == 0, "vx_gemm: part_ld%%8");
    }
    if (prefer160(p.n)) return launch<128, 160, 2, 2, VX_EPI_SPLIT>(p, stream);
    return launch<128, 128, 2, 2, VX_EPI_SPLIT>(p, stream);
  }
  vx_set_error("vx_gemm: unknown epilogue %d", p.epi);
  return VX_ERR_UNSUPPORTED;
}
