// attn3: the d = 40 (64x64-level, SD-1.5 head dim) spatial attention of vx_attention_bounded, rebuilt around the two
// things the rocprofv3 / micro-benchmark numbers of round 1 said about attn2 (profiles/r01h_*): per 64-key tile and wave
// it issued ~190 VALU + 4 LDS stores + 4 global loads beside 28 MFMAs and crossed two block barriers, and a SIMD
// finished a tile every ~560 ns although its matrix pipe needs ~210 ns and its VALU ~190 ns for the same tile.
//
//   * K / V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4, 10 one-KiB pieces per tile shared by the 4 waves) into a
//     ring of NBUF stages: no staging registers, no ds_write, no address arithmetic in the loop, ONE barrier per tile.
//   * The softmax shift is the fixed Cauchy-Schwarz bound of vx_attention_bounded, but it now rides in the MFMA: Q
//     carries -m_q in the (otherwise zero) padding column d = 40 and the K tile has a constant 1.0 there, so the QK^T
//     accumulator already holds s - m.  When the caller has folded scale*log2(e) into K (vx_attention scale = 0: the
//     model folds it into the key projection weights, so no operand is rounded twice) a score costs one v_exp_f32 and
//     half a v_cvt_pk_bf16_f32 - nothing else; otherwise one v_mul_f32 more (Q is NOT pre-multiplied in bf16: that
//     second rounding of q costs ~0.1 log2 units per score at |logit| ~ 50 and showed in the large-logit tests).
//     (m_q is rounded to bf16: a per-query constant factor on every probability of the row, which the normalisation
//     removes; row sums come from the ones row of V^T as in attn2.)
//   * Keys are permuted inside a tile on the DMA source side (LDS row r holds key pi(r) = 32(r>>5) + 8((r>>2)&3) +
//     4((r>>4)&1) + (r&3)) so that the 8 P^T values a lane owns after two S^T tiles are 8 CONSECUTIVE keys: the V^T
//     A-operand is one ds_read_b128 per MFMA instead of two ds_read_b64.
//   * QK32: QK^T on v_mfma_f32_32x32x16_bf16 - its K dimension steps by 16, so d = 40 (+ the shift column) pads to 48
//     instead of 64: 6 x 32 = 192 matrix cycles per 64-key tile instead of 16 x 16 = 256 (the kernel is matrix-pipe
//     bound).  A lane then owns query l & 31 and 16 keys of each 32-key tile; v_permlane16_swap of the packed P^T halves
//     hands every 16-lane row the 8 consecutive keys the 16x16x32 PV MFMA wants (8 swaps per tile), with the matching key
//     permutation on the DMA source side.
//   * PIPE: the S^T tile of key tile t+1 is multiplied while the exponentials of tile t are taken (two S^T register
//     sets, static ping-pong); needs NBUF = 3.
// Rows whose probabilities underflow under the bound (row sum < 2^-100) are detected block-uniformly; the block then
// finds the true row maxima with a QK^T-only pass and repeats the loop with those as the shift.
#include "vx_common.h"
#include "vx_gemm_common.h"

#include <stdlib.h>

#include <type_traits>

// Compile-time ablation switches (tools/exp_attn3_ablate.sh builds one library per mask with -DVX_ATTN3_ABLATE=mask;
// never defined for the product library): 1 no exponentials, 2 no PV MFMAs, 4 no QK^T MFMAs, 8 no DMA after the
// prologue (and no waits), 16 no barriers in the key loop, 32 LDS fragment reads hoisted out (one stage-0 read).
#ifdef VX_ATTN3_ABLATE
#define A3ABL(bit) (((VX_ATTN3_ABLATE) & (bit)) != 0)
#else
#define A3ABL(bit) false
#endif

namespace {

struct Attn3Params {
  const bf16_t* q; int ldq;
  const bf16_t* k; int ldk;
  const bf16_t* vt; int vt_pitch;
  bf16_t* out; int ldo;
  int batch, heads, n_q, n_kv, q_per_kv;
  float c;             // scale * log2(e)
  const float* kmax;   // [kv batches * heads] max key norm of the head slice
};

constexpr int A3_D = 40;                  // head dim
constexpr int A3_K0 = 0;                  // K chunks 0..3: [64 rows][64 B], chunk ^= (-(row>>2))&3
constexpr int A3_K1 = 4096;               // K chunk 4 (d 32..39): [64 rows][16 B]
constexpr int A3_V = 5120;                // V^T: [48 rows][128 B], chunk ^= (row>>1)&7; rows 40..47 constant
constexpr int A3_STAGE = 5120 + 48 * 128; // 11264
constexpr int A3_QT = 2;                  // 16-query tiles per wave (32 queries per wave, 128 per block)

__device__ __forceinline__ int a3_pi16(int r) { return 32 * (r >> 5) + 8 * ((r >> 2) & 3) + 4 * ((r >> 4) & 1) + (r & 3); }
// QK32: row r of a 32-key S^T tile (C register (r&3) + 8 (r>>2) of lane half 4 (l>>5)) holds key
// 16 b2 + 8 b4 + 4 b3 + (r & 3): the 8 packed values a 16-lane row presents to the PV MFMA after the swap are then keys 8 g .. 8 g + 7
__device__ __forceinline__ int a3_pi32(int r) {
  return 32 * (r >> 5) + 16 * ((r >> 2) & 1) + 8 * ((r >> 4) & 1) + 4 * ((r >> 3) & 1) + (r & 3);
}
template <bool QK32>
__device__ __forceinline__ int a3_pi(int r) { return QK32 ? a3_pi32(r) : a3_pi16(r); }

template <int NBUF, bool PIPE, bool UNIT, bool QK32 = false>   // UNIT: p.c == 1 (K already carries scale * log2 e)
__global__ __launch_bounds__(256, PIPE ? 3 : 4) void attn3_kernel(const Attn3Params p) {
  static_assert(!PIPE || NBUF == 3, "the pipelined loop needs three stages");
  static_assert(!(PIPE && QK32), "the 32x32 QK^T body is built for the plain loop only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ONES = NBUF * A3_STAGE, ZERO = ONES + 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nqb = (p.n_q + 64 * A3_QT - 1) / (64 * A3_QT);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqb, qb = lid - bh * nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int kvb = b / p.q_per_kv;
  const int q0 = qb * (64 * A3_QT) + wave * (16 * A3_QT);
  const bf16_t* __restrict__ kbase = p.k + (size_t)kvb * p.n_kv * p.ldk + h * A3_D;
  const bf16_t* __restrict__ vbase = p.vt + (size_t)(kvb * p.heads + h) * A3_D * p.vt_pitch;
  const int n_tiles = (p.n_kv + 63) >> 6;
  const bool ragged = (p.n_kv & 63) != 0;

  // ---- constants in LDS: the ones / zero planes (K column d = 40 and d >= 48) and V^T rows 40..47 of every stage
  for (int idx = tid; idx < 128; idx += 256) {
    *reinterpret_cast<uint4*>(smem + ONES + idx * 16) = idx < 64 ? make_uint4(VX_E16_ONE, 0, 0, 0) : make_uint4(0, 0, 0, 0);
  }
  for (int idx = tid; idx < NBUF * 64; idx += 256) {
    const int st = idx >> 6, r = 40 + ((idx >> 3) & 7), j = idx & 7;
    *reinterpret_cast<uint4*>(smem + st * A3_STAGE + A3_V + r * 128 + j * 16) =
        r == 40 ? make_uint4(VX_E16_ONE2, VX_E16_ONE2, VX_E16_ONE2, VX_E16_ONE2) : make_uint4(0, 0, 0, 0);
  }

  // ---- DMA pieces of this wave: piece pw = wave + 4 s (s = 0..2, pw < 10); 0..3 = K chunks 0..3 of rows 16 pw..,
  // 4 = K chunk 4 of all 64 rows, 5..9 = V^T rows 8 (pw - 5)...  Per-lane source byte offsets for tile 0.
  uint32_t src0[3], dst[3];
  bool is_k[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int pw = wave + 4 * s;
    is_k[s] = pw < 5;
    if (pw < 4) {
      const int r = 16 * pw + (lane >> 2), pos = lane & 3, chunk = pos ^ ((0 - (r >> 2)) & 3);
      src0[s] = ((uint32_t)a3_pi<QK32>(r) * (uint32_t)p.ldk + 8u * chunk) * 2u;
      dst[s] = A3_K0 + pw * 1024;
    } else if (pw == 4) {
      src0[s] = ((uint32_t)a3_pi<QK32>(lane) * (uint32_t)p.ldk + 32u) * 2u;
      dst[s] = A3_K1;
    } else {
      const int r = 8 * (pw - 5) + (lane >> 3), pos = lane & 7, chunk = pos ^ ((r >> 1) & 7);
      src0[s] = ((uint32_t)r * (uint32_t)p.vt_pitch + 8u * chunk) * 2u;
      dst[s] = A3_V + (pw - 5) * 1024;
    }
  }
  const int my_pieces = wave < 2 ? 3 : 2;
  const uint32_t kstep = 128u * (uint32_t)p.ldk, vstep = 128u;     // bytes per key tile
  // the last tile of a ragged key count: keys beyond n_kv re-read the last valid key / the tile's first 8-key chunk
  // (finite values; their scores are masked to -inf).  Loop-invariant, so nothing of it is carried through the loop.
  uint32_t src_edge[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int pw = wave + 4 * s, te = n_tiles - 1;
    uint32_t off = src0[s] + (uint32_t)te * (is_k[s] ? kstep : vstep);
    if (pw < 5) {
      const int r = pw < 4 ? 16 * pw + (lane >> 2) : lane;
      const int over = te * 64 + a3_pi<QK32>(r) - (p.n_kv - 1);
      if (over > 0) off -= (uint32_t)over * (uint32_t)p.ldk * 2u;
    } else {
      const int r = 8 * (pw - 5) + (lane >> 3), pos = lane & 7, chunk = pos ^ ((r >> 1) & 7);
      if (te * 64 + 8 * chunk >= p.n_kv) off -= 16u * chunk;
    }
    src_edge[s] = off;
  }
  const uint32_t lds0 = lds_addr_of(smem);
  auto issue = [&](int t) {
    if (A3ABL(8) && t >= NBUF) return;
    const uint32_t so = lds0 + (uint32_t)(t % NBUF) * A3_STAGE;
    const bool edge = ragged && t == n_tiles - 1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < my_pieces) {
        const uint32_t off = edge ? src_edge[s] : src0[s] + (uint32_t)t * (is_k[s] ? kstep : vstep);
        glds16_s(is_k[s] ? (const void*)kbase : (const void*)vbase, off, so + dst[s]);
      }
    }
  };
  // all of this wave's pieces except those of the newest `keep` tiles have landed
  auto wait_dma = [&](bool keep_newest) {
    if (A3ABL(8)) return;
    if (!keep_newest) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wave < 2) {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  };

  // ---- Q fragments (raw); kk = 1 holds d 32..39 (g = 0), the shift column (g = 1), zeros (g >= 2).  The shift and
  // the accumulators are in units of q.k; the exponent is c * (q.k - m)
  uint4 qf[A3_QT][2];
  float mfix[A3_QT];
#pragma unroll
  for (int qt = 0; qt < A3_QT; ++qt) {
    const int qrow = q0 + 16 * qt + i;
    float ss = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int dcol = 32 * kk + 8 * g;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow < p.n_q && dcol < A3_D) {
        v = *reinterpret_cast<const uint4*>(p.q + (size_t)(b * p.n_q + qrow) * p.ldq + h * A3_D + dcol);
        float f[8];
        unpack_bf16x8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
      }
      qf[qt][kk] = v;
    }
    ss = wave_xor_sum(ss, 16);
    ss = wave_xor_sum(ss, 32);
    mfix[qt] = sqrtf(ss) * p.kmax[kvb * p.heads + h] - (UNIT ? VX_P_HEADROOM : VX_P_HEADROOM / p.c);
  }
  auto set_shift = [&](const float (&m)[A3_QT]) {
#pragma unroll
    for (int qt = 0; qt < A3_QT; ++qt)
      if (g == 1) qf[qt][1] = make_uint4((uint32_t)f32_to_bf16(-m[qt]), 0, 0, 0);
  };

  // ---- QK32: lane (query l & 31, half hi = l >> 5); k-step s covers d 16 s .. 16 s + 15: chunk 2 s + hi
  const int q32 = lane & 31, hi = lane >> 5;
  uint4 qf32[3];
  float mfix32 = 0.f;
  if constexpr (QK32) {
    const int qrow = q0 + q32;
    float ss = 0.f;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int dcol = 16 * st + 8 * hi;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow < p.n_q && dcol < A3_D) {
        v = *reinterpret_cast<const uint4*>(p.q + (size_t)(b * p.n_q + qrow) * p.ldq + h * A3_D + dcol);
        float f[8];
        unpack_bf16x8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
      }
      qf32[st] = v;
    }
    ss = wave_xor_sum(ss, 32);
    mfix32 = sqrtf(ss) * p.kmax[kvb * p.heads + h] - (UNIT ? VX_P_HEADROOM : VX_P_HEADROOM / p.c);
  }
  auto set_shift32 = [&](float m) {
    if (hi == 1) qf32[2] = make_uint4((uint32_t)f32_to_bf16(-m), 0, 0, 0);
  };
  const int sw32 = (0 - (q32 >> 2)) & 3;
  const int a_k32_0 = A3_K0 + q32 * 64 + (((0 + hi) ^ sw32) << 4);                 // + kt32 * 2048   (d  0..15)
  const int a_k32_1 = A3_K0 + q32 * 64 + (((2 + hi) ^ sw32) << 4);                 //                 (d 16..31)
  const int a_k32_2 = (hi == 0 ? A3_K1 : ONES) + q32 * 16;                         // + kt32 * 512    (d 32..47)
  const int k32_stage = hi == 0 ? 1 : 0;

  // ---- per-lane LDS read offsets (stage offset added per tile)
  const int a_k0 = A3_K0 + i * 64 + ((g ^ ((0 - (i >> 2)) & 3)) << 4);             // + kt * 1024
  const int a_k1 = g == 0 ? A3_K1 + i * 16 : (g == 1 ? ONES : ZERO) + i * 16;      // + kt * 256
  const int k1_stage = g == 0 ? 1 : 0;
  const int vx = (i >> 1) & 7;
  const int a_v0 = A3_V + i * 128 + ((g ^ vx) << 4);                               // + dt * 2048   (keys 0..31)
  const int a_v1 = A3_V + i * 128 + (((4 | g) ^ vx) << 4);                         //               (keys 32..63)

  f32x4_t o[3][A3_QT];
  auto zero_o = [&]() {
#pragma unroll
    for (int qt = 0; qt < A3_QT; ++qt)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) o[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };

  // S^T (already shifted) of key tile t
  auto qk = [&](f32x4_t (&s_)[4][A3_QT], int t) {
    const int so = A3ABL(32) ? 0 : (t % NBUF) * A3_STAGE;
    const char* k0 = smem + a_k0 + so;
    const char* k1 = smem + a_k1 + so * k1_stage;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const uint4 kf0 = *reinterpret_cast<const uint4*>(k0 + kt * 1024);
      const uint4 kf1 = *reinterpret_cast<const uint4*>(k1 + kt * 256);
#pragma unroll
      for (int qt = 0; qt < A3_QT; ++qt) {
        if (A3ABL(4)) {
          s_[kt][qt] = f32x4_t{-1.f, -2.f, -3.f, -4.f};
          asm volatile("" ::"v"(kf0.x), "v"(kf1.x));
          continue;
        }
        s_[kt][qt] = mfma16(kf0, qf[qt][0], f32x4_t{0.f, 0.f, 0.f, 0.f});
        s_[kt][qt] = mfma16(kf1, qf[qt][1], s_[kt][qt]);
      }
    }
  };
  // keys of tile t beyond n_kv -> -inf (lane g, register r of S^T tile kt is key 32 (kt>>1) + 8 g + 4 (kt&1) + r)
  auto mask = [&](f32x4_t (&s_)[4][A3_QT], int t) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t * 64 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r >= p.n_kv) {
#pragma unroll
          for (int qt = 0; qt < A3_QT; ++qt) s_[kt][qt][r] = -INFINITY;
        }
  };
  auto expo = [&](f32x4_t (&s_)[4][A3_QT]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < A3_QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (A3ABL(1)) continue;
          s_[kt][qt][r] = __builtin_amdgcn_exp2f(UNIT ? s_[kt][qt][r] : s_[kt][qt][r] * p.c);
        }
  };
  // O^T += V^T(t) P^T
  auto pv = [&](const f32x4_t (&s_)[4][A3_QT], int t) {
    const int so = A3ABL(32) ? 0 : (t % NBUF) * A3_STAGE;
#pragma unroll
    for (int ks_ = 0; ks_ < 2; ++ks_) {
      uint4 pb[A3_QT];
#pragma unroll
      for (int qt = 0; qt < A3_QT; ++qt) {
        const f32x4_t& a = s_[2 * ks_][qt];
        const f32x4_t& c2 = s_[2 * ks_ + 1][qt];
        pb[qt] = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(c2[0], c2[1]),
                            pack_bf16x2(c2[2], c2[3]));
      }
      const char* vp = smem + (ks_ ? a_v1 : a_v0) + so;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const uint4 vf = *reinterpret_cast<const uint4*>(vp + dt * 2048);
#pragma unroll
        for (int qt = 0; qt < A3_QT; ++qt) {
          if (A3ABL(2)) {
            asm volatile("" ::"v"(vf.x), "v"(pb[qt].x), "v"(pb[qt].w));
            continue;
          }
          o[dt][qt] = mfma16(vf, pb[qt], o[dt][qt]);
        }
      }
    }
  };

  // ---- QK32 body: S^T as two 32 x 32 tiles (register r of tile kt32: key 32 kt32 + pi32((r&3) + 8 (r>>2) + 4 hi))
  auto qk32 = [&](f32x16_t (&s_)[2], int t) {
    const int so = (t % NBUF) * A3_STAGE;
    const char* k0 = smem + a_k32_0 + so;
    const char* k1 = smem + a_k32_1 + so;
    const char* k2 = smem + a_k32_2 + so * k32_stage;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const uint4 f0 = *reinterpret_cast<const uint4*>(k0 + kt * 2048);
      const uint4 f1 = *reinterpret_cast<const uint4*>(k1 + kt * 2048);
      const uint4 f2 = *reinterpret_cast<const uint4*>(k2 + kt * 512);
      f32x16_t z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      z = mfma32(f0, qf32[0], z);
      z = mfma32(f1, qf32[1], z);
      s_[kt] = mfma32(f2, qf32[2], z);
    }
  };
  auto mask32 = [&](f32x16_t (&s_)[2], int t) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * 64 + 32 * kt + a3_pi32((r & 3) + 8 * (r >> 2) + 4 * hi) >= p.n_kv) s_[kt][r] = -INFINITY;
  };
  auto expo32 = [&](f32x16_t (&s_)[2]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_[kt][r] = __builtin_amdgcn_exp2f(UNIT ? s_[kt][r] : s_[kt][r] * p.c);
  };
  // O^T += V^T(t) P^T: registers 0..7 / 8..15 of a tile are two key octets of THIS lane's query; swapping the odd 16-lane
  // rows of the first with the even rows of the second gives, per 16-query tile, lane row g the octet g of query l & 15
  auto pv32 = [&](const f32x16_t (&s_)[2], int t) {
    const int so = (t % NBUF) * A3_STAGE;
#pragma unroll
    for (int ks_ = 0; ks_ < 2; ++ks_) {
      const f32x16_t& a = s_[ks_];
      uint32_t fx[4], gx[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t fo = pack_bf16x2(a[2 * e], a[2 * e + 1]), go = pack_bf16x2(a[8 + 2 * e], a[8 + 2 * e + 1]);
        auto sw = __builtin_amdgcn_permlane16_swap(fo, go, false, false);
        fx[e] = sw[0];
        gx[e] = sw[1];
      }
      const uint4 pb[A3_QT] = {make_uint4(fx[0], fx[1], fx[2], fx[3]), make_uint4(gx[0], gx[1], gx[2], gx[3])};
      const char* vp = smem + (ks_ ? a_v1 : a_v0) + so;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const uint4 vf = *reinterpret_cast<const uint4*>(vp + dt * 2048);
#pragma unroll
        for (int qt = 0; qt < A3_QT; ++qt) o[dt][qt] = mfma16(vf, pb[qt], o[dt][qt]);
      }
    }
  };

  // ---- the key loop.  MAXPASS: QK^T only, running row maxima into mrun (the exact-shift fallback).
  float mrun32 = -INFINITY;
  float mrun[A3_QT];
  auto run = [&](auto maxpass_c) {
    constexpr bool MAXPASS = decltype(maxpass_c)::value;
    __syncthreads();                       // constants written / every wave is done with the previous pass's stages
    issue(0);
    if (NBUF == 3 && n_tiles > 1) issue(1);
    if (PIPE && !MAXPASS) {
      f32x4_t sa[4][A3_QT], sb[4][A3_QT];
      wait_dma(n_tiles > 1);
      __syncthreads();                     // tile 0 landed
      qk(sa, 0);
      // body: tile t lives in `cur` (shifted scores, not yet exponentiated); tile t+1 is multiplied into `nxt` in the
      // same basic block as the exponentials of `cur`, so that the scheduler can interleave the two streams
      auto body = [&](auto has_next_c, f32x4_t (&cur)[4][A3_QT], f32x4_t (&nxt)[4][A3_QT], int t) {
        constexpr bool HAS_NEXT = decltype(has_next_c)::value;
        if (ragged && t == n_tiles - 1) mask(cur, t);
        if (HAS_NEXT) {
          wait_dma(false);
          if (!A3ABL(16)) __syncthreads(); // tile t+1 landed; every wave is done with tile t-1 (= the stage of tile t+2)
          if (t + 2 < n_tiles) issue(t + 2);
          qk(nxt, t + 1);
        }
        expo(cur);
        pv(cur, t);
      };
      int t = 0;
      for (; t + 2 < n_tiles; t += 2) {
        body(std::true_type{}, sa, sb, t);
        body(std::true_type{}, sb, sa, t + 1);
      }
      if (t + 2 == n_tiles) {
        body(std::true_type{}, sa, sb, t);
        body(std::false_type{}, sb, sa, t + 1);
      } else {
        body(std::false_type{}, sa, sb, t);
      }
    } else {
      for (int t = 0; t < n_tiles; ++t) {
        wait_dma(NBUF == 3 && t + 1 < n_tiles);
        if (!A3ABL(16)) __syncthreads();   // tile t landed for everyone; everyone is done with tile t-1
        if (t + NBUF - 1 < n_tiles) issue(t + NBUF - 1);
        if constexpr (QK32) {
          f32x16_t c32[2];
          qk32(c32, t);
          if (ragged && t == n_tiles - 1) mask32(c32, t);
          if (MAXPASS) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 16; ++r) mrun32 = fmaxf(mrun32, c32[kt][r]);
          } else {
            expo32(c32);
            pv32(c32, t);
          }
          continue;
        }
        f32x4_t cur[4][A3_QT];
        qk(cur, t);
        if (ragged && t == n_tiles - 1) mask(cur, t);
        if (MAXPASS) {
#pragma unroll
          for (int qt = 0; qt < A3_QT; ++qt) {
            float mx = mrun[qt];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
              mx = fmaxf(fmaxf(fmaxf(mx, cur[kt][qt][0]), fmaxf(cur[kt][qt][1], cur[kt][qt][2])), cur[kt][qt][3]);
            mrun[qt] = mx;
          }
        } else {
          expo(cur);
          pv(cur, t);
        }
      }
    }
  };
  // softmax denominator of query tile qt: O^T row 40 (fragment 2, lane group 2, register 0), for every lane of the column
  auto row_sum = [&](int qt) -> float { return __shfl(o[2][qt][0], i + 32, 64); };

  if constexpr (QK32) set_shift32(mfix32);
  else set_shift(mfix);
  zero_o();
  run(std::false_type{});
  bool bad = false;
#pragma unroll
  for (int qt = 0; qt < A3_QT; ++qt) {
    const float l = row_sum(qt);
    bad |= (q0 + 16 * qt + i < p.n_q) && !(l >= VX_P_MIN_ROWSUM(p.n_kv));   // (bf16: 2^-100); also catches NaN
  }
  if (__syncthreads_or(bad)) {   // block-uniform (the stages are shared by the 4 waves)
    const float zero[A3_QT] = {0.f, 0.f};
    if constexpr (QK32) set_shift32(0.f);
    else set_shift(zero);
    mrun32 = -INFINITY;
#pragma unroll
    for (int qt = 0; qt < A3_QT; ++qt) mrun[qt] = -INFINITY;
    run(std::true_type{});
    if constexpr (QK32) {
      mrun32 = wave_xor_max(mrun32, 32);
      set_shift32(mrun32);
    } else {
#pragma unroll
      for (int qt = 0; qt < A3_QT; ++qt) mrun[qt] = wave_rows_max(mrun[qt]);
      set_shift(mrun);
    }
    zero_o();
    run(std::false_type{});
  }

  // ---- normalise and store: lane holds 4 consecutive d-columns of one query
#pragma unroll
  for (int qt = 0; qt < A3_QT; ++qt) {
    const float inv = 1.0f / row_sum(qt);
    const int qrow = q0 + 16 * qt + i;
    if (qrow >= p.n_q) continue;
    bf16_t* orow = p.out + (size_t)(b * p.n_q + qrow) * p.ldo + h * A3_D;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      const int dcol = 16 * dt + 4 * g;
      if (dcol < A3_D) {
        uint2 w;
        w.x = pack_bf16x2(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
        w.y = pack_bf16x2(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
        *reinterpret_cast<uint2*>(orow + dcol) = w;
      }
    }
  }
}

template <int NBUF, bool PIPE, bool UNIT, bool QK32 = false>
int launch_attn3(const Attn3Params& p, hipStream_t stream) {
  constexpr int smem = NBUF * A3_STAGE + 2048;
  auto kern = attn3_kernel<NBUF, PIPE, UNIT, QK32>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vx_set_error("vx_attention: hipFuncSetAttribute(%d) failed: %s", smem, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)((long)ceil_div(p.n_q, 64 * A3_QT) * p.batch * p.heads));
  static char sym[64] = "";
  if (!sym[0]) {
    auto bs = [](bool v) { return v ? "true" : "false"; };
    snprintf(sym, sizeof(sym), "attn3_kernel<%d, %s, %s, %s>", NBUF, bs(PIPE), bs(UNIT), bs(QK32));
  }
  g_vx_last_kernel = sym;
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
  return vx_check_launch("vx_attention");
}

}  // namespace

// variant: 0 = off (attn2), 1 = NBUF 2 (default), 2 = NBUF 3, 3 = NBUF 3 + in-wave pipeline, 4 = NBUF 2 with QK^T on
// the 32x32x16 MFMA; VX_ATTN3 overrides.
// The three variants measure within 3 % of each other (profiles/r02a_attn_bench.txt, r02b_attn3_ablation.txt): the
// kernel is matrix-pipe / power bound (SQ_VALU_MFMA_BUSY_CYCLES = 61 % of the SIMD cycles at an effective 1.66 GHz,
// profiles/r02b_attn3_pmc.txt), so neither the ring depth nor the in-wave overlap of exp and MFMA moves it.
int vx_attn3_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VX_ATTN3");
    v = e ? atoi(e) : 1;
  }
  return v;
}

int vx_attn3_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int vt_pitch, void* out, int ldo,
                    int batch, int heads, int n_q, int n_kv, int q_per_kv, float c, const float* kmax,
                    hipStream_t stream) {
  Attn3Params p{(const bf16_t*)q, ldq, (const bf16_t*)k, ldk, (const bf16_t*)vt, vt_pitch, (bf16_t*)out, ldo,
                batch, heads, n_q, n_kv, q_per_kv, c, kmax};
  const bool unit = c == 1.0f;
  switch (vx_attn3_variant()) {
    case 2: return unit ? launch_attn3<3, false, true>(p, stream) : launch_attn3<3, false, false>(p, stream);
    case 3: return unit ? launch_attn3<3, true, true>(p, stream) : launch_attn3<3, true, false>(p, stream);
    case 4: return unit ? launch_attn3<2, false, true, true>(p, stream) : launch_attn3<2, false, false, true>(p, stream);
    default: return unit ? launch_attn3<2, false, true>(p, stream) : launch_attn3<2, false, false>(p, stream);
  }
}
