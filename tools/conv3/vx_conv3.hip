// 3x3 convolution with the GroupNorm + SiLU in front of it applied in the A-staging path, on a halo-resident activation
// tile (round 5; ResnetBlock3D.forward: norm1 -> SiLU -> conv1, norm2 -> SiLU -> conv2, modules/resnet.py:217-251).
//
//   out[f, oy, ox, n] = bias[n] (+ rowbias) (+ residual) + sum_{ky, kx, c} act(x[f, oy+ky-1, ox+kx-1, c]) W[n][ky][kx][c]
//   act(v) = silu(v * scale[f, c] + shift[f, c]) inside the image, 0 outside          (scale / shift: vx_groupnorm_scale_shift)
//
// Until now: vx_groupnorm_apply wrote the normalised tensor into a zero-bordered image (read 2 B + write 2 B per
// element) and the persistent ring kernel (vx_gemm_ring.hip) gathered each activation chunk NINE times through the CU's
// vector L1 (the long-K ring GEMM is bound by that path: LABNOTES.md section 7).  Here a tile's activations cross the L1
// ONCE per 32-channel chunk, are normalised in LDS, and the nine taps are formed from the resident tile:
//
//   * tile = 256 output pixels (R image rows x W columns, W = 64 or 32) x 320 output channels, one 512-thread workgroup
//     per CU walking a tile list (persistent), 8 waves as 2 (M) x 4 (N), 8 x 5 accumulator fragments per wave
//     (v_mfma_f32_16x16x32_bf16, C^T form) - the ring kernel's geometry, and its STORE epilogue (bias / time-embedding row /
//     residual / GroupNorm partial sums of the stored values);
//   * K order: 32-channel chunk major, the nine taps innermost: tap-step ts = 9 chunk + tap.  A K-tile = TWO tap-steps
//     (64 K values = one 128-byte weight row): with the weight matrix permuted once at load time to that K order
//     (w'[n][(chunk, tap, 32)]) the B operand is staged EXACTLY like the ring kernel's: five 8-KiB LDS-DMA pieces per
//     K-tile into one of two buffers, issued two K-tiles ahead, fragment reads with the same XOR swizzle;
//   * A operand: a PLANE = the tile's (R + 2) x (W + 2) haloed pixels x 32 channels (64-byte LDS rows), two planes
//     (even / odd chunks).  A plane arrives as three 8-KiB LDS-DMA pieces of the RAW rows (image columns only: the pad
//     columns stay zero from the start of the kernel), each thread then normalises the 16-byte units it copied itself
//     (scale / shift of the tile's frame from an 8-KiB table in LDS, SiLU, zero for rows outside the image) in place, and
//     a fragment of tap (ky, kx) is a ds_read_b128 at row offset ky (W + 2) + kx.  The 16-byte slots of a row are XORed
//     with 2 ((hx >> 2) & 1) (hx = halo column): conflict-free for every tap under the REAL lane groups of ds_read_b128
//     ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS; tools/conv3_emulate.py counts the
//     conflicts of every fragment read - the first version's (hx >> 2) & 3 cost 2 - 3 LDS cycles per lane group); the slot
//     of a lane depends on kx only;
//   * schedule: the ring kernel's slots - L(u, p) = LDS fragment reads (+ copies, + one plane unit to normalise), M(u, p)
//     = 20 MFMAs, one s_barrier after every slot, the two wave rows staggered by one slot - with a period of 9 K-tiles
//     (two chunks).  Position v of the period: planes are copied at v = 0 (odd chunk of this period; pieces H0, H1 in
//     L(0,0), H2 in L(0,3); plus the NEXT tile's table T in L(0,0)) and v = 5 (even chunk of the next period), normalised
//     at v = 2, 2, 3 resp. 7, 7, 8 and first read at v = 4 resp. 0.  The only counted wait is at the end of L(u, 3):
//     vmcnt(N(v)) retires the B pieces of K-tile u + 1, N = 9, 6, 5, 5, 5, 8, 6, 5, 5 (copies retire in order; the plane
//     pieces a wave normalises are its own and older than pieces it has already waited for).
//     tools/conv3_schedule_check.py replays the per-wave operation sequence for both wave rows and proves every LDS read /
//     refill / in-place write against barrier happens-before; tools/conv3_emulate.py restates the address arithmetic
//     lane by lane (both CPU tests);
//   * past the end of its sequence a block keeps issuing copies from clamped (valid) sources into buffers nobody reads:
//     the immediates stay compile-time constants; the kernel drains them before it exits.
#include "../../v-express_amd/csrc/vx_common.h"
#include "../../v-express_amd/csrc/vx_gemm_common.h"
#include "../../include/vexpress_hip.h"
#include "vx_conv3.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int C3_BM = 256, C3_BN = 320, C3_NT = 512;
constexpr int C3_PIECE = 8192;                      // 64 rows x 128 B: one 16-B LDS-DMA per thread
constexpr int C3_BBUF = 5 * C3_PIECE;               // one K-tile of weights: 320 rows x 128 B
// LDS map.  The planes come FIRST: every activation fragment address is (per-lane base of 3 variants) + a compile-time
// offset that must fit the 16-bit offset field of ds_read_b128 (plane + tap + fragment: up to ~41 KB)
constexpr int C3_PLANE = 25344;                     // (4 + 2) x (64 + 2) pixels x 64 B (W = 32: 10 x 34 x 64 = 21760)
constexpr int C3_PL_OFF = 0;                        // two planes
constexpr int C3_B_OFF = 2 * C3_PLANE;              // 50688: two weight buffers
constexpr int C3_TAB = 8192;                        // (scale, shift) of 1024 channels, float2
constexpr int C3_TAB_OFF = C3_B_OFF + 2 * C3_BBUF;  // 132608: two tables
constexpr int C3_DUMP_OFF = C3_TAB_OFF + 2 * C3_TAB;   // 148992: 1 KiB written by the copy slots that map to no plane row (W = 32)
constexpr int C3_SCR_OFF = C3_DUMP_OFF + 1024;      // 150016: GroupNorm-partial-sum scratch of the epilogue, 640 B per wave
constexpr int C3_LDS = C3_SCR_OFF + 8 * 640;        // 155136 of the CU's 163840
#ifdef VX_C3_TRACE
constexpr int C3_LDS_LAUNCH = C3_LDS + 2 * 512 * 8;   // + the slot-trace buffers
#else
constexpr int C3_LDS_LAUNCH = C3_LDS;
#endif
constexpr int C3_MAX_CIN = 1024;

// Compile-time ablation switches (tools/build_conv3_variants.sh; never defined for the product library):
//   1 no MFMA   2 no plane normalisation (raw rows feed the MFMAs)   4 no plane copies after the prologue   8 no B copies
#ifdef VX_C3_ABLATE
#define C3ABL(bit) (((VX_C3_ABLATE) & (bit)) != 0)
#else
#define C3ABL(bit) false
#endif

#ifdef VX_C3_TRACE
// slot timing trace (tools/conv3_trace.py, variant library only): waves 0 and 4 of block 0 stamp the cycle counter at four
// points of every phase into 8 KiB of LDS behind the kernel's own (stamps in global memory would count in vmcnt), dumped
// at the end
__device__ unsigned long long* g_c3_trace = nullptr;
#define C3_TRACE_MAX 512
#define C3_STAMP()                                                                  \
  do {                                                                              \
    if (tr_on && tr_n < C3_TRACE_MAX) tr_buf[tr_n++] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define C3_STAMP() do {} while (0)
#endif

__device__ __forceinline__ void c3_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void c3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void c3_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void c3_swap16(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// W: image width (64: tile = 4 rows, 32: tile = 8 rows).  RES: residual addend.  GNS: GroupNorm partial sums of the stored
// values (vx_conv3_params.gn_ws), as gemm_ring_kernel<..., GNS>.
template <int W, bool RES, bool GNS>
__global__ __launch_bounds__(C3_NT, 2) void conv3_gn_kernel(const vx_conv3_params p) {
  constexpr int R = C3_BM / W, WP = W + 2;
  constexpr int PLANE = (R + 2) * WP * 64;
  constexpr int HSLOTS = (R + 2) * (W / 16);         // 1-KiB wave copies that map to plane rows: 24 (W = 64) / 20 (W = 32)
  static_assert(PLANE <= C3_PLANE && HSLOTS <= 24, "plane geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;         // wave row (= stagger group) / wave column
  const uint32_t lds0 = lds_addr_of(smem);
#ifdef VX_C3_TRACE
  const bool tr_on = g_c3_trace != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
  unsigned long long* tr_buf = reinterpret_cast<unsigned long long*>(smem + C3_LDS) + (wave >> 2) * C3_TRACE_MAX;
  int tr_n = 0;
#endif

  // ---- this block's output tiles (column tile fastest; blocks dealt to the XCDs like the ring kernel's)
  const int n_tiles = p.n / C3_BN;
  const int hw = p.h * W;
  const int total_tiles = (p.frames * hw / C3_BM) * n_tiles;
  const int G = gridDim.x;
  const int lb = xcd_remap(blockIdx.x, G);
  const int my_tiles = (total_tiles - lb + G - 1) / G;
  const int cin = p.c1 + p.c2;
  const int NP = cin / 64;                          // periods (pairs of 32-channel chunks) per tile
  const int nk = 9 * NP;                            // K-tiles per tile
  const int kbytes = 9 * cin * 2;                   // bytes of one weight row

  const char* __restrict__ X1 = (const char*)p.x1;
  const char* __restrict__ X2 = (const char*)p.x2;
  const char* __restrict__ Wt = (const char*)p.w_perm;
  const char* __restrict__ AB = (const char*)p.ab;

  // ---- zero both planes once: the pad columns (hx = 0, W + 1) are never written again
  for (int i = tid * 16; i < 2 * C3_PLANE; i += C3_NT * 16) *reinterpret_cast<uint4*>(smem + C3_PL_OFF + i) = make_uint4(0, 0, 0, 0);

  // ---- B copies: thread (r0, slot) of a piece, the ring kernel's image (slot s of row r holds K chunk s ^ ((r >> 1) & 7))
  const int r0 = tid >> 3;
  const int cc = (tid & 7) ^ ((r0 >> 1) & 7);
  const uint32_t lds_wave = lds0 + wave * 1024;
  const uint32_t boff = (uint32_t)r0 * (uint32_t)kbytes + (uint32_t)cc * 16u;
  const uint32_t bq1 = 64u * (uint32_t)kbytes;      // 64 weight rows (weights < 4 GiB: vx_conv3x3_gn_supported)
  const char* b_cur = nullptr;                      // weights + column tile + K offset of the K-tile at the B issue pointer
  int bi_lid = lb, bi_kt = 0;                       // B issue pointer: tile, K-tile within it
  auto set_b_cur = [&]() {
    const int tile_n = bi_lid % n_tiles;
    b_cur = Wt + (long)(tile_n * C3_BN) * kbytes + (long)bi_kt * 128;
  };
  auto issue_b = [&](uint32_t bbuf, int q) {
    if (C3ABL(8)) return;
    // the piece offset is added per copy (opaque to the compiler: it would keep five loop-invariant sums in registers)
    uint32_t qo = (uint32_t)q * bq1;
    asm volatile("" : "+s"(qo));
    glds16_s(b_cur, boff + qo, lds_wave + C3_B_OFF + bbuf + q * C3_PIECE);
  };
  auto advance_b = [&]() {      // to the next K-tile of the block's sequence; past the end: stay inside the last tile (never read)
    if (++bi_kt == nk) {
      bi_kt = 0;
      if (bi_lid + G < total_tiles) bi_lid += G;
      set_b_cur();
    } else {
      b_cur += 128;
    }
  };

  // ---- geometry of the tile being multiplied and of the block's next tile (none: the same tile), once per tile
  struct TileGeo { int fr, frhw, oy0; };
  auto tile_geo = [&](int lid) {
    TileGeo g;
    const int m0 = (lid / n_tiles) * C3_BM;
    g.fr = m0 / hw;
    g.frhw = g.fr * hw;
    g.oy0 = (m0 - g.frhw) / W;
    return g;
  };
  int cmp_lid = lb;                                 // tile being multiplied
  TileGeo tc = tile_geo(lb);
  TileGeo tn = tile_geo(lb + G < total_tiles ? lb + G : lb);

  // ---- plane copies.  Copy slot sidx = 8 q + wave of piece q covers 16 pixels x 64 B: halo row hy = sidx / (W / 16),
  // halo columns hx0 .. hx0 + 15 with hx0 = 1 + 16 (sidx % (W / 16)) (wave-only: 8 q is a multiple of W / 16).  Lane = (pixel
  // lane >> 2, slot lane & 3); slot s of halo column hx holds the 16-byte channel group s ^ (2 ((hx >> 2) & 1)).
  const int hx0 = 1 + 16 * (wave % (W / 16));
  const int hpx = hx0 + (lane >> 2);
  const int hkg = (lane & 3) ^ (((hpx >> 2) & 1) << 1);   // channel group (8 channels) this lane copies and normalises
  const uint32_t hvoff1 = (uint32_t)((hpx - 1) * p.ldx1 + hkg * 8) * 2u;
  const uint32_t hvoff2 = (uint32_t)((hpx - 1) * p.ldx2 + hkg * 8) * 2u;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  // the plane in flight: set when its first copies are issued, read by its last copy and when its units are normalised
  int h_frhw = 0, h_oy0 = 0, h_ci = 0, h_buf = 0, h_tab = 0;
  auto hy_of = [&](int q) { return (8 * q + wave) / (W / 16); };
  // q-th piece of the plane in flight: rows outside the image come from a clamped (valid) row and are zeroed later.  The
  // tile / row / chunk part of the address is a 32-bit scalar added to the lane's offset (inputs < 4 GiB)
  auto issue_h = [&](int q, bool in_loop) {
    if (C3ABL(4) && in_loop) return;
    const bool first = h_ci < p.c1;
    const int hy = hy_of(q);
    int iy = h_oy0 + hy - 1;
    iy = iy < 0 ? 0 : (iy >= p.h ? p.h - 1 : iy);
    const uint32_t pix = (uint32_t)(h_frhw + iy * W);
    uint32_t so = first ? (pix * (uint32_t)p.ldx1 + (uint32_t)h_ci) * 2u : (pix * (uint32_t)p.ldx2 + (uint32_t)(h_ci - p.c1)) * 2u;
    asm volatile("" : "+s"(so));
    const uint32_t dst = 8 * q + wave < HSLOTS ? (uint32_t)(C3_PL_OFF + h_buf * C3_PLANE + (hy * WP + hx0) * 64)
                                               : (uint32_t)C3_DUMP_OFF;
    glds16_s(first ? X1 : X2, (first ? hvoff1 : hvoff2) + so, lds0 + dst);
  };
  // the (scale, shift) table of frame fr into table buffer tb: 8 KiB, 1 KiB per wave
  auto issue_t = [&](int fr, int tb) {
    glds16_s(AB + (long)fr * (C3_MAX_CIN * 8) + wave * 1024, lane16, lds_wave + C3_TAB_OFF + tb * C3_TAB);
  };
  // normalise the unit this thread copied as piece q of the plane in flight (in place)
  auto transform = [&](int q) {
    if (8 * q + wave >= HSLOTS) return;             // (wave-uniform) this copy went to the dump
    const int hy = hy_of(q);
    char* u = smem + C3_PL_OFF + h_buf * C3_PLANE + (hy * WP + hx0) * 64 + lane * 16;
    const int iy = h_oy0 + hy - 1;
    if (iy < 0 || iy >= p.h) {                      // (wave-uniform) zero padding above / below the image
      *reinterpret_cast<uint4*>(u) = make_uint4(0, 0, 0, 0);
      return;
    }
    if (C3ABL(2)) return;
    const char* tab = smem + C3_TAB_OFF + h_tab * C3_TAB + (h_ci + hkg * 8) * 8;
    // the raw unit and the table entries of two channel pairs at a time ((scale0, shift0, scale1, shift1) per 16 bytes)
    // are read before their arithmetic: as a read -> wait -> compute chain per channel pair the unit took four LDS
    // latencies inside an L slot (all four table reads at once cost 16 registers the GNS instantiations do not have)
    uint4 raw = *reinterpret_cast<const uint4*>(u);
    uint32_t* rw = reinterpret_cast<uint32_t*>(&raw);
    auto body = [&](auto silu_c) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float4 ta = *reinterpret_cast<const float4*>(tab + (2 * h2) * 16);
        const float4 tb = *reinterpret_cast<const float4*>(tab + (2 * h2 + 1) * 16);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float4 t = e ? tb : ta;
          const int e2 = 2 * h2 + e;
          float v0 = __uint_as_float(rw[e2] << 16), v1 = __uint_as_float(rw[e2] & 0xffff0000u);
          v0 = fmaf(v0, t.x, t.y);
          v1 = fmaf(v1, t.z, t.w);
          if constexpr (decltype(silu_c)::value) {
            v0 = silu_f(v0);
            v1 = silu_f(v1);
          }
          rw[e2] = pack_bf16x2(v0, v1);
        }
      }
    };
    if (p.silu) body(std::true_type{});
    else body(std::false_type{});
    *reinterpret_cast<uint4*>(u) = raw;
  };

  // ---- fragment read offsets.  B: the ring kernel's.  A: halo row (oy + ky) WP + ox + kx + lrow of the plane, slot
  // lq ^ (2 (((lrow + kx) >> 2) & 1)) (ox is a multiple of 16)
  const int frow = lane & 15, fgrp = lane >> 4;
  const int sw = (frow >> 1) & 7;
  const int ck0 = ((fgrp ^ sw) << 4), ck1 = (((4 + fgrp) ^ sw) << 4);
  const int b_rd = (wc * 80 + frow) * 128;           // (+ the K-tile's buffer: `bb` below)
  // first output row of the wave row inside the tile: W = 64: 2 grp, W = 32: 4 grp
  const int a_lane = C3_PL_OFF + (grp * (R / 2) * WP + frow) * 64;
  int a_slot[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) a_slot[kx] = (fgrp ^ ((((frow + kx) >> 2) & 1) << 1)) << 4;
  // compile-time part: fragment s of row half hf = output pixels 64 hf + 16 s .. + 15 of the wave row
  auto a_const = [](int hf, int s) {
    const int ml = 64 * hf + 16 * s;
    return ((ml / W) * WP + (ml % W)) * 64;
  };

  f32x4_t acc[8][5];
  uint4 bfr[5], af[4];      // phase ph multiplies tap-step kk = ph >> 1 with the 64 rows of half ph & 1: 5 + 4 fragments live

  // ------------------------------------------------------------------ prologue
  c3_wait_lgkm0();
  c3_barrier();                                     // planes zeroed before the first copy can land
  int tab_cur = 0;                                  // table buffer of the tile being multiplied
  issue_t(tc.fr, 0);
  h_frhw = tc.frhw; h_oy0 = tc.oy0; h_ci = 0; h_buf = 0; h_tab = 0;
  issue_h(0, false); issue_h(1, false); issue_h(2, false);
  set_b_cur();
  issue_b(0, 0); issue_b(0, 1); issue_b(0, 2); issue_b(0, 3); issue_b(0, 4);
  advance_b();
  issue_b(C3_BBUF, 0); issue_b(C3_BBUF, 1); issue_b(C3_BBUF, 2); issue_b(C3_BBUF, 3); issue_b(C3_BBUF, 4);
  advance_b();
  c3_wait_vm<10>();                                 // table + plane landed (this wave's share)
  c3_barrier();                                     // ... and everybody's: the table is read across waves
  transform(0); transform(1); transform(2);
  c3_wait_vm<5>();                                  // B of K-tile 0
  c3_wait_lgkm0();
  c3_barrier();

  int u = 0;                                        // K-tile sequence number of the block (parity = B buffer)
  int pp = 0;                                       // period within the tile

  // One K-tile at position v of the period (wave-uniform, 0 .. 8): four (L slot, barrier, M slot, barrier) phases.  ONE
  // body with scalar branches for the rare actions (nine unrolled bodies made the register allocator spill the lanes'
  // address constants into scratch - and a scratch reload costs an s_waitcnt vmcnt(0) in the middle of the copy pipeline).
  auto ktile = [&](const int v) {
    const uint32_t bbuf = (u & 1) ? C3_BBUF : 0;
    const char* bb = smem + C3_B_OFF + bbuf;
    // the two tap-steps of this K-tile: ts = 2 v + kk within the period; chunk parity = plane buffer, tap = ts % 9
    int aoff[2], akx[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ts = 2 * v + kk;
      const int pl = ts >= 9 ? 1 : 0;
      const int tap = ts - 9 * pl;
      const int ky = (tap >= 3 ? 1 : 0) + (tap >= 6 ? 1 : 0);
      akx[kk] = tap - 3 * ky;
      aoff[kk] = pl * C3_PLANE + (ky * WP + akx[kk]) * 64;
    }
    const bool copy_odd = v == 0, copy_even = v == 5;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      // ---------------- L slot: the five weight fragments of tap-step kk (phases 0 / 2), the four activation fragments of
      // row half hf at that tap
      const int kk = ph >> 1, hf = ph & 1;
      if (hf == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) bfr[j] = *reinterpret_cast<const uint4*>(bb + b_rd + j * 2048 + (kk ? ck1 : ck0));
      }
      {
        const int sl = akx[kk] == 0 ? a_slot[0] : (akx[kk] == 1 ? a_slot[1] : a_slot[2]);
        const char* ab_ = smem + a_lane + sl + aoff[kk];
#pragma unroll
        for (int s = 0; s < 4; ++s) af[s] = *reinterpret_cast<const uint4*>(ab_ + a_const(hf, s));
      }
      if (ph == 0) {
        if (copy_odd) {
          // the NEXT tile's table (every period: 8 KiB against 360 KiB of weights), then the odd chunk of this period
          issue_t(tn.fr, tab_cur ^ 1);
          h_frhw = tc.frhw; h_oy0 = tc.oy0; h_ci = (2 * pp + 1) * 32; h_buf = 1; h_tab = tab_cur;
          issue_h(0, true); issue_h(1, true);
        } else if (copy_even) {
          // the even chunk of the next period: of this tile, or chunk 0 of the block's next tile (none: this tile again)
          const bool same = pp + 1 < NP;
          h_frhw = same ? tc.frhw : tn.frhw; h_oy0 = same ? tc.oy0 : tn.oy0;
          h_ci = same ? (2 * pp + 2) * 32 : 0; h_buf = 0; h_tab = same ? tab_cur : tab_cur ^ 1;
          issue_h(0, true); issue_h(1, true);
        }
      } else if (ph == 3) {
        // K-tile u + 2 reuses this K-tile's weight buffer: its last fragments were read in L(u, 2)
        issue_b(bbuf, 0); issue_b(bbuf, 1); issue_b(bbuf, 2); issue_b(bbuf, 3); issue_b(bbuf, 4);
        advance_b();
        // then: B of K-tile u + 1 landed = all but the copies issued since (see the header for N(v))
        if (copy_odd) {
          issue_h(2, true);
          c3_wait_vm<9>();
        } else if (copy_even) {
          issue_h(2, true);
          c3_wait_vm<8>();
        } else if (v == 1 || v == 6) {
          c3_wait_vm<6>();
        } else {
          c3_wait_vm<5>();
        }
      } else if (ph == 1) {
        if (v == 2 || v == 7) transform(0);
        else if (v == 3 || v == 8) transform(2);
      } else {
        if (v == 2 || v == 7) transform(1);
      }
      c3_wait_lgkm0();
      C3_STAMP();                 // end of the L slot's own work (before it waits at the barrier)
      c3_barrier();
      C3_STAMP();
      // ---------------- M slot
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (C3ABL(1)) {
            asm volatile("" ::"v"(bfr[j].x), "v"(af[s].x));
            continue;
          }
          acc[4 * hf + s][j] = mfma16(bfr[j], af[s], acc[4 * hf + s][j]);
        }
      __builtin_amdgcn_s_setprio(0);
      C3_STAMP();                 // end of the M slot's MFMA issue
      c3_barrier();
      C3_STAMP();
    }
    ++u;
  };

  const float* __restrict__ bias = p.bias;
  const float* __restrict__ rowbias = p.rowbias;
  const int lrow = lane & 15, lq = lane >> 4;

  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (grp == 1) c3_barrier();   // stagger: wave row 1 runs one slot behind wave row 0
    for (pp = 0; pp < NP; ++pp)
      for (int v = 0; v < 9; ++v) ktile(v);
    if (grp == 0) c3_barrier();   // re-align

    // ---------------------------------------------------------------- epilogue (the ring kernel's STORE epilogue)
    // acc[i][j][r] = C[m0 + 128 grp + 16 i + lrow][n0 + 80 wc + 16 j + 4 lq + r]
    const int tile_m = cmp_lid / n_tiles, tile_n = cmp_lid - tile_m * n_tiles;
    const int row_base = tile_m * C3_BM + 128 * grp + lrow;
    {
      const float* rb_row0 = rowbias != nullptr ? rowbias + (size_t)((tile_m * C3_BM) / p.rows_per_group) * p.rowbias_ld : nullptr;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = tile_n * C3_BN + 80 * wc + 16 * j + 4 * lq;
        float4 b4 = bias != nullptr ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (rb_row0 != nullptr) {
          const float4 r4 = *reinterpret_cast<const float4*>(rb_row0 + col);
          b4.x += r4.x; b4.y += r4.y; b4.z += r4.z; b4.w += r4.w;
        }
        if (bias != nullptr || rb_row0 != nullptr) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w;
          }
        }
      }
    }
    {
      constexpr int N_ITEMS = 24, RES_DEPTH = GNS ? 8 : 10;   // (residual loads in flight; with the 40 column sums of GNS: 8 stays spill-free)
      const int col_p = tile_n * C3_BN + 80 * wc + 8 * (lq >> 1) + 16 * (lq & 1);   // + 32 t  (pair t = 0, 1)
      const int col_4 = tile_n * C3_BN + 80 * wc + 64 + 4 * lq;
      const uint32_t ldr2 = (uint32_t)p.ldr * 2u, ldc2 = (uint32_t)p.ldc * 2u;
      const uint32_t res_p = (uint32_t)row_base * ldr2 + (uint32_t)col_p * 2u, res_4 = (uint32_t)row_base * ldr2 + (uint32_t)col_4 * 2u;
      const uint32_t out_p = (uint32_t)row_base * ldc2 + (uint32_t)col_p * 2u, out_4 = (uint32_t)row_base * ldc2 + (uint32_t)col_4 * 2u;
      auto res_off = [&](int k) {
        return (k % 3 == 2 ? res_4 : res_p + (uint32_t)(k % 3) * 64u) + (uint32_t)(k / 3) * (16u * ldr2);
      };
      auto out_off = [&](int k) {
        return (k % 3 == 2 ? out_4 : out_p + (uint32_t)(k % 3) * 64u) + (uint32_t)(k / 3) * (16u * ldc2);
      };
      const char* __restrict__ resb = (const char*)p.residual;
      char* __restrict__ outb = (char*)p.out;
      uint4 rv[N_ITEMS];
      auto load_res = [&](int k) {
        if (k % 3 == 2) {
          const uint2 t2 = *reinterpret_cast<const uint2*>(resb + res_off(k));
          rv[k] = make_uint4(t2.x, t2.y, 0u, 0u);
        } else {
          rv[k] = *reinterpret_cast<const uint4*>(resb + res_off(k));
        }
      };
      if (RES) {
#pragma unroll
        for (int k = 0; k < RES_DEPTH; ++k) load_res(k);
      }
      float gcs[GNS ? 20 : 1], gcq[GNS ? 20 : 1];
      if constexpr (GNS) {
#pragma unroll
        for (int c = 0; c < 20; ++c) gcs[c] = gcq[c] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < N_ITEMS; ++k) {
        if (RES && k + RES_DEPTH < N_ITEMS) load_res(k + RES_DEPTH);
        const int i = k / 3, kind = k % 3;
        if (kind == 2) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][4][r];
          if (RES) {
            v[0] += __uint_as_float(rv[k].x << 16); v[1] += __uint_as_float(rv[k].x & 0xffff0000u);
            v[2] += __uint_as_float(rv[k].y << 16); v[3] += __uint_as_float(rv[k].y & 0xffff0000u);
          }
          const uint2 pk2 = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          *reinterpret_cast<uint2*>(outb + out_off(k)) = pk2;
          if constexpr (GNS) {
            const float g0 = __uint_as_float(pk2.x << 16), g1 = __uint_as_float(pk2.x & 0xffff0000u);
            const float g2 = __uint_as_float(pk2.y << 16), g3 = __uint_as_float(pk2.y & 0xffff0000u);
            gcs[16] += g0; gcs[17] += g1; gcs[18] += g2; gcs[19] += g3;
            gcq[16] = fmaf(g0, g0, gcq[16]); gcq[17] = fmaf(g1, g1, gcq[17]);
            gcq[18] = fmaf(g2, g2, gcq[18]); gcq[19] = fmaf(g3, g3, gcq[19]);
          }
        } else {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][2 * kind][r], y = acc[i][2 * kind + 1][r];
            c3_swap16(x, y);
            v[r] = x;
            v[4 + r] = y;
          }
          if (RES) {
            float rf[8];
            unpack_bf16x8(rv[k], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
          }
          const uint4 pk8 = pack_bf16x8(v);
          *reinterpret_cast<uint4*>(outb + out_off(k)) = pk8;
          if constexpr (GNS) {
            float gg[8];
            unpack_bf16x8(pk8, gg);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              gcs[kind * 8 + e] += gg[e];
              gcq[kind * 8 + e] = fmaf(gg[e], gg[e], gcq[kind * 8 + e]);
            }
          }
        }
      }
      if constexpr (GNS) {
        // as gemm_ring_kernel<..., GNS>: 16-lane DPP rows fold the 16 rows lrow, lanes lrow == 0 park the wave's 80 column
        // sums in its own 640 B of LDS, lane g adds the columns of group g in ascending order: one (sum, sum of squares) per
        // (frame, 128-row slab, group) in vx_groupnorm's workspace layout
#pragma unroll
        for (int c = 0; c < 20; ++c) {
          gcs[c] = row16_sum(gcs[c]);
          gcq[c] = row16_sum(gcq[c]);
        }
        float2* scr = reinterpret_cast<float2*>(smem + C3_SCR_OFF) + wave * 80;
        if (lrow == 0) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              scr[16 * (2 * t + (lq & 1)) + 8 * (lq >> 1) + e] = make_float2(gcs[t * 8 + e], gcq[t * 8 + e]);
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[64 + 4 * lq + r] = make_float2(gcs[16 + r], gcq[16 + r]);
        }
        c3_wait_lgkm0();
        const int cg = p.n / p.gn_groups;
        if (lane < 80 / cg) {
          float a = 0.f, b = 0.f;
          for (int c = 0; c < cg; ++c) {
            const float2 t = scr[lane * cg + c];
            a += t.x;
            b += t.y;
          }
          const int m0w = tile_m * C3_BM + 128 * grp;
          const int frame = m0w / p.gn_hw, slab = (m0w - frame * p.gn_hw) >> 7, slabs = p.gn_hw >> 7;
          const int g = (tile_n * C3_BN + 80 * wc) / cg + lane;
          reinterpret_cast<float2*>(p.gn_ws)[(size_t)(frame * slabs + slab) * p.gn_groups + g] = make_float2(a, b);
        }
        c3_wait_lgkm0();
      }
    }
    // next tile of this block (the copies of its first K-tiles, planes and table are already on their way)
    if (cmp_lid + G < total_tiles) {
      cmp_lid += G;
      tab_cur ^= 1;
      tc = tn;
      tn = tile_geo(cmp_lid + G < total_tiles ? cmp_lid + G : cmp_lid);
    }
  }
  c3_wait_vm<0>();     // run-ahead copies must land before the LDS is handed on
#ifdef VX_C3_TRACE
  if (tr_on) {
    unsigned long long* dst = g_c3_trace + (wave >> 2) * (C3_TRACE_MAX + 1);
    dst[0] = (unsigned long long)tr_n;
    for (int i = 0; i < tr_n; ++i) dst[1 + i] = tr_buf[i];
  }
#endif
}

template <int W, bool RES, bool GNS>
int conv3_launch(const vx_conv3_params& p, hipStream_t stream) {
  static bool attr_set = false;
  static int cus = 256;
  auto kern = conv3_gn_kernel<W, RES, GNS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS_LAUNCH);
    if (e != hipSuccess) {
      vx_set_error("vx_conv3x3_gn: hipFuncSetAttribute(%d B LDS) failed: %s", C3_LDS, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    attr_set = true;
  }
  const long tiles = (long)(p.frames * p.h * p.w / C3_BM) * (p.n / C3_BN);
  static char sym[64] = "";
  if (!sym[0]) snprintf(sym, sizeof(sym), "conv3_gn_kernel<%d, %s, %s>", W, RES ? "true" : "false", GNS ? "true" : "false");
  g_vx_last_kernel = sym;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(C3_NT), C3_LDS_LAUNCH, stream, p);
  return vx_check_launch("vx_conv3x3_gn");
}

// per (frame, channel): scale = gamma rstd, shift = beta - mean scale - what gn_apply_kernel (vx_norm.hip) builds in LDS,
// from the same partial sums in the same order (fp64 re-reduction, fixed order)
constexpr int SS_THREADS = 256, SS_SUBS = 8;
__global__ __launch_bounds__(SS_THREADS) void gn_scale_shift_kernel(const float* __restrict__ ws, int slices, int hw,
                                                                    int groups, float eps, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int C,
                                                                    float2* __restrict__ ab, int ab_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int frame = blockIdx.x, tid = threadIdx.x;
  float* gstat = reinterpret_cast<float*>(smem);                       // [groups][2] mean, rstd
  double* dpart = reinterpret_cast<double*>(gstat + 2 * groups);       // [SS_SUBS][groups][2]
  const int cg = C / groups;
  for (int idx = tid; idx < groups * SS_SUBS; idx += SS_THREADS) {
    const int g = idx % groups, sub = idx / groups;
    double a = 0.0, b = 0.0;
    for (int sl = sub; sl < slices; sl += SS_SUBS) {
      const float2 o = *reinterpret_cast<const float2*>(ws + (((size_t)frame * slices + sl) * groups + g) * 2);
      a += (double)o.x;
      b += (double)o.y;
    }
    dpart[(sub * groups + g) * 2 + 0] = a;
    dpart[(sub * groups + g) * 2 + 1] = b;
  }
  __syncthreads();
  for (int g = tid; g < groups; g += SS_THREADS) {
    double a = 0.0, b = 0.0;
    for (int sub = 0; sub < SS_SUBS; ++sub) {
      a += dpart[(sub * groups + g) * 2 + 0];
      b += dpart[(sub * groups + g) * 2 + 1];
    }
    const double cnt = (double)cg * (double)hw;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    gstat[g * 2 + 0] = (float)mean;
    gstat[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int ch = tid; ch < ab_ld; ch += SS_THREADS) {
    float2 o = make_float2(0.f, 0.f);
    if (ch < C) {
      const int g = ch / cg;
      const float sc = gamma[ch] * gstat[g * 2 + 1];
      o = make_float2(sc, beta[ch] - gstat[g * 2 + 0] * sc);
    }
    ab[(size_t)frame * ab_ld + ch] = o;
  }
}

bool conv3_ok(const vx_conv3_params& p, const char** why) {
  auto no = [&](const char* s) {
    if (why) *why = s;
    return false;
  };
  if (p.x1 == nullptr || p.w_perm == nullptr || p.ab == nullptr || p.out == nullptr) return no("null pointer");
  if ((p.c2 == 0) != (p.x2 == nullptr)) return no("x2 / c2 mismatch");
  if (p.w != 64 && p.w != 32) return no("image width must be 64 or 32");
  if (p.h <= 0 || p.frames <= 0 || (p.h * p.w) % C3_BM != 0) return no("frames must be whole 256-pixel tiles");
  if (p.c1 <= 0 || (p.c1 % 32) != 0 || (p.c2 % 32) != 0 || ((p.c1 + p.c2) % 64) != 0) return no("channel counts: multiples of 32, sum a multiple of 64");
  if (p.c1 + p.c2 > C3_MAX_CIN || p.ab_ld != C3_MAX_CIN) return no("at most 1024 input channels; ab_ld must be 1024");
  if (p.n <= 0 || (p.n % C3_BN) != 0) return no("output channels must be a multiple of 320");
  if ((p.ldx1 % 8) != 0 || p.ldx1 < p.c1 || (p.c2 && ((p.ldx2 % 8) != 0 || p.ldx2 < p.c2))) return no("input pixel strides");
  if ((p.ldc % 8) != 0 || p.ldc < p.n || (p.residual != nullptr && ((p.ldr % 8) != 0 || p.ldr < p.n))) return no("output / residual strides");
  const unsigned long long m = (unsigned long long)p.frames * p.h * p.w;
  if (m * p.ldc * 2ull >= (1ull << 32) || (p.residual != nullptr && m * p.ldr * 2ull >= (1ull << 32))) return no("output larger than 4 GiB");
  if ((unsigned long long)p.w * (p.ldx1 > p.ldx2 ? p.ldx1 : p.ldx2) * 2ull >= (1ull << 31)) return no("input row larger than 2 GiB");
  if ((unsigned long long)p.n * 9ull * (p.c1 + p.c2) * 2ull >= (1ull << 32)) return no("weights larger than 4 GiB");
  if (p.rowbias != nullptr && ((p.rowbias_ld % 4) != 0 || p.rows_per_group <= 0 || (p.rows_per_group % C3_BM) != 0)) return no("rowbias groups must be whole tiles");
  if (p.gn_ws != nullptr) {
    if (p.gn_groups <= 0 || p.gn_hw <= 0 || (p.n % p.gn_groups) != 0 || (m % p.gn_hw) != 0 || (p.gn_hw % 128) != 0) return no("GroupNorm partial sums: geometry");
    const int cg = p.n / p.gn_groups;
    if (cg <= 0 || (80 % cg) != 0) return no("GroupNorm partial sums: groups must not straddle a wave's 80 columns");
  }
  return true;
}

}  // namespace

#ifdef VX_C3_TRACE
// (trace build only; not part of the C ABI header) device buffer of 2 x (1 + 512) uint64: [count, stamps...] per wave row
extern "C" int vx_conv3_set_trace(void* dev_buf) {
  unsigned long long* ptr = (unsigned long long*)dev_buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_c3_trace), &ptr, sizeof(ptr)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int vx_conv3x3_gn_supported(const vx_conv3_params* p) { return p != nullptr && conv3_ok(*p, nullptr) ? 1 : 0; }

extern "C" int vx_conv3x3_gn(const vx_conv3_params* pp, void* stream_) {
  VX_REQUIRE(pp != nullptr, "vx_conv3x3_gn: null parameter block");
  const vx_conv3_params& p = *pp;
  const char* why = "";
  if (!conv3_ok(p, &why)) {
    vx_set_error("vx_conv3x3_gn: unsupported problem (%s): frames=%d h=%d w=%d c1=%d c2=%d n=%d", why, p.frames, p.h, p.w,
                 p.c1, p.c2, p.n);
    return VX_ERR_UNSUPPORTED;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const bool res = p.residual != nullptr, gns = p.gn_ws != nullptr;
#define VX_C3(WW)                                                                              \
  (res ? (gns ? conv3_launch<WW, true, true>(p, stream) : conv3_launch<WW, true, false>(p, stream)) \
       : (gns ? conv3_launch<WW, false, true>(p, stream) : conv3_launch<WW, false, false>(p, stream)))
  return p.w == 64 ? VX_C3(64) : VX_C3(32);
#undef VX_C3
}

extern "C" int vx_groupnorm_scale_shift(const float* ws, int stat_slices, int frames, int hw, int groups, float eps,
                                        const float* gamma, const float* beta, int c, float* ab, int ab_ld, void* stream_) {
  VX_REQUIRE(ws != nullptr && gamma != nullptr && beta != nullptr && ab != nullptr, "vx_groupnorm_scale_shift: null pointer");
  VX_REQUIRE(frames > 0 && hw > 0 && stat_slices > 0 && groups > 0 && c > 0 && (c % groups) == 0 && ab_ld >= c,
             "vx_groupnorm_scale_shift: bad geometry (c=%d groups=%d ab_ld=%d)", c, groups, ab_ld);
  const size_t smem = (size_t)2 * groups * sizeof(float) + (size_t)SS_SUBS * groups * 2 * sizeof(double);
  VX_REQUIRE(smem <= 64 * 1024, "vx_groupnorm_scale_shift: %d groups too many", groups);
  hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(frames), dim3(SS_THREADS), smem, (hipStream_t)stream_, ws, stat_slices, hw,
                     groups, eps, gamma, beta, c, reinterpret_cast<float2*>(ab), ab_ld);
  return vx_check_launch("vx_groupnorm_scale_shift");
}
