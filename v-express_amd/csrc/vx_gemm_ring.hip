// Persistent, ring-staged MFMA GEMM / implicit-GEMM convolution for gfx950: the 256 x 320 x 64 tile of vx_gemm.hip
// re-scheduled so that the matrix pipe never waits for a whole K-tile to arrive.
//
//   * one 512-thread workgroup per CU walks a list of output tiles (persistent: tile = block + i * gridDim);
//     8 waves as 2 (M) x 4 (N), wave tile 128 x 80 = 8 x 5 accumulator fragments (v_mfma_f32_16x16x32_bf16, C^T form);
//   * a K-tile (256 + 320 rows of 128 B) is staged as nine 8-KiB *pieces* (one global_load_lds_dwordx4 per thread
//     each): B0..B4 (64 weight rows each) and A0..A3, where A piece p holds the 2 x 32 activation rows that phase p
//     of both wave rows consumes.  LDS holds two K-tile buffers (144 KiB);
//   * a K-tile is multiplied in four phases (A rows 32p..32p+31 of the wave x all 80 columns = 20 MFMAs); the B
//     fragments are read once per K-tile (phase 0) and stay in registers;
//   * the two wave rows (waves 0-3 / 4-7 = the two waves of every SIMD) run the same slot sequence
//        L(t,0) M(t,0) L(t,1) M(t,1) L(t,2) M(t,2) L(t,3) M(t,3) ...      (one s_barrier after every slot)
//     staggered by one slot, so while one wave of a SIMD is in an M slot (20 back-to-back MFMAs at raised priority)
//     its partner is in an L slot: LDS fragment reads for its next M slot, DMA issue, counted vmcnt wait;
//   * DMA pieces are issued 1.4 .. 2 K-tiles ahead of their first use, into LDS slots whose last reader finished at
//     least one slot earlier, across output-tile boundaries (the next tile's first two K-tiles stream in during the
//     epilogue).  Issue schedule (u = K-tile sequence number of this block, groups g0 = {B0,B1,B2},
//     g1 = {B3,B4,A0}, g2 = {A1,A2}, g3 = {A3}):
//         L(t,0): g3 of u = t+1      L(t,1): g0 of t+2      L(t,2): g1 of t+2      L(t,3): g2 of t+2
//     Waits (end of the L slot, before its barrier; in-order vmcnt, e1 = [t+1 exists], e2 = [t+2 exists]):
//         L(t,0): A1(t) landed   -> vmcnt(2 + 9 e1)          L(t,1): A2(t)          -> vmcnt(1 + 9 e1 + 3 e2)
//         L(t,2): A3(t)          -> vmcnt(9 e1 + 6 e2)       L(t,3): g0, g1 of t+1  -> vmcnt(3 e1 + 8 e2)
//     A piece is read one L slot after the wait that retires it (wait -> barrier -> read), never in the same slot;
//     an LDS slot is refilled at the earliest one L slot after its last read, and every L slot drains its own
//     LDS reads (lgkmcnt(0)) before its barrier.  tools/ring_schedule_check.py replays this schedule for both wave rows
//     and proves every read / refill against these rules by barrier happens-before (CPU test
//     tests/test_host_logic.py::test_ring_dma_schedule_is_race_free; the counts are tight: +1 on any of them fails).
//   * epilogue: both wave rows re-align (one extra barrier each), then every lane exchanges accumulator columns with
//     its 16-lane neighbour (v_permlane16_swap) so that it owns 8 consecutive output columns: bias / residual loads
//     and the output stores are 16 bytes per lane.
//
// Eligibility (vx_gemm_ring_eligible): STORE epilogue into bf16, the FAST addressing conditions of vx_gemm.hip
// (pad 0, no upsample, channel counts multiples of 64, operands < 4 GiB), m % 256 == 0, n % 320 == 0, no split-K.
#include "vx_common.h"
#include "vx_gemm_common.h"
#include "../../include/vexpress_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

namespace {

constexpr int R_BM = 256, R_BN = 320, R_NT = 512;
constexpr int PIECE = 8192;                 // bytes: 64 rows x 128 B, one 16-B DMA per thread of the block
constexpr int NB_PIECES = 5, NA_PIECES = 4;
constexpr int BUF_BYTES = (NB_PIECES + NA_PIECES) * PIECE;   // 73,728
constexpr int B_OFF = 0, A_OFF = NB_PIECES * PIECE;
constexpr int R_LDS_BYTES = 2 * BUF_BYTES;  // 147,456
// behind the two K-tile buffers: exchange area of the STATS epilogue, float2 [256 rows][4 wave columns] = (sum, sumsq)
constexpr int STATS_OFF = R_LDS_BYTES, STATS_BYTES = R_BM * 4 * 8;
constexpr int R_LDS_TOTAL = R_LDS_BYTES + STATS_BYTES;   // 155,648 of the CU's 163,840

// Compile-time ablation switches (tools/exp_ring_ablate.sh builds one library per mask with -DVX_RING_ABLATE=mask; the
// product library is built without): 1 no MFMA, 2 no LDS reads (and no MFMA), 4 every tile gathers tile 0's
// activation rows (all L2 hits), 8 no A copies, 16 no B copies, 32 no epilogue stores
#ifdef VX_RING_ABLATE
#define RABL(bit) (((VX_RING_ABLATE) & (bit)) != 0)
#else
#define RABL(bit) false
#endif
// Experimental issue placement (tools/build_ring_variants.sh builds one library per value; the product library is built
// without): some copies move from the L slots into the issuing wave's own M slot, after its 10th MFMA.
//   1: A0 of u+2 in M(t,2), A2 of u+2 in M(t,3)            2: additionally B2 of u+2 in M(t,1)
//   3: no M-slot issue; A1, A2 of u+1 are issued with A3 of u+1 in L(t,0) instead of one K-tile earlier in L(t-1,3)
// (tools/ring_schedule_check.py --m-issue N proves each variant's vmcnt immediates).
#ifndef VX_RING_MISSUE
#define VX_RING_MISSUE 0
#endif
// (tile walk: VX_XCD_ROWS of vx_gemm_common.h)
// VX_RING_PRIO (experiment): 1 = the M slot runs at s_setprio 1 (product), 0 = no priority changes, 2 = the L slot does
#ifndef VX_RING_PRIO
#define VX_RING_PRIO 1
#endif
// residual loads of the STORE epilogue kept in flight ahead of their use (16 bytes each; A/B knob, product 10: the bias
// registers of the round-2 epilogue are gone, so the prefetch is 4 items deeper at the same register count)
#ifndef VX_RING_RES_DEPTH
#define VX_RING_RES_DEPTH 10
#endif
constexpr bool RING_MI_M = VX_RING_MISSUE == 1 || VX_RING_MISSUE == 2;
constexpr bool RING_MI_LATE = VX_RING_MISSUE == 3;
// Cache policy of the STORE epilogue's output stores (experiment, round 4: -DVX_RING_STORE_MOD='" sc1"' = write-through,
// '" sc0 sc1"', '" nt"'; the product library is built without = plain stores).  The question: does leaving up to 32 MB of
// dirty lines in the L2s for the end-of-kernel write-back cost the NEXT launch more than writing through while the
// kernel still runs?
#ifdef VX_RING_STORE_MOD
typedef uint32_t ring_u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t ring_u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ring_store16(char* base, uint32_t off, const uint4& v) {
  const ring_u32x4_t d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, %2" VX_RING_STORE_MOD ::"v"(off), "v"(d), "s"(base) : "memory");
}
__device__ __forceinline__ void ring_store8(char* base, uint32_t off, const uint2& v) {
  const ring_u32x2_t d = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, %2" VX_RING_STORE_MOD ::"v"(off), "v"(d), "s"(base) : "memory");
}
#else
__device__ __forceinline__ void ring_store16(char* base, uint32_t off, const uint4& v) {
  *reinterpret_cast<uint4*>(base + off) = v;
}
__device__ __forceinline__ void ring_store8(char* base, uint32_t off, const uint2& v) {
  *reinterpret_cast<uint2*>(base + off) = v;
}
#endif
#ifdef VX_RING_TRACE
// slot timing trace (tools/gemm_bench --trace): waves 0 and 4 of block 0 record s_memtime at every barrier
__device__ unsigned long long* g_ring_trace = nullptr;
#define RING_TRACE_MAX 512
#define RING_STAMP()                                                                   \
  do {                                                                                 \
    if (trace != nullptr && trace_n < RING_TRACE_MAX) trace[trace_n++] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define RING_STAMP() do {} while (0)
#endif
#define RING_WAIT_VM(N) do { RING_STAMP(); ring_wait_vm<N>(); RING_STAMP(); } while (0)

__device__ __forceinline__ void ring_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void ring_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ring_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float bits_f(uint32_t u) { return __uint_as_float(u); }

// odd 16-lane rows of x <-> even 16-lane rows of y
__device__ __forceinline__ void swap16(float& x, float& y) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// F8 (round 2): e4m3 operands of the fp8 projection GEMMs (vx_gemm_params.a_fp8): a K-tile is still one 128-byte LDS
// row per operand row = 128 elements, the two 16-byte fragment reads of a lane form ONE K = 128
// v_mfma_scale_f32_16x16x128_f8f6f4 operand (unit block scales) instead of feeding two bf16 MFMAs, so the M slots
// halve while the DMA schedule, the stagger and every vmcnt count stay what tools/ring_schedule_check.py proves.
typedef int ring_i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4_t ring_mfma_f8(const uint4& a_lo, const uint4& a_hi, const uint4& b_lo, const uint4& b_hi,
                                                f32x4_t c) {
  const ring_i32x8_t a = {(int)a_lo.x, (int)a_lo.y, (int)a_lo.z, (int)a_lo.w, (int)a_hi.x, (int)a_hi.y, (int)a_hi.z, (int)a_hi.w};
  const ring_i32x8_t b = {(int)b_lo.x, (int)b_lo.y, (int)b_lo.z, (int)b_lo.w, (int)b_hi.x, (int)b_hi.y, (int)b_hi.z, (int)b_hi.w};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// RES: STORE epilogue with a residual addend.  STATS (STORE, n == 320: one column tile holds whole rows): the epilogue
// also writes (mean, rstd) of every stored bf16 row to p.row_stats_out - the statistics of the LayerNorm that the NEXT
// GEMM folds (vx_gemm_params.ln_stats), so no separate pass re-reads the tensor.
// LNF: a LayerNorm is folded into this GEMM (p.ln_stats / p.ln_colsum); a template flag, not a run-time branch: as a
// branch the epilogue's live values spilled (the lesson of the classic tiles in round 2).
// GNS (round 4; STORE): the epilogue also writes the GroupNorm partial sums of the STORED bf16 values to p.gn_ws in the
// workspace layout of vx_groupnorm ([frame][slab][group] (sum, sum of squares); one slab = the 128 rows of a wave row),
// so the statistics pass of the GroupNorm that reads this tensor next never runs (vx_gemm_params.gn_ws).
// exchange of the cooperative split (SK): explicit cache scopes instead of agent-scope fences.  A release fence at agent
// scope is a write-back of the XCD's whole L2 (buffer_wbl2) and an acquire an invalidate (buffer_inv) - 1024 of each per
// launch cost more than the split saves.  Instead: the parked accumulators are stored WRITE-THROUGH (sc0 sc1: the line is
// updated in this XCD's L2 and in memory; buffer instructions, so that the compiler sees them and counts its own waits), the
// flag is stored / polled at agent scope (sc1), and the reader loads with sc1 (misses its CU's L1).  Partner work items are adjacent walk positions, i.e. run on the same XCD on every part seen so far;
// the flag carries the writer's XCC id, and only a reader on ANOTHER XCD pays the L2 invalidate.
typedef int sk_i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sk_store16(__amdgpu_buffer_rsrc_t ws, uint32_t voff, uint32_t soff, const f32x4_t& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sk_i32x4_t, v), ws, voff, soff, 17);   // sc0 sc1
}
__device__ __forceinline__ f32x4_t sk_load16(__amdgpu_buffer_rsrc_t ws, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ws, voff, soff, 16));    // sc1
}
__device__ __forceinline__ uint32_t sk_xcc_id() {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x;
}

// SK (round 5; STORE, bf16 operands, no LayerNorm fold / row statistics): cooperative two-way K split for launches with
// fewer 256 x 320 tiles than CUs (the 16x16 level: 8192 x 1280 = 128 tiles).  A work item = (tile, K half): items 2t and
// 2t + 1 sit next to each other in the walk, i.e. on the same XCD in the same round.  Per WAVE (the STORE / GNS epilogues
// are wave-local): lane 0 claims flag word [t][wave][0] with an atomic exchange of the launch's epoch (p.coop_epoch); the
// FIRST of the two partner waves (it reads another value back) writes its 160 accumulators per lane to splitk_ws (40 KB per
// wave, lane-linear float4), releases flag word [1] = (epoch, XCC id) and goes on to its next item - it never waits, so no
// co-residency of the two blocks is assumed; the SECOND (it reads the epoch back) waits for [1] to carry the epoch (its
// partner is by construction already past its K loop), adds the partner's accumulators to its own (a + b: the same bits
// whichever half arrives first) and runs the normal epilogue.  The words are never reset (round 6): whatever an aborted or
// timed-out launch leaves behind belongs to an older epoch and is ignored.  The K halves are whole channel
// chunks ((c1 + c2) / 64 even), each walked taps-innermost like the unsplit loop.
template <int EPI, bool RES, bool F8 = false, bool STATS = false, bool LNF = false, bool GNS = false, bool SK = false>
__global__ __launch_bounds__(R_NT, 2) void gemm_ring_kernel(const vx_gemm_params p) {
  static_assert(!SK || (EPI == VX_EPI_STORE && !F8 && !STATS && !LNF), "cooperative split: plain STORE / GNS epilogues");
#ifdef VX_RING_TAP_OUTER
  static_assert(!SK, "cooperative split: taps-innermost K order only");
#endif
  constexpr int ES = F8 ? 1 : 2;      // bytes per operand element
  constexpr int BKE = 128 / ES;       // elements per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;   // wave row (= stagger group) / wave column
#ifdef VX_RING_TRACE
  unsigned long long* trace = nullptr;
  int trace_n = 0;
  if (g_ring_trace != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4) && (tid & 63) == 0)
    trace = g_ring_trace + (wave >> 2) * RING_TRACE_MAX;
#endif

  // ---- this block's output tiles.  Blocks are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8, each with its
  // own L2).  VX_XCD_ROWS (round 4 experiment, off - see vx_gemm_common.h): XCD x owns the CONTIGUOUS x-th eighth of the tile list in every launch -
  // i.e. the same eighth of the rows (frames 4 x .. 4 x + 3 of a 32-frame batch) as in the attention kernels, the classic
  // tiles and every other ring launch, whatever its column-tile count: what one kernel wrote into an XCD's L2 is read by
  // the next kernel on the same XCD (output stores that bypass L2 cost 14 % of a clip, profiles/r04h_*).  Within its
  // range a block takes tiles lb, lb + stride, ... (column tile fastest: co-running blocks share activation row tiles).
  // Off: the round-1 walk lb = xcd_remap(b), stride G (first round contiguous per XCD, later rounds elsewhere).
  const int n_tiles = p.n / R_BN;
  const int total_tiles = (p.m / R_BM) * n_tiles;
  const int total_items = SK ? 2 * total_tiles : total_tiles;   // SK: item = 2 * tile + K half
  const int G = gridDim.x;
#if VX_XCD_ROWS
  const int nxcd = G < 8 ? G : 8;
  const int xcd = (int)blockIdx.x % nxcd, xidx = (int)blockIdx.x / nxcd;
  const int stride = (G - xcd + nxcd - 1) / nxcd;                      // blocks of this launch on this XCD
  static_assert(!SK, "cooperative split: round-1 walk only");
  const int t0 = (int)((long)total_tiles * xcd / nxcd), t1 = (int)((long)total_tiles * (xcd + 1) / nxcd);
  const int lb = t0 + xidx;
  const int my_tiles = lb < t1 ? (t1 - lb + stride - 1) / stride : 0;
  if (my_tiles <= 0) return;                                           // (uniform for the block, before any barrier)
#else
  const int stride = G;
  const int lb = xcd_remap(blockIdx.x, G);
  const int my_tiles = (total_items - lb + G - 1) / G;
#endif
  const int nk = (p.k / BKE) >> (SK ? 1 : 0);   // K-tiles per work item
  const int S = my_tiles * nk;   // K-tile sequence length of this block

  const char* __restrict__ A1 = (const char*)p.a;
  const char* __restrict__ A2 = (const char*)p.a2;
  const char* __restrict__ Wt = (const char*)p.w;
  const int cin = p.c1 + p.c2, c1 = p.c1;
  const int kw = p.kw, kh = p.kh, w_in = p.w_in;
  const int hw_out = p.h_out * p.w_out;
  const uint32_t lda1b = (uint32_t)p.lda1 * (uint32_t)ES, lda2b = (uint32_t)p.lda2 * (uint32_t)ES;
  const bool is_conv = !(kh == 1 && kw == 1 && p.stride == 1 && p.h_in == p.h_out && p.w_in == p.w_out);

  // ---- DMA coordinates: thread (r0, slot) of a piece; the LDS image is lane-linear, so the XOR swizzle is applied to
  // the K-chunk this thread fetches (slot s of row r holds chunk s ^ ((r >> 1) & 7))
  const int r0 = tid >> 3;
  const int cc = (tid & 7) ^ ((r0 >> 1) & 7);
  const uint32_t lds_wave = lds_addr_of(smem) + wave * 1024;
  const uint32_t boff = (uint32_t)r0 * (uint32_t)p.k * (uint32_t)ES + (uint32_t)cc * 16u;   // + q * 64 rows (wave-uniform)
  const long b_piece_stride = (long)p.k * 64 * ES;                                   // 64 weight rows
  // Input pixel of this thread's row in A piece q, RELATIVE to the first row of the output tile.  The eligibility
  // test guarantees that a 256-row tile is either a whole number of frames or a whole number of image rows of one
  // frame, so the relative pixel is the same for every tile and only a wave-uniform base changes per tile.
  uint32_t pixl[NA_PIECES];
  const uint32_t cc16 = (uint32_t)cc * 16u;
#pragma unroll
  for (int q = 0; q < NA_PIECES; ++q) {
    const int ml = (r0 >> 5) * 128 + 32 * q + (r0 & 31);   // A piece q: rows 32q.. of wave row 0, then of wave row 1
    if (is_conv) {
      const int fr = ml / hw_out;
      const int rem = ml - fr * hw_out;
      const int oy = rem / p.w_out;
      const int ox = rem - oy * p.w_out;
      pixl[q] = (uint32_t)(fr * p.h_in * p.w_in + oy * p.stride * w_in + ox * p.stride);
    } else {
      pixl[q] = (uint32_t)ml;
    }
  }

  // issue-side position: output tile, K-tile within it, and the (channel chunk, tap) of that K-tile.  K-tiles are
  // walked channel-chunk-major, taps innermost, so the nine shifted reads of an activation chunk are back to back
  // (they hit the XCD's L2 instead of travelling from the Infinity Cache nine times).
  int iss_lid = lb, iss_kt = 0;
  int s_ci = SK ? (lb & 1) * (cin >> 1) : 0, s_kx = 0, s_ky = 0;
  const char* bbase_tile = nullptr;
  long pix0 = 0;   // input pixel of the issue tile's first output row (wave-uniform)

  auto setup_issue_tile = [&]() {
    const int iss_tile = SK ? iss_lid >> 1 : iss_lid;
    const int tile_m = iss_tile / n_tiles, tile_n = iss_tile - tile_m * n_tiles;
    const int m0 = tile_m * R_BM;
    bbase_tile = Wt + (long)(tile_n * R_BN) * p.k * ES;
    if (p.w_group_rows > 0) bbase_tile += (long)(m0 / p.w_group_rows) * ((long)p.n * p.k * ES);   // per-group weights
    if (is_conv) {
      const int fr = m0 / hw_out;
      const int rem = m0 - fr * hw_out;
      const int oy = rem / p.w_out;   // rem % w_out == 0
      pix0 = (long)fr * p.h_in * p.w_in + (long)oy * p.stride * w_in;
    } else {
      pix0 = m0;
    }
  };

  // wave-uniform sources of the K-tile at the issue pointer, refreshed once per K-tile (not per copy: the scalar
  // 64-bit arithmetic would otherwise dominate the L slots)
  const char* a_cur = nullptr;   // source + channel chunk + (tile origin + tap) * row pitch
  const char* b_cur = nullptr;   // weights + column tile + K offset
  uint32_t a_ld = 0;             // row pitch (bytes) of the source that holds this channel chunk
  auto refresh_issue_bases = [&]() {
    const bool first = s_ci < c1;
    a_ld = first ? lda1b : lda2b;
    a_cur = (first ? A1 + (long)s_ci * ES : A2 + (long)(s_ci - c1) * ES) +
            ((RABL(4) ? 0 : pix0) + s_ky * w_in + s_kx) * (long)a_ld;
    b_cur = bbase_tile + ((long)(s_ky * kw + s_kx) * cin + s_ci) * ES;
  };
  const uint32_t bq1 = (uint32_t)b_piece_stride, bq2 = 2u * bq1, bq3 = 3u * bq1, bq4 = 4u * bq1;   // < 4 GiB (fast_ok)

  auto issue_b = [&](uint32_t buf, int q) {
    const uint32_t qo = q == 0 ? 0u : (q == 1 ? bq1 : (q == 2 ? bq2 : (q == 3 ? bq3 : bq4)));
    if (RABL(16)) return;
    glds16_s(b_cur, boff + qo, buf + B_OFF + q * PIECE);
  };
  auto issue_a = [&](uint32_t buf, int q) {
    // pixl < 2^24 and the row pitch < 2^24 bytes (eligibility): one v_mad_u32_u24
    if (RABL(8)) return;
    glds16_s(a_cur, __umul24(pixl[q], a_ld) + cc16, buf + A_OFF + q * PIECE);
  };
  // group g of the K-tile at the issue pointer, into that K-tile's buffer (sequence parity `par`)
  auto issue_group = [&](int g, int par) {
    const uint32_t buf = lds_wave + par * BUF_BYTES;
    if (g == 0) {
      issue_b(buf, 0); issue_b(buf, 1); issue_b(buf, 2);
    } else if (g == 1) {
      issue_b(buf, 3); issue_b(buf, 4); issue_a(buf, 0);
    } else if (g == 2) {
      issue_a(buf, 1); issue_a(buf, 2);
    } else {
      issue_a(buf, 3);
    }
  };
  // step the issue pointer to the next K-tile of the sequence (`more`: that K-tile exists)
  auto advance_issue = [&](bool more) {
    ++iss_kt;
#ifdef VX_RING_TAP_OUTER
    s_ci += BKE;
    if (s_ci >= cin) {
      s_ci = 0;
      if (++s_kx == kw) { s_kx = 0; ++s_ky; }
    }
#else
    if (++s_kx == kw) {
      s_kx = 0;
      if (++s_ky == kh) { s_ky = 0; s_ci += BKE; }
    }
#endif
    if (iss_kt == nk) {
      iss_kt = 0; s_kx = 0; s_ky = 0;
      iss_lid += stride;
      s_ci = SK ? (iss_lid & 1) * (cin >> 1) : 0;
      if (more) setup_issue_tile();
    }
    if (more) refresh_issue_bases();
  };

  // ---- fragment read offsets (bytes within a buffer).  (row >> 1) & 7 == (frow >> 1) & 7 for every fragment row
  // (piece / wave / fragment bases are multiples of 16 rows), so the swizzle term is per-lane constant.
  const int frow = lane & 15, fgrp = lane >> 4;
  const int sw = (frow >> 1) & 7;
  const int ck0 = ((fgrp ^ sw) << 4), ck1 = (((4 + fgrp) ^ sw) << 4);
  const int a_rd = A_OFF + (32 * grp + frow) * 128;        // + p * PIECE + s * 2048 + ck
  const int b_rd = B_OFF + (wc * 80 + frow) * 128;         // + j * 2048 + ck

  f32x4_t acc[8][5];
  uint4 bfr[5][2], af[2][2];

  // ------------------------------------------------------------------ prologue: K-tiles 0 and 1 of the sequence
  setup_issue_tile();
  refresh_issue_bases();
  issue_group(0, 0); issue_group(1, 0); issue_group(2, 0); issue_group(3, 0);
  advance_issue(S > 1);
  if (S > 1) {
    issue_group(0, 1); issue_group(1, 1);
    if constexpr (RING_MI_LATE) {
      ring_wait_vm<9>();
    } else {
      issue_group(2, 1);
      ring_wait_vm<11>();   // g0, g1 of K-tile 0 landed (younger: g2, g3 of 0 and g0..g2 of 1)
    }
  } else {
    ring_wait_vm<3>();
  }
  ring_barrier();

  int u = 0;                 // sequence number of the K-tile being multiplied
  int cmp_lid = lb;          // its output tile

  // one K-tile: four (L slot, barrier, M slot, barrier) phases.  e1 / e2: K-tiles u+1 / u+2 exist in this block's
  // sequence (wave-uniform; false only for the last two K-tiles of the block)
  auto ktile = [&](const bool e1, const bool e2) {
    const int par = u & 1;
    const char* buf = smem + par * BUF_BYTES;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      // ---------------- L slot
      if (ph == 0 && !RABL(2)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          bfr[j][0] = *reinterpret_cast<const uint4*>(buf + b_rd + j * 2048 + ck0);
          bfr[j][1] = *reinterpret_cast<const uint4*>(buf + b_rd + j * 2048 + ck1);
        }
      }
#pragma unroll
      for (int s = 0; s < 2 && !RABL(2); ++s) {
        af[s][0] = *reinterpret_cast<const uint4*>(buf + a_rd + ph * PIECE + s * 2048 + ck0);
        af[s][1] = *reinterpret_cast<const uint4*>(buf + a_rd + ph * PIECE + s * 2048 + ck1);
      }
      RING_STAMP();
      if (ph == 0) {
        if (e1) {
          if constexpr (RING_MI_LATE) issue_group(2, par ^ 1);
          issue_group(3, par ^ 1);          // A3 of u+1
          advance_issue(e2);                // -> u+2
          RING_WAIT_VM(11);
        } else {
          RING_WAIT_VM(2);
        }
      } else if (ph == 1) {
        if (e2) {
          // u+2 reuses this K-tile's buffer: its B slots were read in L(u,0)
          if constexpr (VX_RING_MISSUE == 2) {
            issue_b(lds_wave + par * BUF_BYTES, 0); issue_b(lds_wave + par * BUF_BYTES, 1);
            RING_WAIT_VM(12);
          } else {
            issue_group(0, par);
            RING_WAIT_VM(13);
          }
        } else if (e1) {
          RING_WAIT_VM(10);
        } else {
          RING_WAIT_VM(1);
        }
      } else if (ph == 2) {
        if (e2) {
          if constexpr (RING_MI_M) {
            issue_b(lds_wave + par * BUF_BYTES, 3); issue_b(lds_wave + par * BUF_BYTES, 4);
            RING_WAIT_VM(14);
          } else {
            issue_group(1, par);
            RING_WAIT_VM(15);
          }
        } else if (e1) {
          RING_WAIT_VM(9);
        } else {
          RING_WAIT_VM(0);
        }
      } else {
        if (e2) {
          if constexpr (RING_MI_M) {
            issue_a(lds_wave + par * BUF_BYTES, 1);
            RING_WAIT_VM(10);
          } else if constexpr (RING_MI_LATE) {
            RING_WAIT_VM(9);
          } else {
            issue_group(2, par);
            RING_WAIT_VM(11);
          }
        } else if (e1) {
          RING_WAIT_VM(3);
        } else {
          RING_WAIT_VM(0);
        }
      }
      ring_wait_lgkm0();
      RING_STAMP();
      ring_barrier();
      RING_STAMP();
      // ---------------- M slot
      if constexpr (VX_RING_PRIO == 1) __builtin_amdgcn_s_setprio(1);
      if constexpr (VX_RING_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      auto m_slot_issue = [&]() {   // VX_RING_MISSUE only: this wave's copies that left the L slots
        if (!e2) return;
        __builtin_amdgcn_sched_barrier(0);
        if (VX_RING_MISSUE == 2 && ph == 1) issue_b(lds_wave + par * BUF_BYTES, 2);
        if (ph == 2) issue_a(lds_wave + par * BUF_BYTES, 0);
        if (ph == 3) issue_a(lds_wave + par * BUF_BYTES, 2);
        __builtin_amdgcn_sched_barrier(0);
      };
      if constexpr (F8) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < 5; ++j)
            acc[2 * ph + s][j] = ring_mfma_f8(bfr[j][0], bfr[j][1], af[s][0], af[s][1], acc[2 * ph + s][j]);
          if constexpr (RING_MI_M) {
            if (s == 0) m_slot_issue();
          }
        }
      } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            if (RABL(3)) {
              asm volatile("" ::"v"(bfr[j][kk].x), "v"(af[s][kk].x));
              continue;
            }
            acc[2 * ph + s][j] = mfma16(bfr[j][kk], af[s][kk], acc[2 * ph + s][j]);
          }
        if constexpr (RING_MI_M) {
          if (kk == 0) m_slot_issue();
        }
      }
      }
      if constexpr (VX_RING_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if constexpr (VX_RING_PRIO == 2) __builtin_amdgcn_s_setprio(1);
      RING_STAMP();
      ring_barrier();
      RING_STAMP();
    }
    ++u;
  };

  const float* __restrict__ bias = p.bias;
  const float* __restrict__ rowbias = p.rowbias;
  const bf16_t* __restrict__ resid = (const bf16_t*)p.residual;
  const bool do_silu = p.act == VX_ACT_SILU;
  const float alpha = p.alpha;
  const int lrow = lane & 15, lq = lane >> 4;

  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (grp == 1) ring_barrier();   // stagger: wave row 1 runs one slot behind wave row 0
    for (int kt = 0; kt < nk; ++kt) {
      ktile(u + 1 < S, u + 2 < S);
    }
    if (grp == 0) ring_barrier();   // re-align

    // ---------------------------------------------------------------- epilogue (no barriers, no LDS)
    // acc[i][j][r] = C[m0 + 128 grp + 16 i + lrow][n0 + 80 wc + 16 j + 4 lq + r].
    const int cmp_tile = SK ? cmp_lid >> 1 : cmp_lid;
    // SK: this wave's 40 KB of the tile's exchange area, lane-linear float4 [i * 5 + j][lane]
    // (wave-uniform base + one 32-bit lane offset: the scalar-base addressing form, no 64-bit address registers per item)
    const __amdgpu_buffer_rsrc_t wsr =
        __builtin_amdgcn_make_buffer_rsrc(SK ? p.splitk_ws : nullptr, 0, SK ? total_tiles * (R_BM * R_BN * 4) : 0, 0x00020000);
    const uint32_t wsb = (uint32_t)(cmp_tile * 8 + wave) * (40u * 1024u);   // < 4 GiB: m * n * 4 bytes (eligibility)
    const uint32_t wsl = (uint32_t)lane * 16u;
    if constexpr (SK) {
      // rendezvous of the two K halves of this tile, per wave (see the kernel's head comment); the partner's accumulators
      // are added in the bias pass below (a separate "acc += partner" pass here makes hipcc spill 100-200 registers)
      int* fl = (int*)((char*)p.splitk_ws + (size_t)total_tiles * (R_BM * R_BN * 4)) + (cmp_tile * 8 + wave) * 2;
      // fl[0] = epoch of the launch whose first partner wave claimed the slot, fl[1] = (epoch << 4 | writer's XCC id) once its
      // accumulators are in memory.  Nothing is reset: a later launch waits for ITS epoch (ABI 14; tools/coop_protocol_check.py)
      const int epoch = p.coop_epoch;
      int claimed = 0;
      if (lane == 0) claimed = __hip_atomic_exchange(fl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      claimed = __builtin_amdgcn_readfirstlane(claimed);
      if (claimed != epoch) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) sk_store16(wsr, wsl, wsb + (i * 5 + j) * 1024, acc[i][j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the write-through stores have reached L2 / memory
        if (lane == 0)
          __hip_atomic_store(fl + 1, (int)(((uint32_t)epoch << 4) | sk_xcc_id()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cmp_lid += stride;
        continue;                                  // (the epilogue below has no barriers: waves may skip it one by one)
      }
      uint32_t pollv, polls, polln, polle;
      {
        // wait for flag word [1] to carry this launch's epoch (agent-scope loads, every lane the same word).  Bounded (~2^20
        // polls): a lost partner must show up as a wrong result in the tests, never as a hung GPU.  An asm loop on purpose: with
        // a C++ loop here the 160 accumulators are "live through a loop" for the register allocator, which then spills ~100-200
        // of them (and puts scratch reloads into the K loop).
        asm volatile("s_mov_b32 %2, 0\n"
                     "1:\n\t"
                     "global_load_dword %0, %4, %5 offset:4 sc1\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_readfirstlane_b32 %1, %0\n\t"
                     "s_lshr_b32 %3, %1, 4\n\t"
                     "s_cmp_eq_u32 %3, %6\n\t"
                     "s_cbranch_scc1 2f\n\t"
                     "s_sleep 2\n\t"
                     "s_add_u32 %2, %2, 1\n\t"
                     "s_cmp_lt_u32 %2, 0x100000\n\t"
                     "s_cbranch_scc1 1b\n"
                     "2:"
                     : "=&v"(pollv), "=&s"(polls), "=&s"(polln), "=&s"(polle)
                     : "v"(0u), "s"(fl), "s"(epoch)
                     : "memory", "scc");
      }
      if ((polls & 15u) != sk_xcc_id() || polle != (uint32_t)epoch)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // writer on another XCD (or lost)
    }
    const int tile_m = cmp_tile / n_tiles, tile_n = cmp_tile - tile_m * n_tiles;
    const int row_base = tile_m * R_BM + 128 * grp + lrow;
    if constexpr (F8) {
      // dequantise: acc[m][n] *= a_scale[m] * w_scale[n]
      const float* __restrict__ asc = p.a_scale;
      const float* __restrict__ wsc = p.w_scale;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float sa_ = asc[row_base + 16 * i];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float4 sw4 = *reinterpret_cast<const float4*>(wsc + tile_n * R_BN + 80 * wc + 16 * j + 4 * lq);
          acc[i][j][0] *= sa_ * sw4.x; acc[i][j][1] *= sa_ * sw4.y;
          acc[i][j][2] *= sa_ * sw4.z; acc[i][j][3] *= sa_ * sw4.w;
        }
      }
    }
    float4 sk_b4[SK ? 5 : 1];   // SK: bias (+ row bias) of this lane's columns, per fragment j
    {
      // Bias (+ the tile's time-embedding / per-item row) and the folded LayerNorm in ONE pass over the raw accumulators,
      // on every lane's OWN columns (before any lane exchange): with the fold, acc <- rstd[m] * acc + (bias[n] -
      // rstd[m] * mean[m] * colsum[n]) = 2 FMA per element instead of FMA + MUL + ADD; without it one ADD.  Both
      // epilogues are VALU / latency bound on the short-K shapes, and the STORE epilogue needs no bias registers while it
      // walks its items (they hold a deeper residual prefetch instead).
      const float* rb_row0 = nullptr;
      if constexpr (EPI == VX_EPI_STORE)
        rb_row0 = rowbias != nullptr ? rowbias + (size_t)((tile_m * R_BM) / p.rows_per_group) * p.rowbias_ld : nullptr;
      float rs[8], rm[8];
      if constexpr (LNF) {
        const float2* __restrict__ st = reinterpret_cast<const float2*>(p.ln_stats);
        if (p.ln_stats_parts == 2) {
          // two-part format (k == 640): (sum, sum of squares) of each half of the row, finished here
          const float inv_k = 1.0f / (float)p.k;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float4 t = reinterpret_cast<const float4*>(st)[row_base + 16 * i];
            const float mean = (t.x + t.z) * inv_k;
            float var = (t.y + t.w) * inv_k - mean * mean;
            var = var > 0.f ? var : 0.f;
            rs[i] = 1.0f / sqrtf(var + p.ln_eps);
            rm[i] = -mean * rs[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float2 t = st[row_base + 16 * i];
            rs[i] = t.y;
            rm[i] = -t.x * t.y;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int col = tile_n * R_BN + 80 * wc + 16 * j + 4 * lq;
        float4 b4 = bias != nullptr ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (rb_row0 != nullptr) {
          const float4 r4 = *reinterpret_cast<const float4*>(rb_row0 + col);
          b4.x += r4.x; b4.y += r4.y; b4.z += r4.z; b4.w += r4.w;
        }
        if constexpr (LNF) {
          const float4 s4 = *reinterpret_cast<const float4*>(p.ln_colsum + col);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc[i][j][0] = fmaf(rs[i], acc[i][j][0], fmaf(rm[i], s4.x, b4.x));
            acc[i][j][1] = fmaf(rs[i], acc[i][j][1], fmaf(rm[i], s4.y, b4.y));
            acc[i][j][2] = fmaf(rs[i], acc[i][j][2], fmaf(rm[i], s4.z, b4.z));
            acc[i][j][3] = fmaf(rs[i], acc[i][j][3], fmaf(rm[i], s4.w, b4.w));
          }
        } else if constexpr (SK) {
          sk_b4[j] = b4;   // added per row block in the item walk below, together with the partner's accumulators
        } else if (bias != nullptr || rb_row0 != nullptr) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w;
          }
        }
      }
    }
    if constexpr (EPI == VX_EPI_STORE) {
      // Pairing fragments (j, j+1) and swapping the odd 16-lane rows of the first with the even rows of the second
      // gives every lane 8 consecutive columns of one fragment: [x0..x3 y0..y3] = fragment j + (lq & 1), columns
      // 8 (lq >> 1) .. + 7 -> 16-byte residual loads / output stores for fragments 0-3; fragment 4 keeps the native
      // 4 columns per lane (8 bytes).  Items are walked row-block-major (i, then {01, 23, 4}) so that a wave finishes
      // the 160 contiguous bytes of an output row within one step (partial lines meet in L2 right away); bias and the
      // tile's time-embedding / per-item row are already in the accumulators (pass above), and the residual loads run
      // RES_DEPTH items ahead of their use: the epilogue of a short-K tile is a string of HBM round trips otherwise.
      constexpr int N_ITEMS = 24, RES_DEPTH = (LNF || SK) ? 4 : VX_RING_RES_DEPTH;   // (LNF + RES: unused by the model, keep it spill-free)
      const int col_p = tile_n * R_BN + 80 * wc + 8 * (lq >> 1) + 16 * (lq & 1);   // + 32 t  (pair t = 0, 1)
      const int col_4 = tile_n * R_BN + 80 * wc + 64 + 4 * lq;
      // byte offsets (32-bit: eligibility bounds m * ld * 2 < 4 GiB) = per-lane base + wave-uniform item part
      const uint32_t ldr2 = (uint32_t)p.ldr * 2u, ldc2 = (uint32_t)p.ldc * 2u;
      const uint32_t res_p = (uint32_t)row_base * ldr2 + (uint32_t)col_p * 2u, res_4 = (uint32_t)row_base * ldr2 + (uint32_t)col_4 * 2u;
      const uint32_t out_p = (uint32_t)row_base * ldc2 + (uint32_t)col_p * 2u, out_4 = (uint32_t)row_base * ldc2 + (uint32_t)col_4 * 2u;
      // item k: row block i = k / 3, kind = k % 3 (0, 1: fragment pair; 2: fragment 4)
      auto res_off = [&](int k) {
        return (k % 3 == 2 ? res_4 : res_p + (uint32_t)(k % 3) * 64u) + (uint32_t)(k / 3) * (16u * ldr2);
      };
      auto out_off = [&](int k) {
        return (k % 3 == 2 ? out_4 : out_p + (uint32_t)(k % 3) * 64u) + (uint32_t)(k / 3) * (16u * ldc2);
      };
      const char* __restrict__ resb = (const char*)resid;
      char* __restrict__ outb = (char*)p.out;
      uint4 rv[N_ITEMS];   // (.x, .y only for the 8-byte items)
      auto load_res = [&](int k) {
        if (k % 3 == 2) {
          const uint2 t2 = *reinterpret_cast<const uint2*>(resb + res_off(k));
          rv[k] = make_uint4(t2.x, t2.y, 0u, 0u);
        } else {
          rv[k] = *reinterpret_cast<const uint4*>(resb + res_off(k));
        }
      };
      if (RES && !RABL(64)) {
#pragma unroll
        for (int k = 0; k < RES_DEPTH; ++k) load_res(k);
      }
      float st_s = 0.f, st_q = 0.f;   // STATS: this lane's (sum, sum of squares) of row 16 i + lrow, 20 columns
      // GNS: this lane's 20 columns, summed over its 8 rows (16 i + lrow): [0..7] pair 0, [8..15] pair 1, [16..19] fragment 4
      float gcs[GNS ? 20 : 1], gcq[GNS ? 20 : 1];
      if constexpr (GNS) {
#pragma unroll
        for (int c = 0; c < 20; ++c) gcs[c] = gcq[c] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < N_ITEMS; ++k) {
        if (RES && !RABL(64) && k + RES_DEPTH < N_ITEMS) load_res(k + RES_DEPTH);
        const int i = k / 3, kind = k % 3;
        if constexpr (SK) {
          // (own half + partner's half) + bias, one row block at a time right before its three items: five 16-byte loads
          // in flight and a shrinking set of live accumulators (all 40 loads in the bias pass: every loaded value was
          // spilled the moment it arrived, one memory round trip each - 50 us per launch)
          if (kind == 0) {
            f32x4_t pv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) pv[j] = sk_load16(wsr, wsl, wsb + (i * 5 + j) * 1024);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
              acc[i][j][0] = (acc[i][j][0] + pv[j][0]) + sk_b4[j].x; acc[i][j][1] = (acc[i][j][1] + pv[j][1]) + sk_b4[j].y;
              acc[i][j][2] = (acc[i][j][2] + pv[j][2]) + sk_b4[j].z; acc[i][j][3] = (acc[i][j][3] + pv[j][3]) + sk_b4[j].w;
            }
          }
        }
        if (kind == 2) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][4][r];
            if (do_silu) v[r] = silu_f(v[r]);
            v[r] *= alpha;
          }
          if (RES && !RABL(64)) {
            v[0] += e16_lo(rv[k].x); v[1] += e16_hi(rv[k].x);
            v[2] += e16_lo(rv[k].y); v[3] += e16_hi(rv[k].y);
          }
          if (RABL(32)) {
            asm volatile("" ::"v"(v[0]), "v"(v[3]));
            continue;
          }
          const uint2 pk2 = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          ring_store8(outb, out_off(k), pk2);
          if constexpr (GNS) {
            const float g0 = e16_lo(pk2.x), g1 = e16_hi(pk2.x);
            const float g2 = e16_lo(pk2.y), g3 = e16_hi(pk2.y);
            gcs[16] += g0; gcs[17] += g1; gcs[18] += g2; gcs[19] += g3;
            gcq[16] = fmaf(g0, g0, gcq[16]); gcq[17] = fmaf(g1, g1, gcq[17]);
            gcq[18] = fmaf(g2, g2, gcq[18]); gcq[19] = fmaf(g3, g3, gcq[19]);
          }
          if constexpr (STATS) {
            // statistics of the STORED (bf16-rounded) values, as vx_row_stats would read them back
            const float r0 = e16_lo(pk2.x), r1 = e16_hi(pk2.x);
            const float r2 = e16_lo(pk2.y), r3 = e16_hi(pk2.y);
            st_s += (r0 + r1) + (r2 + r3);
            st_q = fmaf(r0, r0, st_q); st_q = fmaf(r1, r1, st_q); st_q = fmaf(r2, r2, st_q); st_q = fmaf(r3, r3, st_q);
            // last item of row block i: add up the four lanes (lq = 0..3) that share the row.  permlane16_swap(s, q)
            // leaves (s0, q0, s2, q2) / (s1, q1, s3, q3) in the four 16-lane rows; their sum, then the two halves
            // exchanged by permlane32_swap: 16-lane rows 0 / 2 hold the row's sum, rows 1 / 3 its sum of squares
            auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(st_s), __float_as_uint(st_q), false, false);
            const float t = __uint_as_float(a[0]) + __uint_as_float(a[1]);
            auto b2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
            const float tot = __uint_as_float(b2[0]) + __uint_as_float(b2[1]);
            if (lq < 2)
              *reinterpret_cast<float*>(smem + STATS_OFF + ((128 * grp + 16 * i + lrow) * 4 + wc) * 8 + 4 * lq) = tot;
            st_s = 0.f;
            st_q = 0.f;
          }
        } else {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][2 * kind][r], y = acc[i][2 * kind + 1][r];
            swap16(x, y);
            v[r] = x;
            v[4 + r] = y;
          }
          if (do_silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= alpha;
          if (RES && !RABL(64)) {
            float rf[8];
            unpack_bf16x8(rv[k], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
          }
          if (RABL(32)) {
            asm volatile("" ::"v"(v[0]), "v"(v[7]));
            continue;
          }
          const uint4 pk8 = pack_bf16x8(v);
          ring_store16(outb, out_off(k), pk8);
          if constexpr (GNS) {
            float gg[8];
            unpack_bf16x8(pk8, gg);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              gcs[kind * 8 + e] += gg[e];
              gcq[kind * 8 + e] = fmaf(gg[e], gg[e], gcq[kind * 8 + e]);
            }
          }
          if constexpr (STATS) {
            float rr[8];
            unpack_bf16x8(pk8, rr);
            st_s += ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
#pragma unroll
            for (int e = 0; e < 8; ++e) st_q = fmaf(rr[e], rr[e], st_q);
          }
        }
      }
      if constexpr (STATS) {
        // the four wave columns of a row meet in LDS (fixed order -> deterministic); rows 0..255 of the tile = tid.
        // ring_barrier() is a bare s_barrier (no fence): the partial sums must have LEFT this wave's LDS queue first
        ring_wait_lgkm0();
        ring_barrier();
        if (tid < R_BM) {
          const float4 p01 = *reinterpret_cast<const float4*>(smem + STATS_OFF + tid * 32);
          const float4 p23 = *reinterpret_cast<const float4*>(smem + STATS_OFF + tid * 32 + 16);
          const float s_t = (p01.x + p01.z) + (p23.x + p23.z), q_t = (p01.y + p01.w) + (p23.y + p23.w);
          if (p.row_stats_parts == 2) {
            // n == 640: this tile holds one half of every row - its (sum, sum of squares) is part tile_n of the row
            reinterpret_cast<float2*>(p.row_stats_out)[(size_t)(tile_m * R_BM + tid) * 2 + tile_n] = make_float2(s_t, q_t);
          } else {
            const float inv_n = 1.0f / (float)R_BN;
            const float mean = s_t * inv_n;
            float var = q_t * inv_n - mean * mean;
            var = var > 0.f ? var : 0.f;
            reinterpret_cast<float2*>(p.row_stats_out)[tile_m * R_BM + tid] =
                make_float2(mean, 1.0f / sqrtf(var + p.row_stats_eps));
          }
        }
      }
      if constexpr (GNS) {
        // (a) the 16 lanes of a DPP row hold the 16 rows lrow of the same columns: four rotate-and-add steps leave the
        // 128-row column sums in every lane; (b) lanes lrow == 0 park their 20 columns in this wave's 640 bytes of LDS
        // behind the K-tile buffers (the wave's own writes and reads are ordered in the LDS queue: no barrier);
        // (c) lane g < 80 / cg adds the cg columns of group g in ascending order (groups never straddle a wave: 80 % cg
        // == 0 is part of the eligibility) and writes the slab's (sum, sum of squares).  Fixed order -> same bits for any
        // batch; the consumer (gn_apply / gn_fold_linear) adds the slabs of a frame in float64.
#pragma unroll
        for (int c = 0; c < 20; ++c) {
          gcs[c] = row16_sum(gcs[c]);
          gcq[c] = row16_sum(gcq[c]);
        }
        float2* scr = reinterpret_cast<float2*>(smem + STATS_OFF) + wave * 80;
        if (lrow == 0) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              scr[16 * (2 * t + (lq & 1)) + 8 * (lq >> 1) + e] = make_float2(gcs[t * 8 + e], gcq[t * 8 + e]);
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[64 + 4 * lq + r] = make_float2(gcs[16 + r], gcq[16 + r]);
        }
        ring_wait_lgkm0();
        const int cg = p.n / p.gn_groups;
        if (lane < 80 / cg) {
          float a = 0.f, b = 0.f;
          for (int c = 0; c < cg; ++c) {
            const float2 t = scr[lane * cg + c];
            a += t.x;
            b += t.y;
          }
          const int m0w = tile_m * R_BM + 128 * grp;
          const int frame = m0w / p.gn_hw, slab = (m0w - frame * p.gn_hw) >> 7, slabs = p.gn_hw >> 7;
          const int g = (tile_n * R_BN + 80 * wc) / cg + lane;
          reinterpret_cast<float2*>(p.gn_ws)[(size_t)(frame * slabs + slab) * p.gn_groups + g] = make_float2(a, b);
        }
        ring_wait_lgkm0();   // the scratch is rewritten by this wave's next tile only after these reads
      }
    } else {   // VX_EPI_GEGLU
      // Weight rows are interleaved in blocks of 8 (weights.py: geglu_interleave): columns 16j..16j+7 of a fragment
      // are the VALUES of output channels 8j'..8j'+7 and columns 16j+8..16j+15 their GATES, i.e. lanes 0-31 hold
      // values and lanes 32-63 the matching gates.  v_permlane32_swap on the fragments of two row blocks (i, i+1)
      // gives lanes 0-31 value AND gate of block i, lanes 32-63 those of block i+1: no lane idles in the GELU.
      // Then the packed results of fragments (j, j+1) are exchanged with v_permlane16_swap for 16-byte stores.
      const int ocol_base = (tile_n * R_BN + 80 * wc) / 2 + 4 * (lq & 1);    // + 8 j (+ r)
      const int orow = row_base + 16 * (lane >> 5);                          // + 16 i  (i even)
      bf16_t* __restrict__ outp = (bf16_t*)p.out;
#ifdef VX_GELU_PK
      GeluPk gpk;
      gpk.init();
#endif
#pragma unroll
      for (int jp = 0; jp < 5; jp += 2) {
        const int nj = jp + 1 < 5 ? 2 : 1;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          uint32_t pk[2][2];
#pragma unroll
          for (int jj = 0; jj < nj; ++jj) {
            float o[4];
#ifdef VX_GELU_PK
            float val[4], gat[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][jp + jj][r]),
                                                         __float_as_uint(acc[i + 1][jp + jj][r]), false, false);
              val[r] = __uint_as_float(sw[0]);
              gat[r] = __uint_as_float(sw[1]);
            }
            gpk.mul2(val[0], val[1], gat[0], gat[1], o[0], o[1]);
            gpk.mul2(val[2], val[3], gat[2], gat[3], o[2], o[3]);
#else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][jp + jj][r]),
                                                         __float_as_uint(acc[i + 1][jp + jj][r]), false, false);
              const float val = __uint_as_float(sw[0]);   // lanes 0-31: block i, lanes 32-63: block i+1
              const float gat = __uint_as_float(sw[1]);   // (biases were added before the pairing)
              o[r] = val * (RABL(64) ? gat : gelu_f(gat));
            }
#endif
            pk[jj][0] = pack_bf16x2(o[0], o[1]);
            pk[jj][1] = pack_bf16x2(o[2], o[3]);
          }
          const size_t rowoff = (size_t)(orow + 16 * i) * p.ldc;
          if (RABL(32)) {
            asm volatile("" ::"v"(pk[0][0]), "v"(pk[0][1]));
            continue;
          }
          if (nj == 2) {
            // lane (lq & 1) = 0 keeps fragment jp (columns 0-3 own, 4-7 from its odd neighbour), the odd lane takes jp+1
            auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
            const int oc = (tile_n * R_BN + 80 * wc) / 2 + 8 * (jp + (lq & 1));
            *reinterpret_cast<uint4*>(outp + rowoff + oc) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          } else {
            *reinterpret_cast<uint2*>(outp + rowoff + ocol_base + 8 * jp) = make_uint2(pk[0][0], pk[0][1]);
          }
        }
      }
    }
    cmp_lid += stride;
  }
}

int g_cu_count = 0;

}  // namespace

#ifdef VX_RING_TRACE
extern "C" int vx_gemm_ring_set_trace(void* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_trace), &dev_buf, sizeof(void*)) == hipSuccess ? 0 : VX_ERR_HIP;
}
#endif

bool vx_gemm_ring_eligible(const vx_gemm_params& p) {
  // The kernel choice is a per-call fact (ABI 14): ring_hint < 0 = never; there is no process-wide mode any more (the
  // A/B knobs VX_GEMM_RING / VX_FP8_RING live in the Python layer, which turns them into ring_hint values).
  if (p.ring_hint < 0) return false;
  if (p.a_fp8) {
    // fp8 operands on the ring kernel: correct (tests/test_gpu_kernels.py::test_gemm_fp8_ring_vs_classic_tiles), but
    // its 8-register MFMA operand tuples push the 256-VGPR budget over (67-72 spilled registers in the K loop) and it
    // measures SLOWER than the classic fp8 tiles (287 vs 207 us at 294912 x 320 x 384; 768^2 clip 3.59 vs 3.92
    // frames/s, profiles/r02d_*): only on the explicit per-call request ring_hint == 3
    if (p.ring_hint != 3 || p.epi != VX_EPI_STORE || (p.k % 128) != 0 || p.kh != 1 || p.kw != 1 || p.a2 != nullptr) return false;
  } else if (p.ring_hint == 3) {
    return false;
  }
  // ring_hint == 2: the cooperative two-way K split (gemm_ring_kernel<..., SK>; splitk == 2 + a ZEROED workspace of
  // vx_gemm_splitk_ws_bytes(m, n, 2) bytes).  Any other split-K request belongs to the classic tiles.
  const bool coop = p.ring_hint == 2;
  if (coop && p.splitk != 2) return false;
  if (coop && (p.epi != VX_EPI_STORE || p.a_fp8 || p.ln_stats != nullptr || vx_gemm_ring_writes_row_stats(p) ||
               p.splitk_ws == nullptr || p.w_group_rows != 0 || (unsigned long long)p.m * p.n * 4ull >= (1ull << 32) || (((p.c1 + p.c2) / 64) & 1) != 0 || ((p.c1 + p.c2) % 64) != 0 ||
               VX_XCD_ROWS))
    return false;
  if ((p.epi != VX_EPI_STORE && p.epi != VX_EPI_GEGLU) || p.out_f32 || (p.splitk > 1 && !coop) || p.act == VX_ACT_GELU) return false;
  if ((p.m % R_BM) != 0 || (p.n % R_BN) != 0) return false;
  if (p.w_group_rows != 0 && (p.w_group_rows < 0 || (p.w_group_rows % R_BM) != 0 || p.a_fp8)) return false;
  if ((p.ldc % 8) != 0 || (p.residual != nullptr && (p.ldr % 8) != 0)) return false;
  if ((unsigned long long)p.m * p.ldc * 2ull >= (1ull << 32) ||
      (p.residual != nullptr && (unsigned long long)p.m * p.ldr * 2ull >= (1ull << 32)))
    return false;   // 32-bit epilogue offsets
  // the time-embedding row must be the same for all 256 rows of a tile (true whenever frames * hw is a multiple of 256)
  if (p.rowbias != nullptr && ((p.rowbias_ld % 4) != 0 || (p.rows_per_group % R_BM) != 0)) return false;
  if (!vx_gemm_fast_ok(p)) return false;
  if ((long)p.nb * p.h_in * p.w_in >= (1l << 24) || p.lda1 >= (1 << 23) || p.lda2 >= (1 << 23)) return false;   // umul24
  const int hw_out = p.h_out * p.w_out;
  // a 256-row tile = whole frames, or whole image rows of one frame (tile-invariant per-thread gather offsets)
  if (!((R_BM % hw_out) == 0 || ((hw_out % R_BM) == 0 && (R_BM % p.w_out) == 0))) return false;
  const long tiles = (long)(p.m / R_BM) * (p.n / R_BN);
  // fewer tiles than ~3/4 of the CUs: the 128-row tiles of vx_gemm.hip fill the chip better (ring_hint = 1: the caller
  // made that call from batch-independent facts, e.g. v_express_amd.ops for the CFG-pair launch shape)
  if (p.ring_hint == 0 && tiles < 192) return false;
  // In isolation (tools/gemm_bench, operands warm in L2 / Infinity Cache) the plain two-stage loop is a few % faster on
  // long K loops: an LDS-DMA copy issued beside the partner wave's MFMAs costs 2-3x one issued in a burst
  // (profiles/r01d_ring_ablation.txt).  Inside the model the ring kernel wins on every eligible shape (tap-innermost
  // K order: 89 % vs 57 % L2 hit rate, 6x less fabric traffic; next tile prefetched under the epilogue): 11.01 vs
  // 10.64 frames/s.
  return true;
}

template <int EPI, bool RES, bool F8 = false, bool STATS = false, bool LNF = false, bool GNS = false, bool SK = false>
static int ring_launch(const vx_gemm_params& p, hipStream_t stream) {
  static bool attr_set = false;
  auto kern = gemm_ring_kernel<EPI, RES, F8, STATS, LNF, GNS, SK>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_TOTAL);
    if (e != hipSuccess) {
      vx_set_error("vx_gemm(ring): hipFuncSetAttribute(%d B LDS) failed: %s", R_LDS_TOTAL, hipGetErrorString(e));
      return VX_ERR_HIP;
    }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    g_cu_count = cus;
    attr_set = true;
  }
  const long tiles = (long)(p.m / R_BM) * (p.n / R_BN) * (SK ? 2 : 1);   // work items
  unsigned grid = (unsigned)(tiles < g_cu_count ? tiles : g_cu_count);
  if (SK) grid &= ~1u;   // both K halves of a tile in the same round
  static char sym[104] = "";
  if (!sym[0]) {
    auto b = [](bool v) { return v ? "true" : "false"; };
    if (SK)
      snprintf(sym, sizeof(sym), "gemm_ring_kernel<%d, %s, %s, %s, %s, %s, true>", EPI, b(RES), b(F8), b(STATS), b(LNF), b(GNS));
    else
      snprintf(sym, sizeof(sym), "gemm_ring_kernel<%d, %s, %s, %s, %s, %s>", EPI, b(RES), b(F8), b(STATS), b(LNF), b(GNS));
  }
  g_vx_last_kernel = sym;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(R_NT), R_LDS_TOTAL, stream, p);
  return vx_check_launch("vx_gemm(ring)");
}

int vx_gemm_ring_launch(const vx_gemm_params& p, hipStream_t stream) {
  if (p.a_fp8)
    return p.residual != nullptr ? ring_launch<VX_EPI_STORE, true, true>(p, stream)
                                 : ring_launch<VX_EPI_STORE, false, true>(p, stream);
  const bool ln = p.ln_stats != nullptr;
  if (p.epi == VX_EPI_GEGLU)
    return ln ? ring_launch<VX_EPI_GEGLU, false, false, false, true>(p, stream) : ring_launch<VX_EPI_GEGLU, false>(p, stream);
  const bool res = p.residual != nullptr;
  if (p.splitk == 2) {   // (vx_gemm_ring_eligible: ring_hint == 2, STORE, no fold / row statistics)
    if (p.gn_ws != nullptr)
      return res ? ring_launch<VX_EPI_STORE, true, false, false, false, true, true>(p, stream)
                 : ring_launch<VX_EPI_STORE, false, false, false, false, true, true>(p, stream);
    return res ? ring_launch<VX_EPI_STORE, true, false, false, false, false, true>(p, stream)
               : ring_launch<VX_EPI_STORE, false, false, false, false, false, true>(p, stream);
  }
  if (ln) {
    // the folded projections (q / qkv) never carry a residual in this model; the combination exists for completeness
    if (vx_gemm_ring_writes_row_stats(p))
      return res ? ring_launch<VX_EPI_STORE, true, false, true, true>(p, stream)
                 : ring_launch<VX_EPI_STORE, false, false, true, true>(p, stream);
    return res ? ring_launch<VX_EPI_STORE, true, false, false, true>(p, stream)
               : ring_launch<VX_EPI_STORE, false, false, false, true>(p, stream);
  }
  if (vx_gemm_ring_writes_row_stats(p))
    return res ? ring_launch<VX_EPI_STORE, true, false, true>(p, stream) : ring_launch<VX_EPI_STORE, false, false, true>(p, stream);
  if (p.gn_ws != nullptr)   // (vx_gemm checked vx_gemm_gn_slabs(p) > 0)
    return res ? ring_launch<VX_EPI_STORE, true, false, false, false, true>(p, stream)
               : ring_launch<VX_EPI_STORE, false, false, false, false, true>(p, stream);
  return res ? ring_launch<VX_EPI_STORE, true>(p, stream) : ring_launch<VX_EPI_STORE, false>(p, stream);
}

// whether the ring launch of p fills p.row_stats_out itself (otherwise vx_gemm runs vx_row_stats on the output)
bool vx_gemm_ring_writes_row_stats(const vx_gemm_params& p) {
  return p.row_stats_out != nullptr && !p.a_fp8 && p.epi == VX_EPI_STORE &&
         (p.row_stats_parts == 2 ? p.n == 2 * R_BN : p.n == R_BN);
}

// GroupNorm partial sums from the STORE epilogue (vx_gemm_params.gn_ws): slabs per frame this launch writes, 0 = cannot
int vx_gemm_ring_gn_slabs(const vx_gemm_params& p) {
  if (p.epi != VX_EPI_STORE || p.a_fp8 || p.out_f32 || p.ln_stats != nullptr || p.row_stats_out != nullptr) return 0;
  if (p.gn_groups <= 0 || p.gn_hw <= 0 || (p.n % p.gn_groups) != 0 || (p.m % p.gn_hw) != 0 || (p.gn_hw % 128) != 0) return 0;
  const int cg = p.n / p.gn_groups;
  if (cg <= 0 || (80 % cg) != 0) return 0;
  return p.gn_hw / 128;
}
