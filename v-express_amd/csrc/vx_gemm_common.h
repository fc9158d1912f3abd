// Pieces shared by the GEMM kernels of libvexpress_hip.so (vx_gemm.hip, vx_gemm_ring.hip).
#pragma once
#include "vx_common.h"
#include "../../include/vexpress_hip.h"

constexpr int BK = 64;   // K-tile depth: one LDS row = 64 bf16 = 128 B = eight 16-B chunks

// XOR swizzle of the 16-B chunks of an LDS row (conflict-free ds_read_b128 fragment reads)
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Async 16-B global -> LDS copies (global_load_lds_dwordx4): LDS address = M0 (wave-uniform) + lane * 16.
// Issued through inline asm ON PURPOSE: for the __builtin_amdgcn_global_load_lds form hipcc (ROCm 7.2) tracks the
// copy as a pending LDS store and puts `s_waitcnt vmcnt(0)` in front of the next LDS read of ANY address, i.e. it
// drains the prefetch of the next K-tile before the current one is multiplied (seen in the ISA: the "double
// buffered" loop was serial).  The kernels order DMA -> ds_read themselves (counted s_waitcnt vmcnt + s_barrier),
// so the compiler must not know about these copies.  Its own vmcnt arithmetic for ordinary loads stays correct:
// vmcnt retires in order, so unknown older copies only make its waits conservative.
__device__ __forceinline__ uint32_t lds_addr_of(const void* shared_ptr) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)shared_ptr;
}
// Cache-policy experiment for the copies (build with -DVX_GLDS_MOD='" nt"', '" sc1"', '" sc0 sc1"'; the product library
// is built without, i.e. default policy): the long-K GEMMs are bound by the CU's L1 miss path (LABNOTES.md section 7)
#ifndef VX_GLDS_MOD
#define VX_GLDS_MOD ""
#endif
// wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_wave_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" VX_GLDS_MOD ::"s"(lds_wave_addr),
               "v"(voff), "s"(sbase)
               : "memory", "m0");
}
// per-lane 64-bit address
__device__ __forceinline__ void glds16_v(const void* vaddr, uint32_t lds_wave_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" VX_GLDS_MOD ::"s"(lds_wave_addr),
               "v"(vaddr)
               : "memory", "m0");
}

// VX_XCD_ROWS (round 4 experiment, default 0; 1 = A/B builds): every kernel that walks row tiles gives XCD x (= block id
// % 8) the CONTIGUOUS x-th eighth of the rows, so that what one kernel leaves in an XCD's L2 would be read by the next
// kernel on the same XCD (gemm_ring_kernel's tile walk, ff_fused_kernel's, the GroupNorm kernels' block order; the classic
// tiles and the attention kernels always do through xcd_remap).  Measured: 12.51 vs 12.56 frames/s (-0.4 %,
// profiles/r04i_xcd_row_ownership_negative_result.txt) - an XCD's 4 MB of L2 holds a twentieth of an 84 MB tensor, the
// cross-launch reuse lives in the 256 MB Infinity Cache, which does not care about the XCD.  Off.
#ifndef VX_XCD_ROWS
#define VX_XCD_ROWS 0
#endif

// XCD-aware block remap (8 XCDs, blocks are dealt round-robin): logical ids that are adjacent run on the same
// XCD, so the column tiles of one A row-tile share that XCD's L2.  Bijective for any block count.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// (g_vx_last_kernel: vx_common.h)

// FAST addressing eligibility (see gemm_kernel in vx_gemm.hip)
bool vx_gemm_fast_ok(const vx_gemm_params& p);
// persistent ring-staged kernel (vx_gemm_ring.hip)
bool vx_gemm_ring_eligible(const vx_gemm_params& p);
int vx_gemm_ring_launch(const vx_gemm_params& p, hipStream_t stream);
bool vx_gemm_ring_writes_row_stats(const vx_gemm_params& p);
int vx_gemm_ring_gn_slabs(const vx_gemm_params& p);

// sum over the 16 lanes of a DPP row (v_add_f32 with row_ror:8 / 4 / 2 / 1): every lane ends with the row's total.  In the
// C^T accumulator layout the lanes of a DPP row hold 16 consecutive output rows of the same columns.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0x128>(v);
  v += dpp_mov<0x124>(v);
  v += dpp_mov<0x122>(v);
  v += dpp_mov<0x121>(v);
  return v;
}
