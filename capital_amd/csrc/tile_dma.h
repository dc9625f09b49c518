// LDS-DMA of one K-contiguous operand tile through a buffer descriptor (shared by the fp64 and the bf16 tile kernels).
//
// Tile image in LDS: 128 rows x 128 bytes (16 doubles or 64 bf16 of K), lane-linear as the DMA writes it, with the eight
// 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7 on the SOURCE side so that ds_read_b128 fragment reads are
// bank-conflict free.  One wave-instruction moves 8 rows (1 KiB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The same tile move through a buffer descriptor (buffer_load_dwordx4 ... lds): the per-lane part of the address is ONE
// loop-invariant 32-bit VGPR offset (two: even / odd row groups differ in the swizzle), everything that changes from
// piece to piece and from K tile to K tile (row group, k) is a scalar offset.  No 64-bit VALU address arithmetic in
// the main loop - measured: the LDS-DMA issue sequence was the only thing keeping the MFMA pipe below 97 % busy.
// `rowbytes` = ld * 8;  base = first row of the tile at k = 0;  requires 128 * rowbytes + K * 8 < 2^32.
struct DmaBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff_even, voff_odd;   // per-lane byte offsets of a piece with even / odd q
  uint32_t rowgrp;                // byte stride between row groups of 8 rows
};
__device__ __forceinline__ DmaBuf dma_buf_make(const double* base, int64_t ld) {
  DmaBuf d;
  d.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xffffffffu, 0x00020000);   // offsets are unsigned 32-bit
  const int lane = threadIdx.x & 63;
  const int rsub = lane >> 3, p = lane & 7;
  const uint32_t rowbytes = (uint32_t)(ld * 8);
  // row = r0 + rsub with r0 a multiple of 8: (row >> 1) & 7 = ((rsub >> 1) + (r0 >> 1)) & 7, r0 >> 1 is 0 or 4 mod 8 (q even / odd)
  d.voff_even = rsub * rowbytes + ((p ^ (rsub >> 1)) << 4);
  d.voff_odd = rsub * rowbytes + ((p ^ ((rsub >> 1) ^ 4)) << 4);
  d.rowgrp = 8 * rowbytes;
  return d;
}
// pieces [q0, q1) of the wave's four row groups; wid must be wave-uniform (readfirstlane'd by the caller)
template <int Q0, int Q1>
__device__ __forceinline__ void dma_tile_buf(const DmaBuf& d, int wid, uint32_t kbytes, double* lds_tile) {
#pragma unroll
  for (int q = Q0; q < Q1; q++) {
    const uint32_t g8 = (uint32_t)(wid * 4 + q);        // row group of 8 rows
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rsrc, (__attribute__((address_space(3))) void*)(lds_tile + g8 * 8 * 16), 16,
                                             (int)((q & 1) ? d.voff_odd : d.voff_even), (int)(g8 * d.rowgrp + kbytes), 0, 0);
  }
}

