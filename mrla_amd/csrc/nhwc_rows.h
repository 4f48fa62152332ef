// Row pipeline of the channels_last (NHWC) stencil kernels, C % 64 == 0 ("wide" form).
//
// A wave owns 64 consecutive channels (LANE = CHANNEL in the arithmetic) and a strip of image columns, and walks down
// the rows.  Global memory is touched with 16 bytes per lane only (a 2-byte access costs the address unit as much):
//   * loads are LDS-DMA buffer loads (`buffer_load_dwordx4 ... lds`): a wave-instruction moves 8 pixels x 64 channels
//     (16-bit types) straight into the wave's private LDS row buffer -- no VGPR staging, no ds_write.  The buffer
//     descriptor of a row is built on the scalar unit (base = row pointer, num_records = bytes of the row, 0 for rows
//     above / below the image) and the per-lane byte offset is a per-strip constant (0x80000000 for pixels left / right
//     of the image), so everything outside the image arrives as zeros from the bounds check: no exec masking, no
//     zero-fills, no per-row vector address arithmetic;
//   * the LANE = CHANNEL registers are then filled from LDS with one 2-byte read per pixel; for bf16 that read is
//     `ds_read_u16_d16_hi`, which deposits the value in the upper half of a register whose lower half is zero -- the
//     register then IS the fp32 value (no shift per element).  Registers filled this way are never written by
//     arithmetic (RawRow keeps them apart), so their lower halves stay zero while the arrays rotate by name;
//   * results take the reverse route through a second LDS buffer and leave with buffer stores (pixels beyond the
//     strip's last owned column are dropped by the bounds check).
// The next row's DMA is in flight while the current row is computed (one buffer per tensor suffices: the LDS reads of
// a row have returned before the next DMA into the same buffer is issued).
#pragma once
#include "mrla_device.h"

#ifndef MRLA_ROW_LOAD_AUX
#define MRLA_ROW_LOAD_AUX 0
#endif
#ifndef MRLA_ROW_STORE_AUX
#define MRLA_ROW_STORE_AUX 0
#endif

namespace mrla {

constexpr unsigned kRowOob = 0x80000000u;          // byte offset no row reaches: the buffer bounds check yields zeros
constexpr int kBufFlags = 0x00020000;              // raw buffer, 32-bit data format (gfx9 family descriptor word 3)

typedef __attribute__((address_space(3))) const char* lds_cchar_ptr;

template <typename T, int NPX>
struct RowIO {
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int UPP = 64 / VEC;             // lanes per pixel
  static constexpr int PPL = 64 / UPP;             // pixels per wave-instruction
  static constexpr int NL = (NPX + PPL - 1) / PPL; // wave-instructions per row piece
  static constexpr int kBytes = NL * 1024;         // LDS bytes of the row buffer
  unsigned voff[NL];                               // this lane's byte offset inside an image row (kRowOob: no pixel)
};

// Row piece of NPX pixels starting at image column col0; pixels at index >= npx or outside [0, W) do not exist.
template <typename T, int NPX>
__device__ __forceinline__ void make_row_io(RowIO<T, NPX>& a, int col0, int npx, int W, int C, int cbase, int lane) {
  typedef RowIO<T, NPX> Q;
  const int px = lane / Q::UPP, part = lane - px * Q::UPP;
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) {
    const int p = l * Q::PPL + px, col = col0 + p;
    const bool ok = p < npx && col >= 0 && col < W;
    a.voff[l] = ok ? (unsigned)((col * C + cbase + part * Q::VEC) * (int)sizeof(T)) : kRowOob;
  }
}

// (The buffer builtins exist in the device pass only; the host pass of hipcc sees empty bodies.)
// Start the DMA of row r into the LDS row buffer `buf` (wave-private, RowIO::kBytes).  The buffer descriptor of a row
// outside [0, H) has num_records = 0: every access is out of range and delivers zeros.
template <typename T, int NPX, int AUX = MRLA_ROW_LOAD_AUX>
__device__ __forceinline__ void row_fetch(const RowIO<T, NPX>& a, const T* img, int r, int H, int rowelems, T* buf) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef RowIO<T, NPX> Q;
  const bool live = r >= 0 && r < H;                // wave-uniform
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(img) + (size_t)(live ? r : 0) * rowelems, 0,
                                                    live ? rowelems * (int)sizeof(T) : 0, kBufFlags);
#pragma unroll
  for (int l = 0; l < Q::NL; ++l)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(reinterpret_cast<char*>(buf) + l * 1024), 16, a.voff[l], 0, 0, AUX);
#endif
}

// The same for a workgroup that walks a RANGE of an image's rows (light_nhwc_wide.h: RowCut): `img` points at the range's
// first row, r counts from there (negative: the halo rows above), and the image's rows are [lo, hi) in that frame.
template <typename T, int NPX, int AUX = MRLA_ROW_LOAD_AUX>
__device__ __forceinline__ void row_fetch_in(const RowIO<T, NPX>& a, const T* img, int r, int lo, int hi, int rowelems, T* buf) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef RowIO<T, NPX> Q;
  const bool live = r >= lo && r < hi;              // wave-uniform
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(img) + (ptrdiff_t)(live ? r : 0) * rowelems, 0,
                                                    live ? rowelems * (int)sizeof(T) : 0, kBufFlags);
#pragma unroll
  for (int l = 0; l < Q::NL; ++l)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(reinterpret_cast<char*>(buf) + l * 1024), 16, a.voff[l], 0, 0, AUX);
#endif
}

// All DMA rows issued so far have landed (also orders them against the LDS reads below).
__device__ __forceinline__ void rows_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... except that the NEWEST `KEEP` vector-memory instructions may still be in flight: the row stores a step issued after
// its fetches.  (gfx9-family vmcnt counts loads and stores in issue order, which is also what the compiler relies on.)
template <int KEEP>
__device__ __forceinline__ void rows_landed_keep() {
  static_assert(KEEP >= 0 && KEEP < 64, "vmcnt immediate (6 bits on the gfx9 family)");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)((lds_cchar_ptr)p); }

// Registers that only ever receive raw row data (see the header comment: their low halves stay zero for bf16).
template <int NPX>
struct RawRow {
  float v[NPX];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < NPX; ++j) v[j] = 0.f;
  }
};

// out.v[j] = float(buf[pixel px0 + j][this lane's channel]).  Call rows_landed() first.
// The reads have returned when this comes back (the row buffer may be handed to the next DMA right away), and the
// compiler cannot move them across it.
template <typename T, int NPX>
__device__ __forceinline__ void row_read(const T* buf, int lane, RawRow<NPX>& out, int px0 = 0) {
  typedef __attribute__((address_space(3))) const T* lds_T_ptr;
  lds_T_ptr p = (lds_T_ptr)buf + px0 * kWave + lane;
#pragma unroll
  for (int j = 0; j < NPX; ++j) out.v[j] = to_f(p[j * kWave]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

#define MRLA_D16(i, off) "ds_read_u16_d16_hi %" #i ", %[a] offset:" #off "\n\t"
template <>
__device__ __forceinline__ void row_read<bf16_t, 7>(const bf16_t* buf, int lane, RawRow<7>& o, int px0) {
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  asm volatile(MRLA_D16(0, 0) MRLA_D16(1, 128) MRLA_D16(2, 256) MRLA_D16(3, 384) MRLA_D16(4, 512) MRLA_D16(5, 640)
               MRLA_D16(6, 768) "s_waitcnt lgkmcnt(0)"
               : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6])
               : [a] "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void row_read<bf16_t, 9>(const bf16_t* buf, int lane, RawRow<9>& o, int px0) {
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  asm volatile(MRLA_D16(0, 0) MRLA_D16(1, 128) MRLA_D16(2, 256) MRLA_D16(3, 384) MRLA_D16(4, 512) MRLA_D16(5, 640)
               MRLA_D16(6, 768) MRLA_D16(7, 896) MRLA_D16(8, 1024) "s_waitcnt lgkmcnt(0)"
               : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]),
                 "+v"(o.v[7]), "+v"(o.v[8])
               : [a] "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void row_read<bf16_t, 11>(const bf16_t* buf, int lane, RawRow<11>& o, int px0) {
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  asm volatile(MRLA_D16(0, 0) MRLA_D16(1, 128) MRLA_D16(2, 256) MRLA_D16(3, 384) MRLA_D16(4, 512) MRLA_D16(5, 640)
               MRLA_D16(6, 768) MRLA_D16(7, 896) MRLA_D16(8, 1024) MRLA_D16(9, 1152) MRLA_D16(10, 1280)
               "s_waitcnt lgkmcnt(0)"
               : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]),
                 "+v"(o.v[7]), "+v"(o.v[8]), "+v"(o.v[9]), "+v"(o.v[10])
               : [a] "v"(a) : "memory");
}

// Split form for several rows per step: issue the reads of all rows first, then fence them -- the first fence waits for
// every outstanding LDS read, the others only tie their registers to it (one LDS latency per step instead of one per row).
// For other element types the issue IS the complete (waited-for) read and the fences are no-ops.
template <typename T, int NPX>
__device__ __forceinline__ void row_read_issue(const T* buf, int lane, RawRow<NPX>& out, int px0 = 0) {
  row_read<T, NPX>(buf, lane, out, px0);
}
template <int NPX>
__device__ __forceinline__ void row_read_fence(RawRow<NPX>&, bool) {}
template <>
__device__ __forceinline__ void row_read_issue<bf16_t, 7>(const bf16_t* buf, int lane, RawRow<7>& o, int px0) {
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  asm volatile(MRLA_D16(0, 0) MRLA_D16(1, 128) MRLA_D16(2, 256) MRLA_D16(3, 384) MRLA_D16(4, 512) MRLA_D16(5, 640) MRLA_D16(6, 768) ""
               : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6])
               : [a] "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void row_read_fence<7>(RawRow<7>& o, bool wait) {
  if (wait) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]) : : "memory");
  else      asm volatile("" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]) : : "memory");
}
template <>
__device__ __forceinline__ void row_read_issue<bf16_t, 9>(const bf16_t* buf, int lane, RawRow<9>& o, int px0) {
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  asm volatile(MRLA_D16(0, 0) MRLA_D16(1, 128) MRLA_D16(2, 256) MRLA_D16(3, 384) MRLA_D16(4, 512) MRLA_D16(5, 640) MRLA_D16(6, 768) MRLA_D16(7, 896) MRLA_D16(8, 1024) ""
               : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]), "+v"(o.v[7]), "+v"(o.v[8])
               : [a] "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void row_read_fence<9>(RawRow<9>& o, bool wait) {
  if (wait) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]), "+v"(o.v[7]), "+v"(o.v[8]) : : "memory");
  else      asm volatile("" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]), "+v"(o.v[7]), "+v"(o.v[8]) : : "memory");
}
template <>
__device__ __forceinline__ void row_read_issue<bf16_t, 11>(const bf16_t* buf, int lane, RawRow<11>& o, int px0) {
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  asm volatile(MRLA_D16(0, 0) MRLA_D16(1, 128) MRLA_D16(2, 256) MRLA_D16(3, 384) MRLA_D16(4, 512) MRLA_D16(5, 640) MRLA_D16(6, 768) MRLA_D16(7, 896) MRLA_D16(8, 1024) MRLA_D16(9, 1152) MRLA_D16(10, 1280) ""
               : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]), "+v"(o.v[7]), "+v"(o.v[8]), "+v"(o.v[9]), "+v"(o.v[10])
               : [a] "v"(a) : "memory");
}
template <>
__device__ __forceinline__ void row_read_fence<11>(RawRow<11>& o, bool wait) {
  if (wait) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]), "+v"(o.v[7]), "+v"(o.v[8]), "+v"(o.v[9]), "+v"(o.v[10]) : : "memory");
  else      asm volatile("" : "+v"(o.v[0]), "+v"(o.v[1]), "+v"(o.v[2]), "+v"(o.v[3]), "+v"(o.v[4]), "+v"(o.v[5]), "+v"(o.v[6]), "+v"(o.v[7]), "+v"(o.v[8]), "+v"(o.v[9]), "+v"(o.v[10]) : : "memory");
}
#undef MRLA_D16


// lane = channel values v[0 .. NPX) of one row piece -> global row r (16 B per lane through the LDS buffer `buf`);
// `a` was made with npx = the number of pixels that exist (the rest is dropped by the bounds check).
template <typename T, int NPX>
__device__ __forceinline__ void row_store(const RowIO<T, NPX>& a, T* img, int r, int rowelems, int lane, T* buf,
                                          const float (&v)[NPX]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef RowIO<T, NPX> Q;
  typedef __attribute__((address_space(3))) T* lds_T_ptr;
  lds_T_ptr p = (lds_T_ptr)buf + lane;
#pragma unroll
  for (int j = 0; j < NPX; ++j) p[j * kWave] = from_f<T>(v[j]);
  const auto rs =
      __builtin_amdgcn_make_buffer_rsrc(img + (size_t)r * rowelems, 0, rowelems * (int)sizeof(T), kBufFlags);
  typedef __attribute__((address_space(3))) const u32x4* lds_v4_ptr;
  lds_v4_ptr s4 = (lds_v4_ptr)buf + lane;
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) __builtin_amdgcn_raw_buffer_store_b128(s4[l * kWave], rs, a.voff[l], 0, MRLA_ROW_STORE_AUX);
#endif
}

}  // namespace mrla
