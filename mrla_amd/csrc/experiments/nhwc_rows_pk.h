// Row windows held as PAIRS of neighbouring columns, for packed FP32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32: two lanes' worth of FP32 work per issued instruction -- the only way to the FP32 vector peak on gfx950,
// MI355X_MICROARCH.md).  A packed operand is an EVEN-ALIGNED register pair, so a window of N columns is cut into
//   [head: column 0, iff PH = 1]  pairs (PH, PH+1), (PH+2, PH+3), ...  [tail: column N-1, iff N - PH is odd]
// and the stencil taps whose operand pair would straddle two of these pairs run as two plain FMAs on the halves (a
// sub-register of a pair is an ordinary register: no moves).  Written by profiles/r06_notes.md section 2.
//
// LDS reads: one `ds_read_u16_d16_hi` per bf16 value straight into its half of a pair (the register then IS the fp32
// value, nhwc_rows.h).  The reads are separate asm statements on the scalar halves and the FENCE ties the 64-bit pairs: a
// single asm with every half as a tied operand makes the register coalescer copy each odd half (one v_mov per pair and row).
#pragma once
#include "../nhwc_rows.h"

namespace mrla {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat2(float a) { return (v2f){a, a}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

template <int N, int PH>
struct PkRow {
  static_assert(PH == 0 || PH == 1, "phase of the first pair");
  static constexpr int NP = (N - PH) / 2;
  static constexpr bool TAIL = ((N - PH) & 1) != 0;
  float head;                                      // column 0 (PH = 1 only)
  v2f p[NP];                                       // columns (PH + 2q, PH + 2q + 1)
  float tail;                                      // column N - 1 (TAIL only)
  __device__ __forceinline__ void clear() {
    head = 0.f; tail = 0.f;
#pragma unroll
    for (int q = 0; q < NP; ++q) p[q] = splat2(0.f);
  }
  // column j (a constant after unrolling)
  __device__ __forceinline__ float at(int j) const {
    if (PH && j == 0) return head;
    if (TAIL && j == N - 1) return tail;
    return ((j - PH) & 1) ? p[(j - PH) >> 1].y : p[(j - PH) >> 1].x;
  }
};

// out[column j] = float(buf[pixel px0 + j][this lane's channel]); rows_landed() first, pk_read_fence() before any use.
template <typename T, int N, int PH>
__device__ __forceinline__ void pk_read_issue(const T* buf, int lane, PkRow<N, PH>& o, int px0 = 0) {
  typedef PkRow<N, PH> R;
  typedef __attribute__((address_space(3))) const T* lds_T_ptr;
  lds_T_ptr p = (lds_T_ptr)buf + px0 * kWave + lane;
  if (PH) o.head = to_f(p[0]);
#pragma unroll
  for (int q = 0; q < R::NP; ++q) o.p[q] = (v2f){to_f(p[(PH + 2 * q) * kWave]), to_f(p[(PH + 2 * q + 1) * kWave])};
  if (R::TAIL) o.tail = to_f(p[(N - 1) * kWave]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#define MRLA_PK_D16(dst, j) \
  asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(dst) : "v"(a), "n"((j) * 2 * kWave) : "memory")
template <int N, int PH>
__device__ __forceinline__ void pk_read_issue_bf16(const bf16_t* buf, int lane, PkRow<N, PH>& o, int px0) {
  typedef PkRow<N, PH> R;
  const unsigned a = lds_addr_of(buf) + (px0 * kWave + lane) * 2;
  if (PH) MRLA_PK_D16(o.head, 0);
#pragma unroll
  for (int q = 0; q < R::NP; ++q) {
    float lo = o.p[q].x, hi = o.p[q].y;            // (the low halves of the registers stay zero: nothing else writes them)
    MRLA_PK_D16(lo, PH + 2 * q);
    MRLA_PK_D16(hi, PH + 2 * q + 1);
    o.p[q] = (v2f){lo, hi};
  }
  if (R::TAIL) MRLA_PK_D16(o.tail, N - 1);
}
#undef MRLA_PK_D16
template <> __device__ __forceinline__ void pk_read_issue<bf16_t, 11, 1>(const bf16_t* b, int l, PkRow<11, 1>& o, int px0) { pk_read_issue_bf16<11, 1>(b, l, o, px0); }
template <> __device__ __forceinline__ void pk_read_issue<bf16_t, 9, 1>(const bf16_t* b, int l, PkRow<9, 1>& o, int px0) { pk_read_issue_bf16<9, 1>(b, l, o, px0); }
template <> __device__ __forceinline__ void pk_read_issue<bf16_t, 7, 0>(const bf16_t* b, int l, PkRow<7, 0>& o, int px0) { pk_read_issue_bf16<7, 0>(b, l, o, px0); }

// `wait`: every LDS read issued so far has returned; the row's registers are tied to that point either way (the compiler
// cannot move a use of them above it).
template <int N, int PH>
__device__ __forceinline__ void pk_read_fence(PkRow<N, PH>& o, bool wait) {
  typedef PkRow<N, PH> R;
  if (wait) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (PH) asm volatile("" : "+v"(o.head) : : "memory");
#pragma unroll
  for (int q = 0; q < R::NP; ++q) asm volatile("" : "+v"(o.p[q]) : : "memory");
  if (R::TAIL) asm volatile("" : "+v"(o.tail) : : "memory");
}

// owned columns 0 .. kS-1 of one row -> global row r through the LDS buffer `buf` (16 B per lane, row_store's route); 16-bit
// types are converted two at a time (one v_cvt_pk per pair; its halves leave with ds_write_b16 / ds_write_b16_d16_hi)
template <typename T>
__device__ __forceinline__ void pk_row_store(const RowIO<T, 7>& a, T* img, int r, int rowelems, int lane, T* buf,
                                             const PkRow<7, 0>& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef RowIO<T, 7> Q;
  typedef __attribute__((address_space(3))) T* lds_T_ptr;
  lds_T_ptr p = (lds_T_ptr)buf + lane;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if constexpr (sizeof(T) == 2) {
      typedef T t2 __attribute__((ext_vector_type(2)));
      const t2 h = __builtin_convertvector(v.p[q], t2);
      p[(2 * q) * kWave] = h.x;
      p[(2 * q + 1) * kWave] = h.y;
    } else {
      p[(2 * q) * kWave] = from_f<T>(v.p[q].x);
      p[(2 * q + 1) * kWave] = from_f<T>(v.p[q].y);
    }
  }
  p[6 * kWave] = from_f<T>(v.tail);
  const auto rs =
      __builtin_amdgcn_make_buffer_rsrc(img + (size_t)r * rowelems, 0, rowelems * (int)sizeof(T), kBufFlags);
  typedef __attribute__((address_space(3))) const u32x4* lds_v4_ptr;
  lds_v4_ptr s4 = (lds_v4_ptr)buf + lane;
#pragma unroll
  for (int l = 0; l < Q::NL; ++l) __builtin_amdgcn_raw_buffer_store_b128(s4[l * kWave], rs, a.voff[l], 0, MRLA_ROW_STORE_AUX);
#endif
}

// the value a pair has once it is stored as T (what a later pass reads back)
template <typename T>
__device__ __forceinline__ v2f pk_as_stored(v2f y) {
  if constexpr (sizeof(T) == 2) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    const t2 h = __builtin_convertvector(y, t2);
    return (v2f){to_f(h.x), to_f(h.y)};
  } else {
    return y;
  }
}

}  // namespace mrla
