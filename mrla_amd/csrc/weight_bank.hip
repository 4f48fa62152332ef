// bf16 working copies of fp32 convolution weights, all of a model in one launch (see mrla_weight_bank_refresh in
// include/mrla_hip.h): dst[n, k] = bf16(src[n, k]) and, where asked for, dst_t[k, n] = its transpose (the operand of the
// input-gradient GEMM dX = dY * W, mrla_conv1x1_fwd with w^T).  Replaces one autocast cast kernel per convolution and
// forward plus one transposing copy per input gradient (~70 tiny launches per resnet50_mrlal step).
#include "mrla_device.h"
#include "mrla_kernels.h"

namespace mrla {
namespace {

// grid (max_tiles, entries): a workgroup converts one 64 x 64 tile of one weight
__global__ __launch_bounds__(256) void weight_bank_kernel(const long long* __restrict__ table) {
  __shared__ unsigned short tile[64][66];
  const long long* e = table + (size_t)blockIdx.y * 4;
  const float* src = reinterpret_cast<const float*>(e[0]);
  bf16_t* dst = reinterpret_cast<bf16_t*>(e[1]);
  bf16_t* dst_t = reinterpret_cast<bf16_t*>(e[2]);
  const int n = (int)(e[3] >> 32), k = (int)(e[3] & 0xffffffffLL);
  const int tk = k / 64;
  if ((int)blockIdx.x >= (n / 64) * tk) return;
  const int n0 = (blockIdx.x / tk) * 64, k0 = (blockIdx.x % tk) * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i, r = idx >> 4, c4 = (idx & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(n0 + r) * k + k0 + c4);
    typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 o;
    o[0] = from_f<bf16_t>(v.x); o[1] = from_f<bf16_t>(v.y); o[2] = from_f<bf16_t>(v.z); o[3] = from_f<bf16_t>(v.w);
    *reinterpret_cast<bf16x4*>(dst + (size_t)(n0 + r) * k + k0 + c4) = o;
    if (dst_t) {
      const unsigned short* u = reinterpret_cast<const unsigned short*>(&o);
      tile[r][c4] = u[0]; tile[r][c4 + 1] = u[1]; tile[r][c4 + 2] = u[2]; tile[r][c4 + 3] = u[3];
    }
  }
  if (!dst_t) return;               // (uniform over the workgroup)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i, r = idx >> 4, c4 = (idx & 15) * 4;      // row r of the transposed tile = column r
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    u16x4 o;
    o[0] = tile[c4][r]; o[1] = tile[c4 + 1][r]; o[2] = tile[c4 + 2][r]; o[3] = tile[c4 + 3][r];
    *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(dst_t) + (size_t)(k0 + r) * n + n0 + c4) = o;
  }
}

}  // namespace

int launch_weight_bank_refresh(const long long* table, int entries, int max_tiles, hipStream_t st) {
  hipLaunchKernelGGL(weight_bank_kernel, dim3(max_tiles, entries), dim3(256), 0, st, table);
  return hip_status(hipGetLastError());
}

}  // namespace mrla
