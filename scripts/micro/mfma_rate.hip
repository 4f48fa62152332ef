// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 with 8 independent accumulators and operands held in registers,
// at 1 and 2 waves per SIMD (no LDS, no memory in the loop).  Build + run: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void spin(float* out, int iters) {
  bf16x8 a[4], b[4];
  f32x16 acc[2][4];
  const float s = out[threadIdx.x & 7];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(s + i + e); b[i][e] = (__bf16)(s - i + e); }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][c][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[p][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], b[p], acc[p][c], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) r += acc[p][c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 24);
  (void)hipMemset(d, 0, 1 << 24);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 64 * 4 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(256), dim3(threads), 0, 0, d, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * wps;
    const double flops = 256.0 * 4 * mfma_per_simd * 32 * 32 * 16 * 2;
    printf("%d wave(s)/SIMD: %.3f ms, %.0f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", wps, ms, flops / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / mfma_per_simd);
  }
  return 0;
}
