// Micro-benchmark: wave64 issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (independent and dependent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

template <int CH>
__global__ void scalar_fma(float* out, float a, float b) {
  float acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < kIters; ++it)
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ void packed_fma(float* out, float a, float b) {
  f2 acc[CH];
  const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
  for (int i = 0; i < CH; ++i) acc[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
  for (int it = 0; it < kIters; ++it)
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
float run(K k, float* d, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  const int blocks = 256 * 8;        // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  const double lanes = (double)blocks * 256;
  struct { const char* name; float ms; double fma; } r[] = {
    {"v_fma_f32     1 chain ", run(scalar_fma<1>, d, blocks), lanes * kIters * 1},
    {"v_fma_f32     8 chains", run(scalar_fma<8>, d, blocks), lanes * kIters * 8},
    {"v_pk_fma_f32  1 chain ", run(packed_fma<1>, d, blocks), lanes * kIters * 2},
    {"v_pk_fma_f32  8 chains", run(packed_fma<8>, d, blocks), lanes * kIters * 16},
  };
  for (auto& x : r) printf("%s  %8.3f ms  %7.1f TFLOP/s (fma = 2 flop)\n", x.name, x.ms, 2 * x.fma / x.ms / 1e9);
  return 0;
}
