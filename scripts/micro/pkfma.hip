// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 at 1, 2 and 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/pkfma.hip -o /tmp/pkfma ; run: /tmp/pkfma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ void spin(float* out, int iters) {
  f2 a[8];
  const float s = out[threadIdx.x & 7], t = 1.0f + 1e-7f * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{s + i, s - i};
  const f2 m = {t, t}, c = {1e-3f, 2e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PK) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(m), "v"(c));
      } else {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i].x) : "v"(a[i].x), "v"(m.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i].y) : "v"(a[i].y), "v"(m.y), "v"(c.y));
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 24);
  hipMemset(d, 0, 1 << 24);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int pk = 0; pk < 2; ++pk) {
      const int threads = 64 * 4 * wps;        // 4 SIMDs x wps waves, one workgroup per CU
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(spin<true>, dim3(256), dim3(threads), 0, 0, d, iters);
        else hipLaunchKernelGGL(spin<false>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double fma_per_wave = (double)iters * 16;                   // scalar-FMA equivalents per lane
      const double cyc = ms * 1e-3 * 2.4e9;                              // at 2.4 GHz
      printf("%d wave(s)/SIMD %-12s %.3f ms  %.2f cycles per lane-FMA-pair-of-64 (wave64 FMA equivalents: %.2f cyc each per SIMD)\n",
             wps, pk ? "v_pk_fma_f32" : "v_fma_f32", ms, cyc / fma_per_wave, cyc / (fma_per_wave * wps));
    }
  }
  return 0;
}
