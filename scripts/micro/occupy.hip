// Probe, not product: a persistent copy kernel on a chosen number of workgroups, standing in for RCCL's channel kernels
// (which occupy CUs and move the gradient buckets while the backward pass runs).  Built on the GPU box by
// scripts/cu_contention_probe.py:  hipcc --offload-arch=gfx950 -shared -fPIC scripts/micro/occupy.hip -o /tmp/liboccupy.so
#include <hip/hip_runtime.h>
#include <stdint.h>

// each workgroup copies its slice of src -> dst `rounds` times (16 B per lane, 512 threads: the shape of a collective's
// channel kernel), so n_wg workgroups stay resident for the duration and move n_wg * slice * rounds bytes each way
__global__ __launch_bounds__(512) void occupy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t slice_vec,
                                                     int rounds) {
  const uint4* s = src + (size_t)blockIdx.x * slice_vec;
  uint4* d = dst + (size_t)blockIdx.x * slice_vec;
  for (int r = 0; r < rounds; ++r)
    for (size_t i = threadIdx.x; i < slice_vec; i += blockDim.x) d[i] = s[i];
}

extern "C" int occupy(const void* src, void* dst, size_t bytes_per_wg, int n_wg, int rounds, void* stream) {
  hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(512), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                     bytes_per_wg / 16, rounds);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
