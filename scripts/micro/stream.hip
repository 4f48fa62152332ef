// Micro-benchmark: what a plain streaming kernel reaches on this GPU at the read : write mixes of the MRLA passes -- the
// practical ceiling the roofline fractions of DESIGN.md section 5 are read against (the 8 TB/s of the guide is the pin
// rate; nothing with a write stream in it gets there).
//   R reads and W writes of `n` bytes each, 16 B per lane, every stream its own buffer, two sets of buffers used in turn
//   (the 256 MB Infinity Cache cannot serve the next launch), grid = one 256-thread workgroup per 4 KB x UNROLL chunk or a
//   persistent grid (8 workgroups per CU) striding over the chunks.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/stream.hip -o /tmp/stream ; run: /tmp/stream [MB per stream ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

struct Ptrs { const f4* r[4]; f4* w[2]; };

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void stream(Ptrs p, size_t n16, f4* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < R; ++s) v += NT ? __builtin_nontemporal_load(p.r[s] + i) : p.r[s][i];
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const f4 o = v * (float)(s + 1);
      if (NT) __builtin_nontemporal_store(o, p.w[s] + i); else p.w[s][i] = o;
    }
    if (W == 0) acc += v;
  }
  if (W == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc;      // keeps the loads alive
}

// The same mix in the GEOMETRY of the NHWC row pipeline (nhwc_rows.h) without its LDS stage, halo or arithmetic: bf16
// [B, H, Wd, C] tensors, a wave owns 64 channels (128 B per pixel) x PX columns and walks down a band of rows, lanes
// 8p .. 8p+7 move pixel p's 128 B (16 B per lane; PX = 7: lanes 56 .. 63 idle).  Which wave gets which (channel group,
// strip, image, row band) is the MAP:
//   0  the kernels' map: grid (C/64, B), the waves of a workgroup are the strips of one channel group
//   1  grid (strips, B, C/64/waves): the waves of a workgroup are neighbouring channel groups of one strip (a workgroup
//      touches contiguous waves x 128 B per pixel)
//   2  map 0 with the workgroups renumbered so that the channel groups of an image run on the same XCD (id % 8)
//   3  map 0 with the rows split in bands of 14 (4 x the workgroups on the 56-row stage)
//   4  a WAVE owns a strip x min(C/64, 8) neighbouring channel groups: each of its instructions moves 1 KB that is
//      contiguous in memory (8 channel groups of a pixel; the whole 3.5 KB strip row when C = 256); grid (strips, B, C/512)
struct Geo { int H, Wd, C, nstrips, ncg, band; };

template <int R, int W, int PX, int MAP, bool NT>
__global__ __launch_bounds__(512) void rowcopy(Ptrs p, Geo g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, px = lane >> 3, part = lane & 7;
  int cg, strip, img, r0 = 0, r1 = g.H;
  if (MAP == 1) {
    strip = blockIdx.x; img = blockIdx.y; cg = blockIdx.z * (blockDim.x >> 6) + wave;
  } else if (MAP == 2) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id & 7, k = id >> 3;       // k-th workgroup of this XCD
    const int per = gridDim.x * gridDim.y / 8;                                          // workgroups per XCD
    const int lin = xcd * per + k;                                                      // XCD x runs ids [x*per, (x+1)*per)
    cg = lin % g.ncg; img = lin / g.ncg; strip = wave;
  } else if (MAP == 3) {
    cg = blockIdx.x; img = blockIdx.y; strip = wave; r0 = blockIdx.z * g.band; r1 = min(g.H, r0 + g.band);
  } else {
    cg = blockIdx.x; img = blockIdx.y; strip = wave;
  }
  if (px >= PX || strip * PX + px >= g.Wd) return;
  const size_t rowb = (size_t)g.Wd * g.C * 2;
  const size_t off0 = (size_t)img * g.H * rowb + ((size_t)(strip * PX + px) * g.C + cg * 64 + part * 8) * 2;
  for (int r = r0; r < r1; ++r) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const f4* q = (const f4*)((const char*)p.r[s] + off0 + (size_t)r * rowb);
      v += NT ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int s = 0; s < W; ++s) {
      f4* q = (f4*)((char*)p.w[s] + off0 + (size_t)r * rowb);
      if (NT) __builtin_nontemporal_store(v * (float)(s + 1), q); else *q = v * (float)(s + 1);
    }
  }
}

template <int R, int W, bool NT>
__global__ __launch_bounds__(64) void rowcopy4(Ptrs p, Geo g) {
  const int lane = threadIdx.x, ncgw = g.ncg < 8 ? g.ncg : 8, per_px = ncgw * 8;          // 16-byte units per pixel
  const size_t rowb = (size_t)g.Wd * g.C * 2;
  const size_t base = (size_t)blockIdx.y * g.H * rowb + ((size_t)blockIdx.x * 7 * g.C + blockIdx.z * 512) * 2;
  const int units = 7 * per_px;
  for (int r = 0; r < g.H; ++r) {
    for (int u0 = 0; u0 < units; u0 += 256) {
      f4 v[4];
      size_t off[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int u = u0 + k * 64 + lane, px = u / per_px, in = u - px * per_px;
        off[k] = u < units ? base + (size_t)r * rowb + ((size_t)px * g.C) * 2 + (size_t)in * 16 : (size_t)-1;
        v[k] = f4{0.f, 0.f, 0.f, 0.f};
        if (u < units) {
#pragma unroll
          for (int s = 0; s < R; ++s) {
            const f4* q = (const f4*)((const char*)p.r[s] + off[k]);
            v[k] += NT ? __builtin_nontemporal_load(q) : *q;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (off[k] != (size_t)-1) {
#pragma unroll
          for (int s = 0; s < W; ++s) {
            f4* q = (f4*)((char*)p.w[s] + off[k]);
            if (NT) __builtin_nontemporal_store(v[k] * (float)(s + 1), q); else *q = v[k] * (float)(s + 1);
          }
        }
    }
  }
}

template <int R, int W, int PX, int MAP, bool NT>
static float run_rows(const Ptrs set[2], int B, int HW, int C, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  Geo g = {HW, HW, C, (HW + PX - 1) / PX, C / 64, 14};
  dim3 grid(g.ncg, B), block(g.nstrips * 64);
  if (MAP == 1) { const int wv = g.ncg < 8 ? g.ncg : 8; grid = dim3(g.nstrips, B, g.ncg / wv); block = dim3(wv * 64); }
  if (MAP == 3) grid.z = (HW + g.band - 1) / g.band;
  if (MAP == 4) { grid = dim3(g.nstrips, B, (C + 511) / 512); block = dim3(64); }
  auto go = [&](int i) {
    if (MAP == 4) rowcopy4<R, W, NT><<<grid, block>>>(set[i & 1], g);
    else rowcopy<R, W, PX, MAP, NT><<<grid, block>>>(set[i & 1], g);
  };
  for (int i = 0; i < 2; ++i) go(i);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) go(i);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int R, int W, bool NT>
static float run(const Ptrs set[2], size_t n16, int grid, f4* sink, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) stream<R, W, NT><<<grid, 256>>>(set[i & 1], n16, sink);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) stream<R, W, NT><<<grid, 256>>>(set[i & 1], n16, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  std::vector<size_t> mbs;
  for (int i = 1; i < argc; ++i) mbs.push_back(strtoul(argv[i], 0, 10));
  if (mbs.empty()) mbs = {51, 103, 206, 411};        // the four stages' N x 2 B of resnet50_mrlal at batch 256
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs; GB/s over (R + W) x n bytes; default | nontemporal policy; grid: one chunk per workgroup / %d persistent\n",
         prop.gcnArchName, cus, cus * 8);
  f4* sink;
  hipMalloc(&sink, 64);
  for (size_t mb : mbs) {
    const size_t n = mb << 20, n16 = n / 16;
    Ptrs set[2];
    for (int k = 0; k < 2; ++k) {
      for (int s = 0; s < 4; ++s) { void* q; hipMalloc(&q, n); hipMemset(q, 0, n); set[k].r[s] = (const f4*)q; }
      for (int s = 0; s < 2; ++s) { void* q; hipMalloc(&q, n); hipMemset(q, 0, n); set[k].w[s] = (f4*)q; }
    }
    const int reps = 20;
    for (int persistent = 0; persistent < 2; ++persistent) {
      const int grid = persistent ? cus * 8 : (int)((n16 + 255) / 256);
#define ROW(R, W)                                                                                              \
  {                                                                                                            \
    const float a = run<R, W, false>(set, n16, grid, sink, reps), b = run<R, W, true>(set, n16, grid, sink, reps); \
    printf("n=%4zu MB  %dR:%dW  grid %-10s  %7.1f us %6.0f GB/s | %7.1f us %6.0f GB/s\n", mb, R, W,           \
           persistent ? "persistent" : "chunked", a * 1e3, (R + W) * (double)n / a * 1e-6, b * 1e3,            \
           (R + W) * (double)n / b * 1e-6);                                                                    \
  }
      ROW(1, 0) ROW(3, 0) ROW(1, 1) ROW(2, 1) ROW(3, 2) ROW(4, 2)
    }
    // row geometry at this size: (C, H = Wd) of the ResNet-50 stage whose N x 2 B this is (batch 256)
    const int geo[4][3] = {{51, 2048, 7}, {103, 1024, 14}, {206, 512, 28}, {411, 256, 56}};
    for (int g = 0; g < 4; ++g) {
      if ((size_t)geo[g][0] != mb) continue;
      const int C = geo[g][1], HW = geo[g][2];
#define RROW(R, W, PX, MAP)                                                                                    \
  {                                                                                                            \
    const float a = run_rows<R, W, PX, MAP, false>(set, 256, HW, C, reps), b = run_rows<R, W, PX, MAP, true>(set, 256, HW, C, reps); \
    const double nb = 256.0 * HW * HW * C * 2;                                                                 \
    printf("n=%4zu MB  %dR:%dW  rows px%d map%d c=%4d %2dx%-2d  %7.1f us %6.0f GB/s | %7.1f us %6.0f GB/s\n", mb, R, W, PX, MAP, C, HW, HW, \
           a * 1e3, (R + W) * nb / a * 1e-6, b * 1e3, (R + W) * nb / b * 1e-6);                                \
  }
      RROW(2, 1, 7, 0) RROW(2, 1, 7, 1) RROW(2, 1, 7, 4)
      RROW(4, 2, 7, 0) RROW(4, 2, 7, 1) RROW(4, 2, 7, 4)
    }
    for (int k = 0; k < 2; ++k) {
      for (int s = 0; s < 4; ++s) hipFree((void*)set[k].r[s]);
      for (int s = 0; s < 2; ++s) hipFree(set[k].w[s]);
    }
  }
  return 0;
}
