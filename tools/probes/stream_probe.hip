// Per-CU weight-stream probe: what bounds the imagination kernels' 115 GB/s per CU
// (docs/LABLOG.md, round 5)?  One workgroup per CU streams the same `bytes`-sized buffer
// (fragment-major planes: a wave's load = 1 KB contiguous, 16 B per lane) `reps` times with DEPTH
// 16-byte loads in flight per lane; reports GB/s per CU from the device wall clock (100 MHz) and the
// shader clock per wall-clock tick.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DEPTH, int THREADS, int POLICY>   // POLICY 0 plain, 1 nontemporal
__global__ void __launch_bounds__(THREADS, 1)
k_stream(const u32x4* __restrict__ buf, long n16, int reps, unsigned* sink, unsigned long long* stamps) {
  const int tid = threadIdx.x;
  constexpr int WAVES = THREADS / 64;
  const int wave = tid >> 6, lane = tid & 63;
  // wave w reads 1 KB blocks w, w + WAVES, ... (like the column tiles of imag.hip's Stream)
  const long iters = n16 / 64 / (DEPTH * WAVES);   // whole windows only
  u32x4 r[DEPTH];
  unsigned acc = 0;
  unsigned long long t0 = 0, c0 = 0;
  if (tid == 0) { t0 = wall_clock64(); c0 = clock64(); }
  for (int rep = 0; rep < reps; ++rep) {
    const u32x4* p = buf + (long)wave * 64 + lane;
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const u32x4* q = p + (long)j * WAVES * 64;
      r[j] = POLICY ? __builtin_nontemporal_load(q) : *q;
    }
    for (long it = 1; it < iters; ++it) {
      p += (long)DEPTH * WAVES * 64;
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
        const u32x4* q = p + (long)j * WAVES * 64;
        r[j] = POLICY ? __builtin_nontemporal_load(q) : *q;
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
  __syncthreads();
  if (tid == 0) {
    stamps[blockIdx.x * 2] = wall_clock64() - t0;
    stamps[blockIdx.x * 2 + 1] = clock64() - c0;
  }
}

template <int DEPTH, int THREADS, int POLICY>
void run(const u32x4* buf, long bytes, int wgs, int reps, unsigned* sink, unsigned long long* stamps) {
  const long n16 = bytes / 16;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) {
    CK(hipEventRecord(e0));
    k_stream<DEPTH, THREADS, POLICY><<<wgs, THREADS>>>(buf, n16, reps, sink, stamps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(wgs * 2);
  CK(hipMemcpy(h.data(), stamps, wgs * 16, hipMemcpyDeviceToHost));
  double wall = 0, clk = 0;
  for (int i = 0; i < wgs; ++i) { wall += h[2 * i]; clk += h[2 * i + 1]; }
  wall /= wgs; clk /= wgs;
  const double us = wall / 100.0;             // 100 MHz
  bytes = bytes / 16 / 64 / (DEPTH * (THREADS / 64)) * (DEPTH * (THREADS / 64)) * 1024;   // whole windows only
  const double per_cu = (double)bytes * reps / us * 1e-3;   // GB/s
  printf("bytes %8.2f MB wgs %3d threads %3d depth %2d policy %d : %7.1f us/launch(event) %7.1f us(block) "
         "%6.1f GB/s per CU  %5.2f TB/s total  clk/wall %.2f (GHz %.2f)  B/clk %.1f\n",
         bytes / 1048576.0, wgs, THREADS, DEPTH, POLICY, ms * 1e3, us, per_cu, per_cu * wgs * 1e-3,
         clk / wall, clk / wall * 0.1, (double)bytes * reps / clk);
}

int main() {
  const long maxb = 256l << 20;
  u32x4* buf; unsigned* sink; unsigned long long* stamps;
  CK(hipMalloc(&buf, maxb)); CK(hipMemset(buf, 1, maxb));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stamps, 1024 * 16));
  const long sizes[] = {1l << 20, 3l << 20, 21l << 19 /* 10.5 MB */, 64l << 20};
  for (long bytes : sizes) {
    const int reps = (int)((160l << 20) / bytes);
    for (int wgs : {8, 157, 256}) {
      run<8, 256, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<24, 256, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<48, 256, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<96, 256, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<24, 512, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<48, 512, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<24, 1024, 0>(buf, bytes, wgs, reps, sink, stamps);
      run<48, 256, 1>(buf, bytes, wgs, reps, sink, stamps);
    }
  }
  return 0;
}
