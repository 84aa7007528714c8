// Micro-benchmark (diagnostic only): rate of 16-byte-per-lane vector loads of one 1 KiB block per wave
// for three lane->address maps: 0 = lane l reads bytes 16*l (consecutive lanes, consecutive 16 B);
// 1 = lane (q = l >> 4, j = l & 15) reads bytes 64*j + 16*q (the [column][16 rows] fragment order);
// 2 = lane (q, j) reads bytes 256*q + 16*j (the [4-row block][column][4 rows] order).
// 3 = lane (q, j) reads bytes 3328*q + 16*j (the same with 208 columns between the row blocks).
// Data stay cache resident (each wave re-reads a 16 KiB window).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ta_probe.hip -o tools/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 64, UNR = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int step) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int off = MODE == 0 ? 4 * lane : (MODE == 1 ? 16 * j + 4 * q : (MODE == 2 ? 64 * q + 4 * j : 832 * q + 4 * j));
  const float* p = src + (size_t)(blockIdx.x % 64) * 65536 + wave * 4096 + off;
  f32x4 s{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < ITER; ++it) {
    f32x4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = *(const f32x4*)(p + ((it * step) & 3) * 4096 * 4 + u * 256);
#pragma unroll
    for (int u = 0; u < UNR; ++u) s += v[u];
    __builtin_amdgcn_sched_barrier(0);
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[threadIdx.x] = s[0];
}

template <int MODE>
void run(const float* src, float* out, int wgs_per_cu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, 1);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double loads_per_cu = (double)wgs_per_cu * 4 * ITER * UNR;
  const double us = ms * 1e3 / reps;
  printf("map %d  %d workgroup(s)/CU: %7.1f us  %6.1f ns per wave-load per CU  %6.2f TB/s chip\n", MODE, wgs_per_cu, us,
         us * 1e3 / loads_per_cu, loads_per_cu * 256 * 1024 / (us * 1e-6) / 1e12);
}

int main() {
  float* src; float* out;
  const size_t n_src = 64 * 65536 + 8192;   // map 3 reaches 2304 floats past a block's 65536-float window
  hipMalloc(&src, n_src * 4); hipMemset(src, 0, n_src * 4); hipMalloc(&out, 4096);
  for (int wg = 1; wg <= 4; wg *= 2) { run<0>(src, out, wg); run<1>(src, out, wg); run<2>(src, out, wg); run<3>(src, out, wg); }
  hipDeviceSynchronize();
  return 0;
}
