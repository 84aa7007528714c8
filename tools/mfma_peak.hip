// Micro-benchmark (diagnostic only, not part of the product): what rate of v_mfma_f32_16x16x4_f32 a
// compute unit sustains (a) from registers only, (b) with the A operand re-read from LDS every 12
// MFMAs, (c) additionally with the B operands streamed from an L2-resident 208x208 fragment image,
// (d) additionally with 12 or 48 vector FMAs per 12 MFMAs (vector ALU work is paid in MFMA time:
// about 4 cycles per instruction), for 1..3 workgroups of 4 waves per compute unit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 13, NG = 13, ITER = 16;   // ITER "layers" of 13 groups x 3 tiles x 4 MFMAs per wave

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int lds_pad) {
  extern __shared__ float dyn[];
  __shared__ __attribute__((aligned(16))) float xs[16 * 212];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 212; i += 256) xs[i] = 0.001f * (i % 17);
  __syncthreads();
  const float* xq = xs + (lane & 15) * 212 + (lane >> 4) * 4;
  const float* wl = w + ((wave * 3) * 64 + lane) * 4;
  f32x4 acc[3] = {};
  float fm[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  f32x4 a = *(const f32x4*)xq;
  f32x4 b[3];
  for (int i = 0; i < 3; ++i) b[i] = *(const f32x4*)(wl + i * 256);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f32x4 an = a, bn[3] = {b[0], b[1], b[2]};
      if (MODE >= 1) an = *(const f32x4*)(xq + 16 * ((g + 1) % NG));
      if (MODE >= 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bn[i] = *(const f32x4*)(wl + (((g + 1) % NG) * NT + i) * 256);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[i][m], acc[i], 0, 0, 0);
      }
      if (MODE >= 3) {
        // MODE 3: 12 independent vector FMAs per group (one per MFMA); MODE 4: 48
#pragma unroll
        for (int v = 0; v < (MODE == 3 ? 12 : 48); ++v) fm[v & 7] = __builtin_fmaf(fm[v & 7], 1.0001f, 0.5f);
      }
      a = an; b[0] = bn[0]; b[1] = bn[1]; b[2] = bn[2];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2];
  for (int v = 0; v < 8; ++v) s[0] += fm[v];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[threadIdx.x] = s[0] + dyn[lds_pad & 1];
}

template <int MODE>
void run(const float* w, float* out, int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu;
  const size_t dyn = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 60 * 1024 : 30 * 1024);   // pins the residency
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), dyn, 0, w, out, 0);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), dyn, 0, w, out, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * ITER * NG * 12 * 2048.0 * reps;
  printf("mode %d  %d workgroup(s)/CU: %8.1f us per launch  %7.1f TFLOP/s  (%s)\n", MODE, wgs_per_cu, ms * 1e3 / reps,
         flops / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
  float* w; float* out;
  hipMalloc(&w, NG * NT * 256 * sizeof(float));
  hipMemset(w, 0, NG * NT * 256 * sizeof(float));
  hipMalloc(&out, 4096);
  for (int wg = 1; wg <= 3; ++wg) { run<0>(w, out, wg); run<1>(w, out, wg); run<2>(w, out, wg); run<3>(w, out, wg); run<4>(w, out, wg); }
  hipDeviceSynchronize();
  return 0;
}
