// Diagnostic (not part of the product): the library's own dense16 for the 208x208 layer in a loop of
// ITER layers per workgroup, 1..3 workgroups per compute unit -- what the wide layer sustains by
// itself, with its epilogue, barrier and tile split, but without the rest of a network pass.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irl-gan_amd/csrc tools/layer_probe.hip -o tools/layer_probe
#include "rlgan_device.h"
#include <cstdio>
using namespace rlgan;
constexpr int ITER = 16;

// the same loop for a layer with NG k-groups (NG = 2: the first layer of a network; NG = 1: the least there can be)
template <int NG>
__global__ __launch_bounds__(THREADS) void ks(const float* __restrict__ w, float* out, int pad) {
  extern __shared__ float dyn[];
  __shared__ __attribute__((aligned(16))) float h1[WIDE_FLOATS], h2[WIDE_FLOATS], sp[4 * ROWS * LDP];
  for (int i = threadIdx.x; i < WIDE_FLOATS; i += THREADS) { h1[i] = 0.001f * (i % 13); h2[i] = 0.f; }
  __syncthreads();
  Pre<NT_H, NG> p;
  p.issue(w);
  float* a = h1; float* b = h2;
  for (int it = 0; it < ITER; ++it) {
    float* dst = b;
    dense16(p, a, LD_W, NoPre{}, [=](int r, int c, float z, float) {
      const float v = c < HID ? act_f<ACT_RELU6>(z) : (c == HID ? 1.f : 0.f);
      dst[r * LD_W + c] = v;
      return v;
    }, nullptr, [&] { p.issue(w); }, sp);
    float* t = a; a = b; b = t;
  }
  if (a[threadIdx.x] == 12345.678f) out[threadIdx.x] = a[threadIdx.x + 1] + dyn[pad & 1];
}

__global__ __launch_bounds__(THREADS) void k(const float* __restrict__ w, float* out, int pad) {
  extern __shared__ float dyn[];
  __shared__ __attribute__((aligned(16))) float h1[WIDE_FLOATS], h2[WIDE_FLOATS], sp[4 * ROWS * LDP];
  for (int i = threadIdx.x; i < WIDE_FLOATS; i += THREADS) { h1[i] = 0.001f * (i % 13); h2[i] = 0.f; }
  __syncthreads();
  Pre<NT_H, NT_H> p;
  p.issue(w);
  float* a = h1; float* b = h2;
  for (int it = 0; it < ITER; ++it) {
    float* dst = b;
    dense16(p, a, LD_W, NoPre{}, [=](int r, int c, float z, float) {
      const float v = c < HID ? act_f<ACT_RELU6>(z) : (c == HID ? 1.f : 0.f);
      dst[r * LD_W + c] = v;
      return v;
    }, nullptr, [&] { p.issue(w); }, sp);
    float* t = a; a = b; b = t;
  }
  if (a[threadIdx.x] == 12345.678f) out[threadIdx.x] = a[threadIdx.x + 1] + dyn[pad & 1];
}

int main() {
  float* w; float* out;
  hipMalloc(&w, NT_H * NT_H * 256 * sizeof(float));
  hipMemset(w, 0, NT_H * NT_H * 256 * sizeof(float));
  hipMalloc(&out, 4096);
  for (int wg = 1; wg <= 3; ++wg) {
    const size_t dyn = wg == 1 ? 100 * 1024 : (wg == 2 ? 40 * 1024 : 20 * 1024);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256 * wg), dim3(THREADS), dyn, 0, w, out, 0);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(256 * wg), dim3(THREADS), dyn, 0, w, out, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double flops_alg = 256.0 * wg * ITER * 16 * 201 * 200 * 2;      // algorithmic: [16 x 201] x [201 x 200]
    const double flops_mfma = 256.0 * wg * ITER * 13 * 13 * 4 * 2048.0;    // issued
    printf("%d workgroup(s)/CU: %7.1f us per launch, %.2f us per layer and workgroup, %6.1f TFLOP/s algorithmic, %6.1f issued (%s)\n",
           wg, us, us / ITER / wg, flops_alg / us / 1e6, flops_mfma / us / 1e6, hipGetErrorString(hipGetLastError()));
  }
  for (int ng = 1; ng <= 2; ++ng) {
    for (int wg = 1; wg <= 3; ++wg) {
      const size_t dyn = wg == 1 ? 100 * 1024 : (wg == 2 ? 40 * 1024 : 20 * 1024);
      auto fn = ng == 1 ? ks<1> : ks<2>;
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(256 * wg), dim3(THREADS), dyn, 0, w, out, 0);
      hipEventRecord(e0);
      const int reps = 20;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(256 * wg), dim3(THREADS), dyn, 0, w, out, 0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      printf("%d k-group(s), %d workgroup(s)/CU: %7.1f us per launch, %.3f us per layer and workgroup (MFMA issue alone: %.3f us)\n",
             ng, wg, us, us / ITER / wg, ng * 13 * 4 * 32 / 4.0 / 2400.0);
    }
  }
  return 0;
}
