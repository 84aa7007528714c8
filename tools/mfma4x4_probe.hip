// Probe (diagnostic only, not part of the product): operand and result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, and
// its rate.  Sixteen independent 4x4 blocks per instruction: lane l = 4 * block + i supplies A_block[i] and B_block[i].
// Prints whether register r of lane 4 * block + j holds A_block[r] * B_block[j], and cycles per instruction per SIMD with
// 1, 2 and 4 waves per SIMD (one or two accumulators per wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
  const int l = threadIdx.x, b = l >> 2, i = l & 3;
  const float a = 1.f + i, bb = 10.f + i + 100.f * b;
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate(float* out, float seed, int iters, unsigned long long* cyc) {
  f32x4 acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* d; hipMalloc(&d, 1 << 22);
  unsigned long long* c; hipMalloc(&c, 8);
  layout<<<1, 64>>>(d);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const float want = (1.f + r) * (10.f + (l & 3) + 100.f * (l >> 2));
      if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); ++bad; }
    }
  printf("layout: register r of lane 4*block+j = A_block[r] * B_block[j]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    for (int nacc = 1; nacc <= 2; ++nacc) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const dim3 grid(256 * wps), block(256);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (nacc == 1) rate<1><<<grid, block>>>(d, 1.f, iters, c); else rate<2><<<grid, block>>>(d, 1.f, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
      const double n = (double)iters * 64;
      printf("%d wave(s)/SIMD, %d accumulator(s): %.2f clock-counter ticks per MFMA per wave, %.2f ns per MFMA per SIMD, %.1f TFLOP/s chip\n",
             wps, nacc, cy / n, ms * 1e6 / (n * wps), 512.0 * n * wps * 1024 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
