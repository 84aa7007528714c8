// Micro-benchmark (diagnostic only, not part of the product): what a vector-ALU instruction costs among
// v_mfma_f32_16x16x4_f32 on gfx950, by placement.  A wave runs blocks of 16 MFMAs (two accumulators in turn) with V
// vector FMAs per block either CLUSTERED after the 16 MFMAs or INTERLEAVED (V/16 after every MFMA); 1, 2 or 4 waves per
// SIMD.  Prints cycles per block per SIMD (s_memtime) and the time per block from events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_mix.hip -o tools/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BLOCKS = 2000;

template <int V, bool INTER>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  f32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float fm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fm[i] = seed + i;
  for (int it = 0; it < BLOCKS; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
      if (INTER) {
#pragma unroll
        for (int v = 0; v < V / 16; ++v) fm[(m * (V / 16) + v) & 7] = __builtin_fmaf(fm[(m * (V / 16) + v) & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!INTER) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < V; ++v) fm[v & 7] = __builtin_fmaf(fm[v & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = acc0 + acc1;
  for (int v = 0; v < 8; ++v) s[0] += fm[v];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[threadIdx.x] = s[0];
}

template <int V, bool INTER>
void run(float* out, int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<V, INTER>), dim3(grid), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<V, INTER>), dim3(grid), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_block = ms * 1e6 / reps / BLOCKS / wgs_per_cu;     // per SIMD: every workgroup puts one wave on each
  printf("V=%2d %-11s %d wave(s)/SIMD: %7.1f ns per block of 16 MFMAs per SIMD (16 MFMAs alone = 512 cycles = 213 ns at 2.4 GHz)  %s\n", V,
         INTER ? "interleaved" : "clustered", wgs_per_cu, ns_per_block, hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  for (int w : {1, 2, 4}) {
    run<0, false>(out, w);
    run<16, false>(out, w);
    run<16, true>(out, w);
    run<32, false>(out, w);
    run<32, true>(out, w);
    run<64, false>(out, w);
    run<64, true>(out, w);
  }
  return 0;
}
