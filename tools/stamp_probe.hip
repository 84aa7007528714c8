// Diagnostic only: where a 16-row tile spends its cycles inside one network pass.  Runs the library's
// own device code (rlgan_device.h) with cycle stamps compiled in, on 256 workgroups so the chip is
// loaded as in a real step, and prints the stamps of one workgroup.  Never part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRLGAN_STAMP=17 -Irl-gan_amd/csrc tools/stamp_probe.hip -o /tmp/stamp_probe
#include "rlgan_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rlgan;

struct Args { Geo g; const float* wf; const float* wb; const float* s; int n_rows; float* out; float* acts[8]; unsigned long long* stamps; };

__global__ __launch_bounds__(THREADS) void probe_kernel(const Args A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + X0_FLOATS + 2 * WIDE_FLOATS + T3_FLOATS + O_FLOATS];
  if (threadIdx.x < 4) rl_nst[threadIdx.x] = 0;
  __syncthreads();
  Tiles t;
  float* b = smem;
  t.part = b; b += PART_FLOATS; t.x0 = b; b += X0_FLOATS; t.h1 = b; b += WIDE_FLOATS; t.h2 = b; b += WIDE_FLOATS;
  t.h3 = b; b += T3_FLOATS; t.out = b;
  const int row0 = blockIdx.x * ROWS;
  FwdPre<NET_ACTOR> p1; p1.issue(A.wf + A.g.offf[0]);
  STAMP(100);
  load_x0(t.x0, row0, A.n_rows, MatIn{A.s, 7, 7, 0}, MatIn{nullptr, 0, 0, 0}, MatIn{nullptr, 0, 0, 0}, nullptr);
  __syncthreads();
  STAMP(101);
  const ActsG ws = acts_at((const float* const*)A.acts, row0, A.g.ld3);
  PreAll<1> pb;
  net_forward<NET_ACTOR, true>(A.wf, A.g, t, ws, p1, [&] { pb.issue(A.wb + A.g.offb[3]); });
  STAMP(102);
  {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    t.out[r * LD_O + c] = c < 2 ? 0.01f * t.out[r * LD_O + c] : 0.f;
  }
  __syncthreads();
  net_backward<NET_ACTOR, false, true>(A.wb, A.g, t, ws, pb, NoNext{});
  STAMP(103);
  __syncthreads();
  if (blockIdx.x == RLGAN_STAMP) {
    for (int i = threadIdx.x; i < 4 * 192; i += THREADS) A.stamps[i] = (i % 192) < rl_nst[i / 192] ? rl_stamps[i / 192][i % 192] : 0ull;
  }
}

int main() {  // PROBE_B: minibatch rows (4096 = one tile per compute unit, 8192 = two)
  Geo g; make_geo(NET_ACTOR, 7, 2, g);
  const int B = getenv("PROBE_B") ? atoi(getenv("PROBE_B")) : 4096;
  std::vector<float> wf(g.wf_floats), wb(g.wb_floats), s(B * 7);
  for (auto& v : wf) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : wb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : s) v = rand() / (float)RAND_MAX - 0.5f;
  Args A; A.g = g; A.n_rows = B;
  float *dwf, *dwb, *ds; unsigned long long* dst;
  hipMalloc(&dwf, wf.size() * 4); hipMalloc(&dwb, wb.size() * 4); hipMalloc(&ds, s.size() * 4); hipMalloc(&dst, 4 * 192 * 8);
  hipMemcpy(dwf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dwb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, s.data(), s.size() * 4, hipMemcpyHostToDevice);
  const int lds[8] = {32, 208, 208, 16, 208, 208, 16, 16};
  for (int i = 0; i < 8; ++i) hipMalloc(&A.acts[i], (size_t)B * lds[i] * 4);
  A.wf = dwf; A.wb = dwb; A.s = ds; A.out = nullptr; A.stamps = dst;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(probe_kernel, dim3(B / 16), dim3(THREADS), 0, 0, A);
  hipEventRecord(e0);
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(probe_kernel, dim3(B / 16), dim3(THREADS), 0, 0, A);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("kernel avg %.2f us (%s)\n", ms / 20 * 1e3, hipGetErrorString(hipGetLastError()));
  std::vector<unsigned long long> st(4 * 192);
  hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
  for (int w = 0; w < 4; ++w) {
    printf("wave %d:", w);
    unsigned long long t0 = st[w * 192] & 0xFFFFFFFFFFFFull, prev = t0;
    for (int k = 0; k < 192 && st[w * 192 + k]; ++k) {
      unsigned long long t = st[w * 192 + k] & 0xFFFFFFFFFFFFull; int tag = (int)(st[w * 192 + k] >> 48);
      printf(" [%d]+%llu", tag, t - prev); prev = t;
    }
    printf("  total %llu\n", prev - t0);
  }
  return 0;
}
