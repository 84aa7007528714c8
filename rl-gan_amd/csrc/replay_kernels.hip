// Replay ring (HBM-resident), index sampling on the legacy MT19937 stream, and the Philox4x32-10
// counter generator used for order-independent exploration noise.  gfx950 only.
#include "rlgan_device.h"
#include "rlgan_host.h"
#include "rlgan_rng.h"
#include <algorithm>
#include <cstdint>

using namespace rlgan;

namespace {

// ---------------------------------------------------------------- ring store / gather
// Memory.store_transition (Curious_GAN.py:161-165, DDPG.py:187-191) for n rows
__global__ void replay_store_kernel(float* __restrict__ data, int capacity, int dims, int ld,
                                    long long pointer, const float* __restrict__ rows, int ld_rows, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = e >> 5, c = e & 31;
  if (i >= n) return;
  const long long slot = (pointer + i) % capacity;
  for (int cc = c; cc < dims; cc += 32) data[(size_t)slot * ld + cc] = rows[(size_t)i * ld_rows + cc];
}

// data[indices, :] (Curious_GAN.py:169): half a wave per row, so a 128-byte-padded row is one
// aligned 128-byte segment per half-wave
__global__ void replay_gather_kernel(const float* __restrict__ data, int capacity, int dims, int ld,
                                     const int64_t* __restrict__ idx, int n, float* __restrict__ out, int ld_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = e >> 5, c = e & 31;
  if (i >= n) return;
  const long long j = idx[i];
  const bool ok = j >= 0 && j < capacity;
  for (int cc = c; cc < dims; cc += 32)
    out[(size_t)i * ld_out + cc] = ok ? data[(size_t)j * ld + cc] : __builtin_nanf("");
}

// ---------------------------------------------------------------- MT19937 / Philox launches
constexpr int MT_THREADS = 640;   // 10 waves: one thread per generator word

__global__ void mt_seed_kernel(uint32_t* state, uint32_t seed) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t x = seed;
  state[0] = x;
  for (int i = 1; i < MT_N; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; state[i] = x; }
  state[MT_N] = MT_N;   // position: a twist is due before the first output
}

__global__ __launch_bounds__(MT_THREADS) void mt_choice_kernel(uint32_t* state, uint32_t rng, uint32_t mask,
                                                               int n, int64_t* out) {
  __shared__ uint32_t lds[MT_LDS_WORDS];
  mt_choice_body<MT_THREADS>(state, rng, mask, n, (long long*)out, lds);
}

__global__ void philox_normal_kernel(float* out, int n_pairs, unsigned long long seed, uint32_t stream_id,
                                     unsigned long long index0, const long long* step, long long step_mul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  if (step) index0 += (unsigned long long)(step[0] * step_mul);
  float z0, z1;
  philox_normal_pair(index0 + (unsigned long long)i, stream_id, seed, z0, z1);
  out[2 * (size_t)i] = z0;
  out[2 * (size_t)i + 1] = z1;
}

__global__ void philox_uniform_kernel(double* out, int n_pairs, unsigned long long seed, uint32_t stream_id,
                                      unsigned long long index0, const long long* step, long long step_mul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  if (step) index0 += (unsigned long long)(step[0] * step_mul);
  double u0, u1;
  philox_uniform_pair(index0 + (unsigned long long)i, stream_id, seed, u0, u1);
  out[2 * (size_t)i] = u0;
  out[2 * (size_t)i + 1] = u1;
}

inline int launched() { return (int)hipGetLastError(); }

// ---------------------------------------------------------------- uint8 frame replay (pixel variants)
// The reference keeps Python lists of float32 frame stacks (Conv_model.py:102-157) that the environment
// wrapper produced as uint8 * (1/255) (env_wrapper.py:44,58); here the ring holds the uint8 stacks and the
// gather applies the scale, bit for bit the same float32 product.  Both kernels move 4 pixels per thread
// and step (one 32-bit load, one 128-bit store in the gather): pure HBM streaming.
__global__ __launch_bounds__(256) void frames_store_kernel(uint32_t* __restrict__ ring, long long slots, int words,
                                                           const uint32_t* __restrict__ frames, const int64_t* __restrict__ slot_idx) {
  const long long slot = slot_idx[blockIdx.y];
  if (slot < 0 || slot >= slots) return;
  const uint32_t* src = frames + (size_t)blockIdx.y * words;
  uint32_t* dst = ring + (size_t)slot * words;
  for (int w0 = blockIdx.x * 1024 + threadIdx.x; w0 < words; w0 += gridDim.x * 1024) {
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = w0 + j * 256 < words ? src[w0 + j * 256] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (w0 + j * 256 < words) __builtin_nontemporal_store(v[j], &dst[w0 + j * 256]);   // written once, read much later
  }
}

__global__ __launch_bounds__(256) void frames_gather_kernel(const uint32_t* __restrict__ ring, long long slots, int words,
                                                            const int64_t* __restrict__ idx, float scale, f32x4* __restrict__ out) {
  const long long slot = idx[blockIdx.y];
  const bool ok = slot >= 0 && slot < slots;
  const uint32_t* src = ring + (size_t)(ok ? slot : 0) * words;
  f32x4* dst = out + (size_t)blockIdx.y * words;
  // four independent loads in flight per thread before the first store
  for (int w0 = blockIdx.x * 1024 + threadIdx.x; w0 < words; w0 += gridDim.x * 1024) {
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (ok && w0 + j * 256 < words) ? src[w0 + j * 256] : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (w0 + j * 256 < words)
        __builtin_nontemporal_store(f32x4{(float)(v[j] & 255u) * scale, (float)((v[j] >> 8) & 255u) * scale,
                                          (float)((v[j] >> 16) & 255u) * scale, (float)(v[j] >> 24) * scale}, &dst[w0 + j * 256]);
  }
}

// one channel of every pixel: out[i][p] = float(ring[idx[i]][p * channels + channel]) * scale
__global__ __launch_bounds__(256) void frames_gather_channel_kernel(const uint8_t* __restrict__ ring, long long slots, int pixels,
                                                                    int channels, int channel, const int64_t* __restrict__ idx,
                                                                    float scale, float* __restrict__ out) {
  const long long slot = idx[blockIdx.y];
  const bool ok = slot >= 0 && slot < slots;
  const uint8_t* src = ring + (size_t)(ok ? slot : 0) * pixels * channels + channel;
  float* dst = out + (size_t)blockIdx.y * pixels;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < pixels; p += gridDim.x * 256)
    dst[p] = ok ? (float)src[(size_t)p * channels] * scale : 0.f;
}

// The same for four channels and a pixel count that is a multiple of four: a thread reads four pixels (16 bytes),
// picks the channel's byte of each and writes four floats (a byte per load and a float per store: 23 us for 4096
// frames of 42x42; the workgroup is one frame, 441 threads' worth of work).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void frames_gather_channel4_kernel(const uint8_t* __restrict__ ring, long long slots, int quads,
                                                                     int channel, const int64_t* __restrict__ idx, float scale,
                                                                     float* __restrict__ out) {
  const long long slot = idx[blockIdx.x];
  const bool ok = slot >= 0 && slot < slots;
  const u32x4* src = (const u32x4*)(ring + (size_t)(ok ? slot : 0) * quads * 16);
  f32x4* dst = (f32x4*)out + (size_t)blockIdx.x * quads;
  const int sh = 8 * channel;
  constexpr int U = 2;
  for (int q0 = threadIdx.x; q0 < quads; q0 += 256 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[min(q0 + 256 * u, quads - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (q0 + 256 * u < quads) {
        const f32x4 f{(float)((v[u][0] >> sh) & 255u) * scale, (float)((v[u][1] >> sh) & 255u) * scale,
                      (float)((v[u][2] >> sh) & 255u) * scale, (float)((v[u][3] >> sh) & 255u) * scale};
        dst[q0 + 256 * u] = ok ? f : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}

}  // namespace

extern "C" {

int rlgan_replay_store(float* data, int capacity, int dims, int ld_data, int64_t pointer, const float* rows,
                       int ld_rows, int n_rows, void* stream) {
  if (!data || !rows || capacity <= 0 || dims <= 0 || n_rows <= 0 || pointer < 0) return RLGAN_E_ARG;
  if (ld_data < dims || ld_rows < dims || n_rows > capacity) return RLGAN_E_ALIGN;
  hipLaunchKernelGGL(replay_store_kernel, dim3(cdiv(n_rows * 32, 256)), dim3(256), 0, (hipStream_t)stream,
                     data, capacity, dims, ld_data, (long long)pointer, rows, ld_rows, n_rows);
  return launched();
}

int rlgan_replay_gather(const float* data, int capacity, int dims, int ld_data, const int64_t* idx, int n,
                        float* out, int ld_out, void* stream) {
  if (!data || !idx || !out || capacity <= 0 || dims <= 0 || n <= 0) return RLGAN_E_ARG;
  if (ld_data < dims || ld_out < dims) return RLGAN_E_ALIGN;
  hipLaunchKernelGGL(replay_gather_kernel, dim3(cdiv(n * 32, 256)), dim3(256), 0, (hipStream_t)stream,
                     data, capacity, dims, ld_data, idx, n, out, ld_out);
  return launched();
}

int rlgan_mt19937_seed(uint32_t* state, uint32_t seed, void* stream) {
  if (!state) return RLGAN_E_ARG;
  hipLaunchKernelGGL(mt_seed_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, seed);
  return launched();
}

int rlgan_mt19937_choice(uint32_t* state, int64_t n_pop, int n, int64_t* out_idx, void* stream) {
  if (!state || !out_idx || n <= 0 || n_pop <= 0 || n_pop > 0x100000000ll) return RLGAN_E_ARG;
  const uint32_t rng = (uint32_t)(n_pop - 1);
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  hipLaunchKernelGGL(mt_choice_kernel, dim3(1), dim3(MT_THREADS), 0, (hipStream_t)stream, state, rng, mask, n, out_idx);
  return launched();
}

int rlgan_philox_normal(float* out, int n_pairs, uint64_t seed, uint32_t stream_id, uint64_t index0,
                        const int64_t* step, int64_t step_mul, void* stream) {
  if (!out || n_pairs <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(philox_normal_kernel, dim3(cdiv(n_pairs, 256)), dim3(256), 0, (hipStream_t)stream,
                     out, n_pairs, (unsigned long long)seed, stream_id, (unsigned long long)index0,
                     (const long long*)step, (long long)step_mul);
  return launched();
}

int rlgan_philox_uniform(double* out, int n_pairs, uint64_t seed, uint32_t stream_id, uint64_t index0,
                         const int64_t* step, int64_t step_mul, void* stream) {
  if (!out || n_pairs <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3(cdiv(n_pairs, 256)), dim3(256), 0, (hipStream_t)stream,
                     out, n_pairs, (unsigned long long)seed, stream_id, (unsigned long long)index0,
                     (const long long*)step, (long long)step_mul);
  return launched();
}

int rlgan_frames_store(uint8_t* ring, int64_t slots, int32_t frame_bytes, const uint8_t* frames, int32_t n,
                       const int64_t* slot_idx, void* stream) {
  if (!ring || !frames || !slot_idx || slots <= 0 || frame_bytes <= 0 || n <= 0 || n > 65535) return RLGAN_E_ARG;
  if ((frame_bytes & 3) || ((uintptr_t)ring & 3) || ((uintptr_t)frames & 3)) return RLGAN_E_ALIGN;
  const int words = frame_bytes / 4;
  hipLaunchKernelGGL(frames_store_kernel, dim3(std::min(cdiv(words, 1024), 64), n), dim3(256), 0, (hipStream_t)stream,
                     (uint32_t*)ring, (long long)slots, words, (const uint32_t*)frames, slot_idx);
  return launched();
}

int rlgan_frames_gather(const uint8_t* ring, int64_t slots, int32_t frame_bytes, const int64_t* idx, int32_t n, float scale,
                        float* out, void* stream) {
  if (!ring || !idx || !out || slots <= 0 || frame_bytes <= 0 || n <= 0 || n > 65535) return RLGAN_E_ARG;
  if ((frame_bytes & 3) || ((uintptr_t)ring & 3) || ((uintptr_t)out & 15)) return RLGAN_E_ALIGN;
  const int words = frame_bytes / 4;
  hipLaunchKernelGGL(frames_gather_kernel, dim3(std::min(cdiv(words, 1024), 64), n), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)ring, (long long)slots, words, idx, scale, (f32x4*)out);
  return launched();
}

int rlgan_frames_gather_channel(const uint8_t* ring, int64_t slots, int32_t pixels, int32_t channels, int32_t channel,
                                const int64_t* idx, int32_t n, float scale, float* out, void* stream) {
  if (!ring || !idx || !out || slots <= 0 || pixels <= 0 || n <= 0 || n > 65535) return RLGAN_E_ARG;
  if (channels <= 0 || channel < 0 || channel >= channels) return RLGAN_E_DIMS;
  if (channels == 4 && pixels % 4 == 0 && !(((uintptr_t)ring | (uintptr_t)out) & 15)) {
    hipLaunchKernelGGL(frames_gather_channel4_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ring, (long long)slots, pixels / 4,
                       channel, idx, scale, out);
    return launched();
  }
  hipLaunchKernelGGL(frames_gather_channel_kernel, dim3(std::min(cdiv(pixels, 256), 64), n), dim3(256), 0, (hipStream_t)stream,
                     ring, (long long)slots, pixels, channels, channel, idx, scale, out);
  return launched();
}

}  // extern "C"
