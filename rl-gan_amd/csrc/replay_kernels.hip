// Replay ring (HBM-resident), index sampling on the legacy MT19937 stream, and the Philox4x32-10
// counter generator used for order-independent exploration noise.  gfx950 only.
#include "rlgan_device.h"
#include "rlgan_host.h"

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

// ---------------------------------------------------------------- MT19937
constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_THREADS = 640;   // 10 waves: one thread per state word

__global__ void mt_seed_kernel(uint32_t* state, uint32_t seed) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t x = seed;
  state[0] = x;
  for (int i = 1; i < MT_N; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; state[i] = x; }
  state[MT_N] = MT_N;   // position: a twist is due before the first output
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}

// RandomState.choice(n_pop, size=n): masked rejection over successive 32-bit outputs, kept in
// stream order (Curious_GAN.py:168).  One workgroup; the generator state is regenerated 624 words
// at a time in three dependency phases, candidates are accepted with a workgroup prefix count.
__global__ __launch_bounds__(MT_THREADS) void mt_choice_kernel(uint32_t* state, uint32_t rng, uint32_t mask,
                                                               int n, int64_t* out) {
  __shared__ uint32_t mt[MT_N];
  __shared__ int wave_cnt[MT_THREADS / 64];
  __shared__ int s_pos, s_count, s_newpos;
  const int tid = threadIdx.x;
  if (tid < MT_N) mt[tid] = state[tid];
  if (tid == 0) { s_pos = (int)state[MT_N]; s_count = 0; s_newpos = -1; }
  __syncthreads();
  if (rng == 0) {   // population of one: zeros, no draws consumed
    for (int i = tid; i < n; i += MT_THREADS) out[i] = 0;
    return;
  }
  while (true) {
    int pos = s_pos, count = s_count;
    if (count >= n) break;
    if (pos >= MT_N) {
      // new[i] = new/old[(i+397)%624] ^ mix(old[i], old[i+1]); phase bounds follow the data dependences
      uint32_t v = 0;
      if (tid < 227) v = mt[tid + MT_M] ^ mt_mix(mt[tid], mt[tid + 1]);
      __syncthreads();
      if (tid < 227) mt[tid] = v;
      __syncthreads();
      if (tid >= 227 && tid < 454) v = mt[tid - 227] ^ mt_mix(mt[tid], mt[tid + 1]);
      __syncthreads();
      if (tid >= 227 && tid < 454) mt[tid] = v;
      __syncthreads();
      if (tid >= 454 && tid < 623) v = mt[tid - 227] ^ mt_mix(mt[tid], mt[tid + 1]);
      if (tid == 623) v = mt[396] ^ mt_mix(mt[623], mt[0]);
      __syncthreads();
      if (tid >= 454 && tid < MT_N) mt[tid] = v;
      __syncthreads();
      pos = 0;
    }
    const int j = pos + tid;
    uint32_t val = 0;
    bool acc = false;
    if (j < MT_N) { val = mt_temper(mt[j]) & mask; acc = val <= rng; }
    const unsigned long long bal = __ballot(acc);
    const int lane = tid & 63, wave = tid >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < MT_THREADS / 64; ++w) { const int cw = wave_cnt[w]; if (w < wave) base += cw; total += cw; }
    const int rank = count + base + before;
    if (acc && rank < n) out[rank] = (int64_t)val;
    if (acc && rank == n - 1) s_newpos = j + 1;     // stream position right after the last draw used
    __syncthreads();
    if (tid == 0) {
      if (count + total >= n) { s_count = n; s_pos = s_newpos; }
      else { s_count = count + total; s_pos = MT_N; }
    }
    __syncthreads();
  }
  if (tid < MT_N) state[tid] = mt[tid];
  if (tid == 0) state[MT_N] = (uint32_t)s_pos;
}

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void philox_normal_kernel(float* out, int n_pairs, unsigned long long seed, uint32_t stream_id,
                                     unsigned long long index0, const long long* step, long long step_mul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  if (step) index0 += (unsigned long long)(step[0] * step_mul);
  const unsigned long long idx = index0 + (unsigned long long)i;
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream_id, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float u0 = ((float)c[0] + 0.5f) * 2.3283064365386963e-10f;
  const float u1 = ((float)c[1] + 0.5f) * 2.3283064365386963e-10f;
  u0 = fminf(u0, 0.99999994f);
  const float rad = sqrtf(-2.f * logf(u0));
  const float ang = 6.2831853071795865f * u1;
  out[2 * (size_t)i] = rad * cosf(ang);
  out[2 * (size_t)i + 1] = rad * sinf(ang);
}

__global__ void philox_uniform_kernel(double* out, int n_pairs, unsigned long long seed, uint32_t stream_id,
                                      unsigned long long index0, const long long* step, long long step_mul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  if (step) index0 += (unsigned long long)(step[0] * step_mul);
  const unsigned long long idx = index0 + (unsigned long long)i;
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream_id, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  out[2 * (size_t)i] = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) / 9007199254740992.0;
  out[2 * (size_t)i + 1] = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6)) / 9007199254740992.0;
}

inline int launched() { return (int)hipGetLastError(); }

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

}  // extern "C"
