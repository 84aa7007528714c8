// Random streams on the device: Philox4x32-10 (counter based, for exploration noise and goal
// placement of the vectorised loop) and the legacy MT19937 index draw of np.random.choice.
#pragma once
#include "rlgan_device.h"

namespace rlgan {
#ifdef __HIPCC__

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

// element `idx` of stream `stream_id` under key `seed`: two standard normals (Box-Muller in fp32)
__device__ __forceinline__ void philox_normal_pair(unsigned long long idx, uint32_t stream_id, unsigned long long seed,
                                                   float& z0, float& z1) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream_id, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float u0 = ((float)c[0] + 0.5f) * 2.3283064365386963e-10f;
  const float u1 = ((float)c[1] + 0.5f) * 2.3283064365386963e-10f;
  u0 = fminf(u0, 0.99999994f);
  const float rad = sqrtf(-2.f * logf(u0));
  const float ang = 6.2831853071795865f * u1;
  z0 = rad * cosf(ang);
  z1 = rad * sinf(ang);
}

// two uniform doubles in [0, 1) with 53 random bits each (the construction of random_sample)
__device__ __forceinline__ void philox_uniform_pair(unsigned long long idx, uint32_t stream_id, unsigned long long seed,
                                                    double& u0, double& u1) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), stream_id, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  u0 = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) / 9007199254740992.0;
  u1 = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6)) / 9007199254740992.0;
}

// ---------------------------------------------------------------- MT19937
constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_LDS_WORDS = MT_N + 16;     // key + per-wave counts + three control words

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}

// RandomState.choice(n_pop, size=n) (Curious_GAN.py:168): masked rejection over successive 32-bit
// outputs, kept in stream order.  Runs in ONE workgroup of NTH threads (256 <= NTH, multiple of 64);
// `lds` provides MT_LDS_WORDS words.  The generator key is regenerated 624 words at a time in three
// dependency phases (each at most 227 words wide, so one pass of the workgroup per phase); candidates
// are accepted NTH at a time with a workgroup prefix count.  rng = n_pop-1, mask = next power of two - 1.
template <int NTH>
__device__ __forceinline__ void mt_choice_body(uint32_t* state, uint32_t rng, uint32_t mask, int n, long long* out, uint32_t* lds) {
  static_assert(NTH >= 256 && NTH % 64 == 0, "one pass per twist phase needs at least 227 threads");
  uint32_t* mt = lds;
  int* wave_cnt = (int*)(lds + MT_N);               // up to 10 waves
  int* ctl = (int*)(lds + MT_N + 12);               // pos, count, newpos
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += NTH) mt[i] = state[i];
  if (tid == 0) { ctl[0] = (int)state[MT_N]; ctl[1] = 0; ctl[2] = -1; }
  lds_barrier();
  if (rng == 0) {   // population of one: zeros, no draws consumed
    for (int i = tid; i < n; i += NTH) out[i] = 0;
    return;
  }
  while (true) {
    int pos = ctl[0], count = ctl[1];
    if (count >= n) break;
    if (pos >= MT_N) {
      // new[i] = new/old[(i+397)%624] ^ mix(old[i], old[i+1]); phase bounds follow the data dependences
      uint32_t v = 0;
      if (tid < 227) v = mt[tid + MT_M] ^ mt_mix(mt[tid], mt[tid + 1]);
      lds_barrier();
      if (tid < 227) mt[tid] = v;
      lds_barrier();
      if (tid < 227) v = mt[tid] ^ mt_mix(mt[tid + 227], mt[tid + 228]);
      lds_barrier();
      if (tid < 227) mt[tid + 227] = v;
      lds_barrier();
      if (tid < 169) v = mt[tid + 227] ^ mt_mix(mt[tid + 454], mt[tid + 455]);
      if (tid == 169) v = mt[396] ^ mt_mix(mt[623], mt[0]);
      lds_barrier();
      if (tid < 170) mt[tid + 454] = v;
      lds_barrier();
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
    lds_barrier();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NTH / 64; ++w) { const int cw = wave_cnt[w]; if (w < wave) base += cw; total += cw; }
    const int rank = count + base + before;
    if (acc && rank < n) out[rank] = (long long)val;
    if (acc && rank == n - 1) ctl[2] = j + 1;       // stream position right after the last draw used
    lds_barrier();
    if (tid == 0) {
      if (count + total >= n) { ctl[1] = n; ctl[0] = ctl[2]; }
      else { ctl[1] = count + total; ctl[0] = min(pos + NTH, MT_N); }
    }
    lds_barrier();
  }
  for (int i = tid; i < MT_N; i += NTH) state[i] = mt[i];
  if (tid == 0) state[MT_N] = (uint32_t)ctl[0];
}

#endif  // __HIPCC__
}  // namespace rlgan
