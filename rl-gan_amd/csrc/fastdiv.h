// n / d for 0 <= n < 2^31 as a multiply-high, an add and a shift: index arithmetic with run-time divisors
// would otherwise cost these kernels more vector-ALU time than their real work.
//   m = floor(2^32 (2^l - d) / d) + 1,  l = ceil(log2 d),  q = (mulhi(m, n) + n) >> l.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace rlgan {

struct FastDiv { uint32_t m; int l; };

inline FastDiv make_div(int d) {
  int l = 0;
  while ((1u << l) < (uint32_t)d) ++l;
  return FastDiv{(uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - (uint64_t)d)) / (uint64_t)d + 1), l};
}

__device__ __forceinline__ int fdiv(int n, FastDiv f) { return (int)((__umulhi(f.m, (uint32_t)n) + (uint32_t)n) >> f.l); }

}  // namespace rlgan
