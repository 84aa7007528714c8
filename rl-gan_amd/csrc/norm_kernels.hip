// Batch normalisation and the softmax cross-entropy of the pixel variant's inverse model
// (`StateActionPredictor`, Conv_model.py:787-881): tf.layers.batch_normalization over the feature axis with its
// defaults (epsilon 1e-3, gamma/beta trained) after every dense layer, relu6 after the first two, and
// tf.nn.sparse_softmax_cross_entropy_with_logits against argmax(a).  The reference never runs the moving-average
// update ops (no control dependency on UPDATE_OPS anywhere), so inference normalises with the initial moving
// statistics (mean 0, variance 1): `training == 0` below.  All column sums run in a fixed order.  gfx950 only.
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

constexpr int BN_COLS = 64, BN_PH = 4;      // a workgroup owns 64 columns; four row phases per column

__device__ __forceinline__ float relu6_d(float y) { return y > 0.f && y < 6.f ? 1.f : 0.f; }

// sum over the four row phases of a column, in phase order; every thread of the column gets it
__device__ __forceinline__ float phase_sum(float* sh, float v) {
  const int c = threadIdx.x & (BN_COLS - 1), ph = threadIdx.x / BN_COLS;
  __syncthreads();
  sh[ph * BN_COLS + c] = v;
  __syncthreads();
  return ((sh[c] + sh[BN_COLS + c]) + sh[2 * BN_COLS + c]) + sh[3 * BN_COLS + c];
}

// y = act(gamma * (z - mean) * inv_std + beta) per column; training: mean / biased variance of the m rows
__global__ __launch_bounds__(256) void bn_forward_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int m, int n, float eps, int training, int act,
                                                         float* __restrict__ y, int ldy, float* __restrict__ save_mean,
                                                         float* __restrict__ save_inv_std) {
  __shared__ float sh[256];
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1)), ph = threadIdx.x / BN_COLS;
  const bool in = c < n;
  float mean = 0.f, inv_std = rsqrtf(1.f + eps);
  if (training) {
    float s = 0.f;
    if (in) for (int r = ph; r < m; r += BN_PH) s += z[(size_t)r * ldz + c];
    mean = phase_sum(sh, s) / (float)m;
    float q = 0.f;
    if (in) for (int r = ph; r < m; r += BN_PH) { const float d = z[(size_t)r * ldz + c] - mean; q += d * d; }
    inv_std = 1.f / sqrtf(phase_sum(sh, q) / (float)m + eps);
  } else {
    inv_std = 1.f / sqrtf(1.f + eps);
  }
  if (!in) return;
  if (ph == 0 && save_mean) { save_mean[c] = mean; save_inv_std[c] = inv_std; }
  const float ga = gamma[c], be = beta[c];
  for (int r = ph; r < m; r += BN_PH) {
    float v = ga * ((z[(size_t)r * ldz + c] - mean) * inv_std) + be;
    if (act == 3) v = __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
    y[(size_t)r * ldy + c] = v;
  }
}

// training-mode gradient: dyh = dy * relu6'(y) (act == 3); xh = (z - mean) * inv_std;
// dgamma = sum dyh * xh, dbeta = sum dyh, dz = gamma * inv_std * (dyh - dbeta / m - xh * dgamma / m)
__global__ __launch_bounds__(256) void bn_backward_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                          const float* __restrict__ z, int ldz, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_, const float* __restrict__ inv_std_, int m, int n,
                                                          int act, float* __restrict__ dz, int lddz, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  __shared__ float sh[256];
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1)), ph = threadIdx.x / BN_COLS;
  const bool in = c < n;
  const float mean = in ? mean_[c] : 0.f, inv_std = in ? inv_std_[c] : 0.f, ga = in ? gamma[c] : 0.f;
  float sg = 0.f, sb = 0.f;
  if (in) {
    for (int r = ph; r < m; r += BN_PH) {
      float d = dy[(size_t)r * lddy + c];
      if (act == 3) d *= relu6_d(y[(size_t)r * ldy + c]);
      sg += d * ((z[(size_t)r * ldz + c] - mean) * inv_std);
      sb += d;
    }
  }
  const float dg = phase_sum(sh, sg), db = phase_sum(sh, sb);
  if (!in) return;
  if (ph == 0) { dgamma[c] = dg; dbeta[c] = db; }
  const float k = ga * inv_std, im = 1.f / (float)m;
  for (int r = ph; r < m; r += BN_PH) {
    float d = dy[(size_t)r * lddy + c];
    if (act == 3) d *= relu6_d(y[(size_t)r * ldy + c]);
    const float xh = (z[(size_t)r * ldz + c] - mean) * inv_std;
    dz[(size_t)r * lddz + c] = k * (d - db * im - xh * dg * im);
  }
}

// ---- the same two passes for many rows: the rows are cut into chunks, a (column block, chunk) pair per workgroup ----
constexpr int BN_ONE_LAUNCH_ROWS = 256, BN_MAX_CHUNKS = 128;

__host__ __device__ inline int bn_chunk_rows(int m) { const int r = (m + BN_MAX_CHUNKS - 1) / BN_MAX_CHUNKS; return r < 32 ? 32 : r; }

// chunk mean and chunk sum of squared deviations from it: scratch[(2 * chunk) * n + c], scratch[(2 * chunk + 1) * n + c]
__global__ __launch_bounds__(256) void bn_chunk_stats_kernel(const float* __restrict__ z, int ldz, int m, int n, int rows,
                                                             float* __restrict__ scratch) {
  __shared__ float sh[256];
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1)), ph = threadIdx.x / BN_COLS;
  const bool in = c < n;
  const int r0 = blockIdx.y * rows, r1 = min(m, r0 + rows);
  float s = 0.f;
  if (in) for (int r = r0 + ph; r < r1; r += BN_PH) s += z[(size_t)r * ldz + c];
  const float mean = phase_sum(sh, s) / (float)(r1 - r0);
  float q = 0.f;
  if (in) for (int r = r0 + ph; r < r1; r += BN_PH) { const float d = z[(size_t)r * ldz + c] - mean; q += d * d; }
  q = phase_sum(sh, q);
  if (in && ph == 0) { scratch[(size_t)(2 * blockIdx.y) * n + c] = mean; scratch[(size_t)(2 * blockIdx.y + 1) * n + c] = q; }
}

// chunk statistics joined in chunk order (pairwise update of count, mean and sum of squared deviations)
__global__ void bn_join_stats_kernel(const float* __restrict__ scratch, int m, int n, int rows, int chunks, float eps,
                                     float* __restrict__ save_mean, float* __restrict__ save_inv_std) {
  // The two weights of a chunk's update depend on its position alone (rows before it and in it; the counts are whole
  // numbers below 2^24, which a float sum holds exactly): the lanes work them out side by side, two divisions per
  // chunk in all, instead of every column repeating them inside its chain of updates.
  __shared__ float w_mean[BN_MAX_CHUNKS], w_m2[BN_MAX_CHUNKS];
  for (int k = threadIdx.x; k < chunks; k += blockDim.x) {
    const float cnt = (float)(k * rows), nb = (float)(min(m, (k + 1) * rows) - k * rows), tot = cnt + nb;
    w_mean[k] = nb / tot;
    w_m2[k] = cnt * nb / tot;
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float mean = 0.f, m2 = 0.f;
  constexpr int FL = 32;                                 // chunks whose loads are in flight together, then the dependent updates
  for (int k0 = 0; k0 < chunks; k0 += FL) {              // (eight at a time: sixteen memory round trips for 128 chunks, 18 us)
    float mb[FL], qb[FL];
#pragma unroll
    for (int j = 0; j < FL; ++j) {
      const int k = min(k0 + j, chunks - 1);
      mb[j] = scratch[(size_t)(2 * k) * n + c]; qb[j] = scratch[(size_t)(2 * k + 1) * n + c];
    }
#pragma unroll
    for (int j = 0; j < FL; ++j) {
      const int k = k0 + j;
      if (k < chunks) {
        const float delta = mb[j] - mean;
        mean += delta * w_mean[k];
        m2 += qb[j] + delta * delta * w_m2[k];
      }
    }
  }
  save_mean[c] = mean;
  save_inv_std[c] = 1.f / sqrtf(m2 / (float)m + eps);
}

// the moving statistics as the reference leaves them: mean 0, variance 1
__global__ void bn_const_stats_kernel(int n, float eps, float* __restrict__ save_mean, float* __restrict__ save_inv_std) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) { save_mean[c] = 0.f; save_inv_std[c] = 1.f / sqrtf(1.f + eps); }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean_,
                                                       const float* __restrict__ inv_std_, int m, int n, int rows, int act,
                                                       float* __restrict__ y, int ldy) {
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1)), ph = threadIdx.x / BN_COLS;
  if (c >= n) return;
  const int r0 = blockIdx.y * rows, r1 = min(m, r0 + rows);
  const float mean = mean_[c], inv_std = inv_std_[c], ga = gamma[c], be = beta[c];
  for (int r = r0 + ph; r < r1; r += BN_PH) {
    float v = ga * ((z[(size_t)r * ldz + c] - mean) * inv_std) + be;
    if (act == 3) v = __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
    y[(size_t)r * ldy + c] = v;
  }
}

// per chunk: sum of dyh * xh and of dyh
__global__ __launch_bounds__(256) void bn_bwd_chunk_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ z, int ldz, const float* __restrict__ mean_,
                                                           const float* __restrict__ inv_std_, int m, int n, int rows, int act,
                                                           float* __restrict__ scratch) {
  __shared__ float sh[256];
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1)), ph = threadIdx.x / BN_COLS;
  const bool in = c < n;
  const int r0 = blockIdx.y * rows, r1 = min(m, r0 + rows);
  const float mean = in ? mean_[c] : 0.f, inv_std = in ? inv_std_[c] : 0.f;
  float sg = 0.f, sb = 0.f;
  if (in) {
    for (int r = r0 + ph; r < r1; r += BN_PH) {
      float d = dy[(size_t)r * lddy + c];
      if (act == 3) d *= relu6_d(y[(size_t)r * ldy + c]);
      sg += d * ((z[(size_t)r * ldz + c] - mean) * inv_std);
      sb += d;
    }
  }
  const float dg = phase_sum(sh, sg), db = phase_sum(sh, sb);
  if (in && ph == 0) { scratch[(size_t)(2 * blockIdx.y) * n + c] = dg; scratch[(size_t)(2 * blockIdx.y + 1) * n + c] = db; }
}

__global__ void bn_bwd_join_kernel(const float* __restrict__ scratch, int n, int chunks, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float dg = 0.f, db = 0.f;
  constexpr int FL = 32;                                 // chunks whose loads are in flight together, added in chunk order
  for (int k0 = 0; k0 < chunks; k0 += FL) {
    float a[FL], b[FL];
#pragma unroll
    for (int j = 0; j < FL; ++j) {
      const int k = min(k0 + j, chunks - 1);
      a[j] = scratch[(size_t)(2 * k) * n + c]; b[j] = scratch[(size_t)(2 * k + 1) * n + c];
    }
#pragma unroll
    for (int j = 0; j < FL; ++j) if (k0 + j < chunks) { dg += a[j]; db += b[j]; }
  }
  dgamma[c] = dg; dbeta[c] = db;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ z, int ldz, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_, const float* __restrict__ inv_std_,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta, int m, int n,
                                                           int rows, int act, float* __restrict__ dz, int lddz) {
  const int c = blockIdx.x * BN_COLS + (threadIdx.x & (BN_COLS - 1)), ph = threadIdx.x / BN_COLS;
  if (c >= n) return;
  const int r0 = blockIdx.y * rows, r1 = min(m, r0 + rows);
  const float mean = mean_[c], inv_std = inv_std_[c], k = gamma[c] * inv_std, im = 1.f / (float)m;
  const float dg = dgamma[c], db = dbeta[c];
  for (int r = r0 + ph; r < r1; r += BN_PH) {
    float d = dy[(size_t)r * lddy + c];
    if (act == 3) d *= relu6_d(y[(size_t)r * ldy + c]);
    const float xh = (z[(size_t)r * ldz + c] - mean) * inv_std;
    dz[(size_t)r * lddz + c] = k * (d - db * im - xh * dg * im);
  }
}

// one thread per row: label = first index of the row maximum of a (tf.argmax), loss_rows[r] = logsumexp - logit[label],
// dlogits = (softmax - onehot) / m with the columns from n_classes to ld zeroed
__global__ void softmax_ce_kernel(const float* __restrict__ logits, int ld, int n_classes, const float* __restrict__ a, int ld_a, int m,
                                  float* __restrict__ loss_rows, float* __restrict__ dlogits) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  const float* x = logits + (size_t)r * ld;
  const float* ar = a + (size_t)r * ld_a;
  int label = 0;
  float amax = ar[0], xmax = x[0];
  for (int c = 1; c < n_classes; ++c) {
    if (ar[c] > amax) { amax = ar[c]; label = c; }
    xmax = fmaxf(xmax, x[c]);
  }
  float se = 0.f;
  for (int c = 0; c < n_classes; ++c) se += expf(x[c] - xmax);
  loss_rows[r] = logf(se) + xmax - x[label];
  if (dlogits) {
    float* d = dlogits + (size_t)r * ld;
    const float im = 1.f / (float)m;
    for (int c = 0; c < ld; ++c) d[c] = c < n_classes ? (expf(x[c] - xmax) / se - (c == label ? 1.f : 0.f)) * im : 0.f;
  }
}

// dz = dy * act'(y) from the activated output y: relu6 (3): 0 < y < 6; tanh (4): 1 - y^2
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz, size_t n, int act) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = y[i];
  dz[i] = dy[i] * (act == 3 ? relu6_d(v) : 1.f - v * v);
}

// TD error of the pixel critic: d = q[r][0] - (reward[r] + gamma * q_next[r][0]); parts[block] = sum of d^2 over the block's
// rows (wave sums by shuffles, then the four waves in order); dq[r] = (2 d / m, 0, 0, ...) over the ld columns
__global__ __launch_bounds__(256) void td_error_kernel(const float* __restrict__ q, const float* __restrict__ q_next, int ld,
                                                       const float* __restrict__ reward, float gamma, int m, float* __restrict__ parts,
                                                       float* __restrict__ dq) {
  __shared__ float sh[4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  float d = 0.f;
  if (r < m) {
    d = q[(size_t)r * ld] - (reward[r] + gamma * q_next[(size_t)r * ld]);
    if (dq) {
      dq[(size_t)r * ld] = 2.f * d / (float)m;
      for (int c = 1; c < ld; ++c) dq[(size_t)r * ld + c] = 0.f;
    }
  }
  float s = d * d;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// tf.contrib.rnn.BasicLSTMCell state update from the gate pre-activations [m][4u] in TensorFlow's order i, j, f, o:
// c' = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o)
__global__ void lstm_gates_kernel(const float* __restrict__ gates, const float* __restrict__ c_in, int m, int u, float forget_bias,
                                  float* __restrict__ c_out, float* __restrict__ h_out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)m * u) return;
  const size_t r = t / u, k = t - r * u;
  const float* g = gates + r * 4 * (size_t)u;
  const float i = g[k], j = g[u + k], f = g[2 * u + k], o = g[3 * u + k];
  const float sig_i = 1.f / (1.f + expf(-i)), sig_f = 1.f / (1.f + expf(-(f + forget_bias))), sig_o = 1.f / (1.f + expf(-o));
  const float c = c_in[t] * sig_f + sig_i * tanhf(j);
  c_out[t] = c;
  h_out[t] = tanhf(c) * sig_o;
}

inline int launched() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int rlgan_batchnorm_chunks(int m) { return m <= BN_ONE_LAUNCH_ROWS ? 1 : cdiv(m, bn_chunk_rows(m)); }

int rlgan_batchnorm_forward(const float* z, int ldz, const float* gamma, const float* beta, int m, int n, float eps, int training,
                            int act, float* y, int ldy, float* save_mean, float* save_inv_std, float* scratch, void* stream) {
  if (!z || !gamma || !beta || !y || m <= 0 || n <= 0 || (act != 0 && act != 3) || (training && (!save_mean || !save_inv_std)))
    return RLGAN_E_ARG;
  if (ldz < n || ldy < n) return RLGAN_E_ALIGN;
  const int chunks = rlgan_batchnorm_chunks(m);
  hipStream_t st = (hipStream_t)stream;
  if (chunks == 1 || !training) {
    if (chunks > 1) {                                            // inference over many rows: constant statistics, rows in chunks
      if (!save_mean || !save_inv_std) return RLGAN_E_ARG;
      hipLaunchKernelGGL(bn_const_stats_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, n, eps, save_mean, save_inv_std);
      const int rows = bn_chunk_rows(m);
      hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(n, BN_COLS), chunks), dim3(256), 0, st, z, ldz, gamma, beta, save_mean, save_inv_std, m,
                         n, rows, act, y, ldy);
      return launched();
    }
    hipLaunchKernelGGL(bn_forward_kernel, dim3(cdiv(n, BN_COLS)), dim3(256), 0, st, z, ldz, gamma, beta, m, n, eps, training, act, y, ldy,
                       save_mean, save_inv_std);
    return launched();
  }
  if (!scratch) return RLGAN_E_ARG;
  const int rows = bn_chunk_rows(m);
  hipLaunchKernelGGL(bn_chunk_stats_kernel, dim3(cdiv(n, BN_COLS), chunks), dim3(256), 0, st, z, ldz, m, n, rows, scratch);
  hipLaunchKernelGGL(bn_join_stats_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, scratch, m, n, rows, chunks, eps, save_mean, save_inv_std);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(n, BN_COLS), chunks), dim3(256), 0, st, z, ldz, gamma, beta, save_mean, save_inv_std, m, n,
                     rows, act, y, ldy);
  return launched();
}

int rlgan_batchnorm_backward(const float* dy, int lddy, const float* y, int ldy, const float* z, int ldz, const float* gamma,
                             const float* save_mean, const float* save_inv_std, int m, int n, int act, float* dz, int lddz,
                             float* dgamma, float* dbeta, float* scratch, void* stream) {
  if (!dy || !z || !gamma || !save_mean || !save_inv_std || !dz || !dgamma || !dbeta || m <= 0 || n <= 0 || (act != 0 && act != 3) ||
      (act == 3 && !y))
    return RLGAN_E_ARG;
  if (lddy < n || ldz < n || lddz < n || (y && ldy < n)) return RLGAN_E_ALIGN;
  const int chunks = rlgan_batchnorm_chunks(m);
  hipStream_t st = (hipStream_t)stream;
  if (chunks == 1) {
    hipLaunchKernelGGL(bn_backward_kernel, dim3(cdiv(n, BN_COLS)), dim3(256), 0, st, dy, lddy, y, ldy, z, ldz, gamma, save_mean,
                       save_inv_std, m, n, act, dz, lddz, dgamma, dbeta);
    return launched();
  }
  if (!scratch) return RLGAN_E_ARG;
  const int rows = bn_chunk_rows(m);
  hipLaunchKernelGGL(bn_bwd_chunk_kernel, dim3(cdiv(n, BN_COLS), chunks), dim3(256), 0, st, dy, lddy, y, ldy, z, ldz, save_mean,
                     save_inv_std, m, n, rows, act, scratch);
  hipLaunchKernelGGL(bn_bwd_join_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, scratch, n, chunks, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(n, BN_COLS), chunks), dim3(256), 0, st, dy, lddy, y, ldy, z, ldz, gamma, save_mean,
                     save_inv_std, dgamma, dbeta, m, n, rows, act, dz, lddz);
  return launched();
}

int rlgan_softmax_ce(const float* logits, int ld, int n_classes, const float* a, int ld_a, int m, float* loss_rows, float* dlogits,
                     void* stream) {
  if (!logits || !a || !loss_rows || m <= 0 || n_classes <= 0) return RLGAN_E_ARG;
  if (ld < n_classes || ld_a < n_classes) return RLGAN_E_ALIGN;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, logits, ld, n_classes, a, ld_a, m,
                     loss_rows, dlogits);
  return launched();
}

int rlgan_act_backward(const float* dy, const float* y, float* dz, int64_t n, int act, void* stream) {
  if (!dy || !y || !dz || n <= 0 || (act != 3 && act != 4)) return RLGAN_E_ARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, (size_t)n, act);
  return launched();
}

int rlgan_td_error(const float* q, const float* q_next, int ld, const float* reward, float gamma, int m, float* parts, float* dq,
                   void* stream) {
  if (!q || !q_next || !reward || !parts || m <= 0 || ld <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(td_error_kernel, dim3(cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, q, q_next, ld, reward, gamma, m, parts, dq);
  return launched();
}

int rlgan_lstm_gates(const float* gates, const float* c_in, int m, int units, float forget_bias, float* c_out, float* h_out,
                     void* stream) {
  if (!gates || !c_in || !c_out || !h_out || m <= 0 || units <= 0) return RLGAN_E_ARG;
  const size_t n = (size_t)m * units;
  hipLaunchKernelGGL(lstm_gates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates, c_in, m, units,
                     forget_bias, c_out, h_out);
  return launched();
}

}  // extern "C"
