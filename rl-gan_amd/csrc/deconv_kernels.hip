// Transposed convolution (NHWC, 'valid', square kernel) of the pixel generator's decoder, the reference's
// tf.layers.conv2d_transpose stack 1x1x200 -> 3x3x128 -> 8x8x64 -> 19x19x32 -> 42x42x1 (Conv_model.py:457-473).
// A layer is a matrix product with every input pixel as a row,
//   cols[(n, iy, ix)][(ky, kx, co)] = sum_ci x[n, iy, ix, ci] * W[ky, kx, co, ci]        (dense_kernels.hip),
// followed by the placement kernels of this file: output pixel (oy, ox) collects the entries whose tap lands
// on it, oy = iy * s + ky, ox = ix * s + kx (a gather in a fixed tap order: no atomics, the same bits every run),
// plus bias and ReLU.  Backward mirrors it: dcols is the same gather read the other way, the two products with
// it give dX and dW, and the bias gradient is a two-stage column sum.  gfx950 only.
#include "fastdiv.h"
#include "rlgan_device.h"
#include "rlgan_host.h"
#include <atomic>

using namespace rlgan;

// Partial sums of rlgan_column_sums_noalloc: the library's own copy of the `partial` argument of
// rlgan_column_sums at its largest (512 partial rows of 256 columns, 512 KiB).  Every GPU the library runs
// on holds one copy at its own address until the process ends; in host code the name below is only a handle
// for hipGetSymbolAddress, never an address a kernel may be given.
constexpr int COLSUM_PARTS = 512, COLSUM_MAX_COLS = 256;
static __device__ __align__(256) float scratch[COLSUM_PARTS * COLSUM_MAX_COLS];

namespace {

struct DeconvG {
  int h, w, Co, k, s, Ho, Wo, KK;     // input size, output channels, kernel, stride, output size, k*k*Co
  FastDiv dU, dCo, dk, ds;            // divisors: units per pixel (see V below), Co, k, s
};

inline bool make_deconv(const rlgan_deconv_t* d, DeconvG& g) {
  if (!d || d->h <= 0 || d->w <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->k <= 0 || d->stride <= 0) return false;
  if (d->c_in % 4 != 0 || (d->k * d->k * d->c_out) % 4 != 0) return false;          // 16-byte rows of the two matrices
  g.h = d->h; g.w = d->w; g.Co = d->c_out; g.k = d->k; g.s = d->stride;
  g.Ho = (d->h - 1) * d->stride + d->k; g.Wo = (d->w - 1) * d->stride + d->k;
  g.KK = d->k * d->k * d->c_out;
  if ((long long)g.Wo * g.Co > 0x7fffffffll || (long long)g.w * g.KK > 0x7fffffffll) return false;
  g.dCo = make_div(g.Co); g.dk = make_div(g.k); g.ds = make_div(g.s);
  return true;
}

__device__ __forceinline__ float relu_d(float y) { return y > 0.f ? 1.f : 0.f; }

// V = 4 when Co % 4 == 0 (a thread handles four consecutive channels as 16 bytes), else 1.
template <int V> struct Vec;
template <> struct Vec<4> {
  using T = f32x4;
  static __device__ __forceinline__ T zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ T relu(T v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
  static __device__ __forceinline__ T mask(T d, T y) { return f32x4{d[0] * relu_d(y[0]), d[1] * relu_d(y[1]), d[2] * relu_d(y[2]), d[3] * relu_d(y[3])}; }
};
template <> struct Vec<1> {
  using T = float;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T relu(T v) { return fmaxf(v, 0.f); }
  static __device__ __forceinline__ T mask(T d, T y) { return d * relu_d(y); }
};

// y[n, oy, ox, c..c+V) = act(bias + sum over the taps (ky, kx) = (oy % s + j*s, ox % s + i*s) that land here,
// in ascending (ky, kx) order).  grid = (units of a row / 256, Ho, n); a unit = V channels of one pixel.
template <int V>
__global__ __launch_bounds__(256) void col2im_kernel(const DeconvG g, const float* __restrict__ cols, const float* __restrict__ bias,
                                                     float* __restrict__ y, int act) {
  using T = typename Vec<V>::T;
  const int cu = g.Co / V;                                   // units per pixel
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.Wo * cu) return;
  const int ox = fdiv(t, g.dU), c = (t - ox * cu) * V;
  const int oy = blockIdx.y, n = blockIdx.z;
  const int qy = fdiv(oy, g.ds), ry = oy - qy * g.s, qx = fdiv(ox, g.ds), rx = ox - qx * g.s;
  T acc = *(const T*)(bias + c);
  for (int ky = ry, iy = qy; ky < g.k && iy >= 0; ky += g.s, --iy) {
    if (iy >= g.h) continue;
    for (int kx = rx, ix = qx; kx < g.k && ix >= 0; kx += g.s, --ix) {
      if (ix >= g.w) continue;
      acc += *(const T*)(cols + ((size_t)(n * g.h + iy) * g.w + ix) * g.KK + (ky * g.k + kx) * g.Co + c);
    }
  }
  *(T*)(y + ((size_t)(n * g.Ho + oy) * g.Wo + ox) * g.Co + c) = act ? Vec<V>::relu(acc) : acc;
}

// dcols[(n, iy, ix)][(ky, kx, c..c+V)] = dy[n, iy*s + ky, ix*s + kx, c..] (* ReLU'(y) when y is given).
// grid = (units of an input row / 256, h, n); a unit = V channels of one tap of one input pixel.
template <int V>
__global__ __launch_bounds__(256) void im2col_kernel(const DeconvG g, const float* __restrict__ dy, const float* __restrict__ yact,
                                                     float* __restrict__ dcols) {
  using T = typename Vec<V>::T;
  const int ku = g.KK / V;                                   // units per input pixel
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.w * ku) return;
  const int ix = fdiv(t, g.dU), e = (t - ix * ku) * V;       // e = (ky*k + kx)*Co + c
  const int tap = fdiv(e, g.dCo), c = e - tap * g.Co, ky = fdiv(tap, g.dk), kx = tap - ky * g.k;
  const int iy = blockIdx.y, n = blockIdx.z;
  const size_t at = ((size_t)(n * g.Ho + iy * g.s + ky) * g.Wo + ix * g.s + kx) * g.Co + c;
  T d = *(const T*)(dy + at);
  if (yact) d = Vec<V>::mask(d, *(const T*)(yact + at));
  *(T*)(dcols + ((size_t)(n * g.h + iy) * g.w + ix) * g.KK + e) = d;
}

// Column sums of dz[rows][C] (dz = dy * ReLU'(y) when y is given), C <= 256, in two stages with a fixed order:
// workgroup b adds its contiguous share of the rows (thread = (row phase, column), phases added in order),
// then one workgroup adds the partial rows in workgroup order.
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ dy, const float* __restrict__ yact, long long rows,
                                                          int C, long long rows_per_wg, float* __restrict__ part) {
  __shared__ float sh[256];
  const int phases = 256 / C, ph = threadIdx.x / C, c = threadIdx.x - ph * C;
  const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  float s = 0.f;
  if (ph < phases) {
    // eight rows' loads requested per turn, added in row order (a row per turn left the compiler's unrolling to cover the
    // memory latency: the one-channel layer's 7 M rows took 27 us for 29 MB)
    constexpr int FL = 8;
    for (long long r = r0 + ph; r < r1; r += (long long)phases * FL) {
      float d[FL], yv[FL];
#pragma unroll
      for (int u = 0; u < FL; ++u) {
        const long long rr = min(r + (long long)u * phases, r1 - 1);
        d[u] = dy[rr * C + c];
        yv[u] = yact ? yact[rr * C + c] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < FL; ++u) {
        if (r + (long long)u * phases < r1) s += yact ? d[u] * relu_d(yv[u]) : d[u];
      }
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    float tot = 0.f;
    for (int p = 0; p < phases; ++p) tot += sh[p * C + threadIdx.x];
    part[(size_t)blockIdx.x * C + threadIdx.x] = tot;
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int n_part, int C, float* __restrict__ out) {
  // thread = (phase, column): a phase adds every `phases`-th partial row in order, 32 loads in flight at a time
  // (one load per addition would be a chain of memory latencies; with eight, 512 partial rows were still 16 round
  // trips: 12.5 us), then the phases are added in order
  __shared__ float sh[256];
  const int phases = 256 / C, ph = threadIdx.x / C, c = threadIdx.x - ph * C;
  constexpr int FL = 32;
  float tot = 0.f;
  if (ph < phases) {
    for (int b0 = ph; b0 < n_part; b0 += FL * phases) {
      float v[FL];
#pragma unroll
      for (int j = 0; j < FL; ++j) {
        const int b = b0 + j * phases;
        v[j] = part[(size_t)min(b, n_part - 1) * C + c];
      }
#pragma unroll
      for (int j = 0; j < FL; ++j) tot += b0 + j * phases < n_part ? v[j] : 0.f;
    }
  }
  sh[threadIdx.x] = tot;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int p = 0; p < phases; ++p) t += sh[p * C + threadIdx.x];
    out[threadIdx.x] = t;
  }
}

// dst[i] = map[i] >= 0 ? src[map[i]] : 0
__global__ void gather_map_kernel(const float* __restrict__ src, const int32_t* __restrict__ map, int n, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = map[i] >= 0 ? src[map[i]] : 0.f;
}

__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = dy[i] * relu_d(y[i]);
}

// one thread per destination element; blockIdx.y walks the rows in steps of gridDim.y
__global__ void copy_cols_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, long long rows,
                                 int n_copy, int n_zero) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_copy + n_zero) return;
  for (long long r = blockIdx.y; r < rows; r += gridDim.y) dst[r * ld_dst + c] = c < n_copy ? src[r * ld_src + c] : 0.f;
}
// up to three column blocks side by side, then zeros: dst[r] = [s0[r][0..w0) | s1[r][0..w1) | s2[r][0..w2) | 0 x n_zero]
struct ConcatArgs { const float* s[3]; int ld[3], w[3]; float* dst; int ld_dst, n_zero; long long rows; };
__global__ void concat_cols_kernel(const ConcatArgs A) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int e0 = A.w[0], e1 = e0 + A.w[1], e2 = e1 + A.w[2];
  if (c >= e2 + A.n_zero) return;
  const int which = (c >= e0) + (c >= e1) + (c >= e2);               // 3: a zero column
  const float* src = which == 0 ? A.s[0] : which == 1 ? A.s[1] : A.s[2];
  const int ld = which == 0 ? A.ld[0] : which == 1 ? A.ld[1] : A.ld[2];
  const int cc = c - (which == 0 ? 0 : which == 1 ? e0 : e1);
  for (long long r = blockIdx.y; r < A.rows; r += gridDim.y) A.dst[r * A.ld_dst + c] = which < 3 ? src[r * ld + cc] : 0.f;
}
// narrow rows (n_copy + n_zero <= 4: the channel padding): one thread per row
__global__ void copy_cols_narrow_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, long long rows,
                                        int n_copy, int n_zero) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  for (int c = 0; c < n_copy + n_zero; ++c) dst[r * ld_dst + c] = c < n_copy ? src[r * ld_src + c] : 0.f;
}

// 2x2 blocks of a one-channel image as the four channels of a half-size image, and back:
//   blocks[i][Y][X][2*dy + dx] = image[i][2Y + dy][2X + dx]          image [n][h][w], blocks [n][h/2][w/2][4]
// one thread per block: two 8-byte accesses on the image side (lanes side by side in both rows), one 16-byte access on
// the block side
template <bool TO_BLOCKS>
__global__ __launch_bounds__(256) void blocks2x2_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n_blocks,
                                                        int hb, int wb) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= n_blocks) return;
  const int X = (int)(b % wb);
  const long long iy = b / wb;                         // image * hb + Y
  const long long row0 = (iy * 2) * (2 * wb) + 2 * X;  // image[i][2Y][2X]: images are h * w = 2hb * 2wb floats
  if (TO_BLOCKS) {
    const f2 a = *(const f2*)(src + row0), c = *(const f2*)(src + row0 + 2 * wb);
    *(f32x4*)(dst + 4 * b) = f32x4{a[0], a[1], c[0], c[1]};
  } else {
    const f32x4 v = *(const f32x4*)(src + 4 * b);
    *(f2*)(dst + row0) = f2{v[0], v[1]};
    *(f2*)(dst + row0 + 2 * wb) = f2{v[2], v[3]};
  }
}

constexpr int MSE_PER_WG = 4096;
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ fake, const float* __restrict__ real, long long n, float inv_n2,
                                                  float* __restrict__ parts, float* __restrict__ dfake) {
  __shared__ float sh[256];
  const long long i0 = (long long)blockIdx.x * MSE_PER_WG;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MSE_PER_WG / 256; ++j) {
    const long long i = i0 + j * 256 + threadIdx.x;
    if (i < n) {
      const float d = fake[i] - real[i];
      s += d * d;
      if (dfake) dfake[i] = d * inv_n2;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {                       // fixed tree: the same bits every run
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = sh[0];
}

__global__ void add_cols_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, long long rows, int n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  for (long long r = blockIdx.y; r < rows; r += gridDim.y) dst[r * ld_dst + c] += src[r * ld_src + c];
}

// tf.nn.sigmoid_cross_entropy_with_logits on column 0 of logits[n_real + n_fake][ld]: labels 1 for the first
// n_real rows and 0 for the rest; loss[0] = mean over the real rows, loss[1] = mean over the fake rows;
// dlogits[r][0] = (sigmoid(x) - label) / (rows of r's half), its other columns zero.  One workgroup, fixed order.
__global__ __launch_bounds__(256) void sigmoid_ce_kernel(const float* __restrict__ logits, int ld, int n_real, int n_fake,
                                                         float* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ float sh[2][256];
  float s[2] = {0.f, 0.f};
  // eight rows per thread and turn, their loads requested together (a row per turn was a memory round trip per turn:
  // 32 in a row for 8192 rows); the sums run over a thread's rows in the same order
  const int n = n_real + n_fake;
  constexpr int U = 8;
  for (int r0 = threadIdx.x; r0 < n; r0 += 256 * U) {
    float xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = logits[(size_t)min(r0 + 256 * u, n - 1) * ld];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 256 * u;
      if (r >= n) break;
      const float x = xv[u];
      const bool real = r < n_real;
      const float z = real ? 1.f : 0.f;
      s[real ? 0 : 1] += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
      if (dlogits) {
        const float sig = 1.f / (1.f + expf(-x));
        dlogits[(size_t)r * ld] = (sig - z) / (float)(real ? n_real : n_fake);
        for (int c = 1; c < ld; ++c) dlogits[(size_t)r * ld + c] = 0.f;
      }
    }
  }
  sh[0][threadIdx.x] = s[0]; sh[1][threadIdx.x] = s[1];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = n_real > 0 ? sh[0][0] / (float)n_real : 0.f;
    loss[1] = n_fake > 0 ? sh[1][0] / (float)n_fake : 0.f;
  }
}

__global__ void one_minus_sigmoid_kernel(const float* __restrict__ logits, int ld, int n, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = 1.f - 1.f / (1.f + expf(-logits[(size_t)r * ld]));
}

inline int launched() { return (int)hipGetLastError(); }

// device address of `scratch` per GPU: every GPU holds its own copy of the array at its own address.  Looked up once per
// device with that device current (rlgan_noalloc_init, or the first allocation-free call on the device).
constexpr int MAX_DEVICES = 64;
std::atomic<float*> scratch_ptr[MAX_DEVICES];
int scratch_on(int dev, float** out) {
  if (dev < 0 || dev >= MAX_DEVICES) return RLGAN_E_ARG;
  float* p = scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    int cur = -1;
    if (const hipError_t e = hipGetDevice(&cur)) return (int)e;
    if (cur != dev) if (const hipError_t e = hipSetDevice(dev)) return (int)e;
    const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (cur != dev) (void)hipSetDevice(cur);
    if (e) return (int)e;
    if (!p) return (int)hipErrorInvalidSymbol;
    scratch_ptr[dev].store(p, std::memory_order_release);
  }
  *out = p;
  return 0;
}
// the device a stream belongs to (the current device for the null stream)
int device_of(void* stream, int* dev) {
  if (stream) {
    hipDevice_t d;
    if (hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess) { *dev = (int)d; return 0; }
    (void)hipGetLastError();
  }
  return (int)hipGetDevice(dev);
}

int column_sums(const float* dy, const float* y_relu, int64_t rows, int32_t n_cols, float* partial, float* out, void* stream) {
  const long long per = ((long long)rows + COLSUM_PARTS - 1) / COLSUM_PARTS;
  const int parts = (int)(((long long)rows + per - 1) / per);
  hipLaunchKernelGGL(colsum_part_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, dy, y_relu, (long long)rows, n_cols, per, partial);
  if (const int rc = launched()) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, parts, n_cols, out);
  return launched();
}

}  // namespace

extern "C" {

int rlgan_deconv_geometry(const rlgan_deconv_t* d, rlgan_deconv_geom_t* out) {
  DeconvG g;
  if (!out) return RLGAN_E_ARG;
  if (!make_deconv(d, g)) return RLGAN_E_DIMS;
  out->h_out = g.Ho; out->w_out = g.Wo; out->kk = g.KK;
  out->n_params = g.KK * d->c_in + d->c_out;
  out->wmat_floats = (d->c_in + 1) * g.KK;
  out->colsum_floats = d->c_out <= 256 ? COLSUM_PARTS * d->c_out : 0;
  return 0;
}

int rlgan_deconv_build_map(const rlgan_deconv_t* d, int32_t* map_w) {
  DeconvG g;
  if (!map_w) return RLGAN_E_ARG;
  if (!make_deconv(d, g)) return RLGAN_E_DIMS;
  int p = 0;
  for (int e = 0; e < g.KK; ++e)                      // e = (ky*k + kx)*Co + co, the kernel tensor's first three axes
    for (int ci = 0; ci < d->c_in; ++ci) map_w[p++] = ci * g.KK + e;
  for (int co = 0; co < d->c_out; ++co) map_w[p++] = -1;   // the bias is added after the placement, not by the product
  return 0;
}

int rlgan_deconv_place(const rlgan_deconv_t* d, const float* cols, const float* bias, float* y, int n_images, int act_relu,
                       void* stream) {
  DeconvG g;
  if (!make_deconv(d, g)) return RLGAN_E_DIMS;
  if (!cols || !bias || !y || n_images <= 0 || n_images > 65535 || g.Ho > 65535) return RLGAN_E_ARG;
  const bool v4 = g.Co % 4 == 0;
  if (v4 && (((uintptr_t)cols | (uintptr_t)bias | (uintptr_t)y) & 15)) return RLGAN_E_ALIGN;
  if ((long long)n_images * g.Ho > 0x7fffffffll || (long long)n_images * g.h > 0x7fffffffll) return RLGAN_E_ALIGN;
  const int units = g.Wo * (g.Co / (v4 ? 4 : 1));
  g.dU = make_div(g.Co / (v4 ? 4 : 1));
  const dim3 grid(cdiv(units, 256), g.Ho, n_images), block(256);
  if (v4) hipLaunchKernelGGL(col2im_kernel<4>, grid, block, 0, (hipStream_t)stream, g, cols, bias, y, act_relu);
  else hipLaunchKernelGGL(col2im_kernel<1>, grid, block, 0, (hipStream_t)stream, g, cols, bias, y, act_relu);
  return launched();
}

int rlgan_deconv_gather(const rlgan_deconv_t* d, const float* dy, const float* y_relu, float* dcols, int n_images, void* stream) {
  DeconvG g;
  if (!make_deconv(d, g)) return RLGAN_E_DIMS;
  if (!dy || !dcols || n_images <= 0 || n_images > 65535 || g.h > 65535) return RLGAN_E_ARG;
  const bool v4 = g.Co % 4 == 0;
  if (v4 && (((uintptr_t)dy | (uintptr_t)y_relu | (uintptr_t)dcols) & 15)) return RLGAN_E_ALIGN;
  if ((long long)n_images * g.Ho > 0x7fffffffll || (long long)n_images * g.h > 0x7fffffffll) return RLGAN_E_ALIGN;
  const int units = g.w * (g.KK / (v4 ? 4 : 1));
  g.dU = make_div(g.KK / (v4 ? 4 : 1));
  const dim3 grid(cdiv(units, 256), g.h, n_images), block(256);
  if (v4) hipLaunchKernelGGL(im2col_kernel<4>, grid, block, 0, (hipStream_t)stream, g, dy, y_relu, dcols);
  else hipLaunchKernelGGL(im2col_kernel<1>, grid, block, 0, (hipStream_t)stream, g, dy, y_relu, dcols);
  return launched();
}

int rlgan_column_sums(const float* dy, const float* y_relu, int64_t rows, int32_t n_cols, float* partial, float* out, void* stream) {
  if (!dy || !partial || !out || rows <= 0 || n_cols <= 0) return RLGAN_E_ARG;
  if (n_cols > COLSUM_MAX_COLS) return RLGAN_E_DIMS;
  return column_sums(dy, y_relu, rows, n_cols, partial, out, stream);
}

int rlgan_column_sums_noalloc(const float* dy, const float* y_relu, int64_t rows, int32_t n_cols, float* out, void* stream) {
  if (!dy || !out || rows <= 0 || n_cols <= 0) return RLGAN_E_ARG;
  if (n_cols > COLSUM_MAX_COLS) return RLGAN_E_SCRATCH;     // a partial row would not fit its slot of the fixed array
  float* partial = nullptr;
  int dev = -1;
  if (const int rc = device_of(stream, &dev)) return rc;
  if (const int rc = scratch_on(dev, &partial)) return rc;
  return column_sums(dy, y_relu, rows, n_cols, partial, out, stream);
}

int rlgan_noalloc_init(void) {
  int dev = -1;
  if (const hipError_t e = hipGetDevice(&dev)) return (int)e;
  float* p = nullptr;
  return scratch_on(dev, &p);
}

int rlgan_gather_map(const float* src, const int32_t* map, int32_t n, float* dst, void* stream) {
  if (!src || !map || !dst || n <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(gather_map_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, map, n, dst);
  return launched();
}

int rlgan_relu_backward(const float* dy, const float* y, float* dz, int64_t n, void* stream) {
  if (!dy || !y || !dz || n <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, (size_t)n);
  return launched();
}

int rlgan_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t rows, int n_copy, int n_zero, void* stream) {
  if (!dst || rows <= 0 || n_copy < 0 || n_zero < 0 || n_copy + n_zero <= 0 || (n_copy > 0 && !src)) return RLGAN_E_ARG;
  if (ld_dst < n_copy + n_zero || (n_copy > 0 && ld_src < n_copy)) return RLGAN_E_ALIGN;
  const int nc = n_copy + n_zero;
  if (nc <= 4) {
    hipLaunchKernelGGL(copy_cols_narrow_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, ld_src, dst, ld_dst, (long long)rows, n_copy, n_zero);
  } else {
    const unsigned gy = (unsigned)std::min<int64_t>(rows, 4096);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(cdiv(nc, 256), gy), dim3(256), 0, (hipStream_t)stream,
                       src, ld_src, dst, ld_dst, (long long)rows, n_copy, n_zero);
  }
  return launched();
}

int rlgan_concat_cols(const float* s0, int ld0, int w0, const float* s1, int ld1, int w1, const float* s2, int ld2, int w2,
                      float* dst, int ld_dst, int64_t rows, int n_zero, void* stream) {
  if (!dst || rows <= 0 || w0 < 0 || w1 < 0 || w2 < 0 || n_zero < 0 || (w0 > 0 && !s0) || (w1 > 0 && !s1) || (w2 > 0 && !s2)) return RLGAN_E_ARG;
  const long long nc = (long long)w0 + w1 + w2 + n_zero;
  if (nc <= 0) return RLGAN_E_ARG;
  if (ld_dst < nc || (w0 > 0 && ld0 < w0) || (w1 > 0 && ld1 < w1) || (w2 > 0 && ld2 < w2)) return RLGAN_E_ALIGN;
  ConcatArgs A{{s0, s1, s2}, {ld0, ld1, ld2}, {w0, w1, w2}, dst, ld_dst, n_zero, (long long)rows};
  for (int i = 0; i < 3; ++i) if (A.w[i] == 0) A.s[i] = dst;          // never read; keeps the selects free of null pointers
  const unsigned gy = (unsigned)std::min<int64_t>(rows, 4096);
  hipLaunchKernelGGL(concat_cols_kernel, dim3(cdiv((int)nc, 256), gy), dim3(256), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_blocks2x2(const float* src, float* dst, int64_t n_images, int h, int w, int to_blocks, void* stream) {
  if (!src || !dst || n_images <= 0 || h <= 0 || w <= 0) return RLGAN_E_ARG;
  if ((h & 1) || (w & 1)) return RLGAN_E_DIMS;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return RLGAN_E_ALIGN;   // image rows are then 8-byte aligned (w even)
  const long long nb = (long long)n_images * (h / 2) * (w / 2);
  if (nb > (long long)INT32_MAX * 256) return RLGAN_E_DIMS;
  const dim3 grid((unsigned)((nb + 255) / 256));
  if (to_blocks) hipLaunchKernelGGL(blocks2x2_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, nb, h / 2, w / 2);
  else hipLaunchKernelGGL(blocks2x2_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, nb, h / 2, w / 2);
  return launched();
}

int rlgan_add_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t rows, int n, void* stream) {
  if (!src || !dst || rows <= 0 || n <= 0) return RLGAN_E_ARG;
  if (ld_src < n || ld_dst < n) return RLGAN_E_ALIGN;
  const unsigned gy = (unsigned)std::min<int64_t>(rows, 4096);
  hipLaunchKernelGGL(add_cols_kernel, dim3(cdiv(n, 256), gy), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, (long long)rows, n);
  return launched();
}

int rlgan_sigmoid_ce(const float* logits, int ld, int n_real, int n_fake, float* loss, float* dlogits, void* stream) {
  if (!logits || !loss || ld <= 0 || n_real < 0 || n_fake < 0 || n_real + n_fake <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(sigmoid_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, n_real, n_fake, loss, dlogits);
  return launched();
}

int rlgan_one_minus_sigmoid(const float* logits, int ld, int n, float* out, void* stream) {
  if (!logits || !out || ld <= 0 || n <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(one_minus_sigmoid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, ld, n, out);
  return launched();
}

int rlgan_mse(const float* fake, const float* real, int64_t n, float* parts, float* dfake, void* stream) {
  if (!fake || !real || !parts || n <= 0) return RLGAN_E_ARG;
  const long long blocks = (n + MSE_PER_WG - 1) / MSE_PER_WG;
  if (blocks > 0x7fffffffll) return RLGAN_E_ARG;
  hipLaunchKernelGGL(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, fake, real, (long long)n,
                     (float)(2.0 / (double)n), parts, dfake);
  return launched();
}

}  // extern "C"
