// Pixel encoder: NHWC / VALID / square-kernel convolutions with ELU, the layer type of the
// reference's feature extractors (Conv_model.py:15-24 "doom" head 8@5x5/4,16@3x3/2,32@3x3/2,64@3x3/2;
// Conv_model.py:36-56 "universe" head n x 32@3x3/2; Conv_DDPG.py:83-86).  Forward and weight gradient
// are implicit GEMMs on v_mfma_f32_16x16x4_f32 (patches are never materialised); the input gradient
// is a gather over the output pixels that cover an input pixel (deterministic, no atomics).  gfx950 only.
//
// GEMM view of one layer:  Z[M, Cout] = P[M, K] * W[K, Cout] + b,  M = n*Ho*Wo output pixels,
// K = k*k*Cin patch elements ordered (ky, kx, c) -- the row-major flattening of TensorFlow's HWIO
// kernel, so W is the kernel tensor as stored.  Weights live on the device as MFMA B fragments of
// [W ; b] (the same fragment format the dense layers use), the bias being the row of a constant-1
// patch element at index K.
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

struct ConvG {
  int H, W, Cin, Cout, k, s;      // input height/width/channels, output channels, kernel size, stride
  int Ho, Wo, K, NG, NT, rowlen;  // output size, patch length, k-groups (of 16, incl. bias row), column tiles, k*Cin
};

inline bool make_conv(const rlgan_conv_t* c, ConvG& g) {
  if (!c || c->h <= 0 || c->w <= 0 || c->c_in <= 0 || c->c_out <= 0 || c->k <= 0 || c->stride <= 0) return false;
  if (c->c_in % 4 != 0 || c->c_out % 4 != 0 || c->c_out > 64 || c->k > c->h || c->k > c->w) return false;   // 16-byte segments, <= 4 column tiles
  g.H = c->h; g.W = c->w; g.Cin = c->c_in; g.Cout = c->c_out; g.k = c->k; g.s = c->stride;
  g.Ho = (c->h - c->k) / c->stride + 1; g.Wo = (c->w - c->k) / c->stride + 1;
  g.K = c->k * c->k * c->c_in; g.NG = cdiv(g.K + 1, 16); g.NT = cdiv(c->c_out, 16); g.rowlen = c->k * c->c_in;
  return true;
}

__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }
__device__ __forceinline__ float elu_d(float y) { return y > 0.f ? 1.f : y + 1.f; }   // ELU' from the layer output

// this lane's four consecutive patch elements k0..k0+3 of output pixel `pix` (k0 % 4 == 0; rowlen % 4 == 0,
// so the four never straddle a kernel row); element K is the constant 1 of the bias row, zeros after it
__device__ __forceinline__ f32x4 patch4(const ConvG& g, const float* __restrict__ x, int b, int oy, int ox, int k0) {
  if (k0 >= g.K) return f32x4{k0 == g.K ? 1.f : 0.f, 0.f, 0.f, 0.f};
  const int ky = k0 / g.rowlen, rem = k0 - ky * g.rowlen;
  const float* p = x + (((size_t)b * g.H + (oy * g.s + ky)) * g.W + ox * g.s) * g.Cin + rem;
  return *(const f32x4*)p;
}

// ---------------------------------------------------------------- forward
// One wave = 16 output pixels x all column tiles; A operands straight from global memory in MFMA
// layout (lane = (pixel l&15, k-quad l>>4)), B operands = weight fragments; double-buffered.
template <int NT>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvG g, const float* __restrict__ x, const float* __restrict__ wfrag,
                                                       float* __restrict__ y, int M, int act_elu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pix0 = (blockIdx.x * 4 + wave) * 16;
  if (pix0 >= M) return;
  const int prow = min(pix0 + (lane & 15), M - 1);
  const int b = prow / (g.Ho * g.Wo), r = prow - b * (g.Ho * g.Wo), oy = r / g.Wo, ox = r - oy * g.Wo;
  const int kq = (lane >> 4) * 4;
  const float* wl = wfrag + lane * 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a = patch4(g, x, b, oy, ox, kq);
  f32x4 w[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) w[t] = *(const f32x4*)(wl + t * 256);
  for (int gi = 0; gi < g.NG; ++gi) {
    f32x4 an = a, wn[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wn[t] = w[t];
    if (gi + 1 < g.NG) {
      an = patch4(g, x, b, oy, ox, 16 * (gi + 1) + kq);
#pragma unroll
      for (int t = 0; t < NT; ++t) wn[t] = *(const f32x4*)(wl + ((gi + 1) * NT + t) * 256);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], w[t][m], acc[t], 0, 0, 0);
    }
    a = an;
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = wn[t];
  }
  // D fragment: rows (pixels) 4*(lane>>4)+i, column lane&15
  const int col = lane & 15;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = 16 * t + col;
    if (n >= g.Cout) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = pix0 + (lane >> 4) * 4 + i;
      if (p < M) y[(size_t)p * g.Cout + n] = act_elu ? elu_f(acc[t][i]) : acc[t][i];
    }
  }
}

// ---------------------------------------------------------------- dZ = dY * ELU'(Y)   (ELU' from the output: y > 0 ? 1 : y + 1)
__global__ void elu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dz[i] = dy[i] * elu_d(y[i]);
}

// ---------------------------------------------------------------- input gradient
// dX[b,y,x,c] = sum over (ky,kx) with (y-ky)%s == 0, (x-kx)%s == 0, output pixel inside the map, of
// dZ[b,oy,ox,:] . W[ky,kx,c,:].  One thread per (input pixel, four channels), 16-byte loads over the
// output channels; fixed summation order, no atomics.
__global__ __launch_bounds__(256) void conv_bwd_input_kernel(const ConvG g, const float* __restrict__ dz,
                                                             const float* __restrict__ w /*[K][Cout]*/,
                                                             float* __restrict__ dx, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int c4 = g.Cin >> 2;
  const int c = (int)(i % c4) * 4;
  size_t q = i / c4;
  const int xx = (int)(q % g.W); q /= g.W;
  const int yy = (int)(q % g.H);
  const int b = (int)(q / g.H);
  f32x4 s{0.f, 0.f, 0.f, 0.f};
  for (int ky = yy % g.s; ky < g.k; ky += g.s) {
    const int oy = (yy - ky) / g.s;
    if (yy < ky || oy >= g.Ho) continue;
    for (int kx = xx % g.s; kx < g.k; kx += g.s) {
      const int ox = (xx - kx) / g.s;
      if (xx < kx || ox >= g.Wo) continue;
      const float* zp = dz + (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout;
      const float* wp = w + (size_t)((ky * g.k + kx) * g.Cin + c) * g.Cout;
      for (int n = 0; n < g.Cout; n += 4) {
        const f32x4 z = *(const f32x4*)(zp + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 wv = *(const f32x4*)(wp + j * g.Cout + n);
          s[j] = fmaf(z[0], wv[0], s[j]); s[j] = fmaf(z[1], wv[1], s[j]);
          s[j] = fmaf(z[2], wv[2], s[j]); s[j] = fmaf(z[3], wv[3], s[j]);
        }
      }
    }
  }
  *(f32x4*)(dx + i * 4) = s;
}

// ---------------------------------------------------------------- input gradient on the matrix cores
// Input pixels fall into s*s classes by (y % s, x % s).  Within a class every pixel (y, x) = (s*a + py,
// s*b + px) is covered by the same kernel taps ky = py + s*u, kx = px + s*v, from output pixels
// (a - u, b - v): a stride-1 correlation of dZ with the class's share of the kernel, i.e. the GEMM
//   dX_class[M_c, Cin] = G[M_c, nky*nkx*Cout] * Wc[nky*nkx*Cout, Cin],
// G gathering dZ rows (zeros where the output pixel lies outside the map), Wc[(u, v, co)][c] =
// W[ky, kx, c, co].  The Wc of all classes are kept as MFMA B fragments, one image after the other.
constexpr int MAX_CLASSES = 16;            // stride <= 4
struct DxClass { int py, px, nky, nkx, na, nb, Kc, NG; };
__host__ __device__ inline DxClass dx_class(const ConvG& g, int cls) {
  DxClass c;
  c.py = cls / g.s; c.px = cls % g.s;
  c.nky = c.py < g.k ? (g.k - c.py + g.s - 1) / g.s : 0;
  c.nkx = c.px < g.k ? (g.k - c.px + g.s - 1) / g.s : 0;
  c.na = c.py < g.H ? (g.H - c.py + g.s - 1) / g.s : 0;
  c.nb = c.px < g.W ? (g.W - c.px + g.s - 1) / g.s : 0;
  c.Kc = c.nky * c.nkx * g.Cout;
  c.NG = cdiv(c.Kc, 16);
  return c;
}
struct DxOff { int off[MAX_CLASSES]; };    // float offset of every class's fragment image

// this lane's four consecutive elements kk0..kk0+3 of the gathered row of input pixel (a, bb) of a class
// (Cout % 4 == 0: the four share one tap)
__device__ __forceinline__ f32x4 dz4(const ConvG& g, const DxClass& c, const float* __restrict__ dz, const float* __restrict__ yelu,
                                    int b, int a, int bb, int kk0) {
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  if (kk0 >= c.Kc) return zero;
  const int tap = kk0 / g.Cout, co = kk0 - tap * g.Cout;
  const int u = tap / c.nkx, v = tap - u * c.nkx;
  const int oy = a - u, ox = bb - v;
  if (oy < 0 || oy >= g.Ho || ox < 0 || ox >= g.Wo) return zero;
  const size_t at = (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.Cout + co;
  f32x4 d = *(const f32x4*)(dz + at);
  if (yelu) {                                                     // dz is dY: times ELU'(y) on the way in
    const f32x4 yv = *(const f32x4*)(yelu + at);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] *= elu_d(yv[i]);
  }
  return d;
}

// One wave = 16 input pixels of one class x all Cin column tiles; blockIdx.y = class.
template <int NT>
__global__ __launch_bounds__(256) void conv_dx_kernel(const ConvG g, const DxOff offs, const float* __restrict__ dz,
                                                      const float* __restrict__ yelu, const float* __restrict__ wdx,
                                                      float* __restrict__ dx, int n_images) {
  const DxClass c = dx_class(g, blockIdx.y);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = c.na * c.nb;
  const long long Ml = (long long)n_images * per;
  const long long pix0 = ((long long)blockIdx.x * 4 + wave) * 16;
  if (pix0 >= Ml) return;
  const long long prow = min(pix0 + (lane & 15), Ml - 1);
  const int b = (int)(prow / per), r = (int)(prow - (long long)b * per), a = r / c.nb, bb = r - a * c.nb;
  const int kq = (lane >> 4) * 4;
  const float* wl = wdx + offs.off[blockIdx.y] + lane * 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c.NG > 0) {
    f32x4 av = dz4(g, c, dz, yelu, b, a, bb, kq);
    f32x4 w[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = *(const f32x4*)(wl + t * 256);
    for (int gi = 0; gi < c.NG; ++gi) {
      f32x4 an = av, wn[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wn[t] = w[t];
      if (gi + 1 < c.NG) {
        an = dz4(g, c, dz, yelu, b, a, bb, 16 * (gi + 1) + kq);
#pragma unroll
        for (int t = 0; t < NT; ++t) wn[t] = *(const f32x4*)(wl + ((gi + 1) * NT + t) * 256);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], w[t][m], acc[t], 0, 0, 0);
      }
      av = an;
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t] = wn[t];
    }
  }
  // D fragment: rows (pixels) 4*(lane>>4)+i, column lane&15
  const int col = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long p = pix0 + (lane >> 4) * 4 + i;
    if (p >= Ml) continue;
    const int pb = (int)(p / per), pr = (int)(p - (long long)pb * per), pa = pr / c.nb, pbb = pr - pa * c.nb;
    float* o = dx + (((size_t)pb * g.H + (g.s * pa + c.py)) * g.W + (g.s * pbb + c.px)) * g.Cin;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * t + col;
      if (n < g.Cin) o[n] = acc[t][i];
    }
  }
}

// ---------------------------------------------------------------- weight gradient
// dW[K+1, Cout] = [P 1]^T dZ reduced over all output pixels.  A workgroup walks its share of the
// pixels WG_PIX at a time: patches and dZ rows are staged in LDS, wave w owns the 16-row strips
// w, w+4, ... of K and all column tiles.  Every workgroup writes one partial image (slab); a second
// kernel adds the slabs in workgroup order, so the result is bitwise reproducible.
constexpr int WG_PIX = 16;   // pixels staged per step: small, so that several workgroups share a compute unit and one's
                             // staging overlaps another's MFMAs (64 left one workgroup per unit: 2.4x slower)
template <int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ConvG g, const float* __restrict__ x, const float* __restrict__ dz,
                                                         const float* __restrict__ yelu, float* __restrict__ slabs, int M, int pix_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int KP = g.NG * 16, ldp = KP + 4, ldz = NT * 16 + 4;
  float* P = lds;                       // [WG_PIX][ldp]
  float* Z = lds + WG_PIX * ldp;        // [WG_PIX][ldz]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nstrips = g.NG, my_strips = (nstrips - wave + 3) / 4;
  constexpr int MAXS = 5;               // K+1 <= 320 rows (the widest layer has 288 + 1)
  f32x4 acc[MAXS][NT];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int p_begin = blockIdx.x * pix_per_wg, p_end = min(M, p_begin + pix_per_wg);
  for (int p0 = p_begin; p0 < p_end; p0 += WG_PIX) {
    __syncthreads();
    // stage patches (16 bytes per thread and step) and dZ rows; pixels beyond the range are zero rows
    for (int e = threadIdx.x; e < WG_PIX * (KP / 4); e += 256) {
      const int pl = e / (KP / 4), k0 = (e - pl * (KP / 4)) * 4;
      const int p = p0 + pl;
      f32x4 v{0.f, 0.f, 0.f, 0.f};
      if (p < p_end) {
        const int b = p / (g.Ho * g.Wo), r = p - b * (g.Ho * g.Wo), oy = r / g.Wo, ox = r - oy * g.Wo;
        v = patch4(g, x, b, oy, ox, k0);
      }
      *(f32x4*)(P + pl * ldp + k0) = v;
    }
    for (int e = threadIdx.x; e < WG_PIX * NT * 16; e += 256) {
      const int pl = e / (NT * 16), n = e - pl * (NT * 16);
      const int p = p0 + pl;
      float v = 0.f;
      if (p < p_end && n < g.Cout) {
        v = dz[(size_t)p * g.Cout + n];
        if (yelu) v *= elu_d(yelu[(size_t)p * g.Cout + n]);      // dz is dY: times ELU'(y) on the way in
      }
      Z[pl * ldz + n] = v;
    }
    __syncthreads();
    // A[row = patch element][kk = pixel], B[kk = pixel][col = channel]
#pragma unroll
    for (int pg = 0; pg < WG_PIX / 16; ++pg) {
      const int pq = 16 * pg + (lane >> 4) * 4;
      f32x4 bz[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m) bz[t][m] = Z[(pq + m) * ldz + 16 * t + (lane & 15)];
      }
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        if (s < my_strips) {
          const int krow = 16 * (wave + 4 * s) + (lane & 15);
          f32x4 ap;
#pragma unroll
          for (int m = 0; m < 4; ++m) ap[m] = P[(pq + m) * ldp + krow];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[m], bz[t][m], acc[s][t], 0, 0, 0);
          }
        }
      }
    }
  }
  // slab [KP][NT*16]: D fragment rows = patch elements 4*(lane>>4)+i of the strip, column lane&15
  float* out = slabs + (size_t)blockIdx.x * KP * NT * 16;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    if (s < my_strips) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          out[(size_t)(16 * (wave + 4 * s) + (lane >> 4) * 4 + i) * (NT * 16) + 16 * t + (lane & 15)] = acc[s][t][i];
      }
    }
  }
}

// grad[(k, n)] in canonical order ([K][Cout] kernel, then bias[Cout]) = sum of the slabs.  A workgroup
// handles 16 entries; 16 threads per entry each add a strided share of the slabs in index order and the
// 16 partial sums are added in a fixed order, so the result does not depend on timing.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int KP, int NP,
                                                                int K, int Cout, float* __restrict__ grad) {
  __shared__ float part[16][17];
  const int e = threadIdx.x >> 4, j = threadIdx.x & 15;
  const int i = blockIdx.x * 16 + e;
  const int total = (K + 1) * Cout;
  float s = 0.f;
  if (i < total) {
    const int k = i / Cout, n = i - k * Cout;
    const float* p = slabs + (size_t)k * NP + n;
    for (int q = j; q < n_slabs; q += 16) s += p[(size_t)q * KP * NP];
  }
  part[e][j] = s;
  __syncthreads();
  if (j == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[e][q];
    grad[i] = t;
  }
}

// fragment images of the input-gradient classes: offsets and total size (0 when the layer is outside
// what conv_dx_kernel handles: stride > 4 or more than 64 input channels)
inline int dx_layout(const ConvG& g, DxOff* offs) {
  if (g.s * g.s > MAX_CLASSES || g.Cin > 64) return 0;
  const int ntc = cdiv(g.Cin, 16);
  int total = 0;
  for (int cls = 0; cls < g.s * g.s; ++cls) {
    if (offs) offs->off[cls] = total;
    total += dx_class(g, cls).NG * ntc * 256;
  }
  return total;
}

inline int launched() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int rlgan_conv_geometry(const rlgan_conv_t* conv, rlgan_conv_geom_t* out) {
  ConvG g;
  if (!out) return RLGAN_E_ARG;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  out->h_out = g.Ho; out->w_out = g.Wo; out->k_len = g.K; out->n_params = (g.K + 1) * g.Cout;
  out->wfrag_floats = g.NG * g.NT * 256; out->slab_floats = g.NG * 16 * g.NT * 16;
  out->wdx_floats = dx_layout(g, nullptr);
  return 0;
}

int rlgan_conv_build_dx_map(const rlgan_conv_t* conv, int32_t* map_dx) {
  ConvG g;
  if (!map_dx) return RLGAN_E_ARG;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  DxOff offs;
  if (dx_layout(g, &offs) == 0) return RLGAN_E_DIMS;
  const int ntc = cdiv(g.Cin, 16);
  int p = 0;
  for (int ky = 0; ky < g.k; ++ky)
    for (int kx = 0; kx < g.k; ++kx) {
      const int cls = (ky % g.s) * g.s + (kx % g.s);
      const DxClass c = dx_class(g, cls);
      const int tap = (ky / g.s) * c.nkx + (kx / g.s);
      for (int ci = 0; ci < g.Cin; ++ci)
        for (int co = 0; co < g.Cout; ++co) map_dx[p++] = offs.off[cls] + frag_offset(ntc, tap * g.Cout + co, ci);
    }
  for (int n = 0; n < g.Cout; ++n) map_dx[p++] = -1;     // the bias takes no part
  return 0;
}

int rlgan_conv_build_map(const rlgan_conv_t* conv, int32_t* map_f) {
  ConvG g;
  if (!map_f) return RLGAN_E_ARG;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  int p = 0;
  for (int k = 0; k <= g.K; ++k)                 // kernel rows in HWIO order, then the bias as row K
    for (int n = 0; n < g.Cout; ++n) map_f[p++] = frag_offset(g.NT, k, n);
  return 0;
}

int rlgan_conv_forward(const rlgan_conv_t* conv, const float* x, const float* wfrag, float* y, int n_images,
                       int act_elu, void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!x || !wfrag || !y || n_images <= 0) return RLGAN_E_ARG;
  const long long Ml = (long long)n_images * g.Ho * g.Wo;
  if (Ml > 0x7fffffffll) return RLGAN_E_ALIGN;
  const int M = (int)Ml;
  const dim3 grid(cdiv(M, 64)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (g.NT) {
    case 1: hipLaunchKernelGGL(conv_fwd_kernel<1>, grid, block, 0, st, g, x, wfrag, y, M, act_elu); break;
    case 2: hipLaunchKernelGGL(conv_fwd_kernel<2>, grid, block, 0, st, g, x, wfrag, y, M, act_elu); break;
    case 3: hipLaunchKernelGGL(conv_fwd_kernel<3>, grid, block, 0, st, g, x, wfrag, y, M, act_elu); break;
    default: hipLaunchKernelGGL(conv_fwd_kernel<4>, grid, block, 0, st, g, x, wfrag, y, M, act_elu); break;
  }
  return launched();
}

int rlgan_elu_backward(const float* dy, const float* y, float* dz, int64_t n, void* stream) {
  if (!dy || !y || !dz || n <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, (size_t)n);
  return launched();
}

int rlgan_conv_backward_input(const rlgan_conv_t* conv, const float* dz, const float* y_elu, const float* w_canon, const float* wdx,
                              float* dx, int n_images, void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!dz || (!w_canon && !wdx) || !dx || n_images <= 0 || (y_elu && !wdx)) return RLGAN_E_ARG;
  if (wdx) {
    DxOff offs;
    if (dx_layout(g, &offs) == 0) return RLGAN_E_DIMS;
    const long long per = (long long)cdiv(g.H, g.s) * cdiv(g.W, g.s);          // pixels per image of the largest class
    const long long groups = cdiv((long long)n_images * per, 64);
    if (groups > 0x7fffffffll) return RLGAN_E_ALIGN;
    const dim3 grid((unsigned)groups, g.s * g.s), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (cdiv(g.Cin, 16)) {
      case 1: hipLaunchKernelGGL(conv_dx_kernel<1>, grid, block, 0, st, g, offs, dz, y_elu, wdx, dx, n_images); break;
      case 2: hipLaunchKernelGGL(conv_dx_kernel<2>, grid, block, 0, st, g, offs, dz, y_elu, wdx, dx, n_images); break;
      case 3: hipLaunchKernelGGL(conv_dx_kernel<3>, grid, block, 0, st, g, offs, dz, y_elu, wdx, dx, n_images); break;
      default: hipLaunchKernelGGL(conv_dx_kernel<4>, grid, block, 0, st, g, offs, dz, y_elu, wdx, dx, n_images); break;
    }
    return launched();
  }
  const size_t total4 = (size_t)n_images * g.H * g.W * (g.Cin / 4);
  hipLaunchKernelGGL(conv_bwd_input_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g, dz, w_canon, dx, total4);
  return launched();
}

int rlgan_conv_backward_weights(const rlgan_conv_t* conv, const float* x, const float* dz, const float* y_elu, float* slabs,
                                int n_slabs, float* grad, int n_images, void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!x || !dz || !slabs || !grad || n_images <= 0 || n_slabs <= 0) return RLGAN_E_ARG;
  if (g.NG > 20) return RLGAN_E_DIMS;                       // five 16-row strips per wave
  const long long Ml = (long long)n_images * g.Ho * g.Wo;
  if (Ml > 0x7fffffffll) return RLGAN_E_ALIGN;
  const int M = (int)Ml;
  const int per = cdiv(cdiv(M, n_slabs), WG_PIX) * WG_PIX;  // pixels per workgroup, a multiple of the staging step
  const int used = cdiv(M, per);                            // slabs actually written (<= n_slabs)
  const int KP = g.NG * 16, NP = g.NT * 16;
  const size_t lds = (size_t)(WG_PIX * (KP + 4) + WG_PIX * (NP + 4)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(used), block(256);
#define RLGAN_WG(NTV)                                                                                              \
  {                                                                                                                \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(conv_wgrad_kernel<NTV>, grid, block, lds, st, g, x, dz, y_elu, slabs, M, per);                     \
  }
  switch (g.NT) {
    case 1: RLGAN_WG(1) break;
    case 2: RLGAN_WG(2) break;
    case 3: RLGAN_WG(3) break;
    default: RLGAN_WG(4) break;
  }
#undef RLGAN_WG
  if (const int rc = launched()) return rc;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv((g.K + 1) * g.Cout, 16)), dim3(256), 0, st, slabs, used, KP, NP, g.K, g.Cout, grad);
  return launched();
}

}  // extern "C"
