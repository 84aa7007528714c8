// Stride-2 transposed convolutions of the pixel generator's decoder (Conv_model.py:457-473) without the `cols`
// matrix in memory, for the two layer shapes that carry most of the decoder's time:
//
//   tile layer    8x8x64  -> 19x19x32, 5x5 kernel (dec_deconv3, Conv_model.py:468): output channels in MFMA column
//                 tiles; a wave owns one image and one 16-channel tile, multiplies the image's 64 input pixels with
//                 one kernel tap at a time and adds the 64x16 result into its own copy of the output tile in LDS,
//                 taps in ascending (ky, kx) order (no two waves ever touch the same output element);
//   narrow layer  19x19x32 -> 42x42x1, 6x6 kernel (dec_deconv4, Conv_model.py:473): the 36 taps of the single
//                 output channel are the MFMA columns; the per-image cols matrix (361 x 36) lives in LDS and the
//                 placement gathers from there.
//
//   few-pixel     3x3x128 -> 8x8x64, 4x4 kernel (dec_deconv2, Conv_model.py:463), forward only: sixteen images are the
//   layer         rows of a tile, see deconv_few_fwd_kernel; its backward pass keeps the cols form.
//
// Backward passes read the output gradient from LDS through the same pixel arithmetic: dX = sum over taps of
// dZ[pixel(tap)] * W_tap^T, dW_tap = sum over images and pixels of dZ[pixel(tap)]^T * X, with per-chunk partial
// kernels added in chunk order (the same bits every run).  Other shapes keep the cols form of deconv_kernels.hip.
// gfx950 only.
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

inline int launched() { return (int)hipGetLastError(); }

__device__ __forceinline__ float relu_d(float y) { return y > 0.f ? 1.f : 0.f; }
__device__ __forceinline__ f32x4 mask4(f32x4 d, f32x4 y) {
  return f32x4{d[0] * relu_d(y[0]), d[1] * relu_d(y[1]), d[2] * relu_d(y[2]), d[3] * relu_d(y[3])};
}
// orders the LDS traffic of one wave as written: a wave's LDS operations execute in program order, so a value
// another lane of the same wave stored is visible to later reads as long as the compiler keeps that order
__device__ __forceinline__ void keep_lds_order() { asm volatile("" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// tile layer
// ------------------------------------------------------------------------------------------------
template <int H_, int CI_, int CO_, int K_>
struct TileT {
  static constexpr int H = H_, CI = CI_, CO = CO_, K = K_;
  static constexpr int P = H * H, MT = P / 16, KG = CI / 16, CT = CO / 16, NT = CI / 16, KGO = CO / 16;
  static constexpr int HO = (H - 1) * 2 + K, NPIX = HO * HO, TAPS = K * K, KW = TAPS * CO * CI;
  static constexpr int IPW = 4 / CT;                       // images per workgroup of the forward kernel
  static_assert(P % 16 == 0 && CI % 16 == 0 && CO % 16 == 0 && 4 % CT == 0 && NT == 4, "tile layer shape");
  static constexpr int FWD_LDS = 4 * NPIX * 16 * 4;        // four waves' output tiles [NPIX][16]
  static constexpr int ZS = CO + 4;                        // pixel stride of the dZ image of the dX kernel
  static constexpr int DX_LDS = (NPIX * ZS + 256) * 4;
  // weight-gradient kernel: dZ rows oy = 2 iy + ky as [H][R], pixel stride S.  With H = 8 the four lane groups of
  // an operand read (pixels 4 apart in x, or one input row apart) then start 16 banks apart.
  static constexpr int S = CO + 2, R = ((HO * S + 31) / 64) * 64 + 32;
  static constexpr int DW_LDS = (P * CI + H * R) * 4;
  static constexpr int MAX_CHUNKS = 256;
};
using Mid = TileT<8, 64, 32, 5>;

// The taps of a stride-2 layer fall into classes by the parity of (ky, kx), and the classes write disjoint output
// pixels (oy = 2 iy + ky).  Side 0 of a pair of waves takes the taps with ky + kx even, side 1 the others (13 and 12
// of the 25 taps of a 5x5 kernel): the two add into one output tile at the same time without ever meeting in an
// element, and every element still receives its taps in ascending order.
// NS sides: two are used.  (Three -- the classes as 9, 6 + 4 and 6 taps, three waves per SIMD -- are no faster: 261 against
// 259 us at 4096 images, and the iteration at 32 images, where a workgroup is a chain of dependent products, 1.012 -> 1.010 ms.)
template <class T, int NS> __host__ __device__ constexpr int tile_side_of(int tap) {
  const int ky = tap / T::K, kx = tap % T::K;
  if (NS == 2) return (ky + kx) & 1;
  const int cls = (ky & 1) * 2 + (kx & 1);
  return cls == 0 ? 0 : cls == 2 ? 2 : 1;
}
template <class T, int NS> __host__ __device__ constexpr int tile_side_taps(int side) {
  int n = 0;
  for (int tap = 0; tap < T::TAPS; ++tap) n += tile_side_of<T, NS>(tap) == side;
  return n;
}
template <class T, int NS> __host__ __device__ constexpr int tile_side_tap(int side, int i) {
  for (int tap = 0; tap < T::TAPS; ++tap) {
    if (tile_side_of<T, NS>(tap) == side && i-- == 0) return tap;
  }
  return T::TAPS;
}
template <class T, int NS, int SIDE>
__device__ __forceinline__ void tile_fwd_taps(const f32x4 (&a)[T::MT][T::KG], float* Y, const int (&po)[T::MT][4], const FragSrc& ws) {
  constexpr int N = tile_side_taps<T, NS>(SIDE);
  f32x4 b[2][T::KG];
#pragma unroll
  for (int g = 0; g < T::KG; ++g) b[0][g] = ws.load(tile_side_tap<T, NS>(SIDE, 0) * T::CT * T::KG + g);
  f32x4 prev[T::MT];
#pragma unroll
  for (int i = 0; i <= N; ++i) {
    f32x4 acc[T::MT];
    if (i < N) {
      if (i + 1 < N) {
#pragma unroll
        for (int g = 0; g < T::KG; ++g) b[(i + 1) & 1][g] = ws.load(tile_side_tap<T, NS>(SIDE, i + 1) * T::CT * T::KG + g);
      }
#pragma unroll
      for (int t = 0; t < T::MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < T::KG; ++g) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int t = 0; t < T::MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][g][m], b[i & 1][g][m], acc[t], 0, 0, 0);
        }
      }
    }
    if (i > 0) {
      // the previous tap's products go to their output pixels while this tap's MFMAs run
      const int tap = tile_side_tap<T, NS>(SIDE, i - 1);
      const int off = ((tap / T::K) * T::HO + tap % T::K) * 16;
#pragma unroll
      for (int t = 0; t < T::MT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[po[t][r] + off] += prev[t][r];      // (as ds_add_f32, one instruction each: 256 -> 1112 us)
      }
      keep_lds_order();
    }
#pragma unroll
    for (int t = 0; t < T::MT; ++t) prev[t] = acc[t];
  }
}

// y[n] = act(bias + sum over taps of place(X[n] W_tap)).  grid = ceil(n / IPW); eight waves: waves w and w + 4 (one
// SIMD) are the two sides of (image, channel tile) w, so each SIMD has one wave's additions into LDS, its first
// loads and its last stores to run under the other's products.  (With one wave per tile, all taps in a row, the
// matrix pipe stood idle for 15 of the 40 us a workgroup took.)
template <class T, int NS>
__global__ __launch_bounds__(256 * NS) void deconv_tile_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wff,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int n_images, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  const int tile = wave & 3, side = wave >> 2;
  const int c = tile % T::CT, n = blockIdx.x * T::IPW + tile / T::CT;
  const bool live = n < n_images;                          // a tile past the last image computes the last image again and stores nothing
  float* Y = lds + tile * (T::NPIX * 16);
  FragSrc ws;
  ws.set(wff, c * T::KG);
  // the image's input pixels as A operands, kept for all taps
  f32x4 a[T::MT][T::KG];
  const float* xn = x + (size_t)min(n, n_images - 1) * T::P * T::CI;
#pragma unroll
  for (int t = 0; t < T::MT; ++t) {
#pragma unroll
    for (int g = 0; g < T::KG; ++g) a[t][g] = *(const f32x4*)(xn + (16 * t + j) * T::CI + 16 * g + 4 * q);
  }
  const f32x4 b4 = *(const f32x4*)(bias + 16 * c + 4 * (lane & 3));
  for (int i = lane + 64 * side; i < T::NPIX * 4; i += 64 * NS) ((f32x4*)Y)[i] = b4;
  // LDS position of accumulator element (t, r) = input pixel 16 t + 4 q + r, channel j, for tap (0, 0)
  int po[T::MT][4];
#pragma unroll
  for (int t = 0; t < T::MT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int px = 16 * t + 4 * q + r;
      po[t][r] = ((2 * (px / T::H)) * T::HO + 2 * (px % T::H)) * 16 + j;
    }
  }
  __syncthreads();                                         // both sides' halves of the bias are in place
  if (side == 0) tile_fwd_taps<T, NS, 0>(a, Y, po, ws);
  else if (side == 1) tile_fwd_taps<T, NS, 1>(a, Y, po, ws);
  else tile_fwd_taps<T, NS, NS - 1>(a, Y, po, ws);
  __syncthreads();
  if (!live) return;
  float* yn = y + (size_t)n * T::NPIX * T::CO + 16 * c + 4 * (lane & 3);
  for (int i = lane + 64 * side; i < T::NPIX * 4; i += 64 * NS) {
    f32x4 v = ((const f32x4*)Y)[i];
    if (act) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    *(f32x4*)(yn + (size_t)(i >> 2) * T::CO) = v;
  }
}

// dX[n] = sum over taps of dZ[n][pixels of the tap] W_tap^T, dZ = dy * ReLU'(y); bias_part[n][co] = column sums of
// dZ[n].  grid = n; wave = 16-wide tile of input channels.
template <class T>
__global__ __launch_bounds__(256) void deconv_tile_dx_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                              const float* __restrict__ wfb, float* __restrict__ dx,
                                                              float* __restrict__ bias_part, const float* __restrict__ x_relu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Z = lds;                                          // [NPIX][ZS]
  float* red = lds + T::NPIX * T::ZS;                      // [256]
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  const int n = blockIdx.x;
  constexpr int STEPS = T::TAPS * T::KGO, RB = 4;
  FragSrc ws;
  ws.set(wfb, wave * STEPS);
  f32x4 b[RB];
#pragma unroll
  for (int s = 0; s < RB; ++s) b[s] = ws.load(s);
  constexpr int C4 = T::CO / 4;
  const f32x4* dy4 = (const f32x4*)(dy + (size_t)n * T::NPIX * T::CO);
  const f32x4* y4 = (const f32x4*)(yact + (size_t)n * T::NPIX * T::CO);
  // six pieces per thread and turn, their loads requested together (a piece per turn was a memory round trip per turn)
  constexpr int SU = 6;
  for (int i0 = threadIdx.x; i0 < T::NPIX * C4; i0 += 256 * SU) {
    f32x4 d[SU], yv[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = min(i0 + 256 * u, T::NPIX * C4 - 1);
      d[u] = dy4[i];
      yv[u] = yact ? y4[i] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + 256 * u;
      if (i < T::NPIX * C4) *(f32x4*)(Z + (i / C4) * T::ZS + 4 * (i % C4)) = yact ? mask4(d[u], yv[u]) : d[u];
    }
  }
  __syncthreads();
  int ab[T::MT];
#pragma unroll
  for (int t = 0; t < T::MT; ++t) {
    const int px = 16 * t + j;
    ab[t] = ((2 * (px / T::H)) * T::HO + 2 * (px % T::H)) * T::ZS + 4 * q;
  }
  f32x4 acc[T::MT];
#pragma unroll
  for (int t = 0; t < T::MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int tap = s / T::KGO, g = s % T::KGO;
    const int off = ((tap / T::K) * T::HO + tap % T::K) * T::ZS + 16 * g;
    f32x4 av[T::MT];
#pragma unroll
    for (int t = 0; t < T::MT; ++t) av[t] = *(const f32x4*)(Z + ab[t] + off);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int t = 0; t < T::MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][m], b[s % RB][m], acc[t], 0, 0, 0);
    }
    if (s + RB < STEPS) b[s % RB] = ws.load(s + RB);
  }
  float* dxn = dx + (size_t)n * T::P * T::CI + 16 * wave + j;
  if (x_relu) {
    // the layer input at this lane's results (the ReLU output of the layer below): the gradient leaves as that layer's dZ
    const float* xn = x_relu + (size_t)n * T::P * T::CI + 16 * wave + j;
    float xr[T::MT][4];
#pragma unroll
    for (int t = 0; t < T::MT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xr[t][r] = xn[(16 * t + 4 * q + r) * T::CI];
    }
#pragma unroll
    for (int t = 0; t < T::MT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dxn[(16 * t + 4 * q + r) * T::CI] = acc[t][r] * relu_d(xr[t][r]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < T::MT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dxn[(16 * t + 4 * q + r) * T::CI] = acc[t][r];
    }
  }
  // bias gradient of this image: thread = (pixel phase, channel), phases added in order
  constexpr int PH = 256 / T::CO;
  const int co = threadIdx.x % T::CO, ph = threadIdx.x / T::CO;
  float s = 0.f;
  for (int p = ph; p < T::NPIX; p += PH) s += Z[p * T::ZS + co];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < T::CO) {
    float tot = 0.f;
#pragma unroll
    for (int p = 0; p < PH; ++p) tot += red[p * T::CO + threadIdx.x];
    bias_part[(size_t)n * T::CO + threadIdx.x] = tot;
  }
}

// slabs[chunk][tap][co][ci] = sum over the chunk's images and input pixels of dZ[pixel(tap)][co] X[pixel][ci] for
// the K taps of kernel row ky.  grid = (K, chunks); wave = 16-wide tile of input channels.
template <class T>
__global__ __launch_bounds__(256) void deconv_tile_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ yact, float* __restrict__ slabs,
                                                              int n_images, int per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xt = lds;                                         // [P / 4][CI][4 pixels]
  float* Zr = lds + T::P * T::CI;                          // [H][R], pixel stride S
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  const int ky = blockIdx.x, chunk = blockIdx.y;
  const int n0 = chunk * per_chunk, n1 = min(n_images, n0 + per_chunk);
  f32x4 acc[T::K][T::CT];
#pragma unroll
  for (int kx = 0; kx < T::K; ++kx) {
#pragma unroll
    for (int c = 0; c < T::CT; ++c) acc[kx][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int C4 = T::CO / 4;
  // An image's operands travel memory -> registers while the previous image is multiplied out of LDS, every piece of
  // them requested at once (a piece per loop turn was a memory round trip per turn, nine per image, with nothing of the
  // workgroup's own to cover them).
  constexpr int XI = (T::P / 4) * T::CI, ZI = T::H * T::HO * C4, XU = (XI + 255) / 256, ZU = (ZI + 255) / 256;
  f32x4 xr[XU], dr[ZU], yr[ZU];
  auto fetch = [&](int n) {
    const float* xn = x + (size_t)n * T::P * T::CI;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = min((int)threadIdx.x + 256 * u, XI - 1), pg = i / T::CI, ci = i % T::CI;
      const float* xp = xn + (4 * pg) * T::CI + ci;
      xr[u] = f32x4{xp[0], xp[T::CI], xp[2 * T::CI], xp[3 * T::CI]};
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int i = min((int)threadIdx.x + 256 * u, ZI - 1);
      const int iy = i / (T::HO * C4), rem = i % (T::HO * C4), ox = rem / C4, c4 = rem % C4;
      const size_t at = (((size_t)n * T::HO + 2 * iy + ky) * T::HO + ox) * C4 + c4;
      dr[u] = ((const f32x4*)dy)[at];
      yr[u] = yact ? ((const f32x4*)yact)[at] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  if (n0 < n1) fetch(n0);
  for (int n = n0; n < n1; ++n) {
    __syncthreads();                                       // the previous image's operands have been read
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < XI) *(f32x4*)(Xt + i * 4) = xr[u];
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < ZI) {
        const int iy = i / (T::HO * C4), rem = i % (T::HO * C4), ox = rem / C4, c4 = rem % C4;
        const f32x4 d = yact ? mask4(dr[u], yr[u]) : dr[u];
        float* z = Zr + iy * T::R + ox * T::S + 4 * c4;    // 8-byte aligned (S and R are even)
        *(float2*)z = float2{d[0], d[1]};
        *(float2*)(z + 2) = float2{d[2], d[3]};
      }
    }
    __syncthreads();
    if (n + 1 < n1) fetch(n + 1);
#pragma unroll
    for (int G = 0; G < T::MT; ++G) {
      const f32x4 bx = *(const f32x4*)(Xt + ((4 * G + q) * T::CI + 16 * wave + j) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int px = 16 * G + 4 * q + e;
        const float* zp = Zr + (px / T::H) * T::R + 2 * (px % T::H) * T::S + j;
#pragma unroll
        for (int kx = 0; kx < T::K; ++kx) {
#pragma unroll
          for (int c = 0; c < T::CT; ++c)
            acc[kx][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(zp[kx * T::S + 16 * c], bx[e], acc[kx][c], 0, 0, 0);
        }
      }
    }
  }
  float* out = slabs + (size_t)chunk * T::KW;
#pragma unroll
  for (int kx = 0; kx < T::K; ++kx) {
#pragma unroll
    for (int c = 0; c < T::CT; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[((ky * T::K + kx) * T::CO + 16 * c + 4 * q + r) * T::CI + 16 * wave + j] = acc[kx][c][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// few-pixel layer, forward only: 3x3x128 -> 8x8x64, 4x4 kernel (dec_deconv2, Conv_model.py:463)
// ------------------------------------------------------------------------------------------------
// With nine input pixels per image the rows of a matrix tile are IMAGES: output pixel (oy, ox) of sixteen images is
// one product of the channels of the 1, 2 or 4 input pixels that reach it with the taps they reach it through.  The
// output pixels of one stride phase (oy % 2, ox % 2) form a 4x4 grid fed by the phase's four taps, tap (u, v) of the
// phase carrying input pixel (iy, ix) to grid point (iy + u, ix + v): a wave owns a 16-channel output tile, holds
// the phase's sixteen accumulator tiles and, tap by tap, the tap's kernel fragments (fetched once per tap and used
// for all nine pixels); the sixteen images' inputs sit in LDS.  Taps in ascending (ky, kx) order per output pixel.
template <int H_, int CI_, int CO_, int K_>
struct FewT {
  static constexpr int H = H_, CI = CI_, CO = CO_, K = K_;
  static constexpr int P = H * H, KG = CI / 16, CT = CO / 16, HO = (H - 1) * 2 + K, NPIX = HO * HO, TAPS = K * K, KW = TAPS * CO * CI;
  static constexpr int GA = H + K / 2 - 1;                 // grid points of a phase per dimension
  static_assert(K % 2 == 0 && CT == 4 && CI % 16 == 0 && HO == 2 * GA, "few-pixel layer shape");
  static constexpr int XS = P * CI + 4;                    // image stride in LDS: sixteen rows on sixteen different bank groups
  static constexpr int FWD_LDS = 16 * XS * 4;
};
using Few = FewT<3, 128, 64, 4>;

// One stride phase of sixteen images for one wave: the grid rows 2 HALF and 2 HALF + 1 of the phase's 4x4 grid of output
// pixels.  Every (input pixel, tap) pair feeds exactly one grid point, so the two halves share no accumulator, each
// gets half of the pairs (18 of 36), and a grid point still receives its taps in ascending order.
template <class T, int HALF>
__device__ __forceinline__ void few_fwd_half(const float* xa, const FragSrc& ws, float bj, int ph, float* __restrict__ y, int n0,
                                             int n_images, int c, int act) {
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int ry = ph >> 1, rx = ph & 1;
  constexpr int HR = T::GA / 2;                            // grid rows per half
  f32x4 acc[HR * T::GA];
#pragma unroll
  for (int i = 0; i < HR * T::GA; ++i) acc[i] = f32x4{bj, bj, bj, bj};
  // one set of kernel fragments: a k-group's fragment of the next tap is requested as soon as the group's products of
  // this tap are issued, and has the other groups' products to arrive under
  f32x4 b[T::KG];
#pragma unroll
  for (int g = 0; g < T::KG; ++g) b[g] = ws.load(((ry * T::K + rx) * T::CT) * T::KG + g);
#pragma unroll
  for (int uv = 0; uv < (T::K / 2) * (T::K / 2); ++uv) {
    const int u = uv / (T::K / 2), v = uv % (T::K / 2);
    const int un = (uv + 1) / (T::K / 2), vn = (uv + 1) % (T::K / 2);
    const int tapn = (ry + 2 * un) * T::K + rx + 2 * vn;
#pragma unroll
    for (int g = 0; g < T::KG; ++g) {
      keep_lds_order();                                    // the inputs are read again for every tap: kept across taps they would spill
#pragma unroll
      for (int p = 0; p < T::P; ++p) {
        const int row = p / T::H + u;
        if (row / HR != HALF) continue;
        const int pos = (row - HALF * HR) * T::GA + p % T::H + v;
        const f32x4 a = *(const f32x4*)(xa + p * T::CI + 16 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[g][m], acc[pos], 0, 0, 0);
      }
      if (uv + 1 < (T::K / 2) * (T::K / 2)) b[g] = ws.load((tapn * T::CT) * T::KG + g);
    }
  }
#pragma unroll
  for (int i = 0; i < HR * T::GA; ++i) {
    const int oy = 2 * (i / T::GA + HALF * HR) + ry, ox = 2 * (i % T::GA) + rx;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int img = n0 + 4 * q + r;
      if (img < n_images) {
        const float v = acc[i][r];
        y[((size_t)img * T::NPIX + oy * T::HO + ox) * T::CO + 16 * c + j] = act ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// grid = (ceil(n / 16), 4 stride phases); eight waves: a 16-channel output tile and one half of the phase's grid rows each.
// One stride phase per workgroup: at a batch of 32 the layer is two image tiles, and with all four phases in one
// workgroup it took 108 us at any batch size (a chain of 4608 products per wave).
template <class T>
__global__ __launch_bounds__(512, 4) void deconv_few_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wff,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int n_images, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = wave_id(), c = wave & 3, half = wave >> 2, q = lane >> 4, j = lane & 15;
  const int n0 = blockIdx.x * 16;
  constexpr int ROW4 = T::P * T::CI / 4;                   // float4 per image
  // all pieces of a thread requested together (a piece per turn was a memory round trip per turn)
  constexpr int SU = (16 * ROW4 + 511) / 512;
  {
    f32x4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = min((int)threadIdx.x + 512 * u, 16 * ROW4 - 1), r = i / ROW4, e = i - r * ROW4;
      const int img = min(n0 + r, n_images - 1);           // rows past the last image repeat it; they are not stored
      v[u] = *(const f32x4*)(x + ((size_t)img * ROW4 + e) * 4);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = threadIdx.x + 512 * u, r = i / ROW4, e = i - r * ROW4;
      if (i < 16 * ROW4) *(f32x4*)(lds + r * T::XS + 4 * e) = v[u];
    }
  }
  __syncthreads();
  FragSrc ws;
  ws.set(wff, c * T::KG);
  const float bj = bias[16 * c + j];
  const float* xa = lds + j * T::XS + 4 * q;
  if (half == 0) few_fwd_half<T, 0>(xa, ws, bj, blockIdx.y, y, n0, n_images, c, act);
  else few_fwd_half<T, 1>(xa, ws, bj, blockIdx.y, y, n0, n_images, c, act);
}

// ------------------------------------------------------------------------------------------------
// narrow layer (one output channel)
// ------------------------------------------------------------------------------------------------
template <int H_, int CI_, int K_>
struct NarrowT {
  static constexpr int H = H_, CI = CI_, K = K_;
  static constexpr int P = H * H, MT = (P + 15) / 16, PP = MT * 16, KG = CI / 16, NTC = CI / 16;
  static constexpr int KK = K * K, NTN = (KK + 15) / 16, KS = KK / 4;      // taps as MFMA columns / as k-steps of 4
  static constexpr int HO = (H - 1) * 2 + K, NO = HO * HO, KW = KK * CI;
  static_assert(CI % 16 == 0 && KK % 4 == 0 && NTC == 2, "narrow layer shape");
  static constexpr int CS = KK + 1;                                        // odd row stride of the cols matrix in LDS
  static constexpr int FWD_LDS = P * CS * 4;
  static constexpr int NTILES = NTN * NTC;                                 // weight-gradient tiles per wave
  static constexpr int XT = PP * CI;                                       // transposed input [PP / 4][CI][4]
  static_assert(4 * NTILES * 256 <= XT, "the cross-wave sum reuses the input's space");
  static constexpr int BWD_LDS = (NO + XT + PP) * 4;
  static constexpr int MAX_CHUNKS = 512;
};
using Last = NarrowT<19, 32, 6>;

template <class N>
__global__ __launch_bounds__(256) void deconv_narrow_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wff,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cols = lds;                                       // [P][CS]
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  const int n = blockIdx.x;
  FragSrc ws;
  ws.set(wff, 0);
  f32x4 b[N::KG][N::NTN];
#pragma unroll
  for (int g = 0; g < N::KG; ++g) {
#pragma unroll
    for (int nt = 0; nt < N::NTN; ++nt) b[g][nt] = ws.load(g * N::NTN + nt);
  }
  const float* xn = x + (size_t)n * N::P * N::CI;
  // (Every row tile of the wave requested before the first product -- one memory round trip instead of six -- ran
  // 93 -> 96 us, twice: this kernel is bound by instruction issue, products and placement, not by those loads.)
  for (int t = wave; t < N::MT; t += 4) {
    const int row = min(16 * t + j, N::P - 1);
    f32x4 a[N::KG];
#pragma unroll
    for (int g = 0; g < N::KG; ++g) a[g] = *(const f32x4*)(xn + row * N::CI + 16 * g + 4 * q);
#pragma unroll
    for (int nt = 0; nt < N::NTN; ++nt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < N::KG; ++g) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][m], b[g][nt][m], acc, 0, 0, 0);
      }
      const int col = 16 * nt + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = 16 * t + 4 * q + r;
        if (px < N::P && col < N::KK) cols[px * N::CS + col] = acc[r];
      }
    }
  }
  __syncthreads();
  // Placement: a thread keeps one output column and walks the rows TPR apart, so the column's three taps (kx = rx,
  // rx + 2, rx + 4 from input columns qx, qx - 1, qx - 2) are worked out once; a tap outside the map reads a clamped
  // address and is left out by a select.  No loop with a data-dependent trip count and no branch: the nine reads of an
  // output are requested together.  Ascending (ky, kx) per output: the order of the cols form.
  const float b0 = bias[0];
  float* yn = y + (size_t)n * N::NO;
  constexpr int TPR = 256 / N::HO, HK = N::K / 2;
  static_assert(N::HO % TPR == 0 && N::K % 2 == 0, "rows of outputs per turn");
  const int ox = threadIdx.x % N::HO, oyb = threadIdx.x / N::HO;
  if (oyb >= TPR) return;
  const int qx = ox >> 1, rx = ox & 1;
  int cofs[HK];
  bool cok[HK];
#pragma unroll
  for (int bb = 0; bb < HK; ++bb) {
    const int ix = qx - bb;
    cok[bb] = (unsigned)ix < (unsigned)N::H;
    cofs[bb] = min(max(ix, 0), N::H - 1) * N::CS + rx + 2 * bb;
  }
#pragma unroll
  for (int i = 0; i < N::HO / TPR; ++i) {
    const int oy = oyb + TPR * i, qy = oy >> 1, ry = oy & 1;
    float v[HK][HK];
    bool ok[HK][HK];
#pragma unroll
    for (int aa = 0; aa < HK; ++aa) {
      const int iy = qy - aa;
      const bool rok = (unsigned)iy < (unsigned)N::H;
      const int rofs = min(max(iy, 0), N::H - 1) * (N::H * N::CS) + (ry + 2 * aa) * N::K;
#pragma unroll
      for (int bb = 0; bb < HK; ++bb) { v[aa][bb] = cols[rofs + cofs[bb]]; ok[aa][bb] = rok && cok[bb]; }
    }
    float acc = b0;
#pragma unroll
    for (int aa = 0; aa < HK; ++aa) {
#pragma unroll
      for (int bb = 0; bb < HK; ++bb) { const float t = acc + v[aa][bb]; acc = ok[aa][bb] ? t : acc; }
    }
    yn[oy * N::HO + ox] = act ? fmaxf(acc, 0.f) : acc;
  }
}

// dX[n][pixel][ci] = sum over taps of dZ[pixel(tap)] W[tap][ci] and slabs[chunk][tap][ci] = sum over the chunk's
// images and pixels of dZ[pixel(tap)] X[pixel][ci].  grid = chunks; the 16-pixel tiles of an image go round the waves.
// Waves of the narrow layer's backward kernel.  (Eight, with registers capped for four waves per SIMD: 158 -> 163 us per
// 4096 images; the 23 row tiles of an image do not divide over eight waves and the kernel is not short of waves.)
#ifndef RLGAN_NBW
#define RLGAN_NBW 4
#endif
constexpr int NBW = RLGAN_NBW;
template <class N, bool DX>
__global__ __launch_bounds__(NBW * 64, NBW / 2) void deconv_narrow_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ yact, const float* __restrict__ wfb,
                                                                 float* __restrict__ dx, float* __restrict__ slabs,
                                                                 int n_images, int per_chunk, int dx_relu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Dimg = lds;                                       // [NO] dZ of the image
  float* Xt = lds + N::NO;                                 // [PP / 4][CI][4 pixels], zero beyond the image
  int* pixtab = (int*)(lds + N::NO + N::XT);               // [PP] output pixel of tap (0, 0) per input pixel
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  const int chunk = blockIdx.x;
  const int n0 = chunk * per_chunk, n1 = min(n_images, n0 + per_chunk);
  for (int p = threadIdx.x; p < N::PP; p += NBW * 64) {
    const int px = min(p, N::P - 1);
    pixtab[p] = (2 * (px / N::H)) * N::HO + 2 * (px % N::H);
  }
  int to[N::NTN], td[N::KS];
#pragma unroll
  for (int tt = 0; tt < N::NTN; ++tt) { const int tap = min(16 * tt + j, N::KK - 1); to[tt] = (tap / N::K) * N::HO + tap % N::K; }
#pragma unroll
  for (int e = 0; e < N::KS; ++e) { const int tap = 4 * e + q; td[e] = (tap / N::K) * N::HO + tap % N::K; }
  float bw[N::KS][N::NTC];
  if (DX) {
#pragma unroll
    for (int e = 0; e < N::KS; ++e) {
#pragma unroll
      for (int nt = 0; nt < N::NTC; ++nt) bw[e][nt] = wfb[(e * N::NTC + nt) * 64 + lane];
    }
  }
  f32x4 accw[N::NTN][N::NTC];
#pragma unroll
  for (int tt = 0; tt < N::NTN; ++tt) {
#pragma unroll
    for (int nt = 0; nt < N::NTC; ++nt) accw[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // An image's operands travel memory -> registers while the previous image is multiplied out of LDS, every piece of
  // them requested at once (a piece per loop turn was a memory round trip per turn, eighteen per image).
  constexpr int NTH = NBW * 64, DU = (N::NO + NTH - 1) / NTH, XI = (N::PP / 4) * N::CI, XU = (XI + NTH - 1) / NTH;
  float dreg[DU], yreg[DU];
  f32x4 xreg[XU];
  auto fetch = [&](int n) {
    const float* dyn = dy + (size_t)n * N::NO;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int o = min((int)threadIdx.x + NTH * u, N::NO - 1);
      dreg[u] = dyn[o];
      yreg[u] = yact ? yact[(size_t)n * N::NO + o] : 1.f;
    }
    const float* xn = x + (size_t)n * N::P * N::CI;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = min((int)threadIdx.x + NTH * u, XI - 1), pg = i / N::CI, ci = i % N::CI;
#pragma unroll
      for (int e = 0; e < 4; ++e) xreg[u][e] = xn[min(4 * pg + e, N::P - 1) * N::CI + ci];
    }
  };
  if (n0 < n1) fetch(n0);
  for (int n = n0; n < n1; ++n) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int o = threadIdx.x + NTH * u;
      if (o < N::NO) Dimg[o] = yact ? dreg[u] * relu_d(yreg[u]) : dreg[u];
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = threadIdx.x + NTH * u;
      if (i < XI) {
        const int pg = i / N::CI;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * pg + e < N::P ? xreg[u][e] : 0.f;
        *(f32x4*)(Xt + i * 4) = v;
      }
    }
    __syncthreads();
    if (n + 1 < n1) fetch(n + 1);
    for (int t = wave; t < N::MT; t += NBW) {
      // this lane's input values: pixels 16 t + 4 q .. + 3 of channels 16 nt + j (the weight gradient's operand, and
      // the ReLU' factor of the input gradient when the layer below is to receive its dZ)
      f32x4 bx[N::NTC];
#pragma unroll
      for (int nt = 0; nt < N::NTC; ++nt) bx[nt] = *(const f32x4*)(Xt + ((4 * t + q) * N::CI + 16 * nt + j) * 4);
      if (DX) {
        const int pb = pixtab[16 * t + j];
        f32x4 accx[N::NTC];
#pragma unroll
        for (int nt = 0; nt < N::NTC; ++nt) accx[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < N::KS; ++e) {
          const float av = Dimg[pb + td[e]];
#pragma unroll
          for (int nt = 0; nt < N::NTC; ++nt) accx[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[e][nt], accx[nt], 0, 0, 0);
        }
        float* dxn = dx + (size_t)n * N::P * N::CI + j;
#pragma unroll
        for (int nt = 0; nt < N::NTC; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int px = 16 * t + 4 * q + r;
            if (px < N::P) dxn[px * N::CI + 16 * nt] = dx_relu ? accx[nt][r] * relu_d(bx[nt][r]) : accx[nt][r];
          }
        }
      }
      const i32x4 pt = *(const i32x4*)(pixtab + 16 * t + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int tt = 0; tt < N::NTN; ++tt) {
          const float av = Dimg[pt[e] + to[tt]];
#pragma unroll
          for (int nt = 0; nt < N::NTC; ++nt) accw[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bx[nt][e], accw[tt][nt], 0, 0, 0);
        }
      }
    }
  }
  // the waves' partial kernels, added in wave order
  __syncthreads();
  float* red = lds;                                        // [NBW waves][NTILES][64 lanes][4], over the image and the input
  static_assert(NBW * N::NTILES * 256 <= N::NO + N::XT, "the cross-wave sum reuses the operands' space");
#pragma unroll
  for (int tt = 0; tt < N::NTN; ++tt) {
#pragma unroll
    for (int nt = 0; nt < N::NTC; ++nt) *(f32x4*)(red + ((wave * N::NTILES + tt * N::NTC + nt) * 64 + lane) * 4) = accw[tt][nt];
  }
  __syncthreads();
  float* out = slabs + (size_t)chunk * N::KW;
  for (int i = threadIdx.x; i < N::NTILES * 256; i += NTH) {
    const int tile = i >> 8, el = i & 255, ln = el >> 2, r = el & 3;
    const int tt = tile / N::NTC, nt = tile % N::NTC;
    const int tap = 16 * tt + 4 * (ln >> 4) + r, ci = 16 * nt + (ln & 15);
    if (tap < N::KK) {
      const float* p = red + tile * 256 + el;
      float tot = p[0];
#pragma unroll
      for (int w = 1; w < NBW; ++w) tot += p[w * N::NTILES * 256];
      out[tap * N::CI + ci] = tot;
    }
  }
}

// grad[p] = sum over chunks of slabs[chunk][p], chunks in order, 32 loads in flight (no branch around a load)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int n_chunks, int n, float* __restrict__ grad) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  constexpr int FL = 32;
  float tot = 0.f;
  for (int c0 = 0; c0 < n_chunks; c0 += FL) {
    float v[FL];
#pragma unroll
    for (int i = 0; i < FL; ++i) v[i] = slabs[(size_t)min(c0 + i, n_chunks - 1) * n + p];
#pragma unroll
    for (int i = 0; i < FL; ++i) tot += c0 + i < n_chunks ? v[i] : 0.f;
  }
  grad[p] = tot;
}

enum { FUSED_NONE = 0, FUSED_TILE = 1, FUSED_NARROW = 2, FUSED_FEW_FWD = 3 };

inline int fused_kind(const rlgan_deconv_t* d) {
  if (!d || d->h != d->w || d->stride != 2) return FUSED_NONE;
  if (d->h == Mid::H && d->c_in == Mid::CI && d->c_out == Mid::CO && d->k == Mid::K) return FUSED_TILE;
  if (d->h == Last::H && d->c_in == Last::CI && d->c_out == 1 && d->k == Last::K) return FUSED_NARROW;
  if (d->h == Few::H && d->c_in == Few::CI && d->c_out == Few::CO && d->k == Few::K) return FUSED_FEW_FWD;
  return FUSED_NONE;
}

inline void chunking(int n_images, int max_chunks, int* per_chunk, int* n_chunks) {
  *per_chunk = (n_images + max_chunks - 1) / max_chunks;
  *n_chunks = (n_images + *per_chunk - 1) / *per_chunk;
}

template <class KernelT>
inline void allow_lds(KernelT k, int bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

extern "C" {

int rlgan_deconv_fused_geometry(const rlgan_deconv_t* d, rlgan_deconv_fused_geom_t* out) {
  if (!d || !out) return RLGAN_E_ARG;
  const int kind = fused_kind(d);
  out->kind = kind;
  out->wff_floats = out->wfb_floats = out->kernel_floats = out->max_chunks = out->bias_part_floats_per_image = 0;
  if (kind == FUSED_TILE) {
    out->wff_floats = out->wfb_floats = out->kernel_floats = Mid::KW;
    out->max_chunks = Mid::MAX_CHUNKS;
    out->bias_part_floats_per_image = Mid::CO;
  } else if (kind == FUSED_FEW_FWD) {
    out->wff_floats = out->kernel_floats = Few::KW;        // forward only: no wfb, no workspace
  } else if (kind == FUSED_NARROW) {
    out->wff_floats = Last::KG * Last::NTN * 256;
    out->wfb_floats = Last::KS * Last::NTC * 64;
    out->kernel_floats = Last::KW;
    out->max_chunks = Last::MAX_CHUNKS;
  }
  return 0;
}

int64_t rlgan_deconv_fused_workspace(const rlgan_deconv_t* d, int n_images) {
  const int kind = fused_kind(d);
  if (kind == FUSED_NONE || kind == FUSED_FEW_FWD || n_images <= 0) return 0;
  int per, nch;
  if (kind == FUSED_TILE) {
    chunking(n_images, Mid::MAX_CHUNKS, &per, &nch);
    return (int64_t)nch * Mid::KW + (int64_t)n_images * Mid::CO + 512 * Mid::CO;
  }
  chunking(n_images, Last::MAX_CHUNKS, &per, &nch);
  return (int64_t)nch * Last::KW + 512;
}

int rlgan_deconv_fused_build_maps(const rlgan_deconv_t* d, int32_t* map_ff, int32_t* map_fb) {
  if (!d || !map_ff) return RLGAN_E_ARG;
  const int kind = fused_kind(d);
  if (kind == FUSED_NONE) return RLGAN_E_DIMS;
  if (!map_fb && kind != FUSED_FEW_FWD) return RLGAN_E_ARG;
  const int taps = d->k * d->k, CO = d->c_out, CI = d->c_in;
  int p = 0;
  for (int tap = 0; tap < taps; ++tap)
    for (int co = 0; co < CO; ++co)
      for (int ci = 0; ci < CI; ++ci, ++p) {
        if (kind == FUSED_FEW_FWD) {
          map_ff[p] = (((tap * Few::CT + co / 16) * Few::KG + ci / 16) * 64 + 16 * ((ci % 16) / 4) + co % 16) * 4 + ci % 4;
        } else if (kind == FUSED_TILE) {
          // forward: block (tap, channel tile, k-group of input channels); backward: block (input tile, tap, group of output channels)
          map_ff[p] = (((tap * Mid::CT + co / 16) * Mid::KG + ci / 16) * 64 + 16 * ((ci % 16) / 4) + co % 16) * 4 + ci % 4;
          map_fb[p] = ((((ci / 16) * Mid::TAPS + tap) * Mid::KGO + co / 16) * 64 + 16 * ((co % 16) / 4) + ci % 16) * 4 + co % 4;
        } else {
          map_ff[p] = frag_offset(Last::NTN, ci, tap);
          map_fb[p] = ((tap / 4) * Last::NTC + ci / 16) * 64 + 16 * (tap % 4) + ci % 16;
        }
      }
  for (int co = 0; co < CO; ++co, ++p) {                                // the bias is read from the canonical vector
    map_ff[p] = -1;
    if (map_fb) map_fb[p] = -1;
  }
  return 0;
}

int rlgan_deconv_fused_forward(const rlgan_deconv_t* d, const float* x, const float* wff, const float* bias, float* y,
                               int n_images, int act_relu, void* stream) {
  const int kind = fused_kind(d);
  if (kind == FUSED_NONE) return RLGAN_E_DIMS;
  if (!x || !wff || !bias || !y || n_images <= 0) return RLGAN_E_ARG;
  if (((uintptr_t)x | (uintptr_t)wff | (uintptr_t)bias | (uintptr_t)y) & 15) return RLGAN_E_ALIGN;
  if (kind == FUSED_TILE) {
    allow_lds(deconv_tile_fwd_kernel<Mid, 2>, Mid::FWD_LDS);
    hipLaunchKernelGGL((deconv_tile_fwd_kernel<Mid, 2>), dim3(cdiv(n_images, Mid::IPW)), dim3(512), Mid::FWD_LDS, (hipStream_t)stream,
                       x, wff, bias, y, n_images, act_relu);
  } else if (kind == FUSED_FEW_FWD) {
    allow_lds(deconv_few_fwd_kernel<Few>, Few::FWD_LDS);
    hipLaunchKernelGGL(deconv_few_fwd_kernel<Few>, dim3(cdiv(n_images, 16), 4), dim3(512), Few::FWD_LDS, (hipStream_t)stream,
                       x, wff, bias, y, n_images, act_relu);
  } else {
    allow_lds(deconv_narrow_fwd_kernel<Last>, Last::FWD_LDS);
    hipLaunchKernelGGL(deconv_narrow_fwd_kernel<Last>, dim3(n_images), dim3(256), Last::FWD_LDS, (hipStream_t)stream,
                       x, wff, bias, y, act_relu);
  }
  return launched();
}

int rlgan_deconv_fused_backward(const rlgan_deconv_t* d, const float* x, const float* y_relu, const float* dy, const float* wfb,
                                float* dx, float* workspace, float* grad, int n_images, void* stream) {
  return rlgan_deconv_fused_backward_dz(d, x, y_relu, dy, wfb, dx, 0, workspace, grad, n_images, stream);
}

int rlgan_deconv_fused_backward_dz(const rlgan_deconv_t* d, const float* x, const float* y_relu, const float* dy, const float* wfb,
                                   float* dx, int dx_relu, float* workspace, float* grad, int n_images, void* stream) {
  const int kind = fused_kind(d);
  if (kind == FUSED_NONE || kind == FUSED_FEW_FWD) return RLGAN_E_DIMS;
  if (dx_relu && !dx) return RLGAN_E_ARG;
  if (!x || !dy || !wfb || !workspace || !grad || n_images <= 0) return RLGAN_E_ARG;
  if (((uintptr_t)x | (uintptr_t)y_relu | (uintptr_t)dy | (uintptr_t)wfb | (uintptr_t)dx | (uintptr_t)workspace | (uintptr_t)grad) & 15)
    return RLGAN_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  int per, nch;
  if (kind == FUSED_TILE) {
    chunking(n_images, Mid::MAX_CHUNKS, &per, &nch);
    float* slabs = workspace;
    float* bias_part = slabs + (size_t)nch * Mid::KW;
    float* colsum_scratch = bias_part + (size_t)n_images * Mid::CO;
    hipLaunchKernelGGL(deconv_tile_dw_kernel<Mid>, dim3(Mid::K, nch), dim3(256), Mid::DW_LDS, st, x, dy, y_relu, slabs, n_images, per);
    if (const int rc = launched()) return rc;
    hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(Mid::KW, 256)), dim3(256), 0, st, slabs, nch, Mid::KW, grad);
    if (const int rc = launched()) return rc;
    if (dx) {
      hipLaunchKernelGGL(deconv_tile_dx_kernel<Mid>, dim3(n_images), dim3(256), Mid::DX_LDS, st, dy, y_relu, wfb, dx, bias_part,
                         dx_relu ? x : nullptr);
      if (const int rc = launched()) return rc;
      return rlgan_column_sums(bias_part, nullptr, n_images, Mid::CO, colsum_scratch, grad + Mid::KW, stream);
    }
    return rlgan_column_sums(dy, y_relu, (int64_t)n_images * Mid::NPIX, Mid::CO, colsum_scratch, grad + Mid::KW, stream);
  }
  chunking(n_images, Last::MAX_CHUNKS, &per, &nch);
  float* slabs = workspace;
  float* colsum_scratch = slabs + (size_t)nch * Last::KW;
  if (dx) {
    hipLaunchKernelGGL((deconv_narrow_bwd_kernel<Last, true>), dim3(nch), dim3(NBW * 64), Last::BWD_LDS, st, x, dy, y_relu, wfb, dx, slabs,
                       n_images, per, dx_relu);
  } else {
    hipLaunchKernelGGL((deconv_narrow_bwd_kernel<Last, false>), dim3(nch), dim3(NBW * 64), Last::BWD_LDS, st, x, dy, y_relu, wfb, dx, slabs,
                       n_images, per, 0);
  }
  if (const int rc = launched()) return rc;
  hipLaunchKernelGGL(slab_sum_kernel, dim3(cdiv(Last::KW, 256)), dim3(256), 0, st, slabs, nch, Last::KW, grad);
  if (const int rc = launched()) return rc;
  return rlgan_column_sums(dy, y_relu, (int64_t)n_images * Last::NO, 1, colsum_scratch, grad + Last::KW, stream);
}

}  // extern "C"
