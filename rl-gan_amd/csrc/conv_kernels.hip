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
#include "fastdiv.h"
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

struct ConvG {
  int H, W, Cin, Cout, k, s;      // input height/width/channels, output channels, kernel size, stride
  int Ho, Wo, K, NG, NT, rowlen;  // output size, patch length, k-groups (of 16, incl. bias row), column tiles, k*Cin
  FastDiv dHoWo, dWo, drow;       // divisors Ho*Wo, Wo and rowlen
};

inline bool make_conv(const rlgan_conv_t* c, ConvG& g) {
  if (!c || c->h <= 0 || c->w <= 0 || c->c_in <= 0 || c->c_out <= 0 || c->k <= 0 || c->stride <= 0) return false;
  if (c->c_in % 4 != 0 || c->c_out % 4 != 0 || c->c_out > 64 || c->k > c->h || c->k > c->w) return false;   // 16-byte segments, <= 4 column tiles
  g.H = c->h; g.W = c->w; g.Cin = c->c_in; g.Cout = c->c_out; g.k = c->k; g.s = c->stride;
  g.Ho = (c->h - c->k) / c->stride + 1; g.Wo = (c->w - c->k) / c->stride + 1;
  g.K = c->k * c->k * c->c_in; g.NG = cdiv(g.K + 1, 16); g.NT = cdiv(c->c_out, 16); g.rowlen = c->k * c->c_in;
  if ((long long)c->h * c->w * c->c_in > 0x7fffffffll) return false;
  g.dHoWo = make_div(g.Ho * g.Wo); g.dWo = make_div(g.Wo); g.drow = make_div(g.rowlen);
  return true;
}

__device__ __forceinline__ float elu_d(float y) { return y > 0.f ? 1.f : y + 1.f; }   // ELU' from the layer output

// Where a layer's input comes from.  `image(b)` = offset of image b in elements (negative: no such image, its
// pixels read as zero); `load(base, e)` fetches elements e..e+3 of that image and `value(raw, base)` turns what
// was fetched into float32.  The two are separate so that a kernel can keep several fetches in flight and only
// touch their results when it multiplies them; for the same reason `load` has no branch around its memory
// access (a wave would wait for every load where the branches join).
struct SrcF32 {                       // float32 activations [n][H][W][Cin]
  using Raw = f32x4;
  const float* x;
  size_t img;
  __device__ __forceinline__ long long image(int b) const { return (long long)((size_t)b * img); }
  __device__ __forceinline__ Raw load(long long base, size_t e) const { return *(const f32x4*)(x + base + e); }
  // the same fetch for data read once and never again (kept out of the way of what the caches hold for others)
  __device__ __forceinline__ Raw load_stream(long long base, size_t e) const {
    return __builtin_nontemporal_load((const f32x4*)(x + base + e));
  }
  __device__ __forceinline__ f32x4 value(Raw r, long long) const { return r; }
  const float* floats() const { return x; }            // host side: the plain float32 array
};
struct SrcU8 {                        // uint8 frame ring [slots][H*W*Cin]; image b = slot idx[b] (idx null: slot b); value = byte * scale
  using Raw = uint32_t;
  const uint8_t* ring;
  const int64_t* idx;
  long long slots;
  size_t img;
  float scale;
  __device__ __forceinline__ long long image(int b) const {
    const long long slot = idx ? idx[b] : b;
    return slot >= 0 && slot < slots ? (long long)((size_t)slot * img) : -1;
  }
  __device__ __forceinline__ Raw load(long long base, size_t e) const { return *(const uint32_t*)(ring + (base < 0 ? 0 : base) + e); }
  __device__ __forceinline__ Raw load_stream(long long base, size_t e) const { return load(base, e); }
  __device__ __forceinline__ f32x4 value(Raw v, long long base) const {
    const float sc = base < 0 ? 0.f : scale;
    return f32x4{(float)(v & 255u) * sc, (float)((v >> 8) & 255u) * sc, (float)((v >> 16) & 255u) * sc, (float)(v >> 24) * sc};
  }
  const float* floats() const { return nullptr; }
};

// Patch element table: entry q (patch elements 4q..4q+3; rowlen % 4 == 0, so the four never straddle a kernel
// row) = their offset from the patch's first element in the image, -1 for the constant 1 of the bias row
// (element K), -2 for the zeros after it.  Filled by all threads of a workgroup; a barrier must follow.
__device__ __forceinline__ void fill_ktab(const ConvG& g, int* ktab) {
  for (int q = threadIdx.x; q < g.NG * 4; q += blockDim.x) {
    const int k0 = q * 4, ky = k0 / g.rowlen;
    ktab[q] = k0 > g.K ? -2 : k0 == g.K ? -1 : ky * g.W * g.Cin + (k0 - ky * g.rowlen);
  }
}
// the four patch elements of table entry `koff` for the patch whose first element is `pix` of the image at `base`:
// fetch (always from inside the patch) and, later, the value
template <class S>
__device__ __forceinline__ typename S::Raw patch_load(const S& src, long long base, int pix, int koff) {
  return src.load(base, (size_t)(pix + max(koff, 0)));
}
template <class S>
__device__ __forceinline__ f32x4 patch_value(const S& src, typename S::Raw r, long long base, int koff) {
  const f32x4 v = src.value(r, base);
  return koff >= 0 ? v : f32x4{koff == -1 ? 1.f : 0.f, 0.f, 0.f, 0.f};
}
// image offset and first patch element of output pixel `prow` (< n_images * Ho * Wo)
struct Pos { long long base; int pix; };
template <class S>
__device__ __forceinline__ Pos locate(const ConvG& g, const S& src, int prow) {
  const int b = fdiv(prow, g.dHoWo), r = prow - b * (g.Ho * g.Wo), oy = fdiv(r, g.dWo), ox = r - oy * g.Wo;
  return Pos{src.image(b), (oy * g.W + ox) * (g.s * g.Cin)};
}

// ---------------------------------------------------------------- forward
// A workgroup keeps the layer's weight fragments in LDS and walks a contiguous range of 16-pixel tiles, its four
// waves taking every fourth tile.  The weight fragment is the row operand and the patch (straight from global
// memory in MFMA layout: lane = (pixel l&15, k-quad l>>4)) the column operand, so a lane ends up with four
// consecutive channels of one pixel and writes them as one 16-byte store.  Patches are fetched PF k-groups at
// a time, the next PF (of this tile or of the wave's next one) while the current ones are multiplied.
template <int NT, class S>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvG g, const S src, const float* __restrict__ wfrag,
                                                       float* __restrict__ y, int M, int act_elu) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  constexpr int PF = 4;
  const int nfrag = g.NG * NT * 256;
  int* ktab = (int*)(wlds + nfrag);
  for (int e = threadIdx.x * 4; e < nfrag; e += 1024) *(f32x4*)(wlds + e) = *(const f32x4*)(wfrag + e);
  fill_ktab(g, ktab);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (M + 15) / 16, per_wg = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_end = min(ntiles, ((int)blockIdx.x + 1) * per_wg);
  int tile = blockIdx.x * per_wg + wave;
  if (tile >= t_end) return;                               // no barrier follows
  const int nchunks = (g.NG + PF - 1) / PF;
  const int* kt = ktab + (lane >> 4);
  const float* wl = wlds + lane * 4;
  auto at = [&](int t) { return locate(g, src, min(t * 16 + (lane & 15), M - 1)); };
  using Raw = typename S::Raw;
  auto fetch = [&](const Pos& ps, int g0, Raw (&aa)[PF], int (&kk)[PF]) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {                          // groups past the end repeat the last one and are not multiplied
      kk[j] = kt[4 * min(g0 + j, g.NG - 1)];
      aa[j] = patch_load(src, ps.base, ps.pix, kk[j]);
    }
  };
  Pos cur = at(tile);
  Raw a[PF];
  int ka[PF];
  f32x4 acc[NT];
  fetch(cur, 0, a, ka);
  int chunk = 0;
  for (;;) {
    int ntile = tile, nchunk = chunk + 1;
    Pos nxt = cur;
    if (nchunk == nchunks) {
      nchunk = 0;
      ntile = tile + 4;
      if (ntile < t_end) nxt = at(ntile);
    }
    const bool more = ntile < t_end;
    Raw an[PF];
    int kn[PF];
    if (more) fetch(nxt, nchunk * PF, an, kn);
    if (chunk == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int gi = chunk * PF + j;
      if (gi < g.NG) {
        f32x4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = *(const f32x4*)(wl + (gi * NT + t) * 256);
        const f32x4 av = patch_value(src, a[j], cur.base, ka[j]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m], av[m], acc[t], 0, 0, 0);
        }
      }
    }
    if (chunk == nchunks - 1) {
      // result fragment: rows = channels 16t + 4*(lane>>4) + i, column = pixel lane&15
      const int p = tile * 16 + (lane & 15);
      if (p < M) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n0 = 16 * t + (lane >> 4) * 4;
          if (n0 >= g.Cout) continue;                       // Cout % 4 == 0: four channels are in range together
          f32x4 v = acc[t];
          if (act_elu) v = f32x4{elu_f(v[0]), elu_f(v[1]), elu_f(v[2]), elu_f(v[3])};
          *(f32x4*)(y + (size_t)p * g.Cout + n0) = v;
        }
      }
    }
    if (!more) break;
#pragma unroll
    for (int j = 0; j < PF; ++j) { a[j] = an[j]; ka[j] = kn[j]; }
    cur = nxt; tile = ntile; chunk = nchunk;
  }
}

// The same forward pass for float32 inputs with the number of k-groups known at compile time.  conv_fwd_kernel spends
// more vector-ALU instructions than MFMAs on a layer with many k-groups (counted on 32 -> 32 channels, 3x3: 650 against
// 152 per 16-pixel tile and wave: a table read, a clamp, a 64-bit address and five selects per group, register copies
// per chunk), and on gfx950 those do not overlap with fp32 MFMAs.  Here every group's patch offset sits in a register
// for the whole kernel, a fetch is one 32-bit add and a buffer load (scalar base, so the input must span less than
// 2 GiB), the groups are unrolled with two register sets taking turns (NCH, the number of chunks of PF groups, is even,
// so the turn of a tile's first chunk is always the same), and only the last group, which holds the bias row's 1 and
// the zeros after it, selects.
template <int NT, int NG, int PF>
__global__ __launch_bounds__(256) void conv_fwd_static_kernel(const ConvG g, const float* __restrict__ x, const float* __restrict__ wfrag,
                                                              float* __restrict__ y, int M, int act_elu) {
  constexpr int NCH = (NG + PF - 1) / PF;
  static_assert(NCH % 2 == 0, "two register sets take turns: an even number of chunks per tile");
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  constexpr int nfrag = NG * NT * 256;
  for (int e = threadIdx.x * 4; e < nfrag; e += 1024) *(f32x4*)(wlds + e) = *(const f32x4*)(wfrag + e);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  int koff[NG];                                           // byte offset of this lane's quad of group gi from the patch's first element
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int k0 = min((4 * gi + q) * 4, g.K - 4), ky = k0 / g.rowlen;      // quads past the patch: clamped, replaced below
    koff[gi] = (ky * g.W * g.Cin + (k0 - ky * g.rowlen)) * 4;
  }
  const int kl = (4 * (NG - 1) + q) * 4;                  // the last group's quad: data, the bias row's 1, or zeros
  const bool last_data = kl < g.K;
  const float last_one = kl == g.K ? 1.f : 0.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7fffffff, 0x00020000);
  const int ntiles = (M + 15) / 16, per_wg = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_end = min(ntiles, ((int)blockIdx.x + 1) * per_wg);
  int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * per_wg + wave);   // (the first wave turning with the workgroup, so that a share of 6 + 5 + 5 + 5 tiles does not always favour one SIMD: no change)
  if (tile >= t_end) return;                              // no barrier follows
  const float* wl = wlds + lane * 4;
  const int img_bytes = g.H * g.W * g.Cin * 4;
  auto first = [&](int t) {                               // byte offset of the patch of this lane's pixel of tile t
    const int p = min(t * 16 + j, M - 1);
    const int b = fdiv(p, g.dHoWo), r = p - b * (g.Ho * g.Wo), oy = fdiv(r, g.dWo), ox = r - oy * g.Wo;
    return b * img_bytes + (oy * g.W + ox) * (g.s * g.Cin * 4);
  };
  f32x4 a[2][PF];
  auto fetch = [&](int at, int c, f32x4 (&dst)[PF]) {     // chunk c (a compile-time constant after unrolling) of the patch at `at`
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int gi = min(c * PF + u, NG - 1);
      dst[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, at + koff[gi], 0, 0));
    }
  };
  int cur = first(tile);
  fetch(cur, 0, a[0]);
  for (;;) {
    const int ntile = tile + 4;
    const bool more = ntile < t_end;                      // uniform: tile is a scalar
    const int nxt = first(min(ntile, t_end - 1));         // always computed and fetched from (a clamped tile at the end): no branch around loads
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) fetch(cur, c + 1, a[(c + 1) & 1]);
      else fetch(nxt, 0, a[0]);
      __builtin_amdgcn_sched_barrier(0);                  // keep the fetches ahead of this chunk's products
      // (the weight fragments of group gi + 1 read from LDS before the products of group gi, order pinned: no change --
      // with four waves per SIMD the other waves cover a group's LDS latency)
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int gi = c * PF + u;
        if (gi < NG) {
          f32x4 w[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) w[t] = *(const f32x4*)(wl + (gi * NT + t) * 256);
          f32x4 av = a[c & 1][u];
          if (gi == NG - 1) av = last_data ? av : f32x4{last_one, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m], av[m], acc[t], 0, 0, 0);
          }
        }
      }
    }
    const int p = tile * 16 + j;
    if (p < M) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n0 = 16 * t + q * 4;
        if (n0 >= g.Cout) continue;
        f32x4 v = acc[t];
        if (act_elu) v = f32x4{elu_f(v[0]), elu_f(v[1]), elu_f(v[2]), elu_f(v[3])};
        *(f32x4*)(y + (size_t)p * g.Cout + n0) = v;
      }
    }
    if (!more) break;
    cur = nxt; tile = ntile;
  }
}

// The same products in the same order for a layer with few output pixels in all (the last layer of an encoder: a 1x1
// or 2x2 map, where a whole batch is a few hundred 16-pixel tiles).  conv_fwd_static_kernel would spend its time on a
// copy of all weight fragments into LDS per workgroup and on a chain of memory round trips per tile (a chunk of groups
// at a time), with one tile per wave to hide them behind and most compute units idle.  Here a wave owns one tile and
// NTW of the layer's column tiles and asks for every patch quad and weight fragment it needs before the first product:
// one round trip, no LDS, no barrier, NT / NTW times as many waves.  One accumulator per column tile, groups in order:
// the bits conv_fwd_static_kernel produces.  (Measured at 4096 images: 32 -> 64 channels on a 4x4 map 9.7 -> 5.2 us;
// at 4096 tiles, the layer before it, the fragments fetched per wave cost more than the chain saves.)
template <int NT, int NTW, int NG>
__global__ __launch_bounds__(256) void conv_fwd_once_kernel(const ConvG g, const float* __restrict__ x, const float* __restrict__ wfrag,
                                                            float* __restrict__ y, int M, int act_elu) {
  constexpr int TS = NT / NTW;                            // waves that share a tile, each with its own column tiles
  static_assert(NT % NTW == 0, "column tiles divide evenly");
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  const int tile = wv / TS, t0 = (wv % TS) * NTW;
  if (tile * 16 >= M) return;                             // no barrier anywhere
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7fffffff, 0x00020000);
  const int p = min(tile * 16 + j, M - 1);
  const int b = fdiv(p, g.dHoWo), r = p - b * (g.Ho * g.Wo), oy = fdiv(r, g.dWo), ox = r - oy * g.Wo;
  const int at = b * (g.H * g.W * g.Cin * 4) + (oy * g.W + ox) * (g.s * g.Cin * 4);
  f32x4 a[NG], w[NG][NTW];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int k0 = min((4 * gi + q) * 4, g.K - 4), ky = fdiv(k0, g.drow);      // quads past the patch: clamped, replaced below
    a[gi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, at + (ky * g.W * g.Cin + (k0 - ky * g.rowlen)) * 4, 0, 0));
  }
  const float* wl = wfrag + lane * 4;
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) w[gi][t] = *(const f32x4*)(wl + (gi * NT + t0 + t) * 256);
  }
  __builtin_amdgcn_sched_barrier(0);                      // every load above is on its way before anything waits for one
  const int kl = (4 * (NG - 1) + q) * 4;                  // the last group's quad: data, the bias row's 1, or zeros
  if (kl >= g.K) a[NG - 1] = f32x4{kl == g.K ? 1.f : 0.f, 0.f, 0.f, 0.f};
  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[gi][t][m], a[gi][m], acc[t], 0, 0, 0);
    }
  }
  if (tile * 16 + j < M) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int n0 = 16 * (t0 + t) + q * 4;
      if (n0 >= g.Cout) continue;
      f32x4 v = acc[t];
      if (act_elu) v = f32x4{elu_f(v[0]), elu_f(v[1]), elu_f(v[2]), elu_f(v[3])};
      *(f32x4*)(y + (size_t)p * g.Cout + n0) = v;
    }
  }
}

// Forward pass of a first layer with four input channels: a workgroup stages a band of input rows of one image in
// LDS with coalesced loads (the rows that `band_h` output rows need; a small image is one band) and gathers the patch
// quads from there (a quad is one input pixel's four channels: one ds_read_b128 per k-group); the layer's weight
// fragments (at most MAXG k-groups) stay in registers.  conv_fwd_kernel fetches every quad from memory, lanes 16 * stride
// bytes apart, each input pixel about k*k / (s*s) times.  grid = (images, bands).
template <int NT, int MAXG, class S>
__global__ __launch_bounds__(256) void conv_fwd_c4_kernel(const ConvG g, const S src, const float* __restrict__ wfrag,
                                                          float* __restrict__ y, int band_h, int act_elu) {
  extern __shared__ __attribute__((aligned(16))) float X[];      // [band rows * W][4]
  constexpr int STAGE_U = 4;
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  const int n = blockIdx.x, oy0 = blockIdx.y * band_h, oh = min(band_h, g.Ho - oy0);
  const int in0 = (g.s * oy0) * g.W, nin = ((oh - 1) * g.s + g.k) * g.W, npix = oh * g.Wo;   // input pixels staged, output pixels made
  const long long base = src.image(n);
  for (int i0 = threadIdx.x; i0 < nin; i0 += 256 * STAGE_U) {
    typename S::Raw raw[STAGE_U];
#pragma unroll
    for (int u = 0; u < STAGE_U; ++u) raw[u] = src.load(base, (size_t)4 * (in0 + min(i0 + 256 * u, nin - 1)));
#pragma unroll
    for (int u = 0; u < STAGE_U; ++u) {
      const int i = i0 + 256 * u;
      if (i < nin) *(f32x4*)(X + 4 * i) = src.value(raw[u], base);
    }
  }
  f32x4 w[MAXG][NT];
  int toff[MAXG];
  float live[MAXG], one[MAXG];
#pragma unroll
  for (int gi = 0; gi < MAXG; ++gi) {
#pragma unroll
    for (int t = 0; t < NT; ++t) w[gi][t] = gi < g.NG ? *(const f32x4*)(wfrag + (gi * NT + t) * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int tap = 4 * gi + q, k0 = 4 * tap;               // this lane's quad of the group: one tap's four channels,
    const bool data = k0 < g.K;                             // the bias row's constant 1, or the zeros after it
    const int ky = data ? tap / g.k : 0, kx = data ? tap - ky * g.k : 0;
    toff[gi] = (ky * g.W + kx) * 4;
    live[gi] = data ? 1.f : 0.f;
    one[gi] = k0 == g.K ? 1.f : 0.f;
  }
  __syncthreads();
  float* yn = y + ((size_t)n * g.Ho * g.Wo + (size_t)oy0 * g.Wo) * g.Cout;
  for (int t = wave; t < (npix + 15) / 16; t += 4) {
    const int p = min(16 * t + j, npix - 1), oy = fdiv(p, g.dWo), ox = p - oy * g.Wo;
    const float* xp = X + ((g.s * oy) * g.W + g.s * ox) * 4;
    f32x4 b[MAXG];
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
      b[gi] = *(const f32x4*)(xp + toff[gi]) * live[gi];
      b[gi][0] += one[gi];
    }
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
      if (gi < g.NG) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[gi][i][m], b[gi][m], acc[i], 0, 0, 0);
        }
      }
    }
    if (16 * t + j < npix) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n0 = 16 * i + q * 4;
        if (n0 >= g.Cout) continue;
        f32x4 v = acc[i];
        if (act_elu) v = f32x4{elu_f(v[0]), elu_f(v[1]), elu_f(v[2]), elu_f(v[3])};
        *(f32x4*)(yn + (size_t)(16 * t + j) * g.Cout + n0) = v;
      }
    }
  }
}

// The same layer type as a stream: a few persistent workgroups per compute unit walk (image, band of output rows)
// items; the input rows of the NEXT item travel memory -> registers (whole lines, consecutive lanes 16 bytes apart)
// while the patches of the current one are gathered from LDS and multiplied, so a compute unit always has several
// bands' worth of loads in flight and the memory system sees nothing but full-line sequential reads.  (The patch
// gather straight from memory reads 16 bytes per lane from lines 16 * stride bytes apart: a quarter of every 64-byte
// line at stride 4, and the doom head's first layer ran at half the memory rate because of it.)  The layer's weight
// fragments, every lane's tap offsets and the band geometry stay in registers for the whole kernel; a wave takes the
// tiles wave, wave + NW, ... of a band (NW = 5 where a band has a multiple of five tiles).
//   STG: input pixels of a band per thread (a band has at most STG * NW * 64 input pixels)
// Measured on the doom head's first layer (4096 frames of 84x84x4, one MI355X): 110 -> 104 us from float32 frames, 93 ->
// 86 us from the uint8 ring.  Counters on the float32 run: matrix pipe 39 % busy, other vector instructions 17 %, 43 % of
// the wave cycles waiting -- the memory system delivers 4.9 TB/s of frames in and feature maps out to this stream.
// Two bands in flight per workgroup (registers: three workgroups per compute unit instead of four) ran at 128 us,
// nontemporal result stores changed nothing; the frames are fetched as nontemporal loads (108 -> 104 us).
// Products and their order are those of conv_fwd_c4_kernel and conv_fwd_static_kernel: one accumulator per tile,
// k-groups ascending.
template <int NT, int NG, int NW, int STG, class S>
__global__ __launch_bounds__(NW * 64) void conv_fwd_band_kernel(const ConvG g, const S src, const float* __restrict__ wfrag,
                                                                 float* __restrict__ y, int items, int band_h, FastDiv dnb,
                                                                 int nbands, int act_elu) {
  extern __shared__ __attribute__((aligned(16))) float X[];      // [band rows * W][4]
  constexpr int NTHR = NW * 64;
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  f32x4 w[NG][NT];
  int toff[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
    for (int t = 0; t < NT; ++t) w[gi][t] = *(const f32x4*)(wfrag + (gi * NT + t) * 256 + lane * 4);
    const int tap = 4 * gi + q;                             // this lane's quad of the group: one tap's four channels
    const bool data = 4 * tap < g.K;
    const int ky = data ? tap / g.k : 0, kx = data ? tap - ky * g.k : 0;
    toff[gi] = (ky * g.W + kx) * 4;
  }
  // only the last group can hold the bias row's constant 1 (patch element K) and the zeros after it
  const int kl = 4 * (4 * (NG - 1) + q);
  const bool last_data = kl < g.K;
  const float last_one = kl == g.K ? 1.f : 0.f;

  using Raw = typename S::Raw;
  struct Item { long long base; int n, oy0, oh, nin; };
  auto locate_item = [&](int item) {
    Item it;
    it.n = fdiv(item, dnb);
    it.oy0 = (item - it.n * nbands) * band_h;
    it.oh = min(band_h, g.Ho - it.oy0);
    it.nin = ((it.oh - 1) * g.s + g.k) * g.W;
    it.base = src.image(it.n);
    return it;
  };
  auto prefetch = [&](const Item& it, Raw (&raw)[STG]) {
    const int in0 = (g.s * it.oy0) * g.W;
#pragma unroll
    for (int u = 0; u < STG; ++u) raw[u] = src.load_stream(it.base, (size_t)4 * (in0 + min((int)threadIdx.x + NTHR * u, it.nin - 1)));
  };
  int item = blockIdx.x;
  if (item >= items) return;
  Item cur = locate_item(item);
  Raw raw[STG];
  prefetch(cur, raw);
  for (;;) {
    // the next item is located before this one is staged (a ring slot comes from the index array: that read is then
    // long back when the fetch addresses need it); always located and fetched from, the last one again at the end:
    // no branch around loads
    const int next = item + gridDim.x;
    const bool more = next < items;                          // uniform
    const Item nxt = locate_item(min(next, items - 1));
#pragma unroll
    for (int u = 0; u < STG; ++u) {
      const int i = threadIdx.x + NTHR * u;
      if (i < cur.nin) *(f32x4*)(X + 4 * i) = src.value(raw[u], cur.base);
    }
    lds_barrier();
    prefetch(nxt, raw);
    __builtin_amdgcn_sched_barrier(0);                       // the next band's loads go out before this band's products
    const int npix = cur.oh * g.Wo;
    float* yn = y + ((size_t)cur.n * g.Ho * g.Wo + (size_t)cur.oy0 * g.Wo) * g.Cout;
    for (int t = wave; t < (npix + 15) / 16; t += NW) {
      const int p = min(16 * t + j, npix - 1), oy = fdiv(p, g.dWo), ox = p - oy * g.Wo;
      const float* xp = X + ((g.s * oy) * g.W + g.s * ox) * 4;
      f32x4 b[NG];
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) b[gi] = *(const f32x4*)(xp + toff[gi]);
      b[NG - 1] = last_data ? b[NG - 1] : f32x4{last_one, 0.f, 0.f, 0.f};
      f32x4 acc[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[gi][i][m], b[gi][m], acc[i], 0, 0, 0);
        }
      }
      if (16 * t + j < npix) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int n0 = 16 * i + q * 4;
          if (n0 >= g.Cout) continue;
          f32x4 v = acc[i];
          if (act_elu) v = f32x4{elu_f(v[0]), elu_f(v[1]), elu_f(v[2]), elu_f(v[3])};
          *(f32x4*)(yn + (size_t)(16 * t + j) * g.Cout + n0) = v;
        }
      }
    }
    if (!more) break;
    lds_barrier();                                           // every wave has read its patches: the band may be replaced
    item = next; cur = nxt;
  }
}

// A first layer with EIGHT output channels (the doom head's 5x5 stride-4 layer, Conv_model.py:15-16) fills only half of
// a 16-row matrix tile.  v_mfma_f32_4x4x1_16b_f32 multiplies sixteen independent 4x4 blocks per instruction, one patch
// element at a time: block = (group of four pixels, four of the eight channels), so one instruction covers 32 pixels x
// 8 channels with nothing wasted (8 cycles against the 2 x 32 of two half-empty 16x16x4 tiles; tools/mfma4x4_probe.hip:
// 131-140 TFLOP/s with two accumulators).  Lane 4 * block + i supplies the weight of channel 4 * cg + i (row operand,
// one register per patch element, held for the whole kernel) and patch element k of pixel i of the block's group (column
// operand: the four channels of a tap are one ds_read_b128, four instructions); register r of lane 4 * block + j ends up
// with channel 4 * cg + r of pixel j, i.e. four consecutive channels: one 16-byte store.  All lanes work on the same tap,
// so tap offsets are instruction immediates.  The band of input rows sits in LDS with the pixels of a row de-interleaved
// by x % 4 (CP pixels per class): the 32 pixels of a tile, 4 input pixels apart, are then consecutive 16-byte slots
// (read in place they are 64 bytes apart: four-way bank conflicts), and CP % 8 = 6 keeps the staging stores of eight
// consecutive pixels on eight different slots.  Streaming as in conv_fwd_band_kernel: persistent workgroups, the next
// (image, band) item's rows memory -> registers while this one is multiplied.  The sum runs over the patch elements in
// order on two accumulators taking turns (bias first), so the last bits differ from the 16x16x4 kernels'.
template <int K, int CP, int NW, int STG, int DEPTH, class S>
__global__ __launch_bounds__(NW * 64) void conv_fwd_c8_kernel(const ConvG g, const S src, const float* __restrict__ wfrag,
                                                               float* __restrict__ y, int items, int band_h, FastDiv dnb,
                                                               int nbands, int act_elu) {
  extern __shared__ __attribute__((aligned(16))) float X[];      // [band rows][4 classes][CP pixels][4 channels]
  constexpr int NTHR = NW * 64, NTAP = K * K, RP = 4 * CP * 4;   // RP: floats per staged row
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int blk = lane >> 2, jj = lane & 3, pg = blk >> 1, cg = blk & 1;
  float wk[NTAP][4];                                             // this lane's row operands: W[4 * tap + c][4 * cg + jj]
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
#pragma unroll
    for (int c = 0; c < 4; ++c) wk[t][c] = wfrag[frag_offset(1, 4 * t + c, 4 * cg + jj)];
  }
  f32x4 bias4;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = wfrag[frag_offset(1, 4 * NTAP, 4 * cg + r)];
  // where this thread's staged pixels go: band pixel (row, x) -> row * RP + ((x & 3) * CP + (x >> 2)) * 4
  int wslot[STG];
#pragma unroll
  for (int u = 0; u < STG; ++u) {
    const int i = threadIdx.x + NTHR * u, r = i / g.W, x = i - r * g.W;
    wslot[u] = r * RP + ((x & 3) * CP + (x >> 2)) * 4;
  }
  using Raw = typename S::Raw;
  struct Item { long long base; int n, oy0, oh, nin; };
  auto locate_item = [&](int item) {
    Item it;
    it.n = fdiv(item, dnb);
    it.oy0 = (item - it.n * nbands) * band_h;
    it.oh = min(band_h, g.Ho - it.oy0);
    it.nin = ((it.oh - 1) * 4 + K) * g.W;
    it.base = src.image(it.n);
    return it;
  };
  auto prefetch = [&](const Item& it, Raw (&raw)[STG]) {
    const int in0 = (4 * it.oy0) * g.W;
#pragma unroll
    for (int u = 0; u < STG; ++u) raw[u] = src.load_stream(it.base, (size_t)4 * (in0 + min((int)threadIdx.x + NTHR * u, it.nin - 1)));
  };
  int round = 0;
  auto compute = [&](const Item& cur) {
    const int npix = cur.oh * g.Wo, ntiles = (npix + 31) >> 5;
    float* yn = y + ((size_t)cur.n * g.Ho * g.Wo + (size_t)cur.oy0 * g.Wo) * 8;
    // a band's tiles are dealt to the waves starting at a different wave every round, so that a band whose tiles do
    // not deal evenly does not load the same SIMD every time
    for (int t = (wave - round) & (NW - 1); t < ntiles; t += NW) {
      const int p = min(32 * t + 4 * pg + jj, npix - 1), oy = fdiv(p, g.dWo), ox = p - oy * g.Wo;
      const float* xp = X + (4 * oy) * RP + ox * 4;          // tap (ky, kx) is at xp + ky * RP + ((kx & 3) * CP + (kx >> 2)) * 4
      f32x4 acc0 = bias4, acc1{0.f, 0.f, 0.f, 0.f};
      f32x4 b[2][K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) b[0][kx] = *(const f32x4*)(xp + ((kx & 3) * CP + (kx >> 2)) * 4);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        if (ky + 1 < K) {
#pragma unroll
          for (int kx = 0; kx < K; ++kx) b[(ky + 1) & 1][kx] = *(const f32x4*)(xp + (ky + 1) * RP + ((kx & 3) * CP + (kx >> 2)) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);                   // the next kernel row's reads stay ahead of this row's products
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x4 bv = b[ky & 1][kx];
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[ky * K + kx][0], bv[0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[ky * K + kx][1], bv[1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[ky * K + kx][2], bv[2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wk[ky * K + kx][3], bv[3], acc1, 0, 0, 0);
        }
      }
      if (32 * t + 4 * pg + jj < npix) {
        f32x4 v = acc0 + acc1;
        if (act_elu) v = f32x4{elu_f(v[0]), elu_f(v[1]), elu_f(v[2]), elu_f(v[3])};
        *(f32x4*)(yn + (size_t)(32 * t + 4 * pg + jj) * 8 + 4 * cg) = v;
      }
    }
  };
  // DEPTH items of input rows are in flight per workgroup.  An item is located (a ring slot comes from the index
  // array) a whole item before its rows are requested; the last item is located and fetched again at the end rather
  // than put a branch around loads.
  int item = blockIdx.x;
  if (item >= items) return;
  const int grid = gridDim.x;
  Item its[DEPTH];
  Raw raw[DEPTH][STG];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    its[d] = locate_item(min(item + d * grid, items - 1));
    prefetch(its[d], raw[d]);
  }
  for (;;) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const Item nxt = locate_item(min(item + DEPTH * grid, items - 1));
      const Item cur = its[d];
#pragma unroll
      for (int u = 0; u < STG; ++u) {
        if ((int)threadIdx.x + NTHR * u < cur.nin) *(f32x4*)(X + wslot[u]) = src.value(raw[d][u], cur.base);
      }
      lds_barrier();
      its[d] = nxt;
      if constexpr (DEPTH == 1) {
        prefetch(nxt, raw[d]);                               // the next band's loads go out before this band's products
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
      } else {
        // another band is in flight already; requests after the result stores, so that waiting for the OTHER band's
        // rows (requested before them) never has to wait for these
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        prefetch(nxt, raw[d]);
      }
      item += grid; ++round;
      if (item >= items) return;                             // uniform
      lds_barrier();                                         // every wave has read its patches: the band may be replaced
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
struct DxOff {                             // per class: float offset of its fragment image, divisors na*nb and nb, k-groups
  int off[MAX_CLASSES];
  FastDiv dper[MAX_CLASSES], dnb[MAX_CLASSES];
  int max_ng;
};

// Gathered-row table of a class: entry q (row elements 4q..4q+3; Cout % 4 == 0, so the four share one tap (u, v))
// = {offset of those dZ elements from the dZ row of output pixel (a, bb), u | v << 16}, or {0, -1} past the row's end.
__device__ __forceinline__ void fill_taptab(const ConvG& g, const DxClass& c, int* tapd, int* tapuv) {
  for (int q = threadIdx.x; q < c.NG * 4; q += blockDim.x) {
    const int kk0 = q * 4, tap = kk0 / g.Cout, co = kk0 - tap * g.Cout, u = tap / c.nkx, v = tap - u * c.nkx;
    const bool in = kk0 < c.Kc;
    tapd[q] = in ? (-u * g.Wo - v) * g.Cout + co : 0;
    tapuv[q] = in ? (u | (v << 16)) : -1;
  }
}
// the fetch of one table entry for input pixel (a, bb) whose dZ row would start at element `pz`, and later its
// value.  No branch around the loads (a wave would wait for each where the branches join): element 0 is read
// instead and discarded.
struct DzRaw { f32x4 d, y; bool ok; };
__device__ __forceinline__ DzRaw dz_load(const ConvG& g, const float* __restrict__ dz, const float* __restrict__ yelu,
                                         long long pz, int a, int bb, int td, int tuv) {
  const int oy = a - (tuv & 0xffff), ox = bb - (tuv >> 16);
  DzRaw r;
  r.ok = tuv >= 0 && (unsigned)oy < (unsigned)g.Ho && (unsigned)ox < (unsigned)g.Wo;
  const size_t at = r.ok ? (size_t)(pz + td) : 0;
  r.d = *(const f32x4*)(dz + at);
  r.y = yelu ? *(const f32x4*)(yelu + at) : f32x4{1.f, 1.f, 1.f, 1.f};   // dz is dY: times ELU'(y) when it is used
  return r;
}
__device__ __forceinline__ f32x4 dz_value(const DzRaw& r) {
  return r.ok ? f32x4{r.d[0] * elu_d(r.y[0]), r.d[1] * elu_d(r.y[1]), r.d[2] * elu_d(r.y[2]), r.d[3] * elu_d(r.y[3])}
              : f32x4{0.f, 0.f, 0.f, 0.f};
}

// Input gradient of a layer with FOUR input channels (the first layer of an encoder: a frame stack, or one frame
// padded to four channels): far too narrow for a matrix tile, and bound by reading dZ.  A thread owns one input
// pixel; blockIdx.y = stride phase (y % s, x % s), so the taps a workgroup sees and their kernel rows are the same
// for all its threads (the weights come through scalar loads) and neighbouring threads read neighbouring dZ rows,
// 16 bytes at a time.  Taps in ascending (ky, kx) order, output channels in ascending order.
struct C4Off { FastDiv dper[16], dnb[16]; int na[16], nb[16]; };
// CO > 0: the layer's output channels as a compile-time constant (loops unrolled, kernel rows fetched by a few wide
// scalar loads per tap); CO == 0: any width.
template <int CO>
__global__ __launch_bounds__(256) void conv_dx_c4_kernel(const ConvG g, const C4Off o, const float* __restrict__ dz,
                                                         const float* __restrict__ yelu, const float* __restrict__ w /*[K][Cout]*/,
                                                         float* __restrict__ dx, int n_images) {
  const int cout = CO > 0 ? CO : g.Cout;
  const int ph = blockIdx.y, ry = ph / g.s, rx = ph % g.s;
  const int per = o.na[ph] * o.nb[ph];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_images * per) return;                      // the host checked that the product fits an int
  const int b = fdiv(t, o.dper[ph]), r = t - b * per, qy = fdiv(r, o.dnb[ph]), qx = r - qy * o.nb[ph];
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  for (int ky = ry, jy = 0; ky < g.k; ky += g.s, ++jy) {
    const int oy = qy - jy;
    for (int kx = rx, jx = 0; kx < g.k; kx += g.s, ++jx) {
      const int ox = qx - jx;
      const bool ok = oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
      const size_t at = (((size_t)b * g.Ho + (ok ? oy : 0)) * g.Wo + (ok ? ox : 0)) * cout;
      const float* wp = w + (size_t)((ky * g.k + kx) * 4) * cout;
      if constexpr (CO > 0) {
        f32x4 z[CO / 4];
#pragma unroll
        for (int i = 0; i < CO / 4; ++i) z[i] = *(const f32x4*)(dz + at + 4 * i);
        if (yelu) {
#pragma unroll
          for (int i = 0; i < CO / 4; ++i) {
            const f32x4 y = *(const f32x4*)(yelu + at + 4 * i);
            z[i] = f32x4{z[i][0] * elu_d(y[0]), z[i][1] * elu_d(y[1]), z[i][2] * elu_d(y[2]), z[i][3] * elu_d(y[3])};
          }
        }
        const float keep = ok ? 1.f : 0.f;
        f32x4 part{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int i = 0; i < CO / 4; ++i) {
#pragma unroll
            for (int m = 0; m < 4; ++m) part[c] += z[i][m] * wp[c * CO + 4 * i + m];
          }
        }
        acc += part * keep;
      } else {
        for (int n = 0; n < cout; n += 4) {
          f32x4 z = *(const f32x4*)(dz + at + n);
          if (yelu) {
            const f32x4 y = *(const f32x4*)(yelu + at + n);
            z = f32x4{z[0] * elu_d(y[0]), z[1] * elu_d(y[1]), z[2] * elu_d(y[2]), z[3] * elu_d(y[3])};
          }
          if (!ok) z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += z[m] * wp[c * cout + n + m];
          }
        }
      }
    }
  }
  const int yy = qy * g.s + ry, xx = qx * g.s + rx;
  *(f32x4*)(dx + (((size_t)b * g.H + yy) * g.W + xx) * 4) = acc;
}

// The same for a layer whose whole dZ image fits LDS: a workgroup owns one image, multiplies dZ by ELU' once while
// staging it (pixel stride CO + 1 floats: neighbouring pixels start on neighbouring banks) and walks the stride phases
// one after the other, so that the taps stay uniform inside a wave.  dZ is read from memory once instead of once per
// tap that touches it.
template <int CO>
__global__ __launch_bounds__(256) void conv_dx_c4_lds_kernel(const ConvG g, const C4Off o, const float* __restrict__ dz,
                                                             const float* __restrict__ yelu, const float* __restrict__ w,
                                                             float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float Z[];      // [Ho * Wo][CO + 1]
  constexpr int C4 = CO / 4, ZS = CO + 1;
  const int n = blockIdx.x, npix = g.Ho * g.Wo;
  const f32x4* dz4 = (const f32x4*)(dz + (size_t)n * npix * CO);
  const f32x4* y4 = (const f32x4*)(yelu + (size_t)n * npix * CO);
  // all pieces of a 20x20x32 image per thread in one turn, their loads requested together (a piece per turn was a memory
  // round trip per turn; 4 / 7 / 13 per turn: 157 / 155 / 151 us for the layer's input gradient at 4096 images)
#ifndef RLGAN_DXC4_SU
#define RLGAN_DXC4_SU 13
#endif
  constexpr int SU = RLGAN_DXC4_SU;
  for (int i0 = threadIdx.x; i0 < npix * C4; i0 += 256 * SU) {
    f32x4 zv[SU], yv[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = min(i0 + 256 * u, npix * C4 - 1);
      zv[u] = dz4[i];
      yv[u] = yelu ? y4[i] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + 256 * u;
      if (i < npix * C4) {
        float* zp = Z + (i / C4) * ZS + 4 * (i % C4);
#pragma unroll
        for (int c = 0; c < 4; ++c) zp[c] = yelu ? zv[u][c] * elu_d(yv[u][c]) : zv[u][c];
      }
    }
  }
  __syncthreads();
  float* dxn = dx + (size_t)n * g.H * g.W * 4;
  for (int ph = 0; ph < g.s * g.s; ++ph) {
    const int ry = ph / g.s, rx = ph % g.s, per = o.na[ph] * o.nb[ph];
    for (int i = threadIdx.x; i < per; i += 256) {
      const int qy = fdiv(i, o.dnb[ph]), qx = i - qy * o.nb[ph];
      f32x4 acc{0.f, 0.f, 0.f, 0.f};
      for (int ky = ry, jy = 0; ky < g.k; ky += g.s, ++jy) {
        const int oy = qy - jy;
        for (int kx = rx, jx = 0; kx < g.k; kx += g.s, ++jx) {
          const int ox = qx - jx;
          const bool ok = oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo;
          const float* zp = Z + ((ok ? oy : 0) * g.Wo + (ok ? ox : 0)) * ZS;
          const float* wp = w + (size_t)((ky * g.k + kx) * 4) * CO;
          f32x4 part{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c0 = 0; c0 < CO; c0 += 8) {
            float z[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = zp[c0 + e];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
              for (int e = 0; e < 8; ++e) part[c] += z[e] * wp[c * CO + c0 + e];
            }
          }
          acc += part * (ok ? 1.f : 0.f);
        }
      }
      *(f32x4*)(dxn + ((size_t)(qy * g.s + ry) * g.W + qx * g.s + rx) * 4) = acc;
    }
  }
}

// One wave = 16 input pixels of one class x all Cin column tiles; blockIdx.y = class.  The class's kernel
// fragment is the row operand and the gathered dZ row the column operand: a lane ends up with four consecutive
// input channels of one pixel, its own, and writes them as one 16-byte store.
template <int NT>
__global__ __launch_bounds__(256) void conv_dx_kernel(const ConvG g, const DxOff offs, const float* __restrict__ dz,
                                                      const float* __restrict__ yelu, const float* __restrict__ wdx,
                                                      float* __restrict__ dx, int n_images, const float* __restrict__ x_elu) {
  extern __shared__ int taps[];                 // [2][max_ng * 4]
  const DxClass c = dx_class(g, blockIdx.y);
  int *tapd = taps, *tapuv = taps + offs.max_ng * 4;
  fill_taptab(g, c, tapd, tapuv);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = c.na * c.nb;
  const int Ml = n_images * per;                // the host checked that it fits
  const int pix0 = (blockIdx.x * 4 + wave) * 16;
  if (pix0 >= Ml) return;
  const int prow = min(pix0 + (lane & 15), Ml - 1);
  const int b = fdiv(prow, offs.dper[blockIdx.y]), r = prow - b * per, a = fdiv(r, offs.dnb[blockIdx.y]), bb = r - a * c.nb;
  const long long pz = (((long long)b * g.Ho + a) * g.Wo + bb) * g.Cout;
  const int q0 = lane >> 4;
  const float* wl = wdx + offs.off[blockIdx.y] + lane * 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c.NG > 0) {
    DzRaw av = dz_load(g, dz, yelu, pz, a, bb, tapd[q0], tapuv[q0]);
    f32x4 w[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = *(const f32x4*)(wl + t * 256);
    for (int gi = 0; gi < c.NG; ++gi) {
      DzRaw an = av;
      f32x4 wn[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wn[t] = w[t];
      if (gi + 1 < c.NG) {
        an = dz_load(g, dz, yelu, pz, a, bb, tapd[4 * (gi + 1) + q0], tapuv[4 * (gi + 1) + q0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) wn[t] = *(const f32x4*)(wl + ((gi + 1) * NT + t) * 256);
      }
      const f32x4 ad = dz_value(av);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][m], ad[m], acc[t], 0, 0, 0);
      }
      av = an;
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t] = wn[t];
    }
  }
  // result fragment: rows = input channels 16t + 4*(lane>>4) + i, column = pixel lane&15
  if (pix0 + (lane & 15) >= Ml) return;
  const size_t oo = (((size_t)b * g.H + (g.s * a + c.py)) * g.W + (g.s * bb + c.px)) * g.Cin;
  float* o = dx + oo;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n0 = 16 * t + (lane >> 4) * 4;
    if (n0 >= g.Cin) continue;                            // Cin % 4 == 0: four channels are in range together
    f32x4 v = acc[t];
    if (x_elu) {                                          // the result leaves as the dZ of the layer below
      const f32x4 xe = *(const f32x4*)(x_elu + oo + n0);
      v = f32x4{v[0] * elu_d(xe[0]), v[1] * elu_d(xe[1]), v[2] * elu_d(xe[2]), v[3] * elu_d(xe[3])};
    }
    *(f32x4*)(o + n0) = v;
  }
}

// The same product for a layer whose dZ image fits LDS and whose per-class kernel fragments fit a wave's registers
// (the 3x3 stride-2 layers of the encoders): a workgroup owns ONE image, multiplies dZ by ELU' once while staging it,
// and walks the classes one after the other; inside a class a wave keeps the class's kernel fragments in registers
// and walks its share of the 16-pixel tiles, gathering dZ rows from LDS.  conv_dx_kernel above fetches every dZ row
// (and the layer output beside it) from memory once per tap and every kernel fragment once per 16 pixels.
constexpr int DXI_MAXF = 16;               // kernel fragments (k-groups x input-channel tiles) of a class held by a wave

// One class of one image for one wave.  NGC = the class's k-groups when it is one of the counts instantiated below
// (loops unrolled: the gathers of a tile are all requested before its first MFMA), 0 = any count up to DXI_MAXF / NT.
template <int NT, int NGC>
__device__ __forceinline__ void dx_img_class(const ConvG& g, const DxClass& c, FastDiv dnb, const float* __restrict__ Z, int ZS,
                                             const int* __restrict__ tapd, const int* __restrict__ tapuv,
                                             const float* __restrict__ wl, float* __restrict__ dx0, int imgs, int zero_at,
                                             const float* __restrict__ xe0) {
  constexpr int MAXG = NGC > 0 ? NGC : DXI_MAXF / NT;
  const int lane = threadIdx.x & 63, wave = wave_id(), q0 = lane >> 4;
  const int ng = NGC > 0 ? NGC : c.NG;
  const int per = c.na * c.nb, ntiles = cdiv(per, 16);
  f32x4 w[MAXG * NT];
  int td[MAXG], tu[MAXG], tv[MAXG];
#pragma unroll
  for (int gi = 0; gi < MAXG; ++gi) {
    const bool in = gi < ng;
#pragma unroll
    for (int i = 0; i < NT; ++i) w[gi * NT + i] = in ? *(const f32x4*)(wl + (gi * NT + i) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int tuv = in ? tapuv[4 * gi + q0] : -1;
    td[gi] = in ? tapd[4 * gi + q0] : 0;
    tu[gi] = tuv < 0 ? 0x10000 : (tuv & 0xffff);          // a tap outside the class's list: a row no pixel has
    tv[gi] = tuv < 0 ? 0 : (tuv >> 16);
  }
  // the (image, tile) pairs of the workgroup's images go round the waves
  int img = 0, t = wave;
  while (t >= ntiles && img < imgs) { t -= ntiles; ++img; }
  for (; img < imgs;) {
    const int p = min(16 * t + (lane & 15), per - 1);
    const int a = fdiv(p, dnb), bb = p - a * c.nb;
    const int pz = (img * g.Ho * g.Wo + a * g.Wo + bb) * ZS;
    float* dxn = dx0 + (size_t)img * g.H * g.W * g.Cin;
    // The layer input at this lane's result (xe0: the ELU output of the layer below), requested before the gathers; the
    // result leaves as that layer's dZ.  (Requested a whole tile ahead instead: no change; two tiles per turn for the
    // classes with one or two taps, their gathers requested together: no change either.  Probe builds: the reads cost
    // the kernel 38 us per 4096 maps of 20x20x32, the multiplications nothing -- the 210 MB move here from the layer
    // below's two backward kernels, which are bound by memory, and cost less here than there.)
    const size_t oo = (size_t)img * g.H * g.W * g.Cin + ((size_t)(g.s * a + c.py) * g.W + (g.s * bb + c.px)) * g.Cin;
    f32x4 xe[NT];
    if (xe0) {
#pragma unroll
      for (int i = 0; i < NT; ++i) xe[i] = *(const f32x4*)(xe0 + oo + min(16 * i + q0 * 4, g.Cin - 4));
    }
    f32x4 ad[MAXG];
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
      // a tap outside the map reads the row of zeros behind the images: one select, nothing to clear afterwards.
      // (A border of zeros around every image, which needs no select at all, was measured: the larger images leave
      // three instead of four to a workgroup and the result is 5 % slower.  The tests made once per tile and picked per
      // gather by scalar selects -- with 16 | Cout a gather's tap is the same for the whole wave -- a third fewer vector
      // instructions per tile: no faster either.)
      const bool ok = (unsigned)(a - tu[gi]) < (unsigned)g.Ho && (unsigned)(bb - tv[gi]) < (unsigned)g.Wo;
      ad[gi] = *(const f32x4*)(Z + (ok ? pz + td[gi] : zero_at));
    }
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
      if (NGC > 0 || gi < ng) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[gi * NT + i][m], ad[gi][m], acc[i], 0, 0, 0);
        }
      }
    }
    if (16 * t + (lane & 15) < per) {
      float* o = dxn + ((size_t)(g.s * a + c.py) * g.W + (g.s * bb + c.px)) * g.Cin;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n0 = 16 * i + q0 * 4;
        f32x4 v = acc[i];
        if (xe0) v = f32x4{v[0] * elu_d(xe[i][0]), v[1] * elu_d(xe[i][1]), v[2] * elu_d(xe[i][2]), v[3] * elu_d(xe[i][3])};
        if (n0 < g.Cin) *(f32x4*)(o + n0) = v;
      }
    }
    t += 4;
    while (t >= ntiles && img < imgs) { t -= ntiles; ++img; }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void conv_dx_img_kernel(const ConvG g, const DxOff offs, const float* __restrict__ dz,
                                                          const float* __restrict__ yelu, const float* __restrict__ wdx,
                                                          float* __restrict__ dx, int n_images, int imgs_per_wg,
                                                          const float* __restrict__ x_elu) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ZS = g.Cout + 4, C4 = g.Cout >> 2;
  const int n = blockIdx.x * imgs_per_wg, imgs = min(imgs_per_wg, n_images - n);
  const int npix = imgs * g.Ho * g.Wo;                   // the images of this workgroup, one after the other
  float* Z = lds;                                        // [imgs][Ho * Wo][Cout + 4], then one row of zeros
  const int zero_at = imgs_per_wg * g.Ho * g.Wo * ZS;
  int* tab = (int*)(lds + zero_at + ZS);                 // per class: [max_ng * 4] offsets, then [max_ng * 4] taps
  if (threadIdx.x < ZS) Z[zero_at + threadIdx.x] = 0.f;
  const int ncls = g.s * g.s, tl = offs.max_ng * 4;
  const f32x4* dz4 = (const f32x4*)(dz + (size_t)n * g.Ho * g.Wo * g.Cout);
  const f32x4* y4 = (const f32x4*)(yelu + (size_t)n * g.Ho * g.Wo * g.Cout);
  // staged four pieces per thread at a time, the loads of a turn all requested before the first is used (one load
  // per turn is a memory round trip per turn)
  constexpr int SU = 4;
  for (int i0 = threadIdx.x; i0 < npix * C4; i0 += 256 * SU) {
    f32x4 z[SU], y[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = min(i0 + 256 * u, npix * C4 - 1);
      z[u] = dz4[i];
      y[u] = yelu ? y4[i] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + 256 * u;
      if (i < npix * C4) {
        const int p = i / C4;
        *(f32x4*)(Z + p * ZS + 4 * (i - p * C4)) = f32x4{z[u][0] * elu_d(y[u][0]), z[u][1] * elu_d(y[u][1]), z[u][2] * elu_d(y[u][2]), z[u][3] * elu_d(y[u][3])};
      }
    }
  }
  for (int i = threadIdx.x; i < ncls * tl; i += 256) {
    const int cls = i / tl, q = i - cls * tl;
    const DxClass c = dx_class(g, cls);
    const int kk0 = q * 4, tap = c.nkx > 0 ? kk0 / g.Cout : 0, co = kk0 - tap * g.Cout;
    const int u = c.nkx > 0 ? tap / c.nkx : 0, v = tap - u * c.nkx;
    const bool in = kk0 < c.Kc;
    tab[2 * cls * tl + q] = in ? (-u * g.Wo - v) * ZS + co : 0;
    tab[(2 * cls + 1) * tl + q] = in ? (u | (v << 16)) : -1;
  }
  __syncthreads();
  float* dxn = dx + (size_t)n * g.H * g.W * g.Cin;
  const float* xen = x_elu ? x_elu + (size_t)n * g.H * g.W * g.Cin : nullptr;
  const int imgs_here = imgs;
  for (int cls = 0; cls < ncls; ++cls) {                   // no barrier below: every wave walks the classes at its own pace
    const DxClass c = dx_class(g, cls);
    const int *tapd = tab + 2 * cls * tl, *tapuv = tapd + tl;
    const float* wl = wdx + offs.off[cls] + (threadIdx.x & 63) * 4;
    switch (c.NG) {
      case 0: break;
      case 1: dx_img_class<NT, 1>(g, c, offs.dnb[cls], Z, ZS, tapd, tapuv, wl, dxn, imgs_here, zero_at, xen); break;
      case 2: dx_img_class<NT, 2>(g, c, offs.dnb[cls], Z, ZS, tapd, tapuv, wl, dxn, imgs_here, zero_at, xen); break;
      case 4: dx_img_class<NT, 4>(g, c, offs.dnb[cls], Z, ZS, tapd, tapuv, wl, dxn, imgs_here, zero_at, xen); break;
      case 8: dx_img_class<NT, 8>(g, c, offs.dnb[cls], Z, ZS, tapd, tapuv, wl, dxn, imgs_here, zero_at, xen); break;
      default: dx_img_class<NT, 0>(g, c, offs.dnb[cls], Z, ZS, tapd, tapuv, wl, dxn, imgs_here, zero_at, xen); break;
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
template <int NT, class S>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ConvG g, const S src, const float* __restrict__ dz,
                                                         const float* __restrict__ yelu, float* __restrict__ slabs, int M, int pix_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int KP = g.NG * 16, ldp = KP + 4, ldz = NT * 16 + 4;
  float* P = lds;                       // [WG_PIX][ldp]
  float* Z = lds + WG_PIX * ldp;        // [WG_PIX][ldz]
  int* ktab = (int*)(Z + WG_PIX * ldz); // [KP / 4]
  fill_ktab(g, ktab);                   // the loop's first barrier comes before its first use
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
    // stage patches (sixteen threads per pixel, 16 bytes each per turn) and dZ rows; pixels beyond the range are zero rows
    {
      const int pl = threadIdx.x >> 4, p = p0 + pl;
      const bool in = p < p_end;
      const Pos ps = locate(g, src, in ? p : p_begin);
      typename S::Raw raw[MAXS];                            // KP / 4 <= 16 * MAXS table entries: all fetches first
      int kf[MAXS];
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        kf[i] = ktab[min((int)(threadIdx.x & 15) + 16 * i, KP / 4 - 1)];
        raw[i] = patch_load(src, ps.base, ps.pix, kf[i]);
      }
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        const int q = (threadIdx.x & 15) + 16 * i;
        if (q < KP / 4) *(f32x4*)(P + pl * ldp + 4 * q) = in ? patch_value(src, raw[i], ps.base, kf[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    {
      // no branch around the loads (a wave would wait for each where the branches join): element 0 stands in
      constexpr int ZPER = WG_PIX * NT * 16 / 256;
      float zv[ZPER], yv[ZPER];
      bool zok[ZPER];
#pragma unroll
      for (int i = 0; i < ZPER; ++i) {
        const int e = threadIdx.x + 256 * i, pl = e / (NT * 16), n = e - pl * (NT * 16), p = p0 + pl;
        zok[i] = p < p_end && n < g.Cout;
        const size_t at = zok[i] ? (size_t)p * g.Cout + n : 0;
        zv[i] = dz[at];
        yv[i] = yelu ? yelu[at] : 1.f;                           // dz is dY: times ELU'(y) on the way in
      }
#pragma unroll
      for (int i = 0; i < ZPER; ++i) {
        const int e = threadIdx.x + 256 * i, pl = e / (NT * 16), n = e - pl * (NT * 16);
        Z[pl * ldz + n] = zok[i] ? zv[i] * elu_d(yv[i]) : 0.f;
      }
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

// Weight gradient of layers with few patch elements (at most 8 k-groups and 2 column tiles: the first layers).  There
// the product per pixel is too small to pay for workgroup barriers, so every wave works alone: it walks its own
// range of pixels 16 at a time, fetches patches and dZ rows in 16-byte pieces (lane = (pixel l&15, quad l>>4)),
// turns them through a private piece of LDS into the operand layout (row operand = patch element, summed
// dimension = pixel), multiplies all strips, and writes a slab of its own.  The next 16 pixels are fetched
// while the current ones are multiplied.
template <int NT, int MAXG, class S>
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(const ConvG g, const S src, const float* __restrict__ dz,
                                                               const float* __restrict__ yelu, float* __restrict__ slabs, int M,
                                                               int pix_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NP = NT * 16, ldz = NP % 32 == 16 ? NP : NP + 16;     // row pitches = 16 mod 32: the four pixel rows a
  const int KP = g.NG * 16, ldp = KP % 32 == 16 ? KP : KP + 16;       // transposed read touches fall on two bank halves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* ktab = (int*)lds;                                              // [KP / 4], shared by the waves
  float* P = lds + KP / 4 + wave * 16 * (ldp + ldz);                  // this wave's [16][ldp]
  float* Z = P + 16 * ldp;                                            // and [16][ldz]
  fill_ktab(g, ktab);
  __syncthreads();
  const int wv = blockIdx.x * 4 + wave;
  const int p_begin = wv * pix_per_wave, p_end = min(M, p_begin + pix_per_wave);
  if (p_begin >= p_end) return;                                       // no barrier follows
  const int px = lane & 15, q = lane >> 4;
  f32x4 acc[MAXG][NT];
#pragma unroll
  for (int s = 0; s < MAXG; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  typename S::Raw ra[MAXG];
  int kf[MAXG];
  f32x4 rz[NT], ry[NT];
  long long rbase;
  auto fetch = [&](int p0) {
    const int p = p0 + px;
    const bool in = p < p_end;
    const Pos ps = locate(g, src, in ? p : p_begin);
    rbase = ps.base;
#pragma unroll
    for (int s = 0; s < MAXG; ++s) {
      if (s < g.NG) {
        const int k = ktab[4 * s + q];
        kf[s] = in ? k : -2;                                          // pixels beyond the range: zero rows
        ra[s] = patch_load(src, ps.base, ps.pix, k);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 4 * (q + 4 * t);
      const size_t at = in && col < g.Cout ? (size_t)p * g.Cout + col : 0;
      rz[t] = *(const f32x4*)(dz + at);
      ry[t] = yelu ? *(const f32x4*)(yelu + at) : f32x4{1.f, 1.f, 1.f, 1.f};   // dz is dY: times ELU'(y) on the way in
    }
  };
  fetch(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += 16) {
    const bool in = p0 + px < p_end;
#pragma unroll
    for (int s = 0; s < MAXG; ++s)
      if (s < g.NG) *(f32x4*)(P + px * ldp + 16 * s + 4 * q) = patch_value(src, ra[s], rbase, kf[s]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 4 * (q + 4 * t);
      const bool ok = in && col < g.Cout;
      *(f32x4*)(Z + px * ldz + col) = ok ? f32x4{rz[t][0] * elu_d(ry[t][0]), rz[t][1] * elu_d(ry[t][1]), rz[t][2] * elu_d(ry[t][2]),
                                                 rz[t][3] * elu_d(ry[t][3])}
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (p0 + 16 < p_end) fetch(p0 + 16);
    // the wave's own LDS accesses are carried out in order; the fence keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // A[row = patch element][kk = pixel 4m + q], B[kk = pixel 4m + q][col = channel]
    f32x4 bz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int m = 0; m < 4; ++m) bz[t][m] = Z[(4 * m + q) * ldz + 16 * t + px];
    }
#pragma unroll
    for (int s = 0; s < MAXG; ++s) {
      if (s < g.NG) {
        f32x4 ap;
#pragma unroll
        for (int m = 0; m < 4; ++m) ap[m] = P[(4 * m + q) * ldp + 16 * s + px];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[m], bz[t][m], acc[s][t], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // slab [KP][NT*16]: result rows = patch elements 16s + 4q + i, column = channel 16t + px
  float* out = slabs + (size_t)wv * KP * NP;
#pragma unroll
  for (int s = 0; s < MAXG; ++s) {
    if (s < g.NG) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(size_t)(16 * s + 4 * q + i) * NP + 16 * t + px] = acc[s][t][i];
      }
    }
  }
}

// Weight gradient of a first layer (four input channels, at most 47 patch elements: the 3x3 kernels of the
// "universe" encoders) with whole images in LDS.  The kernels above copy every input pixel into each of the patches
// it belongs to before multiplying; here a workgroup stages the image once as it lies in memory ([H*W][4]) and dZ of
// the image transposed ([pixel/4][channel][4 pixels], times ELU'), and the row operand of the product
//   dW[(tap, ci) | bias][co] += sum over output pixels of X[pixel(tap)][ci] * dZ[pixel][co]
// is gathered from the image in LDS: lane (i, q) reads X at pixel offset (table in LDS, four pixels per read) + its own
// tap offset; the bias row multiplies by one.  The 16-pixel groups of an image go round the four waves; a workgroup
// walks a chunk of images and leaves one slab, the four waves' sums added in wave order.
constexpr int C4W_RT = 3;                  // row tiles: (k*k*4 + 1) <= 48
template <int NT, int XU, int ZU, class S>
__global__ __launch_bounds__(256) void conv_wgrad_c4_kernel(const ConvG g, const S src, const float* __restrict__ dz,
                                                            const float* __restrict__ yelu, float* __restrict__ slabs,
                                                            int n_images, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HW = g.H * g.W, npix = g.Ho * g.Wo, NG16 = cdiv(npix, 16), PP = NG16 * 16;
  float* X = lds;                                        // [H * W][4]
  float* Zt = lds + HW * 4;                              // [PP / 4][Cout][4], zeros beyond the image's pixels
  unsigned short* pixtab = (unsigned short*)(Zt + PP * g.Cout);   // [PP] offset of the patch's first element in X (16 bits: two workgroups must fit a compute unit's LDS)
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  for (int p = threadIdx.x; p < PP; p += 256) {
    const int pc = min(p, npix - 1), oy = fdiv(pc, g.dWo), ox = pc - oy * g.Wo;
    pixtab[p] = (unsigned short)(((g.s * oy) * g.W + g.s * ox) * 4);
  }
  for (int i = threadIdx.x; i < (PP - npix) * g.Cout; i += 256) {   // pixels past the image's last: zero, never overwritten
    const int p = npix + i / g.Cout, co = i % g.Cout;
    Zt[((p >> 2) * g.Cout + co) * 4 + (p & 3)] = 0.f;
  }
  int tapoff[C4W_RT];
  float one[C4W_RT], live[C4W_RT];
#pragma unroll
  for (int rt = 0; rt < C4W_RT; ++rt) {
    const int e = 16 * rt + j, ec = min(e, g.K - 1), tap = ec >> 2, ky = tap / g.k, kx = tap - ky * g.k;
    tapoff[rt] = (ky * g.W + kx) * 4 + (ec & 3);
    one[rt] = e == g.K ? 1.f : 0.f;                      // the bias row
    live[rt] = e < g.K ? 1.f : 0.f;
  }
  f32x4 acc[C4W_RT][NT];
#pragma unroll
  for (int rt = 0; rt < C4W_RT; ++rt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int n0 = blockIdx.x * per_wg, n1 = min(n_images, n0 + per_wg);
  // An image's operands travel memory -> registers while the previous image is multiplied out of LDS: XU input pixels
  // and ZU four-channel pieces of dY and of the layer output per thread (the host checked that these cover an image).
  // (Items of four pixels by four channels, transposed in registers and stored as four 16-byte pieces without bank
  // conflicts, were measured: 115 -> 121 us per 4096 frames -- the stores are not what this kernel waits for.  It
  // waits for memory: a 42x42x4 float32 frame, dY and the layer output of 20x20x32 are 130 KB per image, 533 MB per
  // 4096 images; a build without the products takes 96 of the 115 us, one that multiplies one image again and again 64.)
  const int C4 = g.Cout >> 2, total = npix * C4;
  typename S::Raw xr[XU];
  f32x4 dr[ZU], yr[ZU];
  long long base_next = -1;
  auto fetch = [&](int n) {
    base_next = src.image(n);
#pragma unroll
    for (int u = 0; u < XU; ++u) xr[u] = src.load(base_next, (size_t)4 * min((int)threadIdx.x + 256 * u, HW - 1));
    const f32x4* dz4 = (const f32x4*)(dz + (size_t)n * npix * g.Cout);
    const f32x4* y4 = (const f32x4*)(yelu + (size_t)n * npix * g.Cout);
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int i = min((int)threadIdx.x + 256 * u, total - 1);
      dr[u] = dz4[i];
      yr[u] = yelu ? y4[i] : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  if (n0 < n1) fetch(n0);
  for (int n = n0; n < n1; ++n) {
    __syncthreads();                                     // the previous image's operands have been read
    const long long base = base_next;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < HW) *(f32x4*)(X + 4 * i) = src.value(xr[u], base);
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < total) {
        const int p = i / C4, c0 = 4 * (i - p * C4);
        float* zt = Zt + ((p >> 2) * g.Cout + c0) * 4 + (p & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) zt[4 * c] = dr[u][c] * elu_d(yr[u][c]);
      }
    }
    __syncthreads();
    if (n + 1 < n1) fetch(n + 1);
    for (int G = wave; G < NG16; G += 4) {
      const uint2 pk = *(const uint2*)(pixtab + 16 * G + 4 * q);
      const int pt[4] = {(int)(pk.x & 0xffffu), (int)(pk.x >> 16), (int)(pk.y & 0xffffu), (int)(pk.y >> 16)};
      f32x4 bz[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bz[nt] = *(const f32x4*)(Zt + ((4 * G + q) * g.Cout + 16 * nt + j) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int rt = 0; rt < C4W_RT; ++rt) {
          const float av = X[pt[e] + tapoff[rt]] * live[rt] + one[rt];   // a row tile past the last multiplies zeros: no branch, so the reads go out ahead
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bz[nt][e], acc[rt][nt], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  float* red = lds;                                      // [4 waves][C4W_RT * NT tiles][64 lanes][4] over the image and, for small images, the start of dZ (the host checked the size)
  constexpr int TILES = C4W_RT * NT;
#pragma unroll
  for (int rt = 0; rt < C4W_RT; ++rt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *(f32x4*)(red + ((wave * TILES + rt * NT + nt) * 64 + lane) * 4) = acc[rt][nt];
  }
  __syncthreads();
  const int NP = NT * 16;
  float* out = slabs + (size_t)blockIdx.x * (g.NG * 16) * NP;
  for (int i = threadIdx.x; i < TILES * 256; i += 256) {
    const int tile = i >> 8, el = i & 255, ln = el >> 2, r = el & 3;
    const int rt = tile / NT, nt = tile - rt * NT;
    if (rt < g.NG) {
      const float* pr = red + tile * 256 + el;
      out[(16 * rt + 4 * (ln >> 4) + r) * NP + 16 * nt + (ln & 15)] = ((pr[0] + pr[TILES * 256]) + pr[2 * TILES * 256]) + pr[3 * TILES * 256];
    }
  }
}

// Weight gradient of the 3x3 layers with 32 input channels on 20x20 maps (the encoders' second layer) with whole images
// in LDS and the work of a workgroup divided by KIND of wave.  conv_wgrad_kernel stages 16 pixels of patches at a time
// behind two barriers; a form where every wave both stages and multiplies whole images was measured at the same time as
// that kernel: the layer needs 72 KB per image at the rate the matrix pipe could use them, and a workgroup that alternates
// between the two phases overlaps them only through its neighbour on the compute unit.  Here four waves multiply image n
// out of one LDS buffer while four others fill the second buffer with image n + 1 (the image as it lies in memory, pixels
// IMG_PAD floats further apart than their 32 channels so that the four pixels of a row-operand read fall on four
// quarters of the banks; dY * ELU'(y) transposed to [pixel / 4][channel][4]); one barrier per image.  A 16-row tile of
// the product is half of one tap's channels, contiguous in LDS; with the map width a template parameter the tap offset
// is an immediate of the read.  The multiplying waves split the row tiles -- wave (half, set) takes channels
// 16 half .. +15 of taps 0, 2, 4, 6, 8 (set 0) or of taps 1, 3, 5, 7 and the bias row (set 1; with half 1 that last
// tile multiplies zeros: the others run five tiles anyway) -- so each owns its part of the result outright.
constexpr int IMG_PAD = 2;
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int SET, int NT, int NG16, int W, int PS>
__device__ __forceinline__ void wgrad_img_products(const float* const (&xa)[NG16][4], const float* Zq, int Cout, float bias_a,
                                                   f32x4 (&acc)[5][NT]) {
  // A step = one pixel of each lane's four in a group: five row operands, ten matrix instructions.  The operands of
  // step t + 1 (and dZ of the next group before a group's last step) are read from LDS before step t's products are
  // issued; the fences keep the compiler from reading further ahead (it runs out of registers).
  float av[2][5];
  f32x4 bz[2][NT];
  auto read_step = [&](int t) {
    const int G = t >> 2, e = t & 3;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int tap = 2 * i + SET;
      av[t & 1][i] = tap < 9 ? xa[G][e][((tap / 3) * W + tap % 3) * PS] : bias_a;
    }
  };
  auto read_dz = [&](int G) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bz[G & 1][nt] = *(const f32x4*)(Zq + (4 * G * Cout + 16 * nt) * 4);
  };
  read_dz(0);
  read_step(0);
#pragma unroll
  for (int t = 0; t < 4 * NG16; ++t) {
    const int G = t >> 2, e = t & 3;
    if (t + 1 < 4 * NG16) read_step(t + 1);
    if (e == 2 && G + 1 < NG16) read_dz(G + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t & 1][i], bz[G & 1][nt][e], acc[i][nt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

constexpr int IMG_WAVES = 8, IMG_FILL = 4;                 // waves of a workgroup; how many of them fill (the others multiply, one per SIMD)
constexpr int IMG_XU = 13, IMG_ZU = 3;                     // pieces of an image per filling thread (20x20x32 in, 9x9x32 out)
template <int NT, int NG16, int W, class S>
__global__ __launch_bounds__(IMG_WAVES * 64) void conv_wgrad_img_kernel(const ConvG g, const S src, const float* __restrict__ dz,
                                                                       const float* __restrict__ yelu, float* __restrict__ slabs,
                                                                       int n_images, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CIN = 32, PS = CIN + IMG_PAD, PP = NG16 * 16, FILL_THREADS = IMG_FILL * 64;
  const int HW = g.H * W, npix = g.Ho * g.Wo, XC4 = HW * (CIN / 4);
  const int BUF = HW * PS + PP * g.Cout;                  // floats of one buffer: X [H * W][PS], then Zt [PP / 4][Cout][4]
  const int lane = threadIdx.x & 63, wave = wave_id(), q = lane >> 4, j = lane & 15;
  for (int i = threadIdx.x; i < 2 * (PP - npix) * g.Cout; i += IMG_WAVES * 64) {   // pixels past the image's last, both buffers: zero, never overwritten
    const int b = i / ((PP - npix) * g.Cout), r = i - b * ((PP - npix) * g.Cout);
    const int p = npix + r / g.Cout, co = r % g.Cout;
    lds[b * BUF + HW * PS + ((p >> 2) * g.Cout + co) * 4 + (p & 3)] = 0.f;
  }
  const int n0 = blockIdx.x * per_wg, n1 = min(n_images, n0 + per_wg);
  const int C4 = g.Cout >> 2, total = npix * C4;
  // An image on its way through the filling waves: every piece of it (XU four-channel pieces of the image, ZU of dY and
  // of the layer output per thread; the host checked that these cover an image) is requested before the first is used.
  const int ft = threadIdx.x - (IMG_WAVES - IMG_FILL) * 64;
  typename S::Raw xr[IMG_XU];
  f32x4 dr[IMG_ZU], yr[IMG_ZU];
  long long base_f = -1;
  auto fetch = [&](int n) {
    base_f = src.image(n);
#pragma unroll
    for (int u = 0; u < IMG_XU; ++u) xr[u] = src.load_stream(base_f, (size_t)4 * min(ft + FILL_THREADS * u, XC4 - 1));
    const f32x4* dz4 = (const f32x4*)(dz + (size_t)n * npix * g.Cout);
    const f32x4* y4 = (const f32x4*)(yelu + (size_t)n * npix * g.Cout);
#pragma unroll
    for (int u = 0; u < IMG_ZU; ++u) {
      const int i = min(ft + FILL_THREADS * u, total - 1);
      dr[u] = __builtin_nontemporal_load(dz4 + i);
      yr[u] = yelu ? __builtin_nontemporal_load(y4 + i) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  auto stash = [&](int b) {
    float* X = lds + b * BUF;
    float* Zt = X + HW * PS;
#pragma unroll
    for (int u = 0; u < IMG_XU; ++u) {
      const int i = ft + FILL_THREADS * u;
      if (i < XC4) {
        const int pix = i / (CIN / 4), c = 4 * (i % (CIN / 4));
        const f32x4 v = src.value(xr[u], base_f);
        float* d = X + pix * PS + c;                     // 8-byte aligned (PS is even)
        *(f32x2*)d = f32x2{v[0], v[1]};
        *(f32x2*)(d + 2) = f32x2{v[2], v[3]};
      }
    }
#pragma unroll
    for (int u = 0; u < IMG_ZU; ++u) {
      const int i = ft + FILL_THREADS * u;
      if (i < total) {
        const int p = i / C4, c0 = 4 * (i - p * C4);
        float* zt = Zt + ((p >> 2) * g.Cout + c0) * 4 + (p & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) zt[4 * c] = dr[u][c] * elu_d(yr[u][c]);
      }
    }
  };
  const bool filler = wave >= IMG_WAVES - IMG_FILL;        // wave-uniform
  const int half = wave & 1, set = (wave >> 1) & 1;           // of the multiplying waves 0 .. 3
  // this lane's row operand of tap (0, 0) for each of its pixels, in buffer 0 (floats from the start of LDS)
  int xo[NG16][4];
#pragma unroll
  for (int G = 0; G < NG16; ++G) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pc = min(16 * G + 4 * q + e, npix - 1), oy = fdiv(pc, g.dWo), ox = pc - oy * g.Wo;   // past the last: any pixel (dZ is zero there)
      xo[G][e] = ((g.s * oy) * W + g.s * ox) * PS + 16 * half + j;
    }
  }
  const float bias_a = (half == 0 && j == 0) ? 1.f : 0.f;     // the bias row's operand; the other rows of that tile are dead
  f32x4 acc[5][NT];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Two loops with the same barriers, one per kind of wave (in one loop the registers of both kinds would be live
  // together).  The filling waves request image n + 1 and write it within the iteration that multiplies image n; keeping
  // a further image in registers across the barrier, or a second multiplying wave per SIMD (the pixel groups split in
  // two, summed at the end), measured 95-97 us per 4096 images against 90 for this form.
  if (filler) {
    if (n0 < n1) { fetch(n0); stash(0); }
    __syncthreads();
    for (int n = n0; n < n1; ++n) {
      if (n + 1 < n1) {
        fetch(n + 1);
        stash(((n - n0) & 1) ^ 1);                         // the other buffer: its last readers passed the previous barrier
      }
      __syncthreads();
    }
    return;
  }
  __syncthreads();
  for (int n = n0; n < n1; ++n) {
    const float* X = lds + ((n - n0) & 1) * BUF;
    const float* xa[NG16][4];
#pragma unroll
    for (int G = 0; G < NG16; ++G) {
#pragma unroll
      for (int e = 0; e < 4; ++e) xa[G][e] = X + xo[G][e];
    }
    const float* Zq = X + HW * PS + (q * g.Cout + j) * 4;
    if (set == 0) wgrad_img_products<0, NT, NG16, W, PS>(xa, Zq, g.Cout, bias_a, acc);
    else wgrad_img_products<1, NT, NG16, W, PS>(xa, Zq, g.Cout, bias_a, acc);
    __syncthreads();
  }
  const int NP = NT * 16;
  float* out = slabs + (size_t)blockIdx.x * (g.NG * 16) * NP;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int rt = 2 * (2 * i + set) + half;             // 16 * rt = first patch element of the tile (tap * 32 + 16 * half); 18 = the bias row's tile
    if (rt < g.NG) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(16 * rt + 4 * q + r) * NP + 16 * nt + j] = acc[i][nt][r];
      }
    }
  }
}

// grad[(k, n)] in canonical order ([K][Cout] kernel, then bias[Cout]) = sum of the slabs.  A workgroup
// handles 16 entries; 16 threads per entry each add a strided share of the slabs in index order and the
// 16 partial sums are added in a fixed order, so the result does not depend on timing.
// `every`: the slabs to add are every `every`-th one (1: all of them; FOLD: the group sums conv_wgrad_fold_kernel left).
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int every, int KP, int NP,
                                                                int K, int Cout, float* __restrict__ grad) {
  // 64 consecutive gradient entries per workgroup (neighbouring lanes read neighbouring floats of a slab), four
  // slab phases per entry, each with four loads in flight; the phases meet in LDS and are added in phase order
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  const int total = (K + 1) * Cout;
  float s = 0.f;
  if (i < total) {
    const int k = i / Cout, n = i - k * Cout;
    const float* p = slabs + (size_t)k * NP + n;
    const size_t st = (size_t)KP * NP * every;
    int q = ph;
    for (; q + 12 < n_slabs; q += 16) {
      const float a = p[(size_t)q * st], b = p[(size_t)(q + 4) * st], c = p[(size_t)(q + 8) * st], d = p[(size_t)(q + 12) * st];
      s += a; s += b; s += c; s += d;
    }
    for (; q < n_slabs; q += 4) s += p[(size_t)q * st];
  }
  part[ph][col] = s;
  __syncthreads();
  if (ph == 0 && i < total) grad[i] = ((part[0][col] + part[1][col]) + part[2][col]) + part[3][col];
}

// With thousands of slabs the kernel above is a few workgroups walking long chains of loads.  This one goes first
// then: slabs in groups of FOLD, one workgroup per (64 entries, group), the group's sum left in the group's first slab
// (every entry of that slab is read by the one workgroup that later overwrites it, behind its barrier).  Groups in
// order, four phases per group in order: still one fixed order of additions.
constexpr int FOLD = 32;
__global__ __launch_bounds__(256) void conv_wgrad_fold_kernel(float* __restrict__ slabs, int n_slabs, int slab_floats) {
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  const int q0 = blockIdx.y * FOLD, q1 = min(n_slabs, q0 + FOLD);
  float s = 0.f;
  if (i < slab_floats) {
    const float* p = slabs + i;
    float v[FOLD / 4];
#pragma unroll
    for (int j = 0; j < FOLD / 4; ++j) { const int q = q0 + ph + 4 * j; v[j] = q < q1 ? p[(size_t)q * slab_floats] : 0.f; }
#pragma unroll
    for (int j = 0; j < FOLD / 4; ++j) s += v[j];
  }
  part[ph][col] = s;
  __syncthreads();
  if (ph == 0 && i < slab_floats) slabs[(size_t)q0 * slab_floats + i] = ((part[0][col] + part[1][col]) + part[2][col]) + part[3][col];
}

// all slabs into grad, through the folding step when there are many
inline int reduce_slabs(float* slabs, int used, int KP, int NP, int K, int Cout, float* grad, hipStream_t st) {
  int every = 1;
  if (used > 2 * FOLD) {
    hipLaunchKernelGGL(conv_wgrad_fold_kernel, dim3(cdiv(KP * NP, 64), cdiv(used, FOLD)), dim3(256), 0, st, slabs, used, KP * NP);
    if (const int rc = (int)hipGetLastError()) return rc;
    every = FOLD;
    used = cdiv(used, FOLD);
  }
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv((K + 1) * Cout, 64)), dim3(256), 0, st, slabs, used, every, KP, NP, K, Cout, grad);
  return (int)hipGetLastError();
}

// fragment images of the input-gradient classes: offsets and total size (0 when the layer is outside
// what conv_dx_kernel handles: stride > 4 or more than 64 input channels)
inline int dx_layout(const ConvG& g, DxOff* offs) {
  if (g.s * g.s > MAX_CLASSES || g.Cin > 64) return 0;
  const int ntc = cdiv(g.Cin, 16);
  int total = 0, max_ng = 0;
  for (int cls = 0; cls < g.s * g.s; ++cls) {
    const DxClass c = dx_class(g, cls);
    if (offs) {
      offs->off[cls] = total;
      offs->dper[cls] = make_div(std::max(1, c.na * c.nb));
      offs->dnb[cls] = make_div(std::max(1, c.nb));
    }
    max_ng = std::max(max_ng, c.NG);
    total += c.NG * ntc * 256;
  }
  if (offs) offs->max_ng = max_ng;
  return total;
}

inline int launched() { return (int)hipGetLastError(); }

// Bands for conv_fwd_band_kernel: output rows per band and waves per workgroup such that little is wasted -- wave slots
// (a band's 16-pixel tiles are dealt to nw waves), pixel columns of the last tile of a band, and input rows staged twice
// because neighbouring bands overlap by k - s rows.  A band's input rows take at most 32 KB of LDS (four or five
// workgroups per compute unit) and at most 6 pixels per thread.
struct BandPlan { int band_h, nw, stg; };
inline bool plan_bands(const ConvG& g, BandPlan& best) {
  double best_score = 0.0;
  for (int nw = 4; nw <= 5; ++nw)
    for (int h = 1; h <= g.Ho; ++h) {
      const int nin = ((h - 1) * g.s + g.k) * g.W, stg = cdiv(nin, nw * 64);
      if ((size_t)nin * 16 > 32 * 1024 || stg > 6) break;
      long long slots = 0, rows = 0;
      for (int oy0 = 0; oy0 < g.Ho; oy0 += h) {
        const int oh = std::min(h, g.Ho - oy0);
        slots += (long long)nw * cdiv(cdiv(oh * g.Wo, 16), nw) * 16;
        rows += (oh - 1) * g.s + g.k;
      }
      const double score = (double)(g.Ho * g.Wo) / (double)slots * std::min(1.0, (double)((g.Ho - 1) * g.s + g.k) / (double)rows);
      if (score >= best_score) { best_score = score; best = BandPlan{h, nw, stg}; }
    }
  return best_score > 0.0;
}

template <int NT, int NG, int NW, int STG, class S>
int launch_band(const ConvG& g, const S& src, const float* wfrag, float* y, int n_images, int band_h, int act_elu, hipStream_t st) {
  const int nbands = cdiv(g.Ho, band_h);
  const long long items = (long long)n_images * nbands;
  if (items > 0x7fffffffll) return RLGAN_E_ALIGN;
  const size_t lds = (size_t)((band_h - 1) * g.s + g.k) * g.W * 16;
  // workgroups per compute unit: what the kernel's registers and the bands in LDS allow (asked once per band size)
  static size_t asked_lds = 0;
  static int per_cu = 0;
  if (per_cu == 0 || asked_lds != lds) {
    asked_lds = lds;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_fwd_band_kernel<NT, NG, NW, STG, S>, NW * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  }
  const int grid = (int)std::min<long long>(items, 256ll * per_cu);
  hipLaunchKernelGGL((conv_fwd_band_kernel<NT, NG, NW, STG, S>), dim3(grid), dim3(NW * 64), lds, st, g, src, wfrag, y, (int)items, band_h,
                     make_div(nbands), nbands, act_elu);
  return (int)hipGetLastError();
}

template <class S>
int launch_fwd(const ConvG& g, const S& src, const float* wfrag, float* y, int n_images, int act_elu, void* stream) {
  const long long Ml = (long long)n_images * g.Ho * g.Wo;
  if (Ml > 0x7fffffffll || ((uintptr_t)y & 15)) return RLGAN_E_ALIGN;
  const int M = (int)Ml;
  if (g.Cin == 4 && g.Cout == 8 && g.k == 5 && g.s == 4 && g.W == 84 && g.Ho >= 4 && !(((uintptr_t)wfrag | (uintptr_t)src.floats()) & 15)) {
    // the doom head's first layer on 84-pixel rows: full tiles of the sixteen-block matrix instruction
    // bands of four output rows (17 input rows, 24 KB of LDS), two waves a workgroup, four workgroups per compute unit
    // (the kernels hold 100 weight registers: two waves per SIMD); with eight-row bands and four waves: 102 against 96 us
    constexpr int K = 5, CP = 22, NW = 2, STG = 12, BAND_H = 4;        // 1,428 pixels a band <= STG * NW * 64
    // one item of rows in flight per workgroup: two or three (uint8 rows are few registers) ran 4 to 7 % slower
    constexpr int DEPTH = 1;
    const int nbands = cdiv(g.Ho, BAND_H);
    const long long items = (long long)n_images * nbands;
    if (items <= 0x7fffffffll) {
      const size_t lds = (size_t)((BAND_H - 1) * 4 + K) * 4 * CP * 16;
      static int per_cu = 0;
      if (per_cu == 0) {
        (void)hipFuncSetAttribute((const void*)conv_fwd_c8_kernel<K, CP, NW, STG, DEPTH, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_fwd_c8_kernel<K, CP, NW, STG, DEPTH, S>, NW * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
      }
      const int grid = (int)std::min<long long>(items, 256ll * per_cu);
      hipLaunchKernelGGL((conv_fwd_c8_kernel<K, CP, NW, STG, DEPTH, S>), dim3(grid), dim3(NW * 64), lds, (hipStream_t)stream, g, src, wfrag, y,
                         (int)items, BAND_H, make_div(nbands), nbands, act_elu);
      return launched();
    }
  }
  BandPlan bp;
  // (a 3x3 layer on uint8 frames that fit LDS whole -- 42x42x4 -- stays with the one-image-per-workgroup kernel below:
  // 62 against 64 us for 4096 frames, 118 against 127 us inside the pixel iteration)
  const bool small_u8 = !src.floats() && (size_t)g.H * g.W * 16 <= 48 * 1024 && g.NG <= 3;
  if (g.Cin == 4 && !small_u8 && !(((uintptr_t)wfrag | (uintptr_t)src.floats()) & 15) && ((g.NG == 7 && g.NT == 1) || (g.NG == 3 && g.NT == 2)) && plan_bands(g, bp)) {
    // a first layer as a stream of bands: the 5x5 layer of the doom head, the 3x3 layers of the universe heads
    hipStream_t st = (hipStream_t)stream;
#define RLGAN_FB(NTV, NGV, NWV, STGV) \
    if (g.NT == NTV && g.NG == NGV && bp.nw == NWV && bp.stg <= STGV) return launch_band<NTV, NGV, NWV, STGV, S>(g, src, wfrag, y, n_images, bp.band_h, act_elu, st);
    RLGAN_FB(1, 7, 4, 6) RLGAN_FB(1, 7, 5, 6)
    RLGAN_FB(2, 3, 4, 3) RLGAN_FB(2, 3, 4, 6) RLGAN_FB(2, 3, 5, 3) RLGAN_FB(2, 3, 5, 6)
#undef RLGAN_FB
  }
  if (g.Cin == 4 && g.NG <= 3 && g.NT <= 2 && !((uintptr_t)wfrag & 15)) {      // 3x3 kernels; the 5x5 stride-4 layer measured no faster this way (128 against 122 us)
    // a first layer: one band of one image per workgroup, staged in LDS.  The band: as many output rows as 48 KB of
    // input rows allow (a 42x42 frame is one band, an 84x84 frame three or four); a clamped staging address must stay
    // inside the band's rows, and a gathered quad inside the staged rows, which the row count below guarantees.
    const int row_bytes = g.W * 16;
    const int max_rows = (48 * 1024) / row_bytes;
    int band_h = max_rows >= g.k ? (max_rows - g.k) / g.s + 1 : 0;
    if (band_h >= 1) {
      band_h = std::min(band_h, g.Ho);
      const int bands = cdiv(g.Ho, band_h);
      band_h = cdiv(g.Ho, bands);                             // even bands
      const size_t ilds = (size_t)((band_h - 1) * g.s + g.k) * row_bytes;
      const dim3 grid(n_images, cdiv(g.Ho, band_h));
      hipStream_t st = (hipStream_t)stream;
      if (g.NT == 1) hipLaunchKernelGGL((conv_fwd_c4_kernel<1, 3, S>), grid, dim3(256), ilds, st, g, src, wfrag, y, band_h, act_elu);
      else hipLaunchKernelGGL((conv_fwd_c4_kernel<2, 3, S>), grid, dim3(256), ilds, st, g, src, wfrag, y, band_h, act_elu);
      return launched();
    }
  }
  // workgroups per compute unit: as many as the LDS copies of the weights allow (eight at most: one wave per SIMD each)
  const size_t lds = (size_t)g.NG * g.NT * 1024 + (size_t)g.NG * 16;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
  const dim3 grid(std::min(cdiv(cdiv(M, 16), 4), 256 * per_cu)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (const float* xf = src.floats()) {
    // float32 input of less than 2 GiB and a layer whose group count has an unrolled form
    const bool small = (long long)n_images * g.H * g.W * g.Cin * 4 < 0x7fffffffll && g.K >= 4 && !((uintptr_t)xf & 15);
    // few tiles in all: one per wave and column-tile share, everything requested at once
#define RLGAN_FWO(NTV, NTWV, NGV)                                                                                        \
    if (small && M <= 16384 && g.NG == NGV && g.NT == NTV && !((uintptr_t)wfrag & 15)) {                                 \
      hipLaunchKernelGGL((conv_fwd_once_kernel<NTV, NTWV, NGV>), dim3(cdiv(cdiv(M, 16) * (NTV / NTWV), 4)), block, 0, st, g, xf, wfrag, y, M, act_elu); \
      return launched();                                                                                                 \
    }
    RLGAN_FWO(4, 1, 19)      // 32 -> 64 channels, 3x3
    RLGAN_FWO(2, 1, 19)      // 32 -> 32, 3x3
#undef RLGAN_FWO
#define RLGAN_FWS(NTV, NGV, PFV)                                                                                         \
    if (small && g.NG == NGV && g.NT == NTV) {                                                                           \
      if ((size_t)NGV * NTV * 1024 > 64 * 1024)                                                                          \
        (void)hipFuncSetAttribute((const void*)conv_fwd_static_kernel<NTV, NGV, PFV>, hipFuncAttributeMaxDynamicSharedMemorySize, NGV * NTV * 1024); \
      hipLaunchKernelGGL((conv_fwd_static_kernel<NTV, NGV, PFV>), grid, block, (size_t)NGV * NTV * 1024, st, g, xf, wfrag, y, M, act_elu); \
      return launched();                                                                                                 \
    }
    RLGAN_FWS(2, 19, 5)      // 32 -> 32 channels, 3x3 (the universe head's layers after the first)
    RLGAN_FWS(1, 7, 2)       // 4 -> 8, 5x5 (the doom head's first layer)
    RLGAN_FWS(1, 5, 3)       // 8 -> 16, 3x3
    RLGAN_FWS(2, 10, 3)      // 16 -> 32, 3x3
    RLGAN_FWS(4, 19, 5)      // 32 -> 64, 3x3
#undef RLGAN_FWS
  }
#define RLGAN_FW(NTV)                                                                                              \
  {                                                                                                                \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<NTV, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((conv_fwd_kernel<NTV, S>), grid, block, lds, st, g, src, wfrag, y, M, act_elu);                    \
  }
  switch (g.NT) {
    case 1: RLGAN_FW(1) break;
    case 2: RLGAN_FW(2) break;
    case 3: RLGAN_FW(3) break;
    default: RLGAN_FW(4) break;
  }
#undef RLGAN_FW
  return launched();
}

template <class S>
int launch_wgrad(const ConvG& g, const S& src, const float* dz, const float* y_elu, float* slabs, int n_slabs, float* grad,
                 int n_images, void* stream) {
  if (g.NG > 20) return RLGAN_E_DIMS;                       // five 16-row strips per wave
  const long long Ml = (long long)n_images * g.Ho * g.Wo;
  if (Ml > 0x7fffffffll || ((uintptr_t)dz & 15) || (y_elu && ((uintptr_t)y_elu & 15))) return RLGAN_E_ALIGN;
  const int M = (int)Ml;
  if (g.Cin == 4 && g.NG <= C4W_RT && g.NT <= 2) {            // a first layer with whole images in LDS
    constexpr int XU = 7, ZU = 13;                            // per-thread pieces of an image in flight: 42x42 frames, 20x20x32 maps
    const int HW = g.H * g.W, PP = cdiv(g.Ho * g.Wo, 16) * 16;
    const size_t lds = ((size_t)HW * 4 + (size_t)PP * g.Cout) * sizeof(float) + (size_t)PP * sizeof(unsigned short);
    if (lds <= 79 * 1024 && (size_t)HW * 16 < 65536 && (size_t)4 * C4W_RT * g.NT * 256 <= (size_t)HW * 4 + (size_t)PP * g.Cout &&   // two workgroups per compute unit; 16-bit offsets; the wave sums fit the space of the image and dZ
        HW <= 256 * XU && g.Ho * g.Wo * (g.Cout / 4) <= 256 * ZU) {
      const int wgs = std::min(n_slabs, std::min(n_images, 512));
      const int per = cdiv(n_images, wgs), used = cdiv(n_images, per);
      hipStream_t st = (hipStream_t)stream;
      if (g.NT == 1) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_wgrad_c4_kernel<1, XU, ZU, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((conv_wgrad_c4_kernel<1, XU, ZU, S>), dim3(used), dim3(256), lds, st, g, src, dz, y_elu, slabs, n_images, per);
      } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_wgrad_c4_kernel<2, XU, ZU, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((conv_wgrad_c4_kernel<2, XU, ZU, S>), dim3(used), dim3(256), lds, st, g, src, dz, y_elu, slabs, n_images, per);
      }
      if (const int rc = launched()) return rc;
      return reduce_slabs(slabs, used, g.NG * 16, g.NT * 16, g.K, g.Cout, grad, st);
    }
  }
  static const bool no_img = getenv("RLGAN_NO_WGRAD_IMG") != nullptr;     // measurement switch
  if (g.Cin == 32 && g.k == 3 && g.W == 20 && g.NT == 2 && g.Cout == 32 && g.NG == 19 && n_images >= 1024 && !no_img) {
    // whole padded images in LDS, two buffers: four waves multiply while four fill
    constexpr int NG16 = 6;                                   // up to 96 output pixels an image
    const int HW = g.H * g.W, PP = NG16 * 16, PS = g.Cin + IMG_PAD;
    const size_t lds = (size_t)2 * ((size_t)HW * PS + (size_t)PP * g.Cout) * sizeof(float);
    if (cdiv(g.Ho * g.Wo, 16) == NG16 && lds <= 150 * 1024 && HW * (g.Cin / 4) <= IMG_FILL * 64 * IMG_XU &&
        g.Ho * g.Wo * (g.Cout / 4) <= IMG_FILL * 64 * IMG_ZU) {
      const int wgs = std::min(n_slabs, std::min(n_images, 256));
      const int per = cdiv(n_images, wgs), used = cdiv(n_images, per);
      hipStream_t st = (hipStream_t)stream;
      (void)hipFuncSetAttribute((const void*)conv_wgrad_img_kernel<2, NG16, 20, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((conv_wgrad_img_kernel<2, NG16, 20, S>), dim3(used), dim3(IMG_WAVES * 64), lds, st, g, src, dz, y_elu, slabs, n_images, per);
      if (const int rc = launched()) return rc;
      return reduce_slabs(slabs, used, g.NG * 16, g.NT * 16, g.K, g.Cout, grad, st);
    }
  }
  if (g.NG <= 8 && g.NT <= 2 && M / n_slabs >= 256) {        // few patch elements and long pixel ranges: one slab per wave, no workgroup barriers
    const int KP = g.NG * 16, NP = g.NT * 16;
    const int per = cdiv(cdiv(M, n_slabs), 16) * 16, used = cdiv(M, per);
    const int ldp = KP % 32 == 16 ? KP : KP + 16, ldz = NP % 32 == 16 ? NP : NP + 16;
    const size_t lds = (size_t)(KP / 4 + 4 * 16 * (ldp + ldz)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv(used, 4)), block(256);
    if (g.NT == 1) {
      hipLaunchKernelGGL((conv_wgrad_small_kernel<1, 8, S>), grid, block, lds, st, g, src, dz, y_elu, slabs, M, per);
    } else {
      hipLaunchKernelGGL((conv_wgrad_small_kernel<2, 8, S>), grid, block, lds, st, g, src, dz, y_elu, slabs, M, per);
    }
    if (const int rc = launched()) return rc;
    return reduce_slabs(slabs, used, KP, NP, g.K, g.Cout, grad, st);
  }
  const int per = cdiv(cdiv(M, n_slabs), WG_PIX) * WG_PIX;  // pixels per workgroup, a multiple of the staging step
  const int used = cdiv(M, per);                            // slabs actually written (<= n_slabs)
  const int KP = g.NG * 16, NP = g.NT * 16;
  const size_t lds = (size_t)(WG_PIX * (KP + 4) + WG_PIX * (NP + 4) + KP / 4) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(used), block(256);
#define RLGAN_WG(NTV)                                                                                              \
  {                                                                                                                \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<NTV, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((conv_wgrad_kernel<NTV, S>), grid, block, lds, st, g, src, dz, y_elu, slabs, M, per);               \
  }
  switch (g.NT) {
    case 1: RLGAN_WG(1) break;
    case 2: RLGAN_WG(2) break;
    case 3: RLGAN_WG(3) break;
    default: RLGAN_WG(4) break;
  }
#undef RLGAN_WG
  if (const int rc = launched()) return rc;
  return reduce_slabs(slabs, used, KP, NP, g.K, g.Cout, grad, st);
}

// uint8 source of a first layer: argument checks shared by the two entry points
inline int make_u8(const ConvG& g, const uint8_t* ring, int64_t slots, const int64_t* idx, int n_images, float scale, SrcU8& src) {
  if (!ring || slots <= 0) return RLGAN_E_ARG;
  if ((uintptr_t)ring & 3) return RLGAN_E_ALIGN;
  if (!idx && n_images > slots) return RLGAN_E_ARG;          // without an index array image i is slot i
  src = SrcU8{ring, idx, (long long)slots, (size_t)g.H * g.W * g.Cin, scale};
  return 0;
}

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
  return launch_fwd(g, SrcF32{x, (size_t)g.H * g.W * g.Cin}, wfrag, y, n_images, act_elu, stream);
}

int rlgan_conv_forward_u8(const rlgan_conv_t* conv, const uint8_t* ring, int64_t slots, const int64_t* idx, float scale,
                          const float* wfrag, float* y, int n_images, int act_elu, void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!wfrag || !y || n_images <= 0) return RLGAN_E_ARG;
  SrcU8 src;
  if (const int rc = make_u8(g, ring, slots, idx, n_images, scale, src)) return rc;
  return launch_fwd(g, src, wfrag, y, n_images, act_elu, stream);
}

int rlgan_elu_backward(const float* dy, const float* y, float* dz, int64_t n, void* stream) {
  if (!dy || !y || !dz || n <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, (size_t)n);
  return launched();
}

// x_elu (optional): dx leaves multiplied by ELU'(x_elu); *fused says whether the kernel that ran did that itself
static int conv_dx_launch(const ConvG& g, const float* dz, const float* y_elu, const float* w_canon, const float* wdx, float* dx,
                          const float* x_elu, bool* fused, int n_images, void* stream) {
  *fused = false;
  if (g.Cin == 4 && w_canon && g.s <= 4) {
    C4Off o;
    int most = 0;
    for (int ph = 0; ph < g.s * g.s; ++ph) {
      o.na[ph] = (g.H - ph / g.s + g.s - 1) / g.s; o.nb[ph] = (g.W - ph % g.s + g.s - 1) / g.s;
      o.dper[ph] = make_div(o.na[ph] * o.nb[ph]); o.dnb[ph] = make_div(o.nb[ph]);
      most = std::max(most, o.na[ph] * o.nb[ph]);
    }
    if ((long long)n_images * most > 0x7fffffffll || (((uintptr_t)dx | (uintptr_t)dz | (uintptr_t)y_elu) & 15)) return RLGAN_E_ALIGN;
    const dim3 grid((unsigned)cdiv(n_images * most, 256), g.s * g.s), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t zlds = (size_t)g.Ho * g.Wo * (g.Cout + 1) * sizeof(float);
    if ((g.Cout == 32 || g.Cout == 8) && zlds <= 64 * 1024) {
      const dim3 gi(n_images);
      if (g.Cout == 32) hipLaunchKernelGGL(conv_dx_c4_lds_kernel<32>, gi, block, zlds, st, g, o, dz, y_elu, w_canon, dx);
      else hipLaunchKernelGGL(conv_dx_c4_lds_kernel<8>, gi, block, zlds, st, g, o, dz, y_elu, w_canon, dx);
      return launched();
    }
    if (g.Cout == 32) hipLaunchKernelGGL(conv_dx_c4_kernel<32>, grid, block, 0, st, g, o, dz, y_elu, w_canon, dx, n_images);
    else if (g.Cout == 8) hipLaunchKernelGGL(conv_dx_c4_kernel<8>, grid, block, 0, st, g, o, dz, y_elu, w_canon, dx, n_images);
    else hipLaunchKernelGGL(conv_dx_c4_kernel<0>, grid, block, 0, st, g, o, dz, y_elu, w_canon, dx, n_images);
    return launched();
  }
  if (y_elu && !wdx) return RLGAN_E_ARG;
  if (wdx) {
    DxOff offs;
    if (dx_layout(g, &offs) == 0) return RLGAN_E_DIMS;
    const int ntc = cdiv(g.Cin, 16);
    const size_t zimg = (size_t)g.Ho * g.Wo * (g.Cout + 4) * sizeof(float), tabs = (size_t)2 * g.s * g.s * std::max(1, offs.max_ng) * 4 * sizeof(int);
    // images per workgroup: a class's kernel fragments are fetched once per workgroup, so several small images share them
    // (with few images, fewer per workgroup: at a batch of 32 four images each left eight workgroups for the whole chip)
    int ipw = (int)std::max<size_t>(1, std::min<size_t>(8, (48 * 1024 - tabs) / zimg));
    while (ipw > 1 && cdiv(n_images, ipw) < 512) --ipw;
    const size_t ilds = ipw * zimg + (g.Cout + 4) * sizeof(float) + tabs;    // the images, a row of zeros, the tap tables
    const bool few = n_images < 64;     // not even a workgroup per four compute units: the per-pixel-group kernel below spreads wider
    if (!few && ntc <= 2 && offs.max_ng * ntc <= DXI_MAXF && ilds <= 64 * 1024 && !(((uintptr_t)dx | (uintptr_t)dz | (uintptr_t)y_elu | (uintptr_t)x_elu) & 15)) {
      // pixels no class covers (stride beyond the kernel) keep the zeros of the other kernel's classes: every class
      // with taps writes all of its pixels, a class without taps (NG == 0) must still be cleared
      bool all_have_taps = true;
      for (int cls = 0; cls < g.s * g.s; ++cls) all_have_taps = all_have_taps && dx_class(g, cls).NG > 0;
      if (all_have_taps) {
        hipStream_t st = (hipStream_t)stream;
        const dim3 gi(cdiv(n_images, ipw));
        if (ntc == 1) hipLaunchKernelGGL(conv_dx_img_kernel<1>, gi, dim3(256), ilds, st, g, offs, dz, y_elu, wdx, dx, n_images, ipw, x_elu);
        else hipLaunchKernelGGL(conv_dx_img_kernel<2>, gi, dim3(256), ilds, st, g, offs, dz, y_elu, wdx, dx, n_images, ipw, x_elu);
        *fused = true;
        return launched();
      }
    }
    const long long per = (long long)cdiv(g.H, g.s) * cdiv(g.W, g.s);          // pixels per image of the largest class
    if ((long long)n_images * per > 0x7fffffffll || (((uintptr_t)dx | (uintptr_t)x_elu) & 15)) return RLGAN_E_ALIGN;
    const long long groups = cdiv((long long)n_images * per, 64);
    const dim3 grid((unsigned)groups, g.s * g.s), block(256);
    const size_t lds = (size_t)std::max(1, offs.max_ng) * 4 * 2 * sizeof(int);
    if (lds > 64 * 1024) return RLGAN_E_DIMS;
    hipStream_t st = (hipStream_t)stream;
    switch (cdiv(g.Cin, 16)) {
      case 1: hipLaunchKernelGGL(conv_dx_kernel<1>, grid, block, lds, st, g, offs, dz, y_elu, wdx, dx, n_images, x_elu); break;
      case 2: hipLaunchKernelGGL(conv_dx_kernel<2>, grid, block, lds, st, g, offs, dz, y_elu, wdx, dx, n_images, x_elu); break;
      case 3: hipLaunchKernelGGL(conv_dx_kernel<3>, grid, block, lds, st, g, offs, dz, y_elu, wdx, dx, n_images, x_elu); break;
      default: hipLaunchKernelGGL(conv_dx_kernel<4>, grid, block, lds, st, g, offs, dz, y_elu, wdx, dx, n_images, x_elu); break;
    }
    *fused = true;
    return launched();
  }
  const size_t total4 = (size_t)n_images * g.H * g.W * (g.Cin / 4);
  hipLaunchKernelGGL(conv_bwd_input_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g, dz, w_canon, dx, total4);
  return launched();
}

int rlgan_conv_backward_input_dz(const rlgan_conv_t* conv, const float* dz, const float* y_elu, const float* w_canon, const float* wdx,
                                 float* dx, const float* x_elu, int n_images, void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!dz || (!w_canon && !wdx) || !dx || n_images <= 0) return RLGAN_E_ARG;
  bool fused = false;
  if (const int rc = conv_dx_launch(g, dz, y_elu, w_canon, wdx, dx, x_elu, &fused, n_images, stream)) return rc;
  if (x_elu && !fused) {      // a kernel without the factor: one elementwise pass over dx in place
    const size_t n = (size_t)n_images * g.H * g.W * g.Cin;
    hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, x_elu, dx, n);
    return launched();
  }
  return 0;
}

int rlgan_conv_backward_input(const rlgan_conv_t* conv, const float* dz, const float* y_elu, const float* w_canon, const float* wdx,
                              float* dx, int n_images, void* stream) {
  return rlgan_conv_backward_input_dz(conv, dz, y_elu, w_canon, wdx, dx, nullptr, n_images, stream);
}

int rlgan_conv_backward_weights(const rlgan_conv_t* conv, const float* x, const float* dz, const float* y_elu, float* slabs,
                                int n_slabs, float* grad, int n_images, void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!x || !dz || !slabs || !grad || n_images <= 0 || n_slabs <= 0) return RLGAN_E_ARG;
  return launch_wgrad(g, SrcF32{x, (size_t)g.H * g.W * g.Cin}, dz, y_elu, slabs, n_slabs, grad, n_images, stream);
}

int rlgan_conv_backward_weights_u8(const rlgan_conv_t* conv, const uint8_t* ring, int64_t slots, const int64_t* idx, float scale,
                                   const float* dz, const float* y_elu, float* slabs, int n_slabs, float* grad, int n_images,
                                   void* stream) {
  ConvG g;
  if (!make_conv(conv, g)) return RLGAN_E_DIMS;
  if (!dz || !slabs || !grad || n_images <= 0 || n_slabs <= 0) return RLGAN_E_ARG;
  SrcU8 src;
  if (const int rc = make_u8(g, ring, slots, idx, n_images, scale, src)) return rc;
  return launch_wgrad(g, src, dz, y_elu, slabs, n_slabs, grad, n_images, stream);
}

}  // extern "C"
