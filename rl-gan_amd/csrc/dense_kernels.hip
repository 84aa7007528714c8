// Fully connected head of the pixel encoders: y = elu([x 1] * [W ; b]) and its two gradients
// (`tf.layers.dense(s, 256, activation=tf.nn.elu, name='fc')` Conv_model.py:24; `dense1` + elu
// Conv_model.py:52-56).  One LDS-tiled fp32 GEMM on v_mfma_f32_16x16x4_f32 serves all three products;
// they differ in which operand dimension is contiguous in memory and in where the constant 1 of the
// bias sits:
//   forward   Y[M, N]   = [X 1][M, K+1] * Wb[K+1, N]         A k-contiguous (+ ones column),  B n-contiguous
//   input     dX[M, K]  = dZ[M, N] * Wb[0:K, :]^T             A k-contiguous,                  B k-contiguous
//   weights   dWb[K+1,N] = [X 1]^T[K+1, M] * dZ[M, N]         A m-contiguous (+ ones row),     B n-contiguous
// Wb is the canonical parameter block, the [K, N] kernel followed by the bias row.  gfx950 only.
#include "rlgan_device.h"
#include "rlgan_host.h"
#include "fastdiv.h"
#include <algorithm>
#include <cstdint>
#include <type_traits>

using namespace rlgan;

namespace {

#ifndef RLGAN_BK
#define RLGAN_BK 32
#endif
constexpr int BM = 64, BN = 64, BK = RLGAN_BK;   // workgroup tile of C and the reduction step
constexpr int F4R = BK / 4, TL = BK / 16;   // float4 per tile row; float4 per thread and operand tile
constexpr int LDT = BK + 4;                 // LDS row: 36 floats, the 16 rows of an MFMA lane group fall on disjoint banks
constexpr int TILE_FLOATS = 64 * LDT;

// An operand that is the cols matrix of a transposed convolution without the matrix in memory:
//   cols[(n, iy, ix)][(ky, kx, c)] = z[((n * Ho + iy * s + ky) * Wo + ix * s + kx) * Co + c]
// -- an element's address is a row part plus a column part.  Used for whole tiles only (the host checks the extents),
// with Co a multiple of the tile's extent along (ky, kx, c), so that the tap of a tile is uniform.
struct ColsView {
  int P, w;                   // input pixels per image, per input row
  FastDiv dP, dw, dk;
  int img, rowstep, colstep;  // floats per output image, per s output rows, per s output pixels
  int Co, k, Wo;
  __device__ __forceinline__ int row_off(int r) const {
    const int n = fdiv(r, dP), p = r - n * P, iy = fdiv(p, dw), ix = p - iy * w;
    return n * img + iy * rowstep + ix * colstep;
  }
  // e0: first column of the tile (uniform), de: this lane's distance from it (< Co - e0 % Co)
  __device__ __forceinline__ int col_off(int e0, int de) const {
    const int tap = e0 / Co, ky = fdiv(tap, dk), kx = tap - ky * k;
    return (ky * Wo + kx) * Co + (e0 - tap * Co) + de;
  }
};

struct GemmArgs {
  const float* a; const float* b; float* c;
  int lda, ldb, ldc;
  int M, N, K;          // C is M x N; K = reduction length (a ones index included)
  int a_x, a_k;         // extents of A in memory along its non-reduction / reduction dimension
  int b_x, b_k;         // the same for B
  int ones_x, ones_k;   // A reads as 1 at this non-reduction (reduction) index; -1: none
  int act;              // epilogue: 1 = ELU, 2 = ReLU, 3 = relu6, 4 = tanh
  int quads;            // every aligned group of four along the contiguous dimension of A and of B is inside or outside as a whole
  int k_split;          // reduction indices per blockIdx.y (a multiple of BK); with more than one split the
  size_t slab;          // raw partial products go to c + blockIdx.y * slab and split_sum_kernel finishes
  ColsView view;        // of operand A (gemm_kernel<.., .., 1>) or B (<.., .., 2>)
};

// One operand tile, 64 (x: rows of A / columns of B) by BK (k), as two float4 per thread.
//   KCONT: memory is [x][k], a float4 holds four consecutive k of one x
//   else:  memory is [k][x], a float4 holds four consecutive x of one k
// Out-of-range elements read as 0, the ones index as 1 (operand A only: pass -1 otherwise).  This is the form for
// tiles on an operand's border.  It has no branch around its loads either: every element is fetched from an address
// clamped into the operand (four 4-byte loads, since neither the row end nor the alignment of a clamped address
// allows a 16-byte one) and what lies outside is replaced afterwards.  With conditions around the loads a wave waits
// for each of them where the branches join, and a product whose tiles are all border tiles (an output 4 to 16
// columns wide) spent a memory latency per element group: 19-30 us for layers of a few thousand multiply-adds per row.
template <bool KCONT>
__device__ __forceinline__ void tile_load(f32x4 (&r)[TL], const float* __restrict__ p, int ld, int x0, int k0, int nx, int nk,
                                          int ones_x, int ones_k) {
#pragma unroll
  for (int h = 0; h < TL; ++h) {
    const int u = threadIdx.x + 256 * h;
    int x, k;
    if (KCONT) { x = x0 + u / F4R; k = k0 + (u % F4R) * 4; }
    else       { k = k0 + (u >> 4); x = x0 + (u & 15) * 4; }
    f32x4 v;
    if (KCONT) {
      const float* row = p + (size_t)min(x, nx - 1) * ld;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = row[min(k + i, nk - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (x >= nx || k + i >= nk) v[i] = 0.f;
        if (k + i == ones_k || x == ones_x) v[i] = 1.f;
      }
    } else {
      const float* row = p + (size_t)min(k, nk - 1) * ld;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = row[min(x + i, nx - 1)];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (k >= nk || x + i >= nx) v[i] = 0.f;
        if (x + i == ones_x || k == ones_k) v[i] = 1.f;
      }
    }
    r[h] = v;
  }
}
// A border tile of an operand whose contiguous extent is a multiple of four (every layer width of the pixel networks):
// an aligned group of four is inside or outside as a whole, so one 16-byte load from a clamped address and one
// select per group do.  (With the element form a product paid for its border tiles with four times the loads and
// twice the vector instructions per step: 4096 x 513 -> 500 ran 39 us against 30 us for -> 512.)
template <bool KCONT>
__device__ __forceinline__ void tile_load_quads(f32x4 (&r)[TL], const float* __restrict__ p, int ld, int x0, int k0, int nx, int nk,
                                                int ones_x, int ones_k) {
#pragma unroll
  for (int h = 0; h < TL; ++h) {
    const int u = threadIdx.x + 256 * h;
    int x, k;
    if (KCONT) { x = x0 + u / F4R; k = k0 + (u % F4R) * 4; }
    else       { k = k0 + (u >> 4); x = x0 + (u & 15) * 4; }
    const bool in = x < nx && k < nk;
    f32x4 v = KCONT ? *(const f32x4*)(p + (size_t)min(x, nx - 1) * ld + min(k, nk - 4))
                    : *(const f32x4*)(p + (size_t)min(k, nk - 1) * ld + min(x, nx - 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!in) v[i] = 0.f;
      if (KCONT ? (k + i == ones_k || x == ones_x) : (x + i == ones_x || k == ones_k)) v[i] = 1.f;
    }
    r[h] = v;
  }
}
// The same for a tile that lies wholly inside the operand and holds no ones index: no per-lane condition.
// (Conditions around the loads make the wave wait for them where the branches join, i.e. before the
// multiplications they are meant to overlap with; the main loop below therefore runs on this form alone.)
template <bool KCONT>
__device__ __forceinline__ void tile_load_inside(f32x4 (&r)[TL], const float* __restrict__ p, int ld, int x0, int k0) {
#pragma unroll
  for (int h = 0; h < TL; ++h) {
    const int u = threadIdx.x + 256 * h;
    r[h] = KCONT ? *(const f32x4*)(p + (size_t)(x0 + u / F4R) * ld + k0 + (u % F4R) * 4)
                 : *(const f32x4*)(p + (size_t)(k0 + (u >> 4)) * ld + x0 + (u & 15) * 4);
  }
}
// The same through a ColsView: KCONT = rows of the view are the x, its columns the k (operand A of the input gradient);
// otherwise its rows are the k and its columns the x (operand B of the kernel gradient).
template <bool KCONT>
__device__ __forceinline__ void tile_load_view(f32x4 (&r)[TL], const float* __restrict__ p, const ColsView& v, int x0, int k0) {
#pragma unroll
  for (int h = 0; h < TL; ++h) {
    const int u = threadIdx.x + 256 * h;
    r[h] = KCONT ? *(const f32x4*)(p + v.row_off(x0 + u / F4R) + v.col_off(k0, (u % F4R) * 4))
                 : *(const f32x4*)(p + v.row_off(k0 + (u >> 4)) + v.col_off(x0, (u & 15) * 4));
  }
}
// Position of reduction index k in row x of an LDS tile: the row's eight 4-float groups are permuted by the row's
// 16-row block (k ^ 0, 4, 8 or 12).  A 16-byte operand read stays one aligned group; the transposing store below,
// whose sixteen x-quads per k would otherwise fall on four banks four deep, spreads over all 64 banks.
// (Rows of 32 floats without padding, the groups permuted by (x >> 1) & 7 instead: the weight product with few slabs
// runs a third faster, 484 -> 324 us unsplit at 8192 x 2593 -> 256, but at the slab counts in use it is 2 % slower and
// the forward product 6 % slower; the LDS counters show conflicts in a third to a half of this kernel's LDS cycles
// with either layout.)
__device__ __forceinline__ int swz(int x, int k) { return k ^ (((x >> 4) & 3) << 2); }
// the registers of tile_load into an LDS tile [64 x][LDT], k contiguous
template <bool KCONT>
__device__ __forceinline__ void tile_store(float* t, const f32x4 (&r)[TL]) {
#pragma unroll
  for (int h = 0; h < TL; ++h) {
    const int u = threadIdx.x + 256 * h;
    if (KCONT) {
      *(f32x4*)(t + (u / F4R) * LDT + swz(u / F4R, (u % F4R) * 4)) = r[h];
    } else {
      const int k = u >> 4, x = (u & 15) * 4;        // x .. x+3 lie in one 16-row block
#pragma unroll
      for (int i = 0; i < 4; ++i) t[(x + i) * LDT + swz(x, k)] = r[h][i];
    }
  }
}

// C tile of one workgroup: four waves, 32 x 32 each (2 x 2 MFMA tiles); the next reduction step's
// operands travel global -> registers while the current one is multiplied out of LDS.  (Two or three steps in
// flight through further register sets: 1 to 2 % slower at batch 4096 and no faster at batch 32, where a product with
// few output tiles is a chain of memory round trips per workgroup -- what helps there is cutting the reduction into
// slabs, which every product can now ask for.)
template <bool A_KCONT, bool B_KCONT, int VIEW = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];           // 4 * TILE_FLOATS: A and B tiles, two stages
  // Workgroups go to the eight XCDs in turn, each with an L2 of its own: workgroup f of the launch takes the f/8-th
  // of the run of (tile, slab) pairs that belongs to XCD f%8, so that the column tiles that share the rows of A, and
  // the row tiles that share a slab of B, meet in one L2.
#ifndef RLGAN_GEMM_NO_XCD
  const unsigned flat = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
  const unsigned xcd = flat & 7, pair = xcd * (total >> 3) + min(xcd, total & 7) + (flat >> 3);
  const int tile = (int)(pair % gridDim.x), zi = (int)(pair / gridDim.x);
#else
  const int tile = (int)blockIdx.x, zi = (int)blockIdx.y;
#endif
  const int ntn = cdiv(g.N, BN);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int q4 = (lane >> 4) * 4, j = lane & 15;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int ax = g.a_x, ak = g.a_k;
  f32x4 ra[TL], rb[TL];
  f32x4 ra0[TL], rb0[TL];                  // second register set of the forward product (tiles two steps ahead)
  f32x4 acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int kb = zi * g.k_split;                                    // this slab's reduction range
  const int nkt = cdiv(min(g.K - kb, g.k_split), BK);
  float* cz = g.c + (size_t)zi * g.slab;
  // leading reduction steps whose A and B tiles lie wholly inside the operands (no edge, no ones index)
  int nin = 0;
  if (m0 + BM <= ax && n0 + BN <= g.b_x && (g.ones_x < m0 || g.ones_x >= m0 + BM)) {
    int kin = min(ak, g.b_k);
    if (g.ones_k >= 0) kin = min(kin, g.ones_k);
    nin = max(0, min(nkt, (kin - kb) / BK));
  }
  if constexpr (VIEW != 0) nin = nkt;                               // a view is launched on whole tiles only
  // mode 0: inside, 1: border by groups of four, 2: border by elements.  The mode is chosen outside the loops: a
  // branch between the loads and the products would make the wave wait for the loads where its arms join.
  auto load = [&](int kt, auto mode) {
    if constexpr (mode.value == 0) {
      if constexpr (VIEW == 1) tile_load_view<A_KCONT>(ra, g.a, g.view, m0, kb + kt * BK);
      else tile_load_inside<A_KCONT>(ra, g.a, g.lda, m0, kb + kt * BK);
      if constexpr (VIEW == 2) tile_load_view<B_KCONT>(rb, g.b, g.view, n0, kb + kt * BK);
      else tile_load_inside<B_KCONT>(rb, g.b, g.ldb, n0, kb + kt * BK);
    } else if constexpr (mode.value == 1) {
      tile_load_quads<A_KCONT>(ra, g.a, g.lda, m0, kb + kt * BK, ax, ak, g.ones_x, g.ones_k);
      tile_load_quads<B_KCONT>(rb, g.b, g.ldb, n0, kb + kt * BK, g.b_x, g.b_k, -1, -1);
    } else {
      tile_load<A_KCONT>(ra, g.a, g.lda, m0, kb + kt * BK, ax, ak, g.ones_x, g.ones_k);
      tile_load<B_KCONT>(rb, g.b, g.ldb, n0, kb + kt * BK, g.b_x, g.b_k, -1, -1);
    }
  };
  using Inside = std::integral_constant<int, 0>; using Quads = std::integral_constant<int, 1>; using Elements = std::integral_constant<int, 2>;
  auto multiply = [&](int kt) {
    const float* as = lds + (kt & 1) * 2 * TILE_FLOATS;
    const float* bs = as + TILE_FLOATS;
#pragma unroll
    for (int kg = 0; kg < BK / 16; ++kg) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        a[s] = *(const f32x4*)(as + (wm + 16 * s + j) * LDT + swz(wm + 16 * s, 16 * kg + q4));
        b[s] = *(const f32x4*)(bs + (wn + 16 * s + j) * LDT + swz(wn + 16 * s, 16 * kg + q4));
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][m], b[t][m], acc[s][t], 0, 0, 0);
        }
      }
    }
  };
  auto stash = [&](int kt) {
    float* an = lds + (kt & 1) * 2 * TILE_FLOATS;
    tile_store<A_KCONT>(an, ra);
    tile_store<B_KCONT>(an + TILE_FLOATS, rb);
  };
  if (nin > 0) load(0, Inside{});
  else if (g.quads) load(0, Quads{});
  else load(0, Elements{});
  stash(0);
  __syncthreads();
  int kt = 0;
  // The forward product asks for its tiles two steps ahead, through a second register set (free: the border forms of
  // the loads need more registers than the bulk loop does): 8192 x 2593 -> 256 109 -> 104 us.  The other two products
  // ran 2 to 3 % slower that way and keep one step.
  bool pre = false;
  if constexpr (A_KCONT && !B_KCONT) {
    if (nin >= 4) {
      auto load0 = [&](int k2) {
        tile_load_inside<A_KCONT>(ra0, g.a, g.lda, m0, kb + k2 * BK);
        tile_load_inside<B_KCONT>(rb0, g.b, g.ldb, n0, kb + k2 * BK);
      };
      auto stash0 = [&](int k2) {
        float* an = lds + (k2 & 1) * 2 * TILE_FLOATS;
        tile_store<A_KCONT>(an, ra0);
        tile_store<B_KCONT>(an + TILE_FLOATS, rb0);
      };
      load(1, Inside{});
      for (; kt + 3 < nin; kt += 2) {
        load0(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        multiply(kt);
        __builtin_amdgcn_sched_barrier(0);
        stash(kt + 1);
        __syncthreads();
        load(kt + 3, Inside{});
        __builtin_amdgcn_sched_barrier(0);
        multiply(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        stash0(kt + 2);
        __syncthreads();
      }
      pre = true;                          // step kt + 1 is in the first register set already
    }
  }
  for (; kt + 1 < nin; ++kt) {          // the bulk: the next step's tiles are inside, nothing but loads, products, stores
    if (!pre) load(kt + 1, Inside{});
    pre = false;
    __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the products (the scheduler would sink them to their use)
    multiply(kt);
    __builtin_amdgcn_sched_barrier(0);
    stash(kt + 1);
    __syncthreads();
  }
  auto edge = [&](auto mode) {          // the edge steps (and everything, for a tile on the operands' border): same pipeline
    for (; kt < nkt; ++kt) {
      if (kt + 1 < nkt) load(kt + 1, mode);
      __builtin_amdgcn_sched_barrier(0);
      multiply(kt);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nkt) stash(kt + 1);
      __syncthreads();
    }
  };
  if (g.quads) edge(Quads{});
  else edge(Elements{});
  // lane (q, j) of an accumulator tile holds rows 4q..4q+3 of column j
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + wn + 16 * t + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + 16 * s + q4 + r;
        if (row < g.M && col < g.N) {
          float v = acc[s][t][r];
          if (g.act == 1) v = elu_f(v);
          else if (g.act == 2) v = fmaxf(v, 0.f);
          else if (g.act == 3) v = __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
          else if (g.act == 4) v = tanhf(v);
          cz[(size_t)row * g.ldc + col] = v;
        }
      }
    }
  }
}

// c[row][col] = act(slab 0 + slab 1 + ...), slabs dense [M][N], added in slab order
__global__ __launch_bounds__(256) void split_sum_kernel(const float* __restrict__ slabs, int n_slabs, size_t slab, int M, int N,
                                                        float* __restrict__ c, int ldc, int act) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * N) return;
  // slab order, eight loads in flight at a time (one load per addition is a chain of memory latencies)
  float v = 0.f;
  for (int z0 = 0; z0 < n_slabs; z0 += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slabs[(size_t)min(z0 + u, n_slabs - 1) * slab + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (z0 + u < n_slabs) v += t[u];
  }
  if (act == 1) v = elu_f(v);
  else if (act == 2) v = fmaxf(v, 0.f);
  else if (act == 3) v = __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
  else if (act == 4) v = tanhf(v);
  c[(i / N) * ldc + i % N] = v;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline int launched() { return (int)hipGetLastError(); }

// (The sum over the slabs by the workgroup that finishes a tile last -- a counter per tile, partial tiles stored and read
// back at device scope, no second launch -- was built and measured: slower everywhere, 0.157 -> 0.221 ms for the 84x84
// update at batch 32, 6.74 -> 6.85 ms for the pixel iteration at 4096: 4-byte device-scope stores and loads cost more
// than the launch they save.)
// The reduction is cut into n_slabs ranges when the C tiles alone would leave the chip idle or a
// workgroup alone on its compute unit with nothing to cover its loads: partial products to `slabs`
// ([n_slabs][M][N]), then one pass adds them in order and applies the activation.
template <bool A_KCONT, bool B_KCONT, int VIEW = 0>
int launch_gemm(GemmArgs g, float* slabs, int n_slabs, void* stream) {
  if (!g.a || !g.b || !g.c || g.M <= 0 || g.N <= 0 || g.K <= 0 || n_slabs < 0 || n_slabs > 256 || (n_slabs > 1 && !slabs))
    return RLGAN_E_ARG;
  // float4 loads: 16-byte aligned bases and leading dimensions that are multiples of four
  if (!aligned16(g.a) || !aligned16(g.b) || (g.lda & 3) || (g.ldb & 3)) return RLGAN_E_ALIGN;
  // a view is read by whole tiles only
  if (VIEW != 0 && ((g.M % BM) || (g.N % BN) || (g.K % BK) || g.ones_x >= 0 || g.ones_k >= 0)) return RLGAN_E_DIMS;
  const long long blocks = (long long)cdiv(g.M, BM) * cdiv(g.N, BN);
  if (blocks > 0x7fffffffll) return RLGAN_E_ARG;
  g.k_split = g.K; g.slab = 0;
  g.quads = (A_KCONT ? g.a_k : g.a_x) % 4 == 0 && (B_KCONT ? g.b_k : g.b_x) % 4 == 0;
  constexpr size_t GEMM_LDS = (size_t)4 * TILE_FLOATS * sizeof(float);
  if (GEMM_LDS > 64 * 1024) (void)hipFuncSetAttribute((const void*)gemm_kernel<A_KCONT, B_KCONT, VIEW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS);
  int nz = 1;
  if (n_slabs > 1) {
    g.k_split = BK * cdiv(cdiv(g.K, n_slabs), BK);
    nz = cdiv(g.K, g.k_split);
  }
  if (nz == 1) {
    hipLaunchKernelGGL((gemm_kernel<A_KCONT, B_KCONT, VIEW>), dim3((unsigned)blocks), dim3(256), GEMM_LDS, (hipStream_t)stream, g);
    return launched();
  }
  float* out = g.c; const int ldc = g.ldc, act = g.act;
  g.c = slabs; g.ldc = g.N; g.act = 0; g.slab = (size_t)g.M * g.N;
  hipLaunchKernelGGL((gemm_kernel<A_KCONT, B_KCONT, VIEW>), dim3((unsigned)blocks, nz), dim3(256), GEMM_LDS, (hipStream_t)stream, g);
  hipLaunchKernelGGL(split_sum_kernel, dim3((unsigned)cdiv((long long)g.M * g.N, 256)), dim3(256), 0, (hipStream_t)stream,
                     slabs, nz, g.slab, g.M, g.N, out, ldc, act);
  return launched();
}

}  // namespace

extern "C" {

int rlgan_dense_slabs(int m_out, int n_out, int k_red) {
  if (m_out <= 0 || n_out <= 0 || k_red <= 0) return RLGAN_E_ARG;
  const long long tiles = (long long)cdiv(m_out, BM) * cdiv(n_out, BN);
  if (m_out <= BM) {
    // a result a single row tile high (a batch of 32 or 64): every product is a few workgroups walking a chain of memory
    // round trips, so two reduction steps per slab already pay (the pixel iteration at batch 32 ran 6 % faster)
    const int by_chip = (int)std::min<long long>(256, cdiv(1024, tiles));
    return std::max(1, std::min(by_chip, std::max(1, k_red / (2 * 32))));
  }
  // From a sweep of the pixel networks' products at 4096 and 8192 rows (tools/gemm_slabs.py): a result with two
  // workgroups' worth of tiles per compute unit is not cut at all (the sum over the slabs would cost more than the
  // shorter loops save); below that, about four workgroups per compute unit, eight for a result of few tiles (the weight
  // gradients, whose partial results are small); and slabs of at least eight reduction steps, four when the result is
  // narrow.  The step bound depends on the layer alone, not on the batch: where it is the smaller of the two (the
  // generator's layers), a row's result is the same bits whatever the batch.
  if (tiles >= 512) return 1;
  const int steps = cdiv(n_out, BN) >= 4 ? 8 : 4;
  const int by_work = std::max(1, k_red / (steps * 32));
  const int by_chip = (int)std::min<long long>(32, cdiv(tiles < 200 ? 2048 : 1024, tiles));
  return std::max(1, std::min(by_chip, by_work));
}

int rlgan_dense_forward(const float* x, int ldx, const float* wb, float* y, int ldy, int m, int k, int n, int act,
                        float* slabs, int n_slabs, void* stream) {
  if (ldx < k || ldy < n || act < 0 || act > 4) return RLGAN_E_ARG;
  GemmArgs g{x, wb, y, ldx, n, ldy, m, n, k + 1, /*a_x*/ m, /*a_k*/ k, /*b_x*/ n, /*b_k*/ k + 1, /*ones_x*/ -1, /*ones_k*/ k, act, 0, 0};
  return launch_gemm<true, false>(g, slabs, n_slabs, stream);
}

int rlgan_dense_backward_input_split(const float* dz, int lddz, const float* wb, float* dx, int lddx, int m, int k, int n,
                                     float* slabs, int n_slabs, void* stream) {
  if (lddz < n || lddx < k) return RLGAN_E_ARG;
  // B[kk = hidden unit][x = input feature] = Wb[x][kk]: rows of Wb are the x, kk is contiguous
  GemmArgs g{dz, wb, dx, lddz, n, lddx, m, k, n, m, n, /*b_x*/ k, /*b_k*/ n, -1, -1, 0, 0, 0};
  return launch_gemm<true, true>(g, slabs, n_slabs, stream);
}
int rlgan_dense_backward_input(const float* dz, int lddz, const float* wb, float* dx, int lddx, int m, int k, int n, void* stream) {
  return rlgan_dense_backward_input_split(dz, lddz, wb, dx, lddx, m, k, n, nullptr, 1, stream);
}

int rlgan_dense_backward_weights(const float* x, int ldx, const float* dz, int lddz, float* gwb, int m, int k, int n,
                                 float* slabs, int n_slabs, void* stream) {
  if (ldx < k || lddz < n) return RLGAN_E_ARG;
  // A[row of dWb = input feature or the ones row][kk = batch row] = x[kk][feature]
  GemmArgs g{x, dz, gwb, ldx, lddz, n, k + 1, n, m, /*a_x*/ k, /*a_k*/ m, /*b_x*/ n, /*b_k*/ m, /*ones_x*/ k, /*ones_k*/ -1, 0, 0, 0};
  return launch_gemm<false, false>(g, slabs, n_slabs, stream);
}

int rlgan_dense_backward_kernel(const float* x, int ldx, const float* dz, int lddz, float* gw, int m, int k, int n,
                                float* slabs, int n_slabs, void* stream) {
  if (ldx < k || lddz < n) return RLGAN_E_ARG;
  // the same product without the ones row: a k that is a multiple of the 64-row tile gets no extra row of tiles for it
  GemmArgs g{x, dz, gw, ldx, lddz, n, k, n, m, /*a_x*/ k, /*a_k*/ m, /*b_x*/ n, /*b_k*/ m, /*ones_x*/ -1, /*ones_k*/ -1, 0, 0, 0};
  return launch_gemm<false, false>(g, slabs, n_slabs, stream);
}

// The two products of a transposed convolution's backward pass that read the cols matrix, reading dZ = dY * ReLU'(Y)
// through a ColsView instead (rlgan_deconv_gather wrote 151 MB of cols for the 3x3x128 -> 8x8x64 layer at 4096 images and
// both products read them back).  Same tiles, same products in the same order as the cols form: the same bits.
static bool cols_view(const rlgan_deconv_t* d, int n_images, ColsView* v) {
  if (!d || d->h <= 0 || d->w <= 0 || d->k <= 0 || d->stride <= 0 || d->c_in <= 0 || d->c_out <= 0 || n_images <= 0) return false;
  const long long rows = (long long)n_images * d->h * d->w, kk = (long long)d->k * d->k * d->c_out;
  const int Ho = (d->h - 1) * d->stride + d->k, Wo = (d->w - 1) * d->stride + d->k;
  if ((long long)n_images * Ho * Wo * d->c_out > 0x7fffffffll || rows > 0x7fffffffll) return false;
  if ((rows % BM) || (d->c_in % BM) || (kk % BN) || (d->c_out % BN)) return false;   // whole tiles, one tap per tile
  v->P = d->h * d->w; v->w = d->w;
  v->dP = make_div(v->P); v->dw = make_div(d->w); v->dk = make_div(d->k);
  v->img = Ho * Wo * d->c_out; v->rowstep = d->stride * Wo * d->c_out; v->colstep = d->stride * d->c_out;
  v->Co = d->c_out; v->k = d->k; v->Wo = Wo;
  return true;
}

int rlgan_deconv_view_ok(const rlgan_deconv_t* d, int n_images) {
  ColsView v;
  return cols_view(d, n_images, &v) ? 1 : 0;
}

int rlgan_deconv_backward_input_view(const rlgan_deconv_t* d, const float* dz, const float* wmat, float* dx, int n_images,
                                     float* slabs, int n_slabs, void* stream) {
  ColsView v;
  if (!cols_view(d, n_images, &v)) return RLGAN_E_DIMS;
  const int rows = n_images * d->h * d->w, kk = d->k * d->k * d->c_out;
  GemmArgs g{dz, wmat, dx, kk, kk, d->c_in, rows, d->c_in, kk, rows, kk, /*b_x*/ d->c_in, /*b_k*/ kk, -1, -1, 0, 0, 0, 0, v};
  return launch_gemm<true, true, 1>(g, slabs, n_slabs, stream);
}

int rlgan_deconv_backward_kernel_view(const rlgan_deconv_t* d, const float* x, const float* dz, float* gw, int n_images,
                                      float* slabs, int n_slabs, void* stream) {
  ColsView v;
  if (!cols_view(d, n_images, &v)) return RLGAN_E_DIMS;
  const int rows = n_images * d->h * d->w, kk = d->k * d->k * d->c_out;
  GemmArgs g{x, dz, gw, d->c_in, kk, kk, d->c_in, kk, rows, /*a_x*/ d->c_in, /*a_k*/ rows, /*b_x*/ kk, /*b_k*/ rows, -1, -1, 0, 0, 0, 0, v};
  return launch_gemm<false, false, 2>(g, slabs, n_slabs, stream);
}

}  // extern "C"
