// Device-side building blocks shared by the learner and acting kernels (gfx950 only).
//
// Tiling model: a workgroup of 256 threads (4 waves, one per SIMD) owns 16 consecutive
// minibatch rows and walks a whole network for them.  One dense layer for a 16-row tile is
//   Z[16, N] = X[16, K+1] * [W ; b]                 (bias folded in as the row of a constant-1 column)
// done with v_mfma_f32_16x16x4_f32 (exact f32 FMA chains).  The 16-wide column tiles of a layer are
// dealt to the four waves; X lives in LDS (it is shared by the waves); the weights are streamed
// straight from L2 into registers as ready-made B fragments, several k-steps ahead of their use
// (each wave reads a private share of them exactly once, so staging them in LDS would only add a
// round trip).  Layers with a single column tile split K over the waves instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlgan {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_LINEAR = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_LRELU = 3, ACT_TANH = 4 };
enum { NET_ACTOR = 0, NET_CRITIC = 1, NET_GEN = 2, NET_DISC = 3 };

constexpr int HID = 200;        // DDPG.py:82,159 ; Curious_GAN.py:59,121
constexpr int NT_H = 13;        // 16-wide column tiles of a hidden layer
constexpr int NP_H = 208;       // padded hidden width (global work arrays)
constexpr int LD_W = 209;       // LDS stride of a wide activation tile (odd: conflict-free column reads)
constexpr int LD_X0 = 33;       // LDS stride of the network-input tile (32 columns)
constexpr int LD_O = 17;        // LDS stride of a 16-column tile
constexpr int LD_3W = 65;       // LDS stride of a 64-column tile
constexpr int X0_COLS = 32;     // global leading dimension of x0 work arrays
constexpr int ROWS = 16;        // minibatch rows per workgroup
constexpr int THREADS = 256;
constexpr float LRELU_ALPHA = 0.2f;  // Curious_GAN.py:56

// ---------------------------------------------------------------- geometry (host + device)
struct Geo {
  int S, A, K0, N3, NOUT, NT3;
  int K[4], N[4];
  int nsf[4], ntf[4], offf[4];   // forward fragments: k-steps, column tiles, float offset in wf
  int nsb[4], ntb[4], offb[4];   // backward fragments (W^T)
  int kp[4], np[4], goff[4];     // padded gradient image [kp][np] per layer
  int n_params, wf_floats, wb_floats, gpad_floats, ld3;
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__host__ __device__ inline bool make_geo(int kind, int S, int A, Geo& g) {
  if (S < 1 || A < 1) return false;
  g.S = S; g.A = A;
  switch (kind) {
    case NET_ACTOR:  g.K0 = S;         g.N3 = 10; g.NOUT = A; break;
    case NET_CRITIC: g.K0 = S + A;     g.N3 = 10; g.NOUT = 1; break;
    case NET_GEN:    g.K0 = S + A;     g.N3 = 50; g.NOUT = S; break;
    case NET_DISC:   g.K0 = 2 * S + A; g.N3 = 10; g.NOUT = 1; break;
    default: return false;
  }
  if (g.K0 > 16 || g.NOUT > 15) return false;
  g.NT3 = cdiv(g.N3, 16);
  g.K[0] = g.K0; g.N[0] = HID; g.K[1] = HID; g.N[1] = HID; g.K[2] = HID; g.N[2] = g.N3; g.K[3] = g.N3; g.N[3] = g.NOUT;
  int of = 0, ob = 0, og = 0, np_ = 0;
  for (int l = 0; l < 4; ++l) {
    g.nsf[l] = cdiv(g.K[l] + 1, 4); g.ntf[l] = cdiv(g.N[l], 16); g.offf[l] = of; of += g.nsf[l] * g.ntf[l] * 64;
    g.nsb[l] = cdiv(g.N[l], 4);     g.ntb[l] = cdiv(g.K[l], 16); g.offb[l] = ob; ob += g.nsb[l] * g.ntb[l] * 64;
    g.kp[l] = 16 * cdiv(g.K[l] + 1, 16); g.np[l] = 16 * g.ntf[l]; g.goff[l] = og; og += g.kp[l] * g.np[l];
    np_ += (g.K[l] + 1) * g.N[l];
  }
  g.n_params = np_; g.wf_floats = of; g.wb_floats = ob; g.gpad_floats = og; g.ld3 = g.kp[3];
  return true;
}

// float offset of element (k, n) of layer l in the forward-fragment image (k == K[l] is the bias row)
__host__ __device__ inline int wf_offset(const Geo& g, int l, int k, int n) {
  const int s = k >> 2, t = n >> 4, lane = ((k & 3) << 4) | (n & 15);
  return g.offf[l] + (s * g.ntf[l] + t) * 64 + lane;
}
// float offset of W[k][n] of layer l in the backward-fragment image (operand element (row n, col k))
__host__ __device__ inline int wb_offset(const Geo& g, int l, int k, int n) {
  const int s = n >> 2, t = k >> 4, lane = ((n & 3) << 4) | (k & 15);
  return g.offb[l] + (s * g.ntb[l] + t) * 64 + lane;
}

#ifdef __HIPCC__
// ---------------------------------------------------------------- activations
template <int ACT> __device__ __forceinline__ float act_f(float z) {
  if (ACT == ACT_RELU)  return fmaxf(z, 0.f);
  if (ACT == ACT_RELU6) return fminf(fmaxf(z, 0.f), 6.f);
  if (ACT == ACT_LRELU) return fmaxf(LRELU_ALPHA * z, z);
  if (ACT == ACT_TANH)  return tanhf(z);
  return z;
}
// derivative from the POST-activation value h (relu6: 0<z<6 <=> 0<h<6; lrelu: tie at 0 -> alpha,
// tf.maximum's gradient goes to its first argument when alpha*z >= z; tanh: 1-h^2 like TanhGrad)
template <int ACT> __device__ __forceinline__ float act_d(float h) {
  if (ACT == ACT_RELU)  return h > 0.f ? 1.f : 0.f;
  if (ACT == ACT_RELU6) return (h > 0.f && h < 6.f) ? 1.f : 0.f;
  if (ACT == ACT_LRELU) return h > 0.f ? 1.f : LRELU_ALPHA;
  if (ACT == ACT_TANH)  return 1.f - h * h;
  return 1.f;
}
__device__ __forceinline__ float sigmoid_f(float x) {
  if (x >= 0.f) return 1.f / (1.f + expf(-x));
  const float e = expf(x);
  return e / (1.f + e);
}
// tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))  (Curious_GAN.py:33-39)
__device__ __forceinline__ float sigmoid_ce(float x, float z) {
  return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

// ---------------------------------------------------------------- network traits
template <int KIND> struct NetT;
template <> struct NetT<NET_ACTOR>  { static constexpr int A1 = ACT_RELU6, A2 = ACT_RELU6, A3 = ACT_RELU,  A4 = ACT_TANH,   NT3 = 1, L3 = 17; };
template <> struct NetT<NET_CRITIC> { static constexpr int A1 = ACT_RELU6, A2 = ACT_RELU6, A3 = ACT_RELU,  A4 = ACT_LINEAR, NT3 = 1, L3 = 17; };
template <> struct NetT<NET_GEN>    { static constexpr int A1 = ACT_RELU,  A2 = ACT_RELU,  A3 = ACT_RELU,  A4 = ACT_LINEAR, NT3 = 4, L3 = 65; };
template <> struct NetT<NET_DISC>   { static constexpr int A1 = ACT_LRELU, A2 = ACT_LRELU, A3 = ACT_LRELU, A4 = ACT_LINEAR, NT3 = 1, L3 = 17; };

// ---------------------------------------------------------------- LDS carve (floats)
constexpr int PART_FLOATS = 4 * ROWS * 20;           // four waves' partial sums of one 16-column tile
constexpr int WIDE_FLOATS = ROWS * LD_W;
constexpr int T3_FLOATS = ROWS * LD_3W;              // h3 tile (16 or 64 columns)
constexpr int X0_FLOATS = ROWS * LD_X0;
constexpr int O_FLOATS = ROWS * LD_O;
constexpr int PIPE = 8;                              // k-steps of operands kept in flight per wave

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Cycle stamps for diagnostic builds only (tools/stamp_probe.hip defines RLGAN_STAMP); in the library
// build STAMP expands to nothing.
#ifdef RLGAN_STAMP
__shared__ unsigned long long rl_stamps[4][192];
__shared__ int rl_nst[4];
__device__ __forceinline__ void stamp_here(int tag) {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if ((threadIdx.x & 63) == 0 && blockIdx.x == RLGAN_STAMP) {
    const int w = threadIdx.x >> 6;
    const int k = rl_nst[w];
    if (k < 192) { rl_stamps[w][k] = ((unsigned long long)tag << 48) | (t & 0xFFFFFFFFFFFFull); rl_nst[w] = k + 1; }
  }
}
#define STAMP(tag) stamp_here(tag)
#else
#define STAMP(tag)
#endif

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, not for its
// outstanding global stores (the activations written for the weight-gradient kernel are not read
// again inside the launch, so nothing here has to wait for them the way __syncthreads() does).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- MFMA product, software pipelined
// acc[i] += X[16, 4*(s0..s1)] * fragment tile i, i < T.  `xrow` is this lane's A-operand address for
// k-step 0 (LDS), `wl` this lane's B-operand address of tile 0 at k-step 0 (global, L2-resident),
// consecutive tiles are 64 floats apart, consecutive k-steps `sstride` floats.  Operands of PIPE
// k-steps are kept in flight.  The loop body is branch-free: the step count is rounded up to a
// multiple of PIPE and the surplus steps multiply a zero activation by a re-read valid fragment.
template <int T>
__device__ __forceinline__ void mma_pipe(f32x4 (&acc)[T], const float* __restrict__ wl, int sstride,
                                         const float* xrow, int s0, int s1) {
  const int n = s1 - s0;
  if (n <= 0) return;
  const int nblk = (n + PIPE - 1) / PIPE;
  const int last = s1 - 1;
  float a[PIPE], b[PIPE][T];
#pragma unroll
  for (int j = 0; j < PIPE; ++j) {
    const int s = s0 + j, sc = min(s, last);
    const float av = xrow[4 * sc];
    a[j] = s < s1 ? av : 0.f;
#pragma unroll
    for (int i = 0; i < T; ++i) b[j][i] = wl[sc * sstride + i * 64];
  }
  for (int blk = 1; blk < nblk; ++blk) {
#pragma unroll
    for (int j = 0; j < PIPE; ++j) {
#pragma unroll
      for (int i = 0; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][i], acc[i], 0, 0, 0);
      const int s = s0 + blk * PIPE + j, sc = min(s, last);
      const float av = xrow[4 * sc];
      a[j] = s < s1 ? av : 0.f;
#pragma unroll
      for (int i = 0; i < T; ++i) b[j][i] = wl[sc * sstride + i * 64];
    }
  }
#pragma unroll
  for (int j = 0; j < PIPE; ++j) {
#pragma unroll
    for (int i = 0; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][i], acc[i], 0, 0, 0);
  }
}

// One dense layer of a 16-row tile, Z[16, NT*16] = X * fragments, epilogue f(row, col, z) per element.
// Of the NT column tiles, 4*(NT/4) are dealt to the four waves (NT/4 each, whole K range, epilogue
// straight from the accumulators); the NT%4 left-over tiles are computed by all four waves together,
// each taking a quarter of the K range, and their partial sums meet in LDS (`part`) where they are
// added in wave order.  Ends with a workgroup barrier: afterwards every wave sees what f wrote.
template <int NT, class F>
__device__ __forceinline__ void dense16(const float* __restrict__ wp, int nsteps, const float* xs, int ldx,
                                        float* part, F f) {
  constexpr int TPW = NT / 4, REM = NT % 4, LDP = REM * 16 + 4;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const float* xrow = xs + (lane & 15) * ldx + (lane >> 4);
  STAMP(1);
  if constexpr (REM > 0) {
    const int per = (nsteps + 3) >> 2;
    const int s0 = wave * per, s1 = min(nsteps, s0 + per);
    f32x4 accr[REM];
#pragma unroll
    for (int i = 0; i < REM; ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_pipe<REM>(accr, wp + (4 * TPW) * 64 + lane, NT * 64, xrow, s0, s1);
    float* pw = part + wave * ROWS * LDP + ((lane >> 4) * 4) * LDP + (lane & 15);
#pragma unroll
    for (int i = 0; i < REM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[r * LDP + 16 * i] = accr[i][r];
    }
  }
  STAMP(2);
  if constexpr (TPW > 0) {
    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_pipe<TPW>(acc, wp + (wave * TPW) * 64 + lane, NT * 64, xrow, 0, nsteps);
    STAMP(3);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int col = 16 * (wave * TPW + i) + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) f((lane >> 4) * 4 + r, col, acc[i][r]);
    }
  }
  STAMP(4);
  lds_barrier();
  STAMP(5);
  if constexpr (REM > 0) {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    constexpr int WS = ROWS * LDP;
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      const float* p = part + row * LDP + 16 * i + col;
      f(row, 16 * (4 * TPW + i) + col, ((p[0] + p[WS]) + p[2 * WS]) + p[3 * WS]);
    }
    STAMP(6);
    lds_barrier();
    STAMP(7);
  }
}

// LDS tiles of one network for one 16-row tile
struct Tiles {
  float* x0;   // [16][LD_X0]  inputs, constant-1 column at K0, zeros after
  float* h1;   // [16][LD_W]
  float* h2;   // [16][LD_W]
  float* h3;   // [16][LD_3W or LD_O]
  float* out;  // [16][LD_O]
  float* part; // PART_FLOATS
};

// global work arrays of one network, already offset to this tile's first row (or all null)
struct ActsG {
  float* x0; float* h1; float* h2; float* h3; float* dz1; float* dz2; float* dz3; float* dz4;
};

// forward pass of a whole network for the tile in t.x0; activations stay in t.h1..t.out and, when
// STORE, are also written to the global work arrays for the weight-gradient pass.
template <int KIND, bool STORE>
__device__ __forceinline__ void net_forward(const float* __restrict__ wf, const Geo& g, const Tiles& t, const ActsG& ws) {
  using N = NetT<KIND>;
  constexpr int L3 = N::L3;
  constexpr int NP3 = N::NT3 * 16;
  float* h1 = t.h1; float* h2 = t.h2; float* h3 = t.h3; float* out = t.out;
  dense16<NT_H>(wf + g.offf[0], g.nsf[0], t.x0, LD_X0, t.part, [=](int r, int c, float z) {
    const float v = c < HID ? act_f<N::A1>(z) : (c == HID ? 1.f : 0.f);
    h1[r * LD_W + c] = v;
    if (STORE) ws.h1[r * NP_H + c] = v;
  });
  dense16<NT_H>(wf + g.offf[1], g.nsf[1], h1, LD_W, t.part, [=](int r, int c, float z) {
    const float v = c < HID ? act_f<N::A2>(z) : (c == HID ? 1.f : 0.f);
    h2[r * LD_W + c] = v;
    if (STORE) ws.h2[r * NP_H + c] = v;
  });
  const int n3 = g.N3;
  dense16<N::NT3>(wf + g.offf[2], g.nsf[2], h2, LD_W, t.part, [=](int r, int c, float z) {
    const float v = c < n3 ? act_f<N::A3>(z) : (c == n3 ? 1.f : 0.f);
    h3[r * L3 + c] = v;
    if (STORE) ws.h3[r * NP3 + c] = v;
  });
  const int nout = g.NOUT;
  dense16<1>(wf + g.offf[3], g.nsf[3], h3, L3, t.part, [=](int r, int c, float z) {
    out[r * LD_O + c] = c < nout ? act_f<N::A4>(z) : 0.f;
  });
}

// backward pass: t.out must hold dL/dz4 ([16][LD_O], zeros beyond NOUT), t.h1..h3 the forward
// activations.  Each layer's dz overwrites its activation tile in place and is written to the
// global work arrays.  With NEED_DX0 the gradient w.r.t. the network inputs is left in t.x0
// (columns 0..K0-1, stride LD_X0).
template <int KIND, bool NEED_DX0, bool STORE>
__device__ __forceinline__ void net_backward(const float* __restrict__ wb, const Geo& g, const Tiles& t, const ActsG& ws) {
  using N = NetT<KIND>;
  constexpr int L3 = N::L3;
  constexpr int NP3 = N::NT3 * 16;
  float* h1 = t.h1; float* h2 = t.h2; float* h3 = t.h3; float* x0 = t.x0;
  if (STORE) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    ws.dz4[r * 16 + c] = t.out[r * LD_O + c];
  }
  const int n3 = g.N3;
  dense16<N::NT3>(wb + g.offb[3], g.nsb[3], t.out, LD_O, t.part, [=](int r, int c, float z) {
    const float v = c < n3 ? z * act_d<N::A3>(h3[r * L3 + c]) : 0.f;
    h3[r * L3 + c] = v;
    if (STORE) ws.dz3[r * NP3 + c] = v;
  });
  dense16<NT_H>(wb + g.offb[2], g.nsb[2], h3, L3, t.part, [=](int r, int c, float z) {
    const float v = c < HID ? z * act_d<N::A2>(h2[r * LD_W + c]) : 0.f;
    h2[r * LD_W + c] = v;
    if (STORE) ws.dz2[r * NP_H + c] = v;
  });
  dense16<NT_H>(wb + g.offb[1], g.nsb[1], h2, LD_W, t.part, [=](int r, int c, float z) {
    const float v = c < HID ? z * act_d<N::A1>(h1[r * LD_W + c]) : 0.f;
    h1[r * LD_W + c] = v;
    if (STORE) ws.dz1[r * NP_H + c] = v;
  });
  if (NEED_DX0) {
    dense16<1>(wb + g.offb[0], g.nsb[0], h1, LD_W, t.part, [=](int r, int c, float z) {
      x0[r * LD_X0 + c] = z;
    });
  }
}

// fill the input tile: up to three column blocks, then the constant-1 column, then zeros; rows at or
// beyond n_rows are zero.  Optionally mirrors the tile to the global x0 work array.
struct MatIn { const float* p; int ld; int w; int local; };   // local: an LDS tile indexed by tile row
__device__ __forceinline__ float matin_at(const MatIn& b, int gr, int r, int c) {
  return b.local ? b.p[r * b.ld + c] : b.p[(size_t)gr * b.ld + c];
}
__device__ __forceinline__ void load_x0(float* x0, int row0, int n_rows, MatIn b0, MatIn b1, MatIn b2,
                                        float* x0_global) {
  const int k0 = b0.w + b1.w + b2.w;
  for (int e = threadIdx.x; e < ROWS * X0_COLS; e += THREADS) {
    const int r = e >> 5, c = e & 31;
    const int gr = row0 + r;
    float v = 0.f;
    if (gr < n_rows) {
      if (c < b0.w) v = matin_at(b0, gr, r, c);
      else if (c < b0.w + b1.w) v = matin_at(b1, gr, r, c - b0.w);
      else if (c < k0) v = matin_at(b2, gr, r, c - b0.w - b1.w);
    }
    if (c == k0) v = 1.f;
    x0[r * LD_X0 + c] = v;
    if (x0_global) x0_global[r * X0_COLS + c] = v;
  }
}

__device__ __forceinline__ ActsG acts_at(const float* const* p8, int row0, int ld3) {
  ActsG a;
  a.x0 = (float*)p8[0] + (size_t)row0 * X0_COLS;
  a.h1 = (float*)p8[1] + (size_t)row0 * NP_H;
  a.h2 = (float*)p8[2] + (size_t)row0 * NP_H;
  a.h3 = (float*)p8[3] + (size_t)row0 * ld3;
  a.dz1 = (float*)p8[4] + (size_t)row0 * NP_H;
  a.dz2 = (float*)p8[5] + (size_t)row0 * NP_H;
  a.dz3 = (float*)p8[6] + (size_t)row0 * ld3;
  a.dz4 = (float*)p8[7] + (size_t)row0 * 16;
  return a;
}
#endif  // __HIPCC__

}  // namespace rlgan
