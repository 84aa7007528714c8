// Learner kernels: the four update steps of RL-GAN as 16-row-tile "row passes" (forward, loss,
// backward-to-inputs), the batch-reduction "weight pass" (dW = X^T dZ), gradient reduction,
// TF-1.x RMSProp / Adam and the target-network update.  gfx950 only.
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

struct NetP { const float* wf; const float* wb; };
struct Mat { const float* p; int ld; };
struct Acts8 { float* p[8]; };

__device__ __forceinline__ MatIn gin(Mat m, int w) { return MatIn{m.p, m.ld, w, 0}; }
__device__ __forceinline__ MatIn lin(const float* p, int ld, int w) { return MatIn{p, ld, w, 1}; }
__device__ __forceinline__ MatIn nil() { return MatIn{nullptr, 0, 0, 0}; }

struct TileSet { float* x0; float* h1; float* h2; float* h3; float* out; };

__device__ __forceinline__ Tiles mk(const TileSet& s, float* part) { return Tiles{s.x0, s.h1, s.h2, s.h3, s.out, part}; }

constexpr int SET_FLOATS = X0_FLOATS + 2 * WIDE_FLOATS + T3_FLOATS + O_FLOATS;

__device__ __forceinline__ TileSet carve_set(float* base) {
  TileSet s;
  s.x0 = base; base += X0_FLOATS;
  s.h1 = base; base += WIDE_FLOATS;
  s.h2 = base; base += WIDE_FLOATS;
  s.h3 = base; base += T3_FLOATS;
  s.out = base;
  return s;
}

// sum of v over the first 16 threads' values, returned to thread 0 (deterministic order)
__device__ __forceinline__ float sum16_lds(float* scratch16, float v) {
  if (threadIdx.x < ROWS) scratch16[threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) s += scratch16[i];
  }
  __syncthreads();
  return s;
}

// write a [16][16] "dz4" tile whose column 0..ncols-1 come from `vals16xN` in LDS, zeros elsewhere
__device__ __forceinline__ void fill_out_col0(float* out, const float* v16) {
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  out[r * LD_O + c] = (c == 0) ? v16[r] : 0.f;
}

// ------------------------------------------------------------------------------------------
// (1) critic step   DDPG.py:141-148,174-175
// ------------------------------------------------------------------------------------------
struct CriticArgs {
  Geo gc, ga;
  NetP critic_e, critic_t, actor_t, actor_e;
  Mat s, a, r, s_next;
  float gamma, inv_gb;
  int n_rows, train_actor;
  Acts8 acts_c, acts_a;
  float* loss_parts;
};

__global__ __launch_bounds__(THREADS) void critic_step_kernel(const CriticArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + SET_FLOATS + O_FLOATS + 64];
  float* part = smem;
  TileSet ts = carve_set(smem + PART_FLOATS);
  float* dsave = smem + PART_FLOATS + SET_FLOATS;   // [16][LD_O]
  float* small = dsave + O_FLOATS;                  // y[16], dq[16], scratch[16]
  const Tiles t = mk(ts, part);
  const int row0 = blockIdx.x * ROWS;
  const int S = A.gc.S, Ad = A.gc.A;
  const ActsG none{};

  // target action a' = actor_target(s')
  load_x0(t.x0, row0, A.n_rows, gin(A.s_next, S), nil(), nil(), nullptr);
  __syncthreads();
  net_forward<NET_ACTOR, false>(A.actor_t.wf, A.ga, t, none);
  // q' = critic_target(s', a')
  load_x0(t.x0, row0, A.n_rows, gin(A.s_next, S), lin(t.out, LD_O, Ad), nil(), nullptr);
  __syncthreads();
  net_forward<NET_CRITIC, false>(A.critic_t.wf, A.gc, t, none);
  if (threadIdx.x < ROWS) {
    const int gr = row0 + threadIdx.x;
    const float rew = gr < A.n_rows ? A.r.p[(size_t)gr * A.r.ld] : 0.f;
    small[threadIdx.x] = rew + A.gamma * t.out[threadIdx.x * LD_O];      // y = R + gamma*q'  DDPG.py:142
  }
  // q = critic_eval(s, a)
  const ActsG wc = acts_at(A.acts_c.p, row0, A.gc.ld3);
  load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), wc.x0);
  __syncthreads();
  net_forward<NET_CRITIC, true>(A.critic_e.wf, A.gc, t, wc);
  float l = 0.f;
  if (threadIdx.x < ROWS) {
    const bool valid = row0 + threadIdx.x < A.n_rows;
    const float diff = small[threadIdx.x] - t.out[threadIdx.x * LD_O];
    l = valid ? diff * diff : 0.f;                                        // squared_difference  DDPG.py:145
    small[16 + threadIdx.x] = valid ? -2.f * diff * A.inv_gb : 0.f;       // d mean((y-q)^2) / dq
  }
  const float lsum = sum16_lds(small + 32, l);
  if (threadIdx.x == 0) A.loss_parts[blockIdx.x] = lsum;
  fill_out_col0(t.out, small + 16);
  __syncthreads();
  if (A.train_actor) {
    net_backward<NET_CRITIC, true, true>(A.critic_e.wb, A.gc, t, wc);
    // gradient continues through the tensor actor.a: TanhGrad at the FED action (DDPG.py:148,175)
    {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
      const int gr = row0 + r;
      float v = 0.f;
      if (c < Ad && gr < A.n_rows) {
        const float af = A.a.p[(size_t)gr * A.a.ld + c];
        v = t.x0[r * LD_X0 + S + c] * (1.f - af * af);
      }
      dsave[r * LD_O + c] = v;
    }
    __syncthreads();
    const ActsG wa = acts_at(A.acts_a.p, row0, A.ga.ld3);
    load_x0(t.x0, row0, A.n_rows, gin(A.s, S), nil(), nil(), wa.x0);
    __syncthreads();
    net_forward<NET_ACTOR, true>(A.actor_e.wf, A.ga, t, wa);
    {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
      t.out[r * LD_O + c] = dsave[r * LD_O + c];
    }
    __syncthreads();
    net_backward<NET_ACTOR, false, true>(A.actor_e.wb, A.ga, t, wa);
  } else {
    net_backward<NET_CRITIC, false, true>(A.critic_e.wb, A.gc, t, wc);
  }
}

// ------------------------------------------------------------------------------------------
// (2) actor step   DDPG.py:97-98,107-113,151
// ------------------------------------------------------------------------------------------
struct ActorArgs {
  Geo gc, ga;
  NetP actor_e, critic_e;
  Mat s;
  int n_rows;
  Acts8 acts_a;
};

__global__ __launch_bounds__(THREADS) void actor_step_kernel(const ActorArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + 2 * SET_FLOATS];
  float* part = smem;
  const TileSet sa = carve_set(smem + PART_FLOATS);
  const TileSet sc = carve_set(smem + PART_FLOATS + SET_FLOATS);
  const Tiles ta = mk(sa, part), tc = mk(sc, part);
  const int row0 = blockIdx.x * ROWS;
  const int S = A.gc.S, Ad = A.gc.A;
  const ActsG none{};
  const ActsG wa = acts_at(A.acts_a.p, row0, A.ga.ld3);

  load_x0(ta.x0, row0, A.n_rows, gin(A.s, S), nil(), nil(), wa.x0);
  __syncthreads();
  net_forward<NET_ACTOR, true>(A.actor_e.wf, A.ga, ta, wa);            // a = actor_eval(s), live
  load_x0(tc.x0, row0, A.n_rows, gin(A.s, S), lin(ta.out, LD_O, Ad), nil(), nullptr);
  __syncthreads();
  net_forward<NET_CRITIC, false>(A.critic_e.wf, A.gc, tc, none);
  {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    tc.out[r * LD_O + c] = (c == 0 && row0 + r < A.n_rows) ? 1.f : 0.f;   // d(sum_b q_b)/dq = 1, DDPG.py:151
  }
  __syncthreads();
  net_backward<NET_CRITIC, true, false>(A.critic_e.wb, A.gc, tc, none);
  {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    float v = 0.f;
    if (c < Ad) {
      const float av = ta.out[r * LD_O + c];
      v = tc.x0[r * LD_X0 + S + c] * (1.f - av * av);                  // grad_ys = a_grads, TanhGrad
    }
    ta.out[r * LD_O + c] = v;
  }
  __syncthreads();
  net_backward<NET_ACTOR, false, true>(A.actor_e.wb, A.ga, ta, wa);
}

// ------------------------------------------------------------------------------------------
// (3) discriminator step   Curious_GAN.py:33-45,84-85
// ------------------------------------------------------------------------------------------
struct DiscArgs {
  Geo gd;
  NetP disc;
  Mat s, g, a, s_next;
  float label_real, inv_gb;
  int n_rows, rows16;
  Acts8 acts_d;
  float* loss_parts;
};

__global__ __launch_bounds__(THREADS) void disc_step_kernel(const DiscArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + SET_FLOATS + 64];
  float* part = smem;
  const TileSet ts = carve_set(smem + PART_FLOATS);
  float* small = smem + PART_FLOATS + SET_FLOATS;
  const Tiles t = mk(ts, part);
  const int row0 = blockIdx.x * ROWS;
  const int S = A.gd.S, Ad = A.gd.A;
  float ltot = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const ActsG wd = acts_at(A.acts_d.p, pass * A.rows16 + row0, A.gd.ld3);
    load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(pass == 0 ? A.g : A.s_next, S), gin(A.a, Ad), wd.x0);
    __syncthreads();
    net_forward<NET_DISC, true>(A.disc.wf, A.gd, t, wd);
    const float label = pass == 0 ? 0.f : A.label_real;
    float l = 0.f;
    if (threadIdx.x < ROWS) {
      const bool valid = row0 + threadIdx.x < A.n_rows;
      const float x = t.out[threadIdx.x * LD_O];
      l = valid ? sigmoid_ce(x, label) : 0.f;
      small[threadIdx.x] = valid ? (sigmoid_f(x) - label) * A.inv_gb : 0.f;
    }
    ltot += sum16_lds(small + 16, l);
    fill_out_col0(t.out, small);
    __syncthreads();
    net_backward<NET_DISC, false, true>(A.disc.wb, A.gd, t, wd);
  }
  if (threadIdx.x == 0) A.loss_parts[blockIdx.x] = ltot;
}

// ------------------------------------------------------------------------------------------
// (4) generator step   Curious_GAN.py:137-145
// ------------------------------------------------------------------------------------------
struct GenArgs {
  Geo gg, gd;
  NetP gen, disc;
  Mat s, a;
  float inv_gb;
  int n_rows;
  Acts8 acts_g;
  float* loss_parts;
};

__global__ __launch_bounds__(THREADS) void gen_step_kernel(const GenArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + 2 * SET_FLOATS + 64];
  float* part = smem;
  const TileSet sg = carve_set(smem + PART_FLOATS);
  const TileSet sd = carve_set(smem + PART_FLOATS + SET_FLOATS);
  float* small = smem + PART_FLOATS + 2 * SET_FLOATS;
  const Tiles tg = mk(sg, part), td = mk(sd, part);
  const int row0 = blockIdx.x * ROWS;
  const int S = A.gg.S, Ad = A.gg.A;
  const ActsG none{};
  const ActsG wg = acts_at(A.acts_g.p, row0, A.gg.ld3);

  load_x0(tg.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), wg.x0);
  __syncthreads();
  net_forward<NET_GEN, true>(A.gen.wf, A.gg, tg, wg);                  // g^ = G(s, a), live
  load_x0(td.x0, row0, A.n_rows, gin(A.s, S), lin(tg.out, LD_O, S), gin(A.a, Ad), nullptr);
  __syncthreads();
  net_forward<NET_DISC, false>(A.disc.wf, A.gd, td, none);
  float l = 0.f;
  if (threadIdx.x < ROWS) {
    const bool valid = row0 + threadIdx.x < A.n_rows;
    const float x = td.out[threadIdx.x * LD_O];
    l = valid ? sigmoid_ce(x, 1.f) : 0.f;                               // labels = ones  Curious_GAN.py:139-141
    small[threadIdx.x] = valid ? (sigmoid_f(x) - 1.f) * A.inv_gb : 0.f;
  }
  const float lsum = sum16_lds(small + 16, l);
  if (threadIdx.x == 0) A.loss_parts[blockIdx.x] = lsum;
  fill_out_col0(td.out, small);
  __syncthreads();
  net_backward<NET_DISC, true, false>(A.disc.wb, A.gd, td, none);       // dX only: D is not trained here
  {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    tg.out[r * LD_O + c] = c < S ? td.x0[r * LD_X0 + S + c] : 0.f;      // dL/dg^ (linear output layer)
  }
  __syncthreads();
  net_backward<NET_GEN, false, true>(A.gen.wb, A.gg, tg, wg);
}

// ------------------------------------------------------------------------------------------
// (5)+(6) D and G losses after both updates   Curious_GAN.py:94-95,147-148
// ------------------------------------------------------------------------------------------
struct EvalArgs {
  Geo gg, gd;
  NetP gen, disc;
  Mat s, g, a, s_next;
  float label_real;
  int n_rows, n_tiles;
  float* loss_parts;
};

__global__ __launch_bounds__(THREADS) void gan_eval_kernel(const EvalArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + 2 * SET_FLOATS + 64];
  float* part = smem;
  const TileSet sg = carve_set(smem + PART_FLOATS);
  const TileSet sd = carve_set(smem + PART_FLOATS + SET_FLOATS);
  float* small = smem + PART_FLOATS + 2 * SET_FLOATS;
  const Tiles tg = mk(sg, part), td = mk(sd, part);
  const int row0 = blockIdx.x * ROWS;
  const int S = A.gg.S, Ad = A.gg.A;
  const ActsG none{};
  float ld = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    load_x0(td.x0, row0, A.n_rows, gin(A.s, S), gin(pass == 0 ? A.g : A.s_next, S), gin(A.a, Ad), nullptr);
    __syncthreads();
    net_forward<NET_DISC, false>(A.disc.wf, A.gd, td, none);
    float l = 0.f;
    if (threadIdx.x < ROWS && row0 + threadIdx.x < A.n_rows)
      l = sigmoid_ce(td.out[threadIdx.x * LD_O], pass == 0 ? 0.f : A.label_real);
    ld += sum16_lds(small, l);
  }
  load_x0(tg.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), nullptr);
  __syncthreads();
  net_forward<NET_GEN, false>(A.gen.wf, A.gg, tg, none);
  load_x0(td.x0, row0, A.n_rows, gin(A.s, S), lin(tg.out, LD_O, S), gin(A.a, Ad), nullptr);
  __syncthreads();
  net_forward<NET_DISC, false>(A.disc.wf, A.gd, td, none);
  float l = 0.f;
  if (threadIdx.x < ROWS && row0 + threadIdx.x < A.n_rows) l = sigmoid_ce(td.out[threadIdx.x * LD_O], 1.f);
  const float lg = sum16_lds(small, l);
  if (threadIdx.x == 0) {
    A.loss_parts[blockIdx.x] = ld;
    A.loss_parts[A.n_tiles + blockIdx.x] = lg;
  }
}

// ------------------------------------------------------------------------------------------
// forward only (choose_action / predict / determine)
// ------------------------------------------------------------------------------------------
struct FwdArgs {
  Geo g;
  NetP net;
  Mat s, x, a;
  float* out; int ld_out, out_mode, n_rows;
};

template <int KIND>
__global__ __launch_bounds__(THREADS) void mlp_forward_kernel(const FwdArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + SET_FLOATS];
  const TileSet ts = carve_set(smem + PART_FLOATS);
  const Tiles t = mk(ts, smem);
  const int row0 = blockIdx.x * ROWS;
  const int S = A.g.S, Ad = A.g.A;
  const ActsG none{};
  if (KIND == NET_ACTOR) load_x0(t.x0, row0, A.n_rows, gin(A.s, S), nil(), nil(), nullptr);
  else if (KIND == NET_DISC) load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(A.x, S), gin(A.a, Ad), nullptr);
  else load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), nullptr);
  __syncthreads();
  net_forward<KIND, false>(A.net.wf, A.g, t, none);
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  if (row0 + r < A.n_rows && c < A.g.NOUT) {
    float v = t.out[r * LD_O + c];
    if (A.out_mode == 1) v = sigmoid_f(v);
    else if (A.out_mode == 2) v = 1.f - sigmoid_f(v);
    A.out[(size_t)(row0 + r) * A.ld_out + c] = v;
  }
}

// ------------------------------------------------------------------------------------------
// weight pass: G_l[chunk] = X_l[chunk rows]^T * dZ_l[chunk rows]
// ------------------------------------------------------------------------------------------
struct WItem { const float* X; const float* dZ; int ldx, ldz, nstrips, nt, goff, np; };
struct WArgs {
  WItem it[4];
  int blk0[5];
  int rows, chunk_rows, nchunks, gstride;
  float* gslabs;
};

// acc[i] += X[rows r0..r1, strip]^T * dZ[rows r0..r1, tile i], i < T: `xp` / `zp` are this lane's
// operand addresses for row 0 (consecutive tiles 16 floats apart).  Operands of PIPE 4-row steps are
// kept in flight; the loop body is branch-free (surplus steps multiply a zero by a re-read row).
template <int T>
__device__ __forceinline__ void wgrad_pipe(f32x4 (&acc)[T], const float* __restrict__ xp, int ldx,
                                           const float* __restrict__ zp, int ldz, int r0, int r1) {
  const int n = (r1 - r0) >> 2;
  if (n <= 0) return;
  const int nblk = (n + PIPE - 1) / PIPE;
  const int last = r1 - 4;
  float a[PIPE], b[PIPE][T];
#pragma unroll
  for (int j = 0; j < PIPE; ++j) {
    const int r = r0 + 4 * j, rc = min(r, last);
    const float av = xp[(size_t)rc * ldx];
    a[j] = r < r1 ? av : 0.f;
#pragma unroll
    for (int i = 0; i < T; ++i) b[j][i] = zp[(size_t)rc * ldz + 16 * i];
  }
  for (int blk = 1; blk < nblk; ++blk) {
#pragma unroll
    for (int j = 0; j < PIPE; ++j) {
#pragma unroll
      for (int i = 0; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][i], acc[i], 0, 0, 0);
      const int r = r0 + 4 * (blk * PIPE + j), rc = min(r, last);
      const float av = xp[(size_t)rc * ldx];
      a[j] = r < r1 ? av : 0.f;
#pragma unroll
      for (int i = 0; i < T; ++i) b[j][i] = zp[(size_t)rc * ldz + 16 * i];
    }
  }
#pragma unroll
  for (int j = 0; j < PIPE; ++j) {
#pragma unroll
    for (int i = 0; i < T; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][i], acc[i], 0, 0, 0);
  }
}

// One workgroup = one 16-row strip of dW for one batch chunk.  4*(NT/4) column tiles are dealt to
// the four waves (each walks all rows of the chunk); the NT%4 left-over tiles are computed by all
// four waves together, each taking a quarter of the rows, and summed through LDS in wave order.
template <int NT>
__device__ __forceinline__ void wpass_body(const WItem& it, int strip, int chunk, const WArgs& A, float* part) {
  constexpr int TPW = NT / 4, REM = NT % 4, LDP = REM * 16 + 4;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int rb = chunk * A.chunk_rows;
  const int re = min(A.rows, rb + A.chunk_rows);
  const float* xp = it.X + (size_t)(lane >> 4) * it.ldx + 16 * strip + (lane & 15);
  const float* zp = it.dZ + (size_t)(lane >> 4) * it.ldz + (lane & 15);
  float* gout = A.gslabs + (size_t)chunk * A.gstride + it.goff + (size_t)(16 * strip) * it.np;
  if constexpr (REM > 0) {
    const int rpw = A.chunk_rows >> 2;
    const int w0 = rb + wave * rpw, w1 = min(re, w0 + rpw);
    f32x4 accr[REM];
#pragma unroll
    for (int i = 0; i < REM; ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    wgrad_pipe<REM>(accr, xp, it.ldx, zp + 16 * (4 * TPW), it.ldz, w0, w1);
    float* pw = part + wave * ROWS * LDP + ((lane >> 4) * 4) * LDP + (lane & 15);
#pragma unroll
    for (int i = 0; i < REM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[r * LDP + 16 * i] = accr[i][r];
    }
  }
  if constexpr (TPW > 0) {
    f32x4 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    wgrad_pipe<TPW>(acc, xp, it.ldx, zp + 16 * (wave * TPW), it.ldz, rb, re);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int col = 16 * (wave * TPW + i) + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) gout[((lane >> 4) * 4 + r) * (NT * 16) + col] = acc[i][r];
    }
  }
  if constexpr (REM > 0) {
    __syncthreads();
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    constexpr int WS = ROWS * LDP;
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      const float* p = part + row * LDP + 16 * i + col;
      gout[row * (NT * 16) + 16 * (4 * TPW + i) + col] = ((p[0] + p[WS]) + p[2 * WS]) + p[3 * WS];
    }
  }
}

__global__ __launch_bounds__(THREADS) void wpass_kernel(const WArgs A) {
  __shared__ __attribute__((aligned(16))) float part[PART_FLOATS];
  int item = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if ((int)blockIdx.x >= A.blk0[i]) item = i;
  const int local = blockIdx.x - A.blk0[item];
  // the item record is copied field by field so that it stays in scalar registers
  WItem it;
  it.X = A.it[0].X; it.dZ = A.it[0].dZ; it.ldx = A.it[0].ldx; it.ldz = A.it[0].ldz;
  it.nstrips = A.it[0].nstrips; it.nt = A.it[0].nt; it.goff = A.it[0].goff; it.np = A.it[0].np;
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    if (item == i) {
      it.X = A.it[i].X; it.dZ = A.it[i].dZ; it.ldx = A.it[i].ldx; it.ldz = A.it[i].ldz;
      it.nstrips = A.it[i].nstrips; it.nt = A.it[i].nt; it.goff = A.it[i].goff; it.np = A.it[i].np;
    }
  }
  const int strip = local % it.nstrips, chunk = local / it.nstrips;
  if (it.nt == NT_H) wpass_body<NT_H>(it, strip, chunk, A, part);
  else if (it.nt == 4) wpass_body<4>(it, strip, chunk, A, part);
  else wpass_body<1>(it, strip, chunk, A, part);
}

// ------------------------------------------------------------------------------------------
// reduction over chunks, optimizers, target update, packing
// ------------------------------------------------------------------------------------------
constexpr int MAX_CHUNKS = 16;
__global__ void grad_reduce_kernel(const float* __restrict__ gslabs, int nchunks, int gstride,
                                   const int32_t* __restrict__ map_g, int n, float* __restrict__ grad,
                                   const float* loss_parts, int n_parts, float loss_scale, float* loss_out,
                                   int32_t* counter) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    const float* src = gslabs + map_g[p];
    float v[MAX_CHUNKS];
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) v[c] = c < nchunks ? src[(size_t)c * gstride] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) s += v[c];          // chunk order: bitwise reproducible
    grad[p] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    if (loss_out) {
      // 64 lanes, each sums a strided share in index order, then a fixed-order lane reduction
      float s = 0.f;
      for (int i = threadIdx.x; i < n_parts; i += 64) s += loss_parts[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
      if (threadIdx.x == 0) loss_out[0] = s * loss_scale;
    }
    if (counter && threadIdx.x == 0) counter[0] += 1;
  }
}

__global__ void rmsprop_kernel(float* __restrict__ canon, const float* __restrict__ grad, float* __restrict__ ms,
                               int n, float lr, float decay, float eps, const int32_t* __restrict__ map_f,
                               float* __restrict__ wf, const int32_t* __restrict__ map_b, float* __restrict__ wb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float g = grad[p];
  const float m = decay * ms[p] + (1.f - decay) * g * g;
  ms[p] = m;
  const float w = canon[p] - lr * g / sqrtf(m + eps);
  canon[p] = w;
  wf[map_f[p]] = w;
  if (wb) { const int ob = map_b[p]; if (ob >= 0) wb[ob] = w; }
}

__global__ void adam_kernel(float* __restrict__ canon, const float* __restrict__ grad, float* __restrict__ m1,
                            float* __restrict__ m2, int n, double lr, double b1, double b2, float eps,
                            const int32_t* __restrict__ step, const int32_t* __restrict__ map_f,
                            float* __restrict__ wf, const int32_t* __restrict__ map_b, float* __restrict__ wb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int t = step[0];
  const float lr_t = (float)(lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
  const float fb1 = (float)b1, fb2 = (float)b2;
  const float g = grad[p];
  const float m = fb1 * m1[p] + (1.f - fb1) * g;
  const float v = fb2 * m2[p] + (1.f - fb2) * g * g;
  m1[p] = m; m2[p] = v;
  const float w = canon[p] - lr_t * m / (sqrtf(v) + eps);
  canon[p] = w;
  wf[map_f[p]] = w;
  if (wb) { const int ob = map_b[p]; if (ob >= 0) wb[ob] = w; }
}

__global__ void target_update_kernel(float* __restrict__ tc, const float* __restrict__ ec, int n, float tau,
                                     const int32_t* counter, int period, const int32_t* __restrict__ map_f,
                                     float* __restrict__ twf) {
  if (counter && ((counter[0] - 1) % period) != 0) return;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float w = tau == 1.f ? ec[p] : tau * ec[p] + (1.f - tau) * tc[p];
  tc[p] = w;
  twf[map_f[p]] = w;
}

__global__ void pack_kernel(const float* __restrict__ canon, int n, const int32_t* __restrict__ map_f,
                            float* __restrict__ wf, const int32_t* __restrict__ map_b, float* __restrict__ wb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float w = canon[p];
  wf[map_f[p]] = w;
  if (wb) { const int ob = map_b[p]; if (ob >= 0) wb[ob] = w; }
}

inline Mat M(rlgan_mat_t m) { return Mat{m.p, m.ld}; }
inline NetP NP(rlgan_net_t n) { return NetP{n.wf, n.wb}; }
inline Acts8 A8(const rlgan_acts_t& a) {
  Acts8 r;
  r.p[0] = a.x0; r.p[1] = a.h1; r.p[2] = a.h2; r.p[3] = a.h3;
  r.p[4] = a.dz1; r.p[5] = a.dz2; r.p[6] = a.dz3; r.p[7] = a.dz4;
  return r;
}
inline bool acts_ok(const rlgan_acts_t& a) {
  return a.x0 && a.h1 && a.h2 && a.h3 && a.dz1 && a.dz2 && a.dz3 && a.dz4;
}
inline int launched() { return (int)hipGetLastError(); }

}  // namespace

// ==============================================================================================
// C ABI
// ==============================================================================================
extern "C" {

int rlgan_critic_step_rows(int s_dim, int a_dim, rlgan_net_t critic_eval, rlgan_net_t critic_target,
                           rlgan_net_t actor_target, rlgan_net_t actor_eval,
                           rlgan_mat_t s, rlgan_mat_t a, rlgan_mat_t r, rlgan_mat_t s_next,
                           float gamma, float inv_global_batch, int n_rows,
                           rlgan_acts_t acts_critic, rlgan_acts_t acts_actor,
                           float* loss_parts, void* stream) {
  CriticArgs A;
  if (!make_geo(NET_CRITIC, s_dim, a_dim, A.gc) || !make_geo(NET_ACTOR, s_dim, a_dim, A.ga)) return RLGAN_E_DIMS;
  if (n_rows <= 0 || !critic_eval.wf || !critic_eval.wb || !critic_target.wf || !actor_target.wf ||
      !s.p || !a.p || !r.p || !s_next.p || !loss_parts || !acts_ok(acts_critic)) return RLGAN_E_ARG;
  A.train_actor = actor_eval.wf != nullptr;
  if (A.train_actor && (!actor_eval.wb || !acts_ok(acts_actor))) return RLGAN_E_ARG;
  A.critic_e = NP(critic_eval); A.critic_t = NP(critic_target); A.actor_t = NP(actor_target); A.actor_e = NP(actor_eval);
  A.s = M(s); A.a = M(a); A.r = M(r); A.s_next = M(s_next);
  A.gamma = gamma; A.inv_gb = inv_global_batch; A.n_rows = n_rows;
  A.acts_c = A8(acts_critic); A.acts_a = A8(acts_actor); A.loss_parts = loss_parts;
  hipLaunchKernelGGL(critic_step_kernel, dim3(cdiv(n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_actor_step_rows(int s_dim, int a_dim, rlgan_net_t actor_eval, rlgan_net_t critic_eval,
                          rlgan_mat_t s, int n_rows, rlgan_acts_t acts_actor, void* stream) {
  ActorArgs A;
  if (!make_geo(NET_CRITIC, s_dim, a_dim, A.gc) || !make_geo(NET_ACTOR, s_dim, a_dim, A.ga)) return RLGAN_E_DIMS;
  if (n_rows <= 0 || !actor_eval.wf || !actor_eval.wb || !critic_eval.wf || !critic_eval.wb || !s.p ||
      !acts_ok(acts_actor)) return RLGAN_E_ARG;
  A.actor_e = NP(actor_eval); A.critic_e = NP(critic_eval); A.s = M(s); A.n_rows = n_rows; A.acts_a = A8(acts_actor);
  hipLaunchKernelGGL(actor_step_kernel, dim3(cdiv(n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_disc_step_rows(int s_dim, int a_dim, rlgan_net_t disc,
                         rlgan_mat_t s, rlgan_mat_t g_stored, rlgan_mat_t a, rlgan_mat_t s_next,
                         float label_real, float inv_global_batch, int n_rows,
                         rlgan_acts_t acts_disc, float* loss_parts, void* stream) {
  DiscArgs A;
  if (!make_geo(NET_DISC, s_dim, a_dim, A.gd)) return RLGAN_E_DIMS;
  if (n_rows <= 0 || !disc.wf || !disc.wb || !s.p || !g_stored.p || !a.p || !s_next.p || !loss_parts ||
      !acts_ok(acts_disc)) return RLGAN_E_ARG;
  A.disc = NP(disc); A.s = M(s); A.g = M(g_stored); A.a = M(a); A.s_next = M(s_next);
  A.label_real = label_real; A.inv_gb = inv_global_batch; A.n_rows = n_rows; A.rows16 = cdiv(n_rows, ROWS) * ROWS;
  A.acts_d = A8(acts_disc); A.loss_parts = loss_parts;
  hipLaunchKernelGGL(disc_step_kernel, dim3(cdiv(n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_gen_step_rows(int s_dim, int a_dim, rlgan_net_t gen, rlgan_net_t disc,
                        rlgan_mat_t s, rlgan_mat_t a, float inv_global_batch, int n_rows,
                        rlgan_acts_t acts_gen, float* loss_parts, void* stream) {
  GenArgs A;
  if (!make_geo(NET_GEN, s_dim, a_dim, A.gg) || !make_geo(NET_DISC, s_dim, a_dim, A.gd)) return RLGAN_E_DIMS;
  if (n_rows <= 0 || !gen.wf || !gen.wb || !disc.wf || !disc.wb || !s.p || !a.p || !loss_parts ||
      !acts_ok(acts_gen)) return RLGAN_E_ARG;
  A.gen = NP(gen); A.disc = NP(disc); A.s = M(s); A.a = M(a); A.inv_gb = inv_global_batch; A.n_rows = n_rows;
  A.acts_g = A8(acts_gen); A.loss_parts = loss_parts;
  hipLaunchKernelGGL(gen_step_kernel, dim3(cdiv(n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_gan_eval_rows(int s_dim, int a_dim, rlgan_net_t gen, rlgan_net_t disc,
                        rlgan_mat_t s, rlgan_mat_t g_stored, rlgan_mat_t a, rlgan_mat_t s_next,
                        float label_real, int n_rows, float* loss_parts, void* stream) {
  EvalArgs A;
  if (!make_geo(NET_GEN, s_dim, a_dim, A.gg) || !make_geo(NET_DISC, s_dim, a_dim, A.gd)) return RLGAN_E_DIMS;
  if (n_rows <= 0 || !gen.wf || !disc.wf || !s.p || !g_stored.p || !a.p || !s_next.p || !loss_parts) return RLGAN_E_ARG;
  A.gen = NP(gen); A.disc = NP(disc); A.s = M(s); A.g = M(g_stored); A.a = M(a); A.s_next = M(s_next);
  A.label_real = label_real; A.n_rows = n_rows; A.n_tiles = cdiv(n_rows, ROWS); A.loss_parts = loss_parts;
  hipLaunchKernelGGL(gan_eval_kernel, dim3(A.n_tiles), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_mlp_forward(int kind, int s_dim, int a_dim, rlgan_net_t net,
                      rlgan_mat_t s, rlgan_mat_t x, rlgan_mat_t a,
                      float* out, int ld_out, int out_mode, int n_rows, void* stream) {
  FwdArgs A;
  if (kind < 0 || kind > 3) return RLGAN_E_KIND;
  if (!make_geo(kind, s_dim, a_dim, A.g)) return RLGAN_E_DIMS;
  if (n_rows <= 0 || !net.wf || !s.p || !out || ld_out < A.g.NOUT || out_mode < 0 || out_mode > 2) return RLGAN_E_ARG;
  if (kind != NET_ACTOR && !a.p) return RLGAN_E_ARG;
  if (kind == NET_DISC && !x.p) return RLGAN_E_ARG;
  A.net = NP(net); A.s = M(s); A.x = M(x); A.a = M(a); A.out = out; A.ld_out = ld_out; A.out_mode = out_mode; A.n_rows = n_rows;
  const dim3 grid(cdiv(n_rows, ROWS)), block(THREADS);
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case NET_ACTOR:  hipLaunchKernelGGL(mlp_forward_kernel<NET_ACTOR>, grid, block, 0, st, A); break;
    case NET_CRITIC: hipLaunchKernelGGL(mlp_forward_kernel<NET_CRITIC>, grid, block, 0, st, A); break;
    case NET_GEN:    hipLaunchKernelGGL(mlp_forward_kernel<NET_GEN>, grid, block, 0, st, A); break;
    default:         hipLaunchKernelGGL(mlp_forward_kernel<NET_DISC>, grid, block, 0, st, A); break;
  }
  return launched();
}

int rlgan_wgrad_chunks(int rows, int32_t* chunk_rows, int32_t* n_chunks) {
  if (rows <= 0 || rows % ROWS != 0 || !chunk_rows || !n_chunks) return RLGAN_E_ALIGN;
  const int cr = ROWS * cdiv(rows, ROWS * 16);
  *chunk_rows = cr;
  *n_chunks = cdiv(rows, cr);
  return 0;
}

int rlgan_weight_grads(int kind, int s_dim, int a_dim, rlgan_acts_t acts, int rows, float* gslabs, void* stream) {
  Geo g;
  if (kind < 0 || kind > 3) return RLGAN_E_KIND;
  if (!make_geo(kind, s_dim, a_dim, g)) return RLGAN_E_DIMS;
  if (!acts_ok(acts) || !gslabs) return RLGAN_E_ARG;
  WArgs A;
  int32_t cr, nc;
  const int rc = rlgan_wgrad_chunks(rows, &cr, &nc);
  if (rc) return rc;
  A.rows = rows; A.chunk_rows = cr; A.nchunks = nc; A.gstride = g.gpad_floats; A.gslabs = gslabs;
  const float* X[4] = {acts.x0, acts.h1, acts.h2, acts.h3};
  const float* Z[4] = {acts.dz1, acts.dz2, acts.dz3, acts.dz4};
  const int ldx[4] = {X0_COLS, NP_H, NP_H, g.ld3};
  const int ldz[4] = {NP_H, NP_H, g.ld3, 16};
  int b = 0;
  for (int l = 0; l < 4; ++l) {
    A.it[l] = WItem{X[l], Z[l], ldx[l], ldz[l], g.kp[l] / 16, g.ntf[l], g.goff[l], g.np[l]};
    A.blk0[l] = b;
    b += (g.kp[l] / 16) * nc;
  }
  A.blk0[4] = b;
  hipLaunchKernelGGL(wpass_kernel, dim3(b), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_grad_reduce(const float* gslabs, int n_chunks, int gpad_floats, const int32_t* map_g,
                      int n_params, float* grad, const float* loss_parts, int n_parts,
                      float loss_scale, float* loss_out, int32_t* counter, void* stream) {
  if (!gslabs || !map_g || !grad || n_params <= 0 || n_chunks <= 0 || n_chunks > MAX_CHUNKS || (loss_out && !loss_parts)) return RLGAN_E_ARG;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(cdiv(n_params, 256)), dim3(256), 0, (hipStream_t)stream,
                     gslabs, n_chunks, gpad_floats, map_g, n_params, grad, loss_parts, n_parts, loss_scale,
                     loss_out, counter);
  return launched();
}

int rlgan_rmsprop_apply(float* canon, const float* grad, float* ms, int n_params, float lr, float decay, float eps,
                        const int32_t* map_f, float* wf, const int32_t* map_b, float* wb, void* stream) {
  if (!canon || !grad || !ms || !map_f || !wf || n_params <= 0 || (wb && !map_b)) return RLGAN_E_ARG;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(cdiv(n_params, 256)), dim3(256), 0, (hipStream_t)stream,
                     canon, grad, ms, n_params, lr, decay, eps, map_f, wf, map_b, wb);
  return launched();
}

int rlgan_adam_apply(float* canon, const float* grad, float* m, float* v, int n_params,
                     double lr, double beta1, double beta2, float eps, const int32_t* step,
                     const int32_t* map_f, float* wf, const int32_t* map_b, float* wb, void* stream) {
  if (!canon || !grad || !m || !v || !step || !map_f || !wf || n_params <= 0 || (wb && !map_b)) return RLGAN_E_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n_params, 256)), dim3(256), 0, (hipStream_t)stream,
                     canon, grad, m, v, n_params, lr, beta1, beta2, eps, step, map_f, wf, map_b, wb);
  return launched();
}

int rlgan_target_update(float* target_canon, const float* eval_canon, int n_params, float tau,
                        const int32_t* counter, int period, const int32_t* map_f, float* target_wf, void* stream) {
  if (!target_canon || !eval_canon || !map_f || !target_wf || n_params <= 0 || (counter && period <= 0)) return RLGAN_E_ARG;
  hipLaunchKernelGGL(target_update_kernel, dim3(cdiv(n_params, 256)), dim3(256), 0, (hipStream_t)stream,
                     target_canon, eval_canon, n_params, tau, counter, period, map_f, target_wf);
  return launched();
}

int rlgan_pack_weights(const float* canon, int n_params, const int32_t* map_f, float* wf,
                       const int32_t* map_b, float* wb, void* stream) {
  if (!canon || !map_f || !wf || n_params <= 0 || (wb && !map_b)) return RLGAN_E_ARG;
  hipLaunchKernelGGL(pack_kernel, dim3(cdiv(n_params, 256)), dim3(256), 0, (hipStream_t)stream,
                     canon, n_params, map_f, wf, map_b, wb);
  return launched();
}

}  // extern "C"
