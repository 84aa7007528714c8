// Update steps of the pixel variant's actor and critic (Conv_model.py:569-784) as 16-row-tile row passes on the
// shared MFMA core: the networks have the arm variant's 200-200-10 body behind a first layer of 17 k-groups (256
// encoder features, the action columns, the constant 1), so one workgroup walks whole networks for its 16 minibatch
// rows exactly as the arm kernels do -- three forward passes and one backward pass for the critic's step, two of each
// for the actor's -- and leaves activations and pre-activation gradients in the work arrays of the weight-gradient
// kernel.  As 41 separate matrix products the two steps took 0.5 ms at batch 4096.  gfx950 only.
#include "rlgan_tiles.h"
#include <cstdlib>

using namespace rlgan;

namespace {

constexpr int SETW_FLOATS = XW_FLOATS + 2 * WIDE_FLOATS + T3_FLOATS + O_FLOATS;
__device__ __forceinline__ TileSet carve_setw(float* base) {
  TileSet s;
  s.x0 = base; base += XW_FLOATS;
  s.h1 = base; base += WIDE_FLOATS;
  s.h2 = base; base += WIDE_FLOATS;
  s.h3 = base; base += T3_FLOATS;
  s.out = base;
  return s;
}

// the action columns of an input tile from an LDS tile `a` [16][lda] (aw columns), then the constant 1, then zeros
__device__ __forceinline__ void set_xw_tail(float* x0, const float* a, int lda, int aw) {
  if (threadIdx.x >= 256) return;                        // one thread per element of the 16x16 tail (workgroups of eight waves)
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  x0[r * LD_XW + PIX_FEATURES + c] = c < aw ? a[r * lda + c] : (c == aw ? 1.f : 0.f);
}

// ------------------------------------------------------------------------------------------
// critic step   Conv_model.py:610-616,653-663
// ------------------------------------------------------------------------------------------
struct PCriticArgs {
  Geo gc, ga;
  NetP critic_e, critic_t, actor_t;
  const float* phis; const float* phis_next; const float* a; const float* r;
  int ldp, ldpn, lda, a_cols;
  float gamma, inv_n;
  int n_rows;
  Acts8 acts_c;
  float* loss_parts;
  int32_t* counter;
};
template <int NW> constexpr int pcritic_lds() { return PartL<NW>::FLOATS + SETW_FLOATS + 64; }

// NW: waves of the workgroup.  A step is one tile per compute unit at 4096 rows, and a tile is a chain of passes whose
// layers each wave walks alone: with eight waves every SIMD holds two, each with half the column tiles.  Same bits.
template <int NW>
__global__ __launch_bounds__(NW * 64) void pixel_critic_step_kernel(const PCriticArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[pcritic_lds<NW>()];
  const int tile = blockIdx.x;
  if (tile == 0 && threadIdx.x == 0 && A.counter) A.counter[0] += 1;   // nothing in this launch reads it
  float* part = smem;
  const TileSet ts = carve_setw(smem + PartL<NW>::FLOATS);
  float* small = smem + PartL<NW>::FLOATS + SETW_FLOATS;   // y[16], dq[16], scratch[16]
  const Tiles t = mk(ts, part);
  const int row0 = tile * ROWS, Ap = A.ga.NOUT;
  const ActsG none{};

  // a' = actor_target(phis')
  FwdPre<NET_PACTOR, NW> p_at; p_at.issue(A.actor_t.wf + A.ga.offf[0]);
  load_xw<NW>(t.x0, row0, A.n_rows, A.phis_next, A.ldpn, nullptr, 0, 0, 0, 0, nullptr);
  lds_barrier();
  FwdPre<NET_PCRITIC, NW> p_ct;
  net_forward<NET_PACTOR, false, NW>(A.actor_t.wf, A.ga, t, none, p_at, [&] { p_ct.issue(A.critic_t.wf + A.gc.offf[0]); });
  // q' = critic_target(phis', a'): the features are in place, only the action columns change
  set_xw_tail(t.x0, t.out, LD_O, Ap);
  lds_barrier();
  FwdPre<NET_PCRITIC, NW> p_ce;
  net_forward<NET_PCRITIC, false, NW>(A.critic_t.wf, A.gc, t, none, p_ct, [&] { p_ce.issue(A.critic_e.wf + A.gc.offf[0]); });
  if (threadIdx.x < ROWS) {
    const int gr = row0 + threadIdx.x;
    const float rew = gr < A.n_rows ? A.r[gr] : 0.f;
    small[threadIdx.x] = rew + A.gamma * t.out[threadIdx.x * LD_O];        // target_q = R + gamma * q'   Conv_model.py:611
  }
  // q = critic_eval(phis, a)
  const ActsG wc = acts_at(A.acts_c.p, row0, A.gc.ld3, XW_COLS);
  lds_barrier();
  load_xw<NW>(t.x0, row0, A.n_rows, A.phis, A.ldp, A.a, A.lda, 0, A.a_cols, Ap, wc.x0);
  lds_barrier();
  PreAll<1> p_cb;
  net_forward<NET_PCRITIC, true, NW>(A.critic_e.wf, A.gc, t, wc, p_ce, [&] { p_cb.issue(A.critic_e.wb + A.gc.offb[3]); });
  float l = 0.f;
  if (threadIdx.x < ROWS) {
    const bool valid = row0 + threadIdx.x < A.n_rows;
    const float diff = t.out[threadIdx.x * LD_O] - small[threadIdx.x];
    l = valid ? diff * diff : 0.f;                                          // squared_difference   Conv_model.py:613
    small[16 + threadIdx.x] = valid ? 2.f * diff * A.inv_n : 0.f;          // d mean((q - target_q)^2) / dq
  }
  const float lsum = sum16_lds(small + 32, l);
  if (threadIdx.x == 0) A.loss_parts[tile] = lsum;
  fill_out_col0(t.out, small + 16);
  lds_barrier();
  net_backward<NET_PCRITIC, false, true, NW>(A.critic_e.wb, A.gc, t, wc, p_cb, NoNext{});
}

// ------------------------------------------------------------------------------------------
// actor step   Conv_model.py:619,746-754,778-784
// ------------------------------------------------------------------------------------------
struct PActorArgs {
  Geo gc, ga;
  NetP actor_e, critic_e;
  const float* phis;
  int ldp, n_rows;
  Acts8 acts_a;
  int32_t* counter;
};
template <int NW> constexpr int pactor_lds() { return PartL<NW>::FLOATS + 2 * SETW_FLOATS; }

template <int NW>
__global__ __launch_bounds__(NW * 64) void pixel_actor_step_kernel(const PActorArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NTHR = NW * 64;
  const bool tile_thread = NW == 4 || threadIdx.x < 256;              // one thread per element of a 16x16 tile
  const int tile = blockIdx.x;
  if (tile == 0 && threadIdx.x == 0 && A.counter) A.counter[0] += 1;
  float* part = smem;
  const TileSet sa = carve_setw(smem + PartL<NW>::FLOATS);
  const TileSet sc = carve_setw(smem + PartL<NW>::FLOATS + SETW_FLOATS);
  const Tiles ta = mk(sa, part), tc = mk(sc, part);
  const int row0 = tile * ROWS, Ap = A.ga.NOUT;
  const ActsG none{};
  const ActsG wa = acts_at(A.acts_a.p, row0, A.ga.ld3, XW_COLS);

  FwdPre<NET_PACTOR, NW> p_a; p_a.issue(A.actor_e.wf + A.ga.offf[0]);
  load_xw<NW>(ta.x0, row0, A.n_rows, A.phis, A.ldp, nullptr, 0, 0, 0, 0, wa.x0);
  lds_barrier();
  FwdPre<NET_PCRITIC, NW> p_c;
  net_forward<NET_PACTOR, true, NW>(A.actor_e.wf, A.ga, ta, wa, p_a, [&] { p_c.issue(A.critic_e.wf + A.gc.offf[0]); });   // a = actor_eval(phis), live
  // the critic's input: the same features, the action just computed
  for (int e = threadIdx.x; e < ROWS * (PIX_FEATURES / 4); e += NTHR) {
    const int r = e / (PIX_FEATURES / 4), c4 = e - r * (PIX_FEATURES / 4);
    *(f32x4*)(tc.x0 + r * LD_XW + 4 * c4) = *(const f32x4*)(ta.x0 + r * LD_XW + 4 * c4);
  }
  set_xw_tail(tc.x0, ta.out, LD_O, Ap);
  lds_barrier();
  PreAll<1> p_cb;
  net_forward<NET_PCRITIC, false, NW>(A.critic_e.wf, A.gc, tc, none, p_c, [&] { p_cb.issue(A.critic_e.wb + A.gc.offb[3]); });
  if (tile_thread) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    tc.out[r * LD_O + c] = (c == 0 && row0 + r < A.n_rows) ? 1.f : 0.f;     // d(sum_b q_b)/dq = 1: tf.gradients(q, a)   Conv_model.py:619
  }
  lds_barrier();
  PreAll<1> p_ab;
  net_backward<NET_PCRITIC, true, false, NW>(A.critic_e.wb, A.gc, tc, none, p_cb, [&] { p_ab.issue(A.actor_e.wb + A.ga.offb[3]); });
  if (tile_thread) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    float v = 0.f;
    if (c < Ap) {
      const float av = ta.out[r * LD_O + c];
      v = tc.x0[r * LD_XW + PIX_FEATURES + c] * (1.f - av * av);            // grad_ys = a_grads through tanh   Conv_model.py:780
    }
    ta.out[r * LD_O + c] = v;
  }
  lds_barrier();
  net_backward<NET_PACTOR, false, true, NW>(A.actor_e.wb, A.ga, ta, wa, p_ab, NoNext{});
}

inline bool wide_tiles(int n_rows) {
  static const bool off = getenv("RLGAN_PIXEL_NW4") != nullptr;      // measurement switch
  return !off && cdiv(n_rows, ROWS) <= 256;
}

}  // namespace

extern "C" {

int rlgan_pixel_critic_step_rows(const rlgan_pixel_critic_step_t* c, void* stream) {
  if (!c) return RLGAN_E_ARG;
  PCriticArgs A;
  if (!make_geo(NET_PCRITIC, PIX_FEATURES, c->a_pad, A.gc) || !make_geo(NET_PACTOR, PIX_FEATURES, c->a_pad, A.ga)) return RLGAN_E_DIMS;
  if (c->n_rows <= 0 || !c->critic_eval.wf || !c->critic_eval.wb || !c->critic_target.wf || !c->actor_target.wf || !c->phis.p ||
      !c->phis_next.p || !c->a.p || !c->r || !c->loss_parts || !acts_ok(c->acts_critic)) return RLGAN_E_ARG;
  if (c->a_cols < 1 || c->a_cols > c->a_pad || c->a.ld < c->a_cols) return RLGAN_E_ARG;
  if (c->phis.ld < PIX_FEATURES || c->phis_next.ld < PIX_FEATURES || (c->phis.ld & 3) || (c->phis_next.ld & 3) ||
      (((uintptr_t)c->phis.p | (uintptr_t)c->phis_next.p) & 15)) return RLGAN_E_ALIGN;
  A.critic_e = NP(c->critic_eval); A.critic_t = NP(c->critic_target); A.actor_t = NP(c->actor_target);
  A.phis = c->phis.p; A.ldp = c->phis.ld; A.phis_next = c->phis_next.p; A.ldpn = c->phis_next.ld;
  A.a = c->a.p; A.lda = c->a.ld; A.a_cols = c->a_cols; A.r = c->r;
  A.gamma = c->gamma; A.inv_n = 1.f / (float)c->n_rows; A.n_rows = c->n_rows;
  A.acts_c = A8(c->acts_critic); A.loss_parts = c->loss_parts; A.counter = c->counter;
  // eight waves per tile while a compute unit holds one tile at most
  if (wide_tiles(A.n_rows)) hipLaunchKernelGGL(pixel_critic_step_kernel<8>, dim3(cdiv(A.n_rows, ROWS)), dim3(512), 0, (hipStream_t)stream, A);
  else hipLaunchKernelGGL(pixel_critic_step_kernel<4>, dim3(cdiv(A.n_rows, ROWS)), dim3(256), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_pixel_actor_step_rows(const rlgan_pixel_actor_step_t* c, void* stream) {
  if (!c) return RLGAN_E_ARG;
  PActorArgs A;
  if (!make_geo(NET_PCRITIC, PIX_FEATURES, c->a_pad, A.gc) || !make_geo(NET_PACTOR, PIX_FEATURES, c->a_pad, A.ga)) return RLGAN_E_DIMS;
  if (c->n_rows <= 0 || !c->actor_eval.wf || !c->actor_eval.wb || !c->critic_eval.wf || !c->critic_eval.wb || !c->phis.p ||
      !acts_ok(c->acts_actor)) return RLGAN_E_ARG;
  if (c->phis.ld < PIX_FEATURES || (c->phis.ld & 3) || ((uintptr_t)c->phis.p & 15)) return RLGAN_E_ALIGN;
  A.actor_e = NP(c->actor_eval); A.critic_e = NP(c->critic_eval);
  A.phis = c->phis.p; A.ldp = c->phis.ld; A.n_rows = c->n_rows;
  A.acts_a = A8(c->acts_actor); A.counter = c->counter;
  if (wide_tiles(A.n_rows)) {
    constexpr int lds = pactor_lds<8>() * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)pixel_actor_step_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(pixel_actor_step_kernel<8>, dim3(cdiv(A.n_rows, ROWS)), dim3(512), lds, (hipStream_t)stream, A);
  } else {
    constexpr int lds = pactor_lds<4>() * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)pixel_actor_step_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(pixel_actor_step_kernel<4>, dim3(cdiv(A.n_rows, ROWS)), dim3(256), lds, (hipStream_t)stream, A);
  }
  return launched();
}

}  // extern "C"
