// Learner kernels: the four update steps of RL-GAN as 16-row-tile "row passes" (forward, loss,
// backward-to-inputs), the batch-reduction "weight pass" (dW = X^T dZ), gradient reduction,
// TF-1.x RMSProp / Adam and the target-network update.  gfx950 only.
#include "rlgan_tiles.h"
#include <algorithm>

using namespace rlgan;

namespace {

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
  int32_t* counter;           // optional: += 1 per launch (t_replace_counter, DDPG.py:64,178)
};
template <int NW> constexpr int critic_lds() { return PartL<NW>::FLOATS + SET_FLOATS + O_FLOATS + 64 + STAGE_FLOATS; }

// NW: waves of the workgroup (4 in the single-step kernels, 8 in the fused phases, where two tiles per compute unit
// then make four waves per SIMD instead of two; the results do not depend on it)
template <int NW>
__device__ __forceinline__ void critic_tile(const CriticArgs& A, float* smem, int tile) {
  if (tile == 0 && threadIdx.x == 0 && A.counter) A.counter[0] += 1;   // nothing in this launch reads it
  const bool tile_thread = NW == 4 || threadIdx.x < 256;              // one thread per element of a 16x16 tile
  float* part = smem;
  TileSet ts = carve_set(smem + PartL<NW>::FLOATS);
  float* dsave = smem + PartL<NW>::FLOATS + SET_FLOATS;   // [16][LD_O]
  float* small = dsave + O_FLOATS;                  // y[16], dq[16], scratch[16]
  float* stage = small + 64;                        // [16][STAGE_LD]: s | a | r | s' of the tile's rows
  const Tiles t = mk(ts, part);
  const int row0 = tile * ROWS;
  const int S = A.gc.S, Ad = A.gc.A;
  const ActsG none{};
  const float* st_s = stage; const float* st_a = stage + S; const float* st_r = stage + S + Ad; const float* st_sn = stage + S + Ad + 1;

  // target action a' = actor_target(s')
  FwdPre<NET_ACTOR, NW> p_at; p_at.issue(A.actor_t.wf + A.ga.offf[0]);
  stage_rows<NW>(stage, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), gin(A.r, 1), gin(A.s_next, S));
  lds_barrier();
  load_x0<NW>(t.x0, row0, A.n_rows, lin(st_sn, STAGE_LD, S), nil(), nil(), nullptr);
  lds_barrier();
  FwdPre<NET_CRITIC, NW> p_ct;
  net_forward<NET_ACTOR, false, NW>(A.actor_t.wf, A.ga, t, none, p_at, [&] { p_ct.issue(A.critic_t.wf + A.gc.offf[0]); });
  // q' = critic_target(s', a')
  load_x0<NW>(t.x0, row0, A.n_rows, lin(st_sn, STAGE_LD, S), lin(t.out, LD_O, Ad), nil(), nullptr);
  lds_barrier();
  FwdPre<NET_CRITIC, NW> p_ce;
  net_forward<NET_CRITIC, false, NW>(A.critic_t.wf, A.gc, t, none, p_ct, [&] { p_ce.issue(A.critic_e.wf + A.gc.offf[0]); });
  if (threadIdx.x < ROWS) {
    const float rew = st_r[threadIdx.x * STAGE_LD];                        // 0 beyond n_rows
    small[threadIdx.x] = rew + A.gamma * t.out[threadIdx.x * LD_O];      // y = R + gamma*q'  DDPG.py:142
  }
  // q = critic_eval(s, a)
  const ActsG wc = acts_at(A.acts_c.p, row0, A.gc.ld3);
  load_x0<NW>(t.x0, row0, A.n_rows, lin(st_s, STAGE_LD, S), lin(st_a, STAGE_LD, Ad), nil(), wc.x0);
  lds_barrier();
  PreAll<1> p_cb;
  net_forward<NET_CRITIC, true, NW>(A.critic_e.wf, A.gc, t, wc, p_ce, [&] { p_cb.issue(A.critic_e.wb + A.gc.offb[3]); });
  float l = 0.f;
  if (threadIdx.x < ROWS) {
    const bool valid = row0 + threadIdx.x < A.n_rows;
    const float diff = small[threadIdx.x] - t.out[threadIdx.x * LD_O];
    l = valid ? diff * diff : 0.f;                                        // squared_difference  DDPG.py:145
    small[16 + threadIdx.x] = valid ? -2.f * diff * A.inv_gb : 0.f;       // d mean((y-q)^2) / dq
  }
  const float lsum = sum16_lds(small + 32, l);
  if (threadIdx.x == 0) A.loss_parts[tile] = lsum;
  fill_out_col0(t.out, small + 16);
  lds_barrier();
  if (A.train_actor) {
    FwdPre<NET_ACTOR, NW> p_ae;
    net_backward<NET_CRITIC, true, true, NW>(A.critic_e.wb, A.gc, t, wc, p_cb, [&] { p_ae.issue(A.actor_e.wf + A.ga.offf[0]); });
    // gradient continues through the tensor actor.a: TanhGrad at the FED action (DDPG.py:148,175)
    if (tile_thread) {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
      const int gr = row0 + r;
      float v = 0.f;
      if (c < Ad && gr < A.n_rows) {
        const float af = st_a[r * STAGE_LD + c];
        v = t.x0[r * LD_X0 + S + c] * (1.f - af * af);
      }
      dsave[r * LD_O + c] = v;
    }
    lds_barrier();
    const ActsG wa = acts_at(A.acts_a.p, row0, A.ga.ld3);
    load_x0<NW>(t.x0, row0, A.n_rows, lin(st_s, STAGE_LD, S), nil(), nil(), wa.x0);
    lds_barrier();
    PreAll<1> p_ab;
    net_forward<NET_ACTOR, true, NW>(A.actor_e.wf, A.ga, t, wa, p_ae, [&] { p_ab.issue(A.actor_e.wb + A.ga.offb[3]); });
    if (tile_thread) {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
      t.out[r * LD_O + c] = dsave[r * LD_O + c];
    }
    lds_barrier();
    net_backward<NET_ACTOR, false, true, NW>(A.actor_e.wb, A.ga, t, wa, p_ab, NoNext{});
  } else {
    net_backward<NET_CRITIC, false, true, NW>(A.critic_e.wb, A.gc, t, wc, p_cb, NoNext{});
  }
}

__global__ __launch_bounds__(THREADS) void critic_step_kernel(const CriticArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[critic_lds<4>()];
  critic_tile<4>(A, smem, blockIdx.x);
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
  int32_t* counter;           // optional: += 1 per launch (t_replace_counter, DDPG.py:64,101)
};
// both networks of this tile have a 10-wide third layer: their tile sets keep a 16-column h3 (SET10_FLOATS)
template <int NW> constexpr int actor_lds() { return PartL<NW>::FLOATS + 2 * SET10_FLOATS; }

template <int NW>
__device__ __forceinline__ void actor_tile(const ActorArgs& A, float* smem, int tile) {
  if (tile == 0 && threadIdx.x == 0 && A.counter) A.counter[0] += 1;   // nothing in this launch reads it
  const bool tile_thread = NW == 4 || threadIdx.x < 256;              // one thread per element of a 16x16 tile
  float* part = smem;
  const TileSet sa = carve_set10(smem + PartL<NW>::FLOATS);
  const TileSet sc = carve_set10(smem + PartL<NW>::FLOATS + SET10_FLOATS);
  const Tiles ta = mk(sa, part), tc = mk(sc, part);
  const int row0 = tile * ROWS;
  const int S = A.gc.S, Ad = A.gc.A;
  const ActsG none{};
  const ActsG wa = acts_at(A.acts_a.p, row0, A.ga.ld3);

  FwdPre<NET_ACTOR, NW> p_a; p_a.issue(A.actor_e.wf + A.ga.offf[0]);
  load_x0<NW>(ta.x0, row0, A.n_rows, gin(A.s, S), nil(), nil(), wa.x0);
  lds_barrier();
  FwdPre<NET_CRITIC, NW> p_c;
  net_forward<NET_ACTOR, true, NW>(A.actor_e.wf, A.ga, ta, wa, p_a, [&] { p_c.issue(A.critic_e.wf + A.gc.offf[0]); });   // a = actor_eval(s), live
  load_x0<NW>(tc.x0, row0, A.n_rows, lin(ta.x0, LD_X0, S), lin(ta.out, LD_O, Ad), nil(), nullptr);   // s: from the actor's input tile
  lds_barrier();
  PreAll<1> p_cb;
  net_forward<NET_CRITIC, false, NW>(A.critic_e.wf, A.gc, tc, none, p_c, [&] { p_cb.issue(A.critic_e.wb + A.gc.offb[3]); });
  if (tile_thread) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    tc.out[r * LD_O + c] = (c == 0 && row0 + r < A.n_rows) ? 1.f : 0.f;   // d(sum_b q_b)/dq = 1, DDPG.py:151
  }
  lds_barrier();
  PreAll<1> p_ab;
  net_backward<NET_CRITIC, true, false, NW>(A.critic_e.wb, A.gc, tc, none, p_cb, [&] { p_ab.issue(A.actor_e.wb + A.ga.offb[3]); });
  if (tile_thread) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    float v = 0.f;
    if (c < Ad) {
      const float av = ta.out[r * LD_O + c];
      v = tc.x0[r * LD_X0 + S + c] * (1.f - av * av);                  // grad_ys = a_grads, TanhGrad
    }
    ta.out[r * LD_O + c] = v;
  }
  lds_barrier();
  net_backward<NET_ACTOR, false, true, NW>(A.actor_e.wb, A.ga, ta, wa, p_ab, NoNext{});
}

__global__ __launch_bounds__(THREADS) void actor_step_kernel(const ActorArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[actor_lds<4>()];
  actor_tile<4>(A, smem, blockIdx.x);
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
  int32_t* counter;           // optional: += 1 per launch (the optimizer's step count)
};
template <int NW> constexpr int disc_lds() { return PartL<NW>::FLOATS + SET_FLOATS + 64 + STAGE_FLOATS; }

template <int NW>
__device__ __forceinline__ void disc_tile(const DiscArgs& A, float* smem, int tile) {
  if (tile == 0 && threadIdx.x == 0 && A.counter) A.counter[0] += 1;   // nothing in this launch reads it
  float* part = smem;
  const TileSet ts = carve_set(smem + PartL<NW>::FLOATS);
  float* small = smem + PartL<NW>::FLOATS + SET_FLOATS;
  const Tiles t = mk(ts, part);
  const int row0 = tile * ROWS;
  const int S = A.gd.S, Ad = A.gd.A;
  float* stage = small + 64;                        // [16][STAGE_LD]: s | g | a | s' of the tile's rows
  float ltot = 0.f;
  FwdPre<NET_DISC, NW> p_f; p_f.issue(A.disc.wf + A.gd.offf[0]);
  stage_rows<NW>(stage, row0, A.n_rows, gin(A.s, S), gin(A.g, S), gin(A.a, Ad), gin(A.s_next, S));
  lds_barrier();
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const ActsG wd = acts_at(A.acts_d.p, pass * A.rows16 + row0, A.gd.ld3);
    load_x0<NW>(t.x0, row0, A.n_rows, lin(stage, STAGE_LD, S), lin(stage + (pass == 0 ? S : 2 * S + Ad), STAGE_LD, S),
            lin(stage + 2 * S, STAGE_LD, Ad), wd.x0);
    lds_barrier();
    PreAll<1> p_b;
    net_forward<NET_DISC, true, NW>(A.disc.wf, A.gd, t, wd, p_f, [&] { p_b.issue(A.disc.wb + A.gd.offb[3]); });
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
    lds_barrier();
    net_backward<NET_DISC, false, true, NW>(A.disc.wb, A.gd, t, wd, p_b, [&] { p_f.issue(A.disc.wf + A.gd.offf[0]); });
  }
  if (threadIdx.x == 0) A.loss_parts[tile] = ltot;
}

__global__ __launch_bounds__(THREADS) void disc_step_kernel(const DiscArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[disc_lds<4>()];
  disc_tile<4>(A, smem, blockIdx.x);
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
  int32_t* counter;           // optional: += 1 per launch (the optimizer's step count)
};
// the generator's set keeps the 64-column h3, the discriminator's the 16-column one
template <int NW> constexpr int gen_lds() { return PartL<NW>::FLOATS + SET_FLOATS + SET10_FLOATS + 64; }

template <int NW>
__device__ __forceinline__ void gen_tile(const GenArgs& A, float* smem, int tile) {
  if (tile == 0 && threadIdx.x == 0 && A.counter) A.counter[0] += 1;   // nothing in this launch reads it
  const bool tile_thread = NW == 4 || threadIdx.x < 256;              // one thread per element of a 16x16 tile
  float* part = smem;
  const TileSet sg = carve_set(smem + PartL<NW>::FLOATS);
  const TileSet sd = carve_set10(smem + PartL<NW>::FLOATS + SET_FLOATS);
  float* small = smem + PartL<NW>::FLOATS + SET_FLOATS + SET10_FLOATS;
  const Tiles tg = mk(sg, part), td = mk(sd, part);
  const int row0 = tile * ROWS;
  const int S = A.gg.S, Ad = A.gg.A;
  const ActsG none{};
  const ActsG wg = acts_at(A.acts_g.p, row0, A.gg.ld3);

  FwdPre<NET_GEN, NW> p_g; p_g.issue(A.gen.wf + A.gg.offf[0]);
  load_x0<NW>(tg.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), wg.x0);
  lds_barrier();
  FwdPre<NET_DISC, NW> p_d;
  net_forward<NET_GEN, true, NW>(A.gen.wf, A.gg, tg, wg, p_g, [&] { p_d.issue(A.disc.wf + A.gd.offf[0]); });   // g^ = G(s, a), live
  load_x0<NW>(td.x0, row0, A.n_rows, lin(tg.x0, LD_X0, S), lin(tg.out, LD_O, S), lin(tg.x0 + S, LD_X0, Ad), nullptr);   // s, a: from G's input tile
  lds_barrier();
  PreAll<1> p_db;
  net_forward<NET_DISC, false, NW>(A.disc.wf, A.gd, td, none, p_d, [&] { p_db.issue(A.disc.wb + A.gd.offb[3]); });
  float l = 0.f;
  if (threadIdx.x < ROWS) {
    const bool valid = row0 + threadIdx.x < A.n_rows;
    const float x = td.out[threadIdx.x * LD_O];
    l = valid ? sigmoid_ce(x, 1.f) : 0.f;                               // labels = ones  Curious_GAN.py:139-141
    small[threadIdx.x] = valid ? (sigmoid_f(x) - 1.f) * A.inv_gb : 0.f;
  }
  const float lsum = sum16_lds(small + 16, l);
  if (threadIdx.x == 0) A.loss_parts[tile] = lsum;
  fill_out_col0(td.out, small);
  lds_barrier();
  Pre<4, 1, NW> p_gb;
  net_backward<NET_DISC, true, false, NW>(A.disc.wb, A.gd, td, none, p_db, [&] { p_gb.issue(A.gen.wb + A.gg.offb[3]); });   // dX only
  if (tile_thread) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    tg.out[r * LD_O + c] = c < S ? td.x0[r * LD_X0 + S + c] : 0.f;      // dL/dg^ (linear output layer)
  }
  lds_barrier();
  net_backward<NET_GEN, false, true, NW>(A.gen.wb, A.gg, tg, wg, p_gb, NoNext{});
}

__global__ __launch_bounds__(THREADS) void gen_step_kernel(const GenArgs A) {
  __shared__ __attribute__((aligned(16))) float smem[gen_lds<4>()];
  gen_tile<4>(A, smem, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// fused phases: two independent steps share one launch so that every CU holds a tile of each
// ------------------------------------------------------------------------------------------
// Waves per tile in the fused phases.  The tiles take the count as a template parameter (4 or 8; the results are the
// same bits).  Eight waves per tile -- sixteen per compute unit at batch 4096, four per SIMD instead of two, 128 registers
// per wave, -DRLGAN_PNW=8 -- was built and measured: phase 1 48.5 us against 47.1, phase 2 38.0 against 37.8, wave-cycles
// doubled, the share of them spent waiting 0.41 -> 0.47, 14 % more vector instructions (every wave repeats the per-wave
// set-up and the narrow stages) and the matrix pipe exactly as busy as before: the launches are bound by what the pipe
// is given, not by waves to hide latencies behind, so the default stays at four.
#ifndef RLGAN_PNW
#define RLGAN_PNW 4
#endif
constexpr int PNW = RLGAN_PNW, PTHREADS = PNW * 64;
#if RLGAN_PNW == 8
#define RLGAN_PHASE_BOUNDS __launch_bounds__(512, 4)      // four waves per SIMD: at most 128 registers
#else
#define RLGAN_PHASE_BOUNDS __launch_bounds__(256)
#endif
constexpr int PHASE1_LDS = critic_lds<PNW>() > disc_lds<PNW>() ? critic_lds<PNW>() : disc_lds<PNW>();
constexpr int PHASE2_LDS = actor_lds<PNW>() > gen_lds<PNW>() ? actor_lds<PNW>() : gen_lds<PNW>();
static_assert(2 * PHASE1_LDS * 4 <= 160 * 1024 && 2 * PHASE2_LDS * 4 <= 160 * 1024, "two workgroups per compute unit");

// Role of a workgroup in a two-role launch: the first half of the grid walks the tiles of role 0, the
// second half those of role 1.  Workgroups are placed breadth-first (one per compute unit before any
// gets a second), so with one tile per compute unit and role every compute unit ends up holding one
// tile of each role -- the same amount of work everywhere -- and the longer role starts first.
__device__ __forceinline__ void role_of(int& role, int& tile) {
  const int b = blockIdx.x, half = gridDim.x >> 1;
  role = b >= half;
  tile = role ? b - half : b;
}

__global__ RLGAN_PHASE_BOUNDS void phase1_kernel(const CriticArgs C, const DiscArgs D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // PHASE1_LDS floats
  int role, tile;
  role_of(role, tile);
  if (role) disc_tile<PNW>(D, smem, tile);
  else critic_tile<PNW>(C, smem, tile);
}

__global__ RLGAN_PHASE_BOUNDS void phase2_kernel(const ActorArgs Aa, const GenArgs G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // PHASE2_LDS floats
  int role, tile;
  role_of(role, tile);
  if (role) gen_tile<PNW>(G, smem, tile);
  else actor_tile<PNW>(Aa, smem, tile);
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
  FwdPre<KIND> p1; p1.issue(A.net.wf + A.g.offf[0]);
  if (KIND == NET_ACTOR) load_x0(t.x0, row0, A.n_rows, gin(A.s, S), nil(), nil(), nullptr);
  else if (KIND == NET_DISC) load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(A.x, S), gin(A.a, Ad), nullptr);
  else load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), nullptr);
  lds_barrier();
  net_forward<KIND, false>(A.net.wf, A.g, t, none, p1, NoNext{});
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  if (row0 + r < A.n_rows && c < A.g.NOUT) {
    float v = t.out[r * LD_O + c];
    if (A.out_mode == 1) v = sigmoid_f(v);
    else if (A.out_mode == 2) v = 1.f - sigmoid_f(v);
    A.out[(size_t)(row0 + r) * A.ld_out + c] = v;
  }
}

// ------------------------------------------------------------------------------------------
// weight pass: G_l[chunk] = X_l[chunk rows]^T * dZ_l[chunk rows] for every layer of up to three networks
// ------------------------------------------------------------------------------------------
#ifndef RLGAN_WD
#define RLGAN_WD 4
#endif
constexpr int WD = RLGAN_WD;     // steps of operands kept in flight
constexpr int MAX_WITEMS = 12;   // 3 networks x 4 layers
struct WItem {
  const float* X; const float* dZ; float* gout;     // gout: chunk 0 of this layer's padded gradient image
  int ldx, ldz, nstrips, nt, np;
  int n_groups, groups_per_chunk, gstride;          // 16-row groups in all / per chunk; floats between chunks
  int blk0;                                         // first workgroup of this item
};
struct WArgs { WItem it[MAX_WITEMS]; int n_items; };

// One workgroup = one 16-row strip of dW for one batch chunk of n 16-row groups.  Column tiles
// 0..4*(NT/4)-1 are dealt to the four waves, T = NT/4 each; a wave with such tiles walks every group of
// the chunk.  The R = NT%4 left-over tiles are shared by rows: group g is wave (g & 3)'s, and the four
// partial sums meet in LDS in wave order.  Inside every block of four groups a wave starts at its own
// group ((step + wave) & 3), so that its duty falls on steps 0, 4, 8, ... and register indices stay
// static; a wave without dealt tiles (NT < 4) visits only its own groups.  One pipeline of D steps
// of operands in flight covers both kinds of tile: the chain of dependent memory round trips per
// workgroup is ceil(steps / D) long.
// xsrc / zt / zr: buffer sources whose lane offset addresses this lane's float4 in group 0 (lane
// (q, j) reads rows 4q..4q+3 of column j, see work_off; the next column tile is 256 bytes on); xs / zs:
// bytes between groups.  Buffer loads keep the address arithmetic on the scalar unit (vector ALU work would
// not overlap with the MFMAs).
template <int T, int R, int D>
__device__ __forceinline__ void wgrad_pipe(f32x4* acc, f32x4* accr, const FragSrc& xsrc, int xs, const FragSrc& zt,
                                           const FragSrc& zr, int zs, int g0, int n, int wave) {
  constexpr bool ALL = T > 0;
  constexpr int DUTY = ALL ? 4 : 1, NR = D / DUTY;
  static_assert(D % 4 == 0, "blocks of four groups");
  const int nsteps = ALL ? ((n + 3) & ~3) : ((n + 3) >> 2);
  const int nblk = (nsteps + D - 1) / D;
  f32x4 a[D], b[D][ALL ? T : 1], br[NR][R > 0 ? R : 1];
#define RLGAN_WFILL(j, k)                                                                        \
  {                                                                                              \
    const int g = ALL ? (((k) & ~3) | (((k) + wave) & 3)) : 4 * (k) + wave;                      \
    const int gc = g0 + min(g, n - 1);                                                           \
    a[j] = xsrc.load_at(gc * xs);                                                                \
    _Pragma("unroll") for (int i = 0; i < T; ++i) b[j][i] = zt.load_at(gc * zs + i * 256);      \
    if (R > 0 && (j) % DUTY == 0) {                                                              \
      _Pragma("unroll") for (int i = 0; i < R; ++i) br[(j) / DUTY][i] = zr.load_at(gc * zs + i * 256); \
    }                                                                                            \
  }
#define RLGAN_WUSE(j, k)                                                                         \
  if ((ALL ? (((k) & ~3) | (((k) + wave) & 3)) : 4 * (k) + wave) < n) {                          \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                              \
      _Pragma("unroll") for (int i = 0; i < T; ++i)                                              \
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], b[j][i][m], acc[i], 0, 0, 0);     \
      if (R > 0 && (j) % DUTY == 0) {                                                            \
        _Pragma("unroll") for (int i = 0; i < R; ++i)                                            \
          accr[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m], br[(j) / DUTY][i][m], accr[i], 0, 0, 0); \
      }                                                                                          \
    }                                                                                            \
  }
#pragma unroll
  for (int j = 0; j < D; ++j) RLGAN_WFILL(j, j)
  for (int blk = 1; blk < nblk; ++blk) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      RLGAN_WUSE(j, (blk - 1) * D + j)
      RLGAN_WFILL(j, blk * D + j)
      __builtin_amdgcn_sched_barrier(0);   // keep the refill D steps ahead of its use
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j) RLGAN_WUSE(j, (nblk - 1) * D + j)
#undef RLGAN_WFILL
#undef RLGAN_WUSE
}

template <int NT>
__device__ __forceinline__ void wpass_body(const WItem& it, int strip, int chunk, float* part) {
  constexpr int TPW = NT / 4, REM = NT % 4, LDP = REM * 16 + 4;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int q4 = (lane >> 4) * 4, j = lane & 15;
  const int gb = chunk * it.groups_per_chunk;
  const int n = min(it.n_groups, gb + it.groups_per_chunk) - gb;
  const int xs = it.ldx * 64, zs = it.ldz * 64;            // bytes per 16-row group
  FragSrc xsrc, zsrc_t, zsrc_r;
  xsrc.set_bytes(it.X, work_off(it.ldx, q4, 16 * strip + j) * 4);
  zsrc_t.set_bytes(it.dZ, work_off(it.ldz, q4, 16 * (wave * TPW) + j) * 4);
  zsrc_r.set_bytes(it.dZ, work_off(it.ldz, q4, 16 * (4 * TPW) + j) * 4);
  float* gout = it.gout + (size_t)chunk * it.gstride + (size_t)(16 * strip) * it.np;
  f32x4 acc[TPW > 0 ? TPW : 1], accr[REM > 0 ? REM : 1];
#pragma unroll
  for (int i = 0; i < (TPW > 0 ? TPW : 1); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (REM > 0 ? REM : 1); ++i) accr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  wgrad_pipe<TPW, REM, WD>(acc, accr, xsrc, xs, zsrc_t, zsrc_r, zs, gb, n, wave);
  if constexpr (REM > 0) {
    float* pw = part + wave * ROWS * LDP + q4 * LDP + j;
#pragma unroll
    for (int i = 0; i < REM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[r * LDP + 16 * i] = accr[i][r];
    }
  }
  if constexpr (TPW > 0) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int col = 16 * (wave * TPW + i) + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) gout[(q4 + r) * (NT * 16) + col] = acc[i][r];
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
  __shared__ __attribute__((aligned(16))) float part[4 * ROWS * 20];   // four waves' partial sums of one 16-column tile
  int item = 0;
#pragma unroll
  for (int i = 1; i < MAX_WITEMS; ++i) if (i < A.n_items && (int)blockIdx.x >= A.it[i].blk0) item = i;
  // the item record is copied field by field under a uniform select so that it stays in scalar registers
  WItem it = A.it[0];
#pragma unroll
  for (int i = 1; i < MAX_WITEMS; ++i) if (item == i) it = A.it[i];
  // Workgroups go to the eight XCDs round-robin (blockIdx % 8), each with its own L2.  All strips of a
  // batch chunk read the same dZ rows, so a chunk is kept on one XCD: items start at a multiple of 8
  // and their chunk counts are padded to a multiple of 8 (workgroups of the padding exit).
  const int local = blockIdx.x - it.blk0;
  const int xcd = local & 7, j = local >> 3;
  const int strip = j % it.nstrips, chunk = xcd + 8 * (j / it.nstrips);
  if (chunk * it.groups_per_chunk >= it.n_groups) return;
  if (it.nt == NT_H) wpass_body<NT_H>(it, strip, chunk, part);
  else if (it.nt == 4) wpass_body<4>(it, strip, chunk, part);
  else wpass_body<1>(it, strip, chunk, part);
}

// ------------------------------------------------------------------------------------------
// reduction over chunks, optimizers, target update, packing -- each for up to three networks a launch
// ------------------------------------------------------------------------------------------
constexpr int MAX_CHUNKS = 16;
constexpr int MAX_JOBS = 4;

struct ReduceJob {
  const float* gslabs; const int32_t* map_g; float* grad;
  const float* loss_parts; float* loss_out; int32_t* counter;
  int nchunks, gstride, n, n_parts; float loss_scale; int blk0;
};
struct ReduceArgs { ReduceJob job[MAX_JOBS]; int n_jobs; };

__global__ __launch_bounds__(256) void grad_reduce_kernel(const ReduceArgs A) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < MAX_JOBS; ++i) if (i < A.n_jobs && (int)blockIdx.x >= A.job[i].blk0) ji = i;
  ReduceJob J = A.job[0];
#pragma unroll
  for (int i = 1; i < MAX_JOBS; ++i) if (ji == i) J = A.job[i];
  const int blk = blockIdx.x - J.blk0;
  const int p = blk * 256 + threadIdx.x;
  if (p < J.n) {
    const float* src = J.gslabs + J.map_g[p];
    float v[MAX_CHUNKS];
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) v[c] = c < J.nchunks ? src[(size_t)c * J.gstride] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) s += v[c];          // chunk order: bitwise reproducible
    J.grad[p] = s;
  }
  if (blk == 0 && threadIdx.x < 64) {
    if (J.loss_out) {
      // 64 lanes, each sums a strided share in index order, then a fixed-order lane reduction
      float s = 0.f;
      for (int i = threadIdx.x; i < J.n_parts; i += 64) s += J.loss_parts[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
      if (threadIdx.x == 0) J.loss_out[0] = s * J.loss_scale;
    }
    if (J.counter && threadIdx.x == 0) J.counter[0] += 1;
  }
}

// one optimizer step of one network, optionally followed by its target-network update
struct ApplyJob {
  float* canon; const float* grad; float* slot0; float* slot1;
  const int32_t* map_f; float* wf; const int32_t* map_b; float* wb;
  const int32_t* step;                 // Adam: bias-correction step count (already incremented)
  float* target_canon; float* target_wf; const int32_t* counter;   // target update (NULL: none)
  double lr, beta1, beta2;             // RMSProp: beta1 = decay
  float eps, tau;
  int opt, n, period, blk0;            // opt: 0 = RMSProp, 1 = Adam
  // optional reduction over batch chunks in the same launch (gslabs null: the gradient is read from grad)
  const float* gslabs; const int32_t* map_g; float* grad_out;
  const float* loss_parts; float* loss_out;
  int nchunks, gstride, n_parts; float loss_scale;
};
struct ApplyArgs { ApplyJob job[MAX_JOBS]; int n_jobs; };

__global__ __launch_bounds__(256) void apply_kernel(const ApplyArgs A) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < MAX_JOBS; ++i) if (i < A.n_jobs && (int)blockIdx.x >= A.job[i].blk0) ji = i;
  ApplyJob J = A.job[0];
#pragma unroll
  for (int i = 1; i < MAX_JOBS; ++i) if (ji == i) J = A.job[i];
  const int blk = blockIdx.x - J.blk0;
  const int p = blk * 256 + threadIdx.x;
  if (J.gslabs && J.loss_out && blk == 0 && threadIdx.x < 64) {
    float s = 0.f;
    for (int i = threadIdx.x; i < J.n_parts; i += 64) s += J.loss_parts[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (threadIdx.x == 0) J.loss_out[0] = s * J.loss_scale;
  }
  if (p >= J.n) return;
  // everything the update reads goes out first, so that these round trips overlap with the chunk sum and the
  // step-size arithmetic instead of following them (job fields are the same for the whole workgroup: scalar branches)
  const float w_old = J.canon[p], s0_old = J.slot0[p];
  const float s1_old = J.opt == 1 ? J.slot1[p] : 0.f;
  const int t_adam = J.opt == 1 ? J.step[0] : 1;
  const int og = J.gslabs ? J.map_g[p] : 0;
  const int of = J.wf ? J.map_f[p] : -1;                   // -1: this parameter has no place in the image (or there is no image)
  const int ob = J.wb ? J.map_b[p] : -1;
  const bool to_target = J.target_canon && (!J.counter || ((J.counter[0] - 1) % J.period) == 0);
  __builtin_amdgcn_sched_barrier(0);
  float g;
  if (J.gslabs) {
    const float* src = J.gslabs + og;
    float v[MAX_CHUNKS];
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) v[c] = c < J.nchunks ? src[(size_t)c * J.gstride] : 0.f;
    g = 0.f;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) g += v[c];          // chunk order: bitwise reproducible
    if (J.grad_out) J.grad_out[p] = g;
  } else {
    g = J.grad[p];
  }
  float w;
  if (J.opt == 0) {
    // tf.train.RMSPropOptimizer: ms <- decay*ms + (1-decay)*g^2 ; w <- w - lr*g/sqrt(ms + eps)
    const float decay = (float)J.beta1;
    const float m = decay * s0_old + (1.f - decay) * g * g;
    J.slot0[p] = m;
    w = w_old - (float)J.lr * g / sqrtf(m + J.eps);
  } else {
    // tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w <- w - lr_t*m/(sqrt(v)+eps)
    const int t = t_adam;
    // beta^t by repeated squaring (t is an integer and the same for the whole launch: a scalar loop of at most 31 steps,
    // far shorter than the general pow(); relative error of a few 2^-53)
    double p1 = 1.0, p2 = 1.0, b1 = J.beta1, b2 = J.beta2;
    for (int e = t; e > 0; e >>= 1) {
      if (e & 1) { p1 *= b1; p2 *= b2; }
      b1 *= b1; b2 *= b2;
    }
    const float lr_t = (float)(J.lr * sqrt(1.0 - p2) / (1.0 - p1));
    const float fb1 = (float)J.beta1, fb2 = (float)J.beta2;
    const float m = fb1 * s0_old + (1.f - fb1) * g;
    const float v = fb2 * s1_old + (1.f - fb2) * g * g;
    J.slot0[p] = m; J.slot1[p] = v;
    w = w_old - lr_t * m / (sqrtf(v) + J.eps);
  }
  J.canon[p] = w;
  if (of >= 0) J.wf[of] = w;
  if (ob >= 0) J.wb[ob] = w;
  if (to_target) {
    // hard copy (tau = 1, DDPG.py:99-101,176-178) or Polyak average
    const float tw = J.tau == 1.f ? w : J.tau * w + (1.f - J.tau) * J.target_canon[p];
    J.target_canon[p] = tw;
    if (of >= 0) J.target_wf[of] = tw;
  }
}

// out[k] = scale * sum_i parts[k*n_parts + i], one wave per sum, fixed order
__global__ void sum_parts_kernel(const float* __restrict__ parts, int n_parts, float scale, float* __restrict__ out) {
  const float* p = parts + (size_t)blockIdx.x * n_parts;
  float s = 0.f;
  constexpr int FL = 16;                                  // loads in flight per lane and turn (one per turn: a memory round trip per addition)
  for (int i0 = threadIdx.x; i0 < n_parts; i0 += 64 * FL) {
    float v[FL];
#pragma unroll
    for (int u = 0; u < FL; ++u) v[u] = p[min(i0 + 64 * u, n_parts - 1)];
#pragma unroll
    for (int u = 0; u < FL; ++u) s += i0 + 64 * u < n_parts ? v[u] : 0.f;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (threadIdx.x == 0) out[blockIdx.x] = s * scale;
}

__global__ void target_update_kernel(float* __restrict__ tc, const float* __restrict__ ec, int n, float tau,
                                     const int32_t* counter, int period, const int32_t* __restrict__ map_f,
                                     float* __restrict__ twf) {
  if (counter && ((counter[0] - 1) % period) != 0) return;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float w = tau == 1.f ? ec[p] : tau * ec[p] + (1.f - tau) * tc[p];
  tc[p] = w;
  if (map_f) {
    const int of = map_f[p];
    if (of >= 0) twf[of] = w;
  }
}

__global__ void pack_kernel(const float* __restrict__ canon, int n, const int32_t* __restrict__ map_f,
                            float* __restrict__ wf, const int32_t* __restrict__ map_b, float* __restrict__ wb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float w = canon[p];
  const int of = map_f[p];
  if (of >= 0) wf[of] = w;
  if (wb) { const int ob = map_b[p]; if (ob >= 0) wb[ob] = w; }
}

// ---- argument validation / packing shared by the single-step and the fused-phase entry points
int make_critic_args(CriticArgs& A, const rlgan_critic_step_t* c) {
  if (!c) return RLGAN_E_ARG;
  if (!make_geo(NET_CRITIC, c->s_dim, c->a_dim, A.gc) || !make_geo(NET_ACTOR, c->s_dim, c->a_dim, A.ga)) return RLGAN_E_DIMS;
  if (c->n_rows <= 0 || !c->critic_eval.wf || !c->critic_eval.wb || !c->critic_target.wf || !c->actor_target.wf ||
      !c->s.p || !c->a.p || !c->r.p || !c->s_next.p || !c->loss_parts || !acts_ok(c->acts_critic)) return RLGAN_E_ARG;
  A.train_actor = c->actor_eval.wf != nullptr;
  if (A.train_actor && (!c->actor_eval.wb || !acts_ok(c->acts_actor))) return RLGAN_E_ARG;
  A.critic_e = NP(c->critic_eval); A.critic_t = NP(c->critic_target); A.actor_t = NP(c->actor_target); A.actor_e = NP(c->actor_eval);
  A.s = M(c->s, c->row_idx); A.a = M(c->a, c->row_idx); A.r = M(c->r, c->row_idx); A.s_next = M(c->s_next, c->row_idx);
  A.gamma = c->gamma; A.inv_gb = c->inv_global_batch; A.n_rows = c->n_rows;
  A.acts_c = A8(c->acts_critic); A.acts_a = A8(c->acts_actor); A.loss_parts = c->loss_parts; A.counter = c->counter;
  return 0;
}
int make_actor_args(ActorArgs& A, const rlgan_actor_step_t* c) {
  if (!c) return RLGAN_E_ARG;
  if (!make_geo(NET_CRITIC, c->s_dim, c->a_dim, A.gc) || !make_geo(NET_ACTOR, c->s_dim, c->a_dim, A.ga)) return RLGAN_E_DIMS;
  if (c->n_rows <= 0 || !c->actor_eval.wf || !c->actor_eval.wb || !c->critic_eval.wf || !c->critic_eval.wb || !c->s.p ||
      !acts_ok(c->acts_actor)) return RLGAN_E_ARG;
  A.actor_e = NP(c->actor_eval); A.critic_e = NP(c->critic_eval); A.s = M(c->s, c->row_idx); A.n_rows = c->n_rows; A.acts_a = A8(c->acts_actor); A.counter = c->counter;
  return 0;
}
int make_disc_args(DiscArgs& A, const rlgan_disc_step_t* c) {
  if (!c) return RLGAN_E_ARG;
  if (!make_geo(NET_DISC, c->s_dim, c->a_dim, A.gd)) return RLGAN_E_DIMS;
  if (c->n_rows <= 0 || !c->disc.wf || !c->disc.wb || !c->s.p || !c->g_stored.p || !c->a.p || !c->s_next.p || !c->loss_parts ||
      !acts_ok(c->acts_disc)) return RLGAN_E_ARG;
  A.disc = NP(c->disc); A.s = M(c->s, c->row_idx); A.g = M(c->g_stored, c->row_idx); A.a = M(c->a, c->row_idx); A.s_next = M(c->s_next, c->row_idx);
  A.label_real = c->label_real; A.inv_gb = c->inv_global_batch; A.n_rows = c->n_rows; A.rows16 = cdiv(c->n_rows, ROWS) * ROWS;
  A.acts_d = A8(c->acts_disc); A.loss_parts = c->loss_parts; A.counter = c->counter;
  return 0;
}
int make_gen_args(GenArgs& A, const rlgan_gen_step_t* c) {
  if (!c) return RLGAN_E_ARG;
  if (!make_geo(NET_GEN, c->s_dim, c->a_dim, A.gg) || !make_geo(NET_DISC, c->s_dim, c->a_dim, A.gd)) return RLGAN_E_DIMS;
  if (c->n_rows <= 0 || !c->gen.wf || !c->gen.wb || !c->disc.wf || !c->disc.wb || !c->s.p || !c->a.p || !c->loss_parts ||
      !acts_ok(c->acts_gen)) return RLGAN_E_ARG;
  A.gen = NP(c->gen); A.disc = NP(c->disc); A.s = M(c->s, c->row_idx); A.a = M(c->a, c->row_idx); A.inv_gb = c->inv_global_batch; A.n_rows = c->n_rows;
  A.acts_g = A8(c->acts_gen); A.loss_parts = c->loss_parts; A.counter = c->counter;
  return 0;
}

}  // namespace

// ==============================================================================================
// C ABI
// ==============================================================================================
extern "C" {

int rlgan_critic_step_rows(const rlgan_critic_step_t* c, void* stream) {
  CriticArgs A;
  if (const int rc = make_critic_args(A, c)) return rc;
  hipLaunchKernelGGL(critic_step_kernel, dim3(cdiv(A.n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_actor_step_rows(const rlgan_actor_step_t* c, void* stream) {
  ActorArgs A;
  if (const int rc = make_actor_args(A, c)) return rc;
  hipLaunchKernelGGL(actor_step_kernel, dim3(cdiv(A.n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_disc_step_rows(const rlgan_disc_step_t* c, void* stream) {
  DiscArgs A;
  if (const int rc = make_disc_args(A, c)) return rc;
  hipLaunchKernelGGL(disc_step_kernel, dim3(cdiv(A.n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_gen_step_rows(const rlgan_gen_step_t* c, void* stream) {
  GenArgs A;
  if (const int rc = make_gen_args(A, c)) return rc;
  hipLaunchKernelGGL(gen_step_kernel, dim3(cdiv(A.n_rows, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_phase1_rows(const rlgan_critic_step_t* c, const rlgan_disc_step_t* d, void* stream) {
  CriticArgs C; DiscArgs D;
  if (const int rc = make_critic_args(C, c)) return rc;
  if (const int rc = make_disc_args(D, d)) return rc;
  if (C.n_rows != D.n_rows) return RLGAN_E_ALIGN;
  static const int lds_ok = (int)hipFuncSetAttribute((const void*)phase1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PHASE1_LDS * 4);
  if (lds_ok) return lds_ok;
  hipLaunchKernelGGL(phase1_kernel, dim3(2 * cdiv(C.n_rows, ROWS)), dim3(PTHREADS), PHASE1_LDS * 4, (hipStream_t)stream, C, D);
  return launched();
}

int rlgan_phase2_rows(const rlgan_actor_step_t* a, const rlgan_gen_step_t* g, void* stream) {
  ActorArgs Aa; GenArgs G;
  if (const int rc = make_actor_args(Aa, a)) return rc;
  if (const int rc = make_gen_args(G, g)) return rc;
  if (Aa.n_rows != G.n_rows) return RLGAN_E_ALIGN;
  static const int lds_ok = (int)hipFuncSetAttribute((const void*)phase2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PHASE2_LDS * 4);
  if (lds_ok) return lds_ok;
  hipLaunchKernelGGL(phase2_kernel, dim3(2 * cdiv(Aa.n_rows, ROWS)), dim3(PTHREADS), PHASE2_LDS * 4, (hipStream_t)stream, Aa, G);
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
  A.net = NP(net); A.s = M(s, nullptr); A.x = M(x, nullptr); A.a = M(a, nullptr); A.out = out; A.ld_out = ld_out; A.out_mode = out_mode; A.n_rows = n_rows;
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
  // batch chunks of the weight-gradient launch (a build parameter for the sweep: 8 / 10 / 12 / 16 chunks at 4096 rows give
  // 0.1716 / 0.1793 / 0.1750 / 0.1691 ms per step -- fewer, fatter slabs save the optimizer launch less than they cost this one)
#ifndef RLGAN_WCHUNKS
#define RLGAN_WCHUNKS 16
#endif
  const int cr = ROWS * cdiv(rows, ROWS * RLGAN_WCHUNKS);
  *chunk_rows = cr;
  *n_chunks = cdiv(rows, cr);
  return 0;
}

int rlgan_weight_grads(const rlgan_wgrad_job_t* jobs, int n_jobs, int s_dim, int a_dim, void* stream) {
  if (!jobs || n_jobs <= 0 || 4 * n_jobs > MAX_WITEMS) return RLGAN_E_ARG;
  WArgs A;
  A.n_items = 0;
  int b = 0;
  for (int ji = 0; ji < n_jobs; ++ji) {
    const rlgan_wgrad_job_t& J = jobs[ji];
    Geo g;
    if (J.kind < 0 || J.kind > NET_PCRITIC) return RLGAN_E_KIND;
    if (!make_geo(J.kind, s_dim, a_dim, g)) return RLGAN_E_DIMS;
    if (!acts_ok(J.acts) || !J.gslabs) return RLGAN_E_ARG;
    int32_t cr, nc;
    if (const int rc = rlgan_wgrad_chunks(J.rows, &cr, &nc)) return rc;
    const float* X[4] = {J.acts.x0, J.acts.h1, J.acts.h2, J.acts.h3};
    const float* Z[4] = {J.acts.dz1, J.acts.dz2, J.acts.dz3, J.acts.dz4};
    const int ldx[4] = {x0_cols(J.kind), NP_H, NP_H, g.ld3};
    const int ldz[4] = {NP_H, NP_H, g.ld3, 16};
    for (int l = 0; l < 4; ++l) {
      WItem& it = A.it[A.n_items++];
      it.X = X[l]; it.dZ = Z[l]; it.gout = J.gslabs + g.goff[l];
      it.ldx = ldx[l]; it.ldz = ldz[l]; it.nstrips = g.kp[l] / 16; it.nt = g.ntf[l]; it.np = g.np[l];
      it.n_groups = J.rows / ROWS; it.groups_per_chunk = cr / ROWS; it.gstride = g.gpad_floats;
    }
  }
  // Workgroups are placed on the compute units in blockIdx order, all of them resident at once.  With
  // the long workgroups (many column tiles x many row groups) first, each unit gets an even share of
  // them and the short ones fill in behind.
  std::stable_sort(A.it, A.it + A.n_items, [](const WItem& a, const WItem& c) {
    return a.nt * a.groups_per_chunk > c.nt * c.groups_per_chunk;
  });
  for (int i = 0; i < A.n_items; ++i) {
    WItem& it = A.it[i];
    it.blk0 = b;
    const int nc = cdiv(it.n_groups, it.groups_per_chunk);
    b += it.nstrips * (cdiv(nc, 8) * 8);            // a multiple of 8: see the XCD mapping in wpass_kernel
  }
  hipLaunchKernelGGL(wpass_kernel, dim3(b), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_grad_reduce(const rlgan_reduce_job_t* jobs, int n_jobs, void* stream) {
  if (!jobs || n_jobs <= 0 || n_jobs > MAX_JOBS) return RLGAN_E_ARG;
  ReduceArgs A;
  A.n_jobs = n_jobs;
  int b = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const rlgan_reduce_job_t& J = jobs[i];
    if (!J.gslabs || !J.map_g || !J.grad || J.n_params <= 0 || J.n_chunks <= 0 || J.n_chunks > MAX_CHUNKS ||
        (J.loss_out && !J.loss_parts)) return RLGAN_E_ARG;
    A.job[i] = ReduceJob{J.gslabs, J.map_g, J.grad, J.loss_parts, J.loss_out, J.counter,
                         J.n_chunks, J.gpad_floats, J.n_params, J.n_parts, J.loss_scale, b};
    b += cdiv(J.n_params, 256);
  }
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_apply(const rlgan_apply_job_t* jobs, int n_jobs, void* stream) {
  if (!jobs || n_jobs <= 0 || n_jobs > MAX_JOBS) return RLGAN_E_ARG;
  ApplyArgs A;
  A.n_jobs = n_jobs;
  int b = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const rlgan_apply_job_t& J = jobs[i];
    if (!J.canon || (!J.grad && !J.reduce) || !J.slot0 || (J.wf && !J.map_f) || J.n_params <= 0 || (J.wb && !J.map_b)) return RLGAN_E_ARG;
    if (J.opt != 0 && J.opt != 1) return RLGAN_E_ARG;
    if (J.opt == 1 && (!J.slot1 || !J.step)) return RLGAN_E_ARG;
    if (J.target_canon && ((J.wf && !J.target_wf) || (J.counter && J.period <= 0))) return RLGAN_E_ARG;
    ApplyJob& K = A.job[i];
    K.canon = J.canon; K.grad = J.grad; K.slot0 = J.slot0; K.slot1 = J.slot1;
    K.map_f = J.map_f; K.wf = J.wf; K.map_b = J.map_b; K.wb = J.wb; K.step = J.step;
    K.target_canon = J.target_canon; K.target_wf = J.target_wf; K.counter = J.counter;
    K.lr = J.lr; K.beta1 = J.beta1; K.beta2 = J.beta2; K.eps = J.eps; K.tau = J.tau;
    K.opt = J.opt; K.n = J.n_params; K.period = J.period; K.blk0 = b;
    K.gslabs = nullptr; K.map_g = nullptr; K.grad_out = nullptr; K.loss_parts = nullptr; K.loss_out = nullptr;
    K.nchunks = 0; K.gstride = 0; K.n_parts = 0; K.loss_scale = 0.f;
    if (const rlgan_reduce_job_t* R = J.reduce) {
      // a counter cannot be advanced here: other workgroups of this launch read it
      if (!R->gslabs || !R->map_g || R->n_params != J.n_params || R->n_chunks <= 0 || R->n_chunks > MAX_CHUNKS ||
          (R->loss_out && !R->loss_parts) || R->counter) return RLGAN_E_ARG;
      K.gslabs = R->gslabs; K.map_g = R->map_g; K.grad_out = R->grad; K.loss_parts = R->loss_parts; K.loss_out = R->loss_out;
      K.nchunks = R->n_chunks; K.gstride = R->gpad_floats; K.n_parts = R->n_parts; K.loss_scale = R->loss_scale;
    }
    b += cdiv(J.n_params, 256);
  }
  hipLaunchKernelGGL(apply_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, A);
  return launched();
}

int rlgan_sum_parts(const float* parts, int n_parts, int n_sums, float scale, float* out, void* stream) {
  if (!parts || !out || n_parts <= 0 || n_sums <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(sum_parts_kernel, dim3(n_sums), dim3(64), 0, (hipStream_t)stream, parts, n_parts, scale, out);
  return launched();
}

int rlgan_target_update(float* target_canon, const float* eval_canon, int n_params, float tau,
                        const int32_t* counter, int period, const int32_t* map_f, float* target_wf, void* stream) {
  if (!target_canon || !eval_canon || (map_f != nullptr) != (target_wf != nullptr) || n_params <= 0 || (counter && period <= 0)) return RLGAN_E_ARG;
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
