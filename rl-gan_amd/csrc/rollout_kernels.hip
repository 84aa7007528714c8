// Rollout-side kernels: the vectorised arm environment (step / reset / episode cut), the fused
// acting step
//   actor -> exploration noise -> generator -> env.step -> discriminator -> intrinsic reward -> replay row
//   -> episode cut -> counters, noise-scale decay
// the post-update loss evaluation of D and G, and one launch that runs both side by side (they only
// read the networks) together with the replay index draw for the next minibatch.  gfx950 only.
#include "rlgan_tiles.h"
#include "rlgan_env.h"
#include "rlgan_rng.h"

using namespace rlgan;

namespace {

__global__ void env_step_kernel(EnvP E, int n, const void* action, int f64, int lda,
                                double* so, float* sof, double* reward, uint8_t* done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a0, a1;
  if (f64) { const double* a = (const double*)action + (size_t)i * lda; a0 = a[0]; a1 = a[1]; }
  else { const float* a = (const float*)action + (size_t)i * lda; a0 = (double)a[0]; a1 = (double)a[1]; }
  double s[7]; uint8_t d;
  const double r = env_step_one(E, i, a0, a1, s, &d);
#pragma unroll
  for (int c = 0; c < 7; ++c) { if (so) so[(size_t)i * 7 + c] = s[c]; if (sof) sof[(size_t)i * 7 + c] = (float)s[c]; }
  if (reward) reward[i] = r;
  if (done) done[i] = d;
}

__global__ void env_reset_kernel(EnvP E, int n, int mode_hard, const uint8_t* mask, const double* u,
                                 double* so, float* sof) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mask && !mask[i]) return;
  double s[7];
  env_reset_one(E, i, mode_hard, u[2 * i], u[2 * i + 1], s);
#pragma unroll
  for (int c = 0; c < 7; ++c) { if (so) so[(size_t)i * 7 + c] = s[c]; if (sof) sof[(size_t)i * 7 + c] = (float)s[c]; }
}

// episode cut `t == MAX_EP_STEPS-1 or done` then `env.reset()` (Curious_GAN.py:180,222-234)
__global__ void env_autoreset_kernel(EnvP E, int n, int mode_hard, const uint8_t* done, int32_t* ep_t,
                                     int max_ep_steps, const double* u, float* sof, int32_t* n_resets,
                                     long long* counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && counters) { counters[0] += n; counters[1] += 1; }   // replay pointer, vector-step count
  if (i >= n) return;
  const int t = ep_t[i];
  if (done[i] || t >= max_ep_steps - 1) {
    double s[7];
    env_reset_one(E, i, mode_hard, u[2 * i], u[2 * i + 1], s);
#pragma unroll
    for (int c = 0; c < 7; ++c) sof[(size_t)i * 7 + c] = (float)s[c];
    ep_t[i] = 0;
    if (n_resets) atomicAdd(n_resets, 1);
  } else {
    ep_t[i] = t + 1;
  }
}

// ------------------------------------------------------------------------------------------
// acting step for 16 environments per workgroup (Curious_GAN.py:190-198,222-234)
// ------------------------------------------------------------------------------------------
constexpr uint32_t STREAM_NOISE = 0, STREAM_GOAL = 1;   // Philox stream ids of the in-kernel draws

struct ActArgs {
  Geo ga, gg, gd;
  NetP actor, gen, disc;
  EnvP env;
  int n_env, n_tiles;
  float* state;               // [n,7] fp32, in: s, out: s' (or the first state of the next episode)
  const float* noise;         // [n,2] standard normal, or null
  double* var;                // device scalar: noise scale
  float ita;
  float* replay; int capacity, dims, ld_replay;
  long long pointer, step;    // host values, used when counters is null
  long long* counters;        // device {replay pointer, vector steps}; advanced by the last workgroup
  double* reward_out; uint8_t* done_out;
  unsigned long long rng_seed;  // != 0: draw noise (when `noise` is null) and reset uniforms (when `reset_u` is null) in the kernel
  int32_t* ep_t; int max_ep_steps, mode_hard; const double* reset_u; int32_t* n_resets;   // episode cut (ep_t null: none)
  double var_decay, var_min;  // var <- max(var*var_decay, var_min) once the ring is full (var_decay <= 0: never)
  long long* counters_out; double* var_out;   // when set: advanced values go here (written by tile 0), inputs stay
};

constexpr int ACT_LDS = PART_FLOATS + SET_FLOATS + 5 * ROWS * 16 + 2 * ROWS * 2;

__device__ __forceinline__ void act_tile(const ActArgs& A, float* smem, int tile) {
  float* part = smem;
  const TileSet ts = carve_set(smem + PART_FLOATS);
  const Tiles t = mk(ts, part);
  float* b = smem + PART_FLOATS + SET_FLOATS;
  float* sbuf = b; b += ROWS * 16;   // s
  float* abuf = b; b += ROWS * 16;   // noisy clipped action (fp32 view)
  float* gbuf = b; b += ROWS * 16;   // g^
  float* nbuf = b; b += ROWS * 16;   // s'
  float* rbuf = b; b += ROWS * 16;   // r, curious_r
  double* a64 = (double*)b;          // [16][2] the action env.step sees (float64)
  const int row0 = tile * ROWS;
  const int S = A.ga.S, Ad = A.ga.A;
  const ActsG none{};
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int gr = row0 + r;
  const bool valid = gr < A.n_env;
  const long long ptr0 = A.counters ? A.counters[0] : A.pointer;
  const long long step = A.counters ? A.counters[1] : A.step;

  FwdPre<NET_ACTOR> p_a; p_a.issue(A.actor.wf + A.ga.offf[0]);
  sbuf[r * 16 + c] = (valid && c < S) ? A.state[(size_t)gr * S + c] : 0.f;
  // exploration noise of this row: given, drawn here, or none
  float z = 0.f;
  if (c < 2 && valid) {
    if (A.noise) z = A.noise[(size_t)gr * Ad + c];
    else if (A.rng_seed) {
      float z0, z1;
      philox_normal_pair((unsigned long long)(step * A.n_env + gr), STREAM_NOISE, A.rng_seed, z0, z1);
      z = c == 0 ? z0 : z1;
    }
  }
  const double var = A.var ? A.var[0] : 0.0;
  if (A.counters_out && tile == 0 && threadIdx.x == 0) {
    // nothing in this launch reads the output pair, so no workgroup has to wait for the others
    const long long p = ptr0 + A.n_env;
    A.counters_out[0] = p; A.counters_out[1] = step + 1;
    if (A.var_out) A.var_out[0] = (A.var_decay > 0.0 && p > A.capacity) ? fmax(var * A.var_decay, A.var_min) : var;
  }
  lds_barrier();
  load_x0(t.x0, row0, A.n_env, lin(sbuf, 16, S), nil(), nil(), nullptr);
  lds_barrier();
  FwdPre<NET_GEN> p_g;
  net_forward<NET_ACTOR, false>(A.actor.wf, A.ga, t, none, p_a, [&] { p_g.issue(A.gen.wf + A.gg.offf[0]); });   // a = actor.choose_action(s)
  // a = clip(normal(a, var), -1, 1) evaluated in float64 like NumPy (Curious_GAN.py:191)
  if (c < Ad) {
    double av = (double)t.out[r * LD_O + c];
    if (c < 2) av += var * (double)z;
    av = clipd(av, -1.0, 1.0);
    if (c < 2) a64[r * 2 + c] = av;
    abuf[r * 16 + c] = (float)av;
  } else {
    abuf[r * 16 + c] = 0.f;
  }
  lds_barrier();
  load_x0(t.x0, row0, A.n_env, lin(sbuf, 16, S), lin(abuf, 16, Ad), nil(), nullptr);
  lds_barrier();
  FwdPre<NET_DISC> p_d;
  net_forward<NET_GEN, false>(A.gen.wf, A.gg, t, none, p_g, [&] { p_d.issue(A.disc.wf + A.gd.offf[0]); });   // g = generator.predict(s, a)
  gbuf[r * 16 + c] = c < S ? t.out[r * LD_O + c] : 0.f;
  double s7[7]; uint8_t done = 0;
  const int ei = row0 + (int)threadIdx.x;                                  // the environment of threads 0..15
  const bool env_thread = threadIdx.x < ROWS && ei < A.n_env;
  if (env_thread) {                                                        // s_, r, done = env.step(a)
    const double rew = env_step_one(A.env, ei, a64[threadIdx.x * 2], a64[threadIdx.x * 2 + 1], s7, &done);
#pragma unroll
    for (int k = 0; k < 7; ++k) nbuf[threadIdx.x * 16 + k] = (float)s7[k];
    rbuf[threadIdx.x * 2] = (float)rew;
    if (A.reward_out) A.reward_out[ei] = rew;
    if (A.done_out) A.done_out[ei] = done;
  }
  lds_barrier();
  load_x0(t.x0, row0, A.n_env, lin(sbuf, 16, S), lin(gbuf, 16, S), lin(abuf, 16, Ad), nullptr);
  lds_barrier();
  net_forward<NET_DISC, false>(A.disc.wf, A.gd, t, none, p_d, NoNext{});
  if (threadIdx.x < ROWS)                                                  // curious_r = r + ITA*(1 - D_fake)
    rbuf[threadIdx.x * 2 + 1] = rbuf[threadIdx.x * 2] + A.ita * (1.f - sigmoid_f(t.out[threadIdx.x * LD_O]));
  lds_barrier();
  // M.store_transition(s, s_, g, a, r, curious_r)  (Curious_GAN.py:161-165,195)
  for (int e = threadIdx.x; e < ROWS * 32; e += THREADS) {
    const int rr = e >> 5, cc = e & 31;
    const int i = row0 + rr;
    if (i >= A.n_env || cc >= A.dims) continue;
    float v;
    if (cc < S) v = sbuf[rr * 16 + cc];
    else if (cc < 2 * S) v = nbuf[rr * 16 + cc - S];
    else if (cc < 3 * S) v = gbuf[rr * 16 + cc - 2 * S];
    else if (cc < 3 * S + Ad) v = abuf[rr * 16 + cc - 3 * S];
    else v = rbuf[rr * 2 + (cc - 3 * S - Ad)];
    const long long slot = (ptr0 + i) % A.capacity;
    A.replay[(size_t)slot * A.ld_replay + cc] = v;
  }
  // the state the next step starts from: s', or a fresh episode (Curious_GAN.py:180,222-234)
  if (env_thread) {
    if (A.ep_t) {
      const int et = A.ep_t[ei];
      if (done || et >= A.max_ep_steps - 1) {
        double u0, u1;
        if (A.reset_u) { u0 = A.reset_u[2 * ei]; u1 = A.reset_u[2 * ei + 1]; }
        else philox_uniform_pair((unsigned long long)(A.n_env + step * A.n_env + ei), STREAM_GOAL, A.rng_seed, u0, u1);
        env_reset_one(A.env, ei, A.mode_hard, u0, u1, s7);
        A.ep_t[ei] = 0;
        if (A.n_resets) atomicAdd(A.n_resets, 1);
      } else {
        A.ep_t[ei] = et + 1;
      }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) A.state[(size_t)ei * S + k] = (float)s7[k];
  }
}

// what the last workgroup of a launch does for the acting step: advance the counters, decay the
// noise scale once learning has started (`if M.pointer > MEMORY_CAPACITY: var = max(var*.9999, VAR_MIN)`,
// Curious_GAN.py:197-198)
__device__ __forceinline__ void act_finish(const ActArgs& A) {
  if (threadIdx.x != 0 || A.counters_out) return;
  long long p = A.pointer + A.n_env;
  if (A.counters) { p = A.counters[0] + A.n_env; A.counters[0] = p; A.counters[1] += 1; }
  if (A.var && A.var_decay > 0.0 && p > A.capacity) A.var[0] = fmax(A.var[0] * A.var_decay, A.var_min);
}

// ------------------------------------------------------------------------------------------
// (5)+(6) D and G losses after both updates   Curious_GAN.py:94-95,147-148
// two independent tiles of work per 16 minibatch rows: the D loss (fake + real pass) and the G loss
// ------------------------------------------------------------------------------------------
struct EvalArgs {
  Geo gg, gd;
  NetP gen, disc;
  Mat s, g, a, s_next;
  const float* x0d; const float* x0g;   // optional: input tiles left by the D step (fake rows, then real rows) and the G step
  float label_real, loss_scale;
  int n_rows, n_tiles;
  float* loss_parts;          // [2][n_tiles]: D loss sums, then G loss sums
  float* loss_out;            // optional [2]: loss_scale * totals, written by the last workgroup
};
constexpr int EVAL_LDS = PART_FLOATS + SET_FLOATS + 64;

// input tile from a global x0 work array tile (as load_x0 mirrors it, layout: work_off), read in
// memory order
__device__ __forceinline__ void load_x0_tile(float* x0, const float* __restrict__ g) {
  for (int e = threadIdx.x; e < ROWS * X0_COLS; e += THREADS) {
    const int r = 4 * (e / (4 * X0_COLS)) + (e & 3), c = (e >> 2) % X0_COLS;
    x0[r * LD_X0 + c] = g[e];
  }
}

__device__ __forceinline__ void eval_d_tile(const EvalArgs& A, float* smem, int tile) {
  const TileSet sd = carve_set(smem + PART_FLOATS);
  float* small = smem + PART_FLOATS + SET_FLOATS;
  const Tiles td = mk(sd, smem);
  const int row0 = tile * ROWS;
  const int S = A.gd.S, Ad = A.gd.A;
  const ActsG none{};
  FwdPre<NET_DISC> p_d; p_d.issue(A.disc.wf + A.gd.offf[0]);
  float ld = 0.f;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (A.x0d) load_x0_tile(td.x0, A.x0d + (size_t)(pass * A.n_tiles + tile) * (ROWS * X0_COLS));
    else load_x0(td.x0, row0, A.n_rows, gin(A.s, S), gin(pass == 0 ? A.g : A.s_next, S), gin(A.a, Ad), nullptr);
    lds_barrier();
    net_forward<NET_DISC, false>(A.disc.wf, A.gd, td, none, p_d, [&] { p_d.issue(A.disc.wf + A.gd.offf[0]); });
    float l = 0.f;
    if (threadIdx.x < ROWS && row0 + threadIdx.x < A.n_rows) l = sigmoid_ce(td.out[threadIdx.x * LD_O], pass == 0 ? 0.f : A.label_real);
    ld += sum16_lds(small, l);
  }
  if (threadIdx.x == 0) publish(A.loss_parts + tile, ld);
}

__device__ __forceinline__ void eval_g_tile(const EvalArgs& A, float* smem, int tile) {
  const TileSet ts = carve_set(smem + PART_FLOATS);
  float* small = smem + PART_FLOATS + SET_FLOATS;
  const Tiles t = mk(ts, smem);
  const int row0 = tile * ROWS;
  const int S = A.gg.S, Ad = A.gg.A;
  const ActsG none{};
  FwdPre<NET_GEN> p_g; p_g.issue(A.gen.wf + A.gg.offf[0]);
  FwdPre<NET_DISC> p_d;
  if (A.x0g) load_x0_tile(t.x0, A.x0g + (size_t)tile * (ROWS * X0_COLS));
  else load_x0(t.x0, row0, A.n_rows, gin(A.s, S), gin(A.a, Ad), nil(), nullptr);
  lds_barrier();
  net_forward<NET_GEN, false>(A.gen.wf, A.gg, t, none, p_g, [&] { p_d.issue(A.disc.wf + A.gd.offf[0]); });
  // the generated state moves out of t.out before the discriminator's forward pass reuses the tiles
  float* gsave = t.h3;                                   // free until the discriminator's third layer; x0 is built first
  {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    gsave[r * 16 + c] = t.out[r * LD_O + c];
  }
  lds_barrier();
  if (A.x0d) {
    // [s, x, a, 1] of the fake rows with x replaced by the generated state (zero rows beyond the batch)
    load_x0_tile(t.x0, A.x0d + (size_t)tile * (ROWS * X0_COLS));
    lds_barrier();
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (c < S) t.x0[r * LD_X0 + S + c] = row0 + r < A.n_rows ? gsave[r * 16 + c] : 0.f;
  } else {
    load_x0(t.x0, row0, A.n_rows, gin(A.s, S), lin(gsave, 16, S), gin(A.a, Ad), nullptr);
  }
  lds_barrier();
  net_forward<NET_DISC, false>(A.disc.wf, A.gd, t, none, p_d, NoNext{});
  float l = 0.f;
  if (threadIdx.x < ROWS && row0 + threadIdx.x < A.n_rows) l = sigmoid_ce(t.out[threadIdx.x * LD_O], 1.f);
  const float lg = sum16_lds(small, l);
  if (threadIdx.x == 0) publish(A.loss_parts + A.n_tiles + tile, lg);
}

__device__ __forceinline__ void eval_finish(const EvalArgs& A) {
  if (!A.loss_out || threadIdx.x >= 64) return;
  const float d = wave_sum_parts(A.loss_parts, A.n_tiles, A.loss_scale);
  const float g = wave_sum_parts(A.loss_parts + A.n_tiles, A.n_tiles, A.loss_scale);
  if (threadIdx.x == 0) { A.loss_out[0] = d; A.loss_out[1] = g; }
}

// ------------------------------------------------------------------------------------------
// one launch: acting tiles, D-loss tiles, G-loss tiles (one of each per compute unit when the tile counts
// are equal), plus one workgroup drawing the next minibatch's replay indices
// ------------------------------------------------------------------------------------------
struct SampleArgs { uint32_t* mt_state; uint32_t rng, mask; int n; long long* out; };

constexpr int ROLLOUT_LDS = ACT_LDS > EVAL_LDS ? ACT_LDS : EVAL_LDS;
static_assert(ROLLOUT_LDS >= MT_LDS_WORDS, "the sampler borrows the tile memory");

__global__ __launch_bounds__(THREADS) void rollout_eval_kernel(const ActArgs A, const EvalArgs E, const SampleArgs Sm,
                                                               int n_act, int n_eval, int32_t* ticket) {
  __shared__ __attribute__((aligned(16))) float smem[ROLLOUT_LDS];
  __shared__ int flag;
  // The index draw is one workgroup's serial chain (about 13 us alone for 4096 indices): it goes first
  // and at raised priority, because the youngest waves of a compute unit get the issue slots the older
  // ones leave, and behind three MFMA-bound tiles the chain would end well after them.  Then ranges,
  // longest role first: acting tiles, D-loss tiles, G-loss tiles.  Workgroups are placed breadth-first
  // over the compute units, so equal tile counts give every unit one of each.
  const int n_draw = Sm.mt_state ? 1 : 0;
  const int b = (int)blockIdx.x - n_draw;
  if (b < 0) {
    __builtin_amdgcn_s_setprio(3);
    mt_choice_body<THREADS>(Sm.mt_state, Sm.rng, Sm.mask, Sm.n, Sm.out, (uint32_t*)smem);
  } else if (b < n_act) {
    act_tile(A, smem, b);
  } else if (b < n_act + n_eval) {
    eval_d_tile(E, smem, b - n_act);
  } else {
    eval_g_tile(E, smem, b - n_act - n_eval);
  }
  if (!ticket) return;
  if (last_workgroup(ticket, gridDim.x, &flag)) {
    if (n_act) act_finish(A);
    if (n_eval) eval_finish(E);
  }
}

inline EnvP EP(const rlgan_env_t& e) { return EnvP{e.theta, e.ends, e.goal, e.grab, e.got}; }
inline bool env_ok(const rlgan_env_t& e) { return e.theta && e.ends && e.goal && e.grab && e.got; }

int make_act_args(ActArgs& A, const rlgan_act_t* a) {
  if (!a) return RLGAN_E_ARG;
  if (!make_geo(NET_ACTOR, a->s_dim, a->a_dim, A.ga) || !make_geo(NET_GEN, a->s_dim, a->a_dim, A.gg) ||
      !make_geo(NET_DISC, a->s_dim, a->a_dim, A.gd) || a->s_dim != 7 || a->a_dim != 2) return RLGAN_E_DIMS;
  if (!env_ok(a->env) || a->n_env <= 0 || !a->actor_eval.wf || !a->gen.wf || !a->disc.wf || !a->state_f32 || !a->replay ||
      ((a->noise || a->rng_seed) && !a->var) || a->capacity <= 0 || a->pointer < 0 || a->step < 0) return RLGAN_E_ARG;
  if (a->ep_t && (a->max_ep_steps <= 0 || (!a->reset_u && !a->rng_seed))) return RLGAN_E_ARG;
  if (a->dims != 3 * a->s_dim + a->a_dim + 2 || a->ld_replay < a->dims || a->dims > 32 || a->n_env > a->capacity) return RLGAN_E_ALIGN;
  A.actor = NetP{a->actor_eval.wf, nullptr}; A.gen = NetP{a->gen.wf, nullptr}; A.disc = NetP{a->disc.wf, nullptr};
  A.env = EP(a->env); A.n_env = a->n_env; A.n_tiles = cdiv(a->n_env, ROWS);
  A.state = a->state_f32; A.noise = a->noise; A.var = a->var; A.ita = a->ita;
  A.replay = a->replay; A.capacity = a->capacity; A.dims = a->dims; A.ld_replay = a->ld_replay;
  A.pointer = a->pointer; A.step = a->step; A.counters = (long long*)a->counters;
  A.reward_out = a->reward_out; A.done_out = a->done_out; A.rng_seed = a->rng_seed;
  A.ep_t = a->ep_t; A.max_ep_steps = a->max_ep_steps; A.mode_hard = a->mode_hard; A.reset_u = a->reset_u; A.n_resets = a->n_resets;
  A.var_decay = a->var_decay; A.var_min = a->var_min;
  if (a->counters_out && (!a->counters || a->counters_out == a->counters)) return RLGAN_E_ARG;
  if (a->counters_out && a->var && (!a->var_out || a->var_out == a->var)) return RLGAN_E_ARG;
  if (!a->counters_out && a->var_out) return RLGAN_E_ARG;
  A.counters_out = (long long*)a->counters_out; A.var_out = a->var_out;
  return 0;
}

int make_eval_args(EvalArgs& A, const rlgan_eval_t* e) {
  if (!e) return RLGAN_E_ARG;
  if (!make_geo(NET_GEN, e->s_dim, e->a_dim, A.gg) || !make_geo(NET_DISC, e->s_dim, e->a_dim, A.gd)) return RLGAN_E_DIMS;
  const bool from_x0 = e->x0_disc && e->x0_gen;
  if (e->n_rows <= 0 || !e->gen.wf || !e->disc.wf || !e->loss_parts || (e->x0_disc != nullptr) != (e->x0_gen != nullptr)) return RLGAN_E_ARG;
  if (!from_x0 && (!e->s.p || !e->g_stored.p || !e->a.p || !e->s_next.p)) return RLGAN_E_ARG;
  A.x0d = e->x0_disc; A.x0g = e->x0_gen;
  A.gen = NP(e->gen); A.disc = NP(e->disc);
  A.s = M(e->s, e->row_idx); A.g = M(e->g_stored, e->row_idx); A.a = M(e->a, e->row_idx); A.s_next = M(e->s_next, e->row_idx);
  A.label_real = e->label_real; A.loss_scale = e->loss_scale; A.n_rows = e->n_rows; A.n_tiles = cdiv(e->n_rows, ROWS);
  A.loss_parts = e->loss_parts; A.loss_out = e->loss_out;
  return 0;
}

int make_sample_args(SampleArgs& Sm, const rlgan_sample_t* s) {
  Sm = SampleArgs{nullptr, 0, 0, 0, nullptr};
  if (!s) return 0;
  if (!s->mt_state || !s->out_idx || s->n <= 0 || s->n_pop <= 0 || s->n_pop > 0x100000000ll) return RLGAN_E_ARG;
  const uint32_t rng = (uint32_t)(s->n_pop - 1);
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  Sm = SampleArgs{s->mt_state, rng, mask, s->n, (long long*)s->out_idx};
  return 0;
}

int launch_rollout(const rlgan_act_t* act, const rlgan_eval_t* eval, const rlgan_sample_t* sample, int32_t* ticket, void* stream) {
  ActArgs A{}; EvalArgs E{}; SampleArgs Sm;
  int n_act = 0, n_eval = 0;
  if (act) { if (const int rc = make_act_args(A, act)) return rc; n_act = A.n_tiles; }
  if (eval) { if (const int rc = make_eval_args(E, eval)) return rc; n_eval = E.n_tiles; }
  if (const int rc = make_sample_args(Sm, sample)) return rc;
  // the last-workgroup step is needed for the counters, the noise-scale decay and the loss totals
  const bool need_ticket = (act && !act->counters_out && (act->counters || act->var_decay > 0.0)) || (eval && eval->loss_out);
  if (need_ticket && !ticket) return RLGAN_E_ARG;
  const int grid = n_act + 2 * n_eval + (Sm.mt_state ? 1 : 0);
  if (grid <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(rollout_eval_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, A, E, Sm, n_act, n_eval,
                     need_ticket ? ticket : nullptr);
  return launched();
}

}  // namespace

extern "C" {

int rlgan_env_step(rlgan_env_t env, int n_env, const void* action, int action_f64, int ld_action,
                   double* state_out, float* state_out_f32, double* reward, uint8_t* done, void* stream) {
  if (!env_ok(env) || n_env <= 0 || !action || ld_action < 2) return RLGAN_E_ARG;
  hipLaunchKernelGGL(env_step_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, (hipStream_t)stream,
                     EP(env), n_env, action, action_f64, ld_action, state_out, state_out_f32, reward, done);
  return launched();
}

int rlgan_env_reset(rlgan_env_t env, int n_env, int mode_hard, const uint8_t* mask, const double* u,
                    double* state_out, float* state_out_f32, void* stream) {
  if (!env_ok(env) || n_env <= 0 || !u) return RLGAN_E_ARG;
  hipLaunchKernelGGL(env_reset_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, (hipStream_t)stream,
                     EP(env), n_env, mode_hard, mask, u, state_out, state_out_f32);
  return launched();
}

int rlgan_env_autoreset(rlgan_env_t env, int n_env, int mode_hard, const uint8_t* done, int32_t* ep_t,
                        int max_ep_steps, const double* u, float* state_f32, int32_t* n_resets,
                        int64_t* counters, void* stream) {
  if (!env_ok(env) || n_env <= 0 || !u || !done || !ep_t || !state_f32 || max_ep_steps <= 0) return RLGAN_E_ARG;
  hipLaunchKernelGGL(env_autoreset_kernel, dim3(cdiv(n_env, 256)), dim3(256), 0, (hipStream_t)stream,
                     EP(env), n_env, mode_hard, done, ep_t, max_ep_steps, u, state_f32, n_resets, (long long*)counters);
  return launched();
}

int rlgan_act_rows(const rlgan_act_t* act, int32_t* ticket, void* stream) {
  if (!act) return RLGAN_E_ARG;
  return launch_rollout(act, nullptr, nullptr, ticket, stream);
}

int rlgan_gan_eval_rows(const rlgan_eval_t* eval, int32_t* ticket, void* stream) {
  if (!eval) return RLGAN_E_ARG;
  return launch_rollout(nullptr, eval, nullptr, ticket, stream);
}

int rlgan_rollout_eval_rows(const rlgan_act_t* act, const rlgan_eval_t* eval, const rlgan_sample_t* sample,
                            int32_t* ticket, void* stream) {
  if (!act && !eval && !sample) return RLGAN_E_ARG;
  return launch_rollout(act, eval, sample, ticket, stream);
}

}  // extern "C"
