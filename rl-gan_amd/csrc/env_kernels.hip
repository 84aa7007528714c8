// Environment and acting-path kernels: the 2-link arm (float64, one thread per environment), its
// reset, the per-episode auto-reset, and the fused acting step
//   actor -> exploration noise -> generator -> env.step -> discriminator -> intrinsic reward -> replay row.
// gfx950 only.  Compiled with -ffp-contract=off so the float64 environment arithmetic keeps the
// reference's operation order (no fused multiply-adds).
#include "rlgan_device.h"
#include "rlgan_host.h"

using namespace rlgan;

namespace {

constexpr double DT = 0.1;          // arm_env.py:22
constexpr double ARM_L = 100.0;     // arm_env.py:23-24
constexpr double VIEW = 400.0;      // arm_env.py:26
constexpr double CENTER = 200.0;    // arm_env.py:41
constexpr double POINT_L = 15.0;    // arm_env.py:29
constexpr int GRAB_T = 50;          // arm_env.py:102
constexpr double TWO_PI = 6.283185307179586;   // np.pi * 2

struct EnvP { double* theta; double* ends; double* goal; int32_t* grab; uint8_t* got; };

// Python's float modulo (sign of the divisor), what `arm_info[:, 1] %= np.pi * 2` does (arm_env.py:47)
__device__ __forceinline__ double pymod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; }
  else m = copysign(0.0, b);
  return m;
}
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// _get_state (arm_env.py:91-99)
__device__ __forceinline__ void env_state(const double* e, const double* g, int grab, double* s) {
  s[0] = grab > 0 ? 1.0 : 0.0;
  s[1] = (e[0] - g[0]) / 200.0; s[2] = (e[1] - g[1]) / 200.0;
  s[3] = (e[2] - g[0]) / 200.0; s[4] = (e[3] - g[1]) / 200.0;
  s[5] = (CENTER - g[0]) / 200.0; s[6] = (CENTER - g[1]) / 200.0;
}
__device__ __forceinline__ void env_ends(const double* th, double* e) {
  const double dx1 = ARM_L * cos(th[0]), dy1 = ARM_L * sin(th[0]);
  const double dx2 = ARM_L * cos(th[1]), dy2 = ARM_L * sin(th[1]);
  e[0] = CENTER + dx1; e[1] = CENTER + dy1;
  e[2] = e[0] + dx2;   e[3] = e[1] + dy2;
}

// step + _r_func (arm_env.py:43-59,101-115) for environment i; returns the reward
__device__ __forceinline__ double env_step_one(const EnvP& E, int i, double a0, double a1, double* s, uint8_t* done) {
  double th[2] = {E.theta[2 * i], E.theta[2 * i + 1]};
  double e[4];
  const double g[2] = {E.goal[2 * i], E.goal[2 * i + 1]};
  int grab = E.grab[i];
  bool got = E.got[i] != 0;
  th[0] = pymod(th[0] + clipd(a0, -1.0, 1.0) * DT, TWO_PI);
  th[1] = pymod(th[1] + clipd(a1, -1.0, 1.0) * DT, TWO_PI);
  env_ends(th, e);
  env_state(e, g, grab, s);                       // in_point is read BEFORE the counter moves
  const double dx = e[2] - g[0], dy = e[3] - g[1];
  const double dist = sqrt(dx * dx + dy * dy);
  double r = 0.0;
  if (dist < POINT_L && !got) {
    grab += 1;
    if (grab > GRAB_T) { r += 10.0; got = true; }
  } else if (dist > POINT_L) {
    grab = 0; got = false;
  }
  E.theta[2 * i] = th[0]; E.theta[2 * i + 1] = th[1];
  E.ends[4 * i] = e[0]; E.ends[4 * i + 1] = e[1]; E.ends[4 * i + 2] = e[2]; E.ends[4 * i + 3] = e[3];
  E.grab[i] = grab; E.got[i] = got ? 1 : 0;
  *done = got ? 1 : 0;
  return r;
}

// reset (arm_env.py:61-78)
__device__ __forceinline__ void env_reset_one(const EnvP& E, int i, int mode_hard, double u0, double u1, double* s) {
  E.got[i] = 0; E.grab[i] = 0;
  double e[4] = {E.ends[4 * i], E.ends[4 * i + 1], E.ends[4 * i + 2], E.ends[4 * i + 3]};
  double g[2];
  if (mode_hard) {
    // assignment into the integer goal array truncates toward zero (arm_env.py:39,66-67); angles and
    // link ends are left as they are
    g[0] = trunc(clipd(u0 * VIEW, 100.0, 300.0));
    g[1] = trunc(clipd(u1 * VIEW, 100.0, 300.0));
  } else {
    double th[2] = {u0 * TWO_PI, u1 * TWO_PI};
    E.theta[2 * i] = th[0]; E.theta[2 * i + 1] = th[1];
    env_ends(th, e);
    E.ends[4 * i] = e[0]; E.ends[4 * i + 1] = e[1]; E.ends[4 * i + 2] = e[2]; E.ends[4 * i + 3] = e[3];
    g[0] = 250.0; g[1] = 303.0;
  }
  E.goal[2 * i] = g[0]; E.goal[2 * i + 1] = g[1];
  env_state(e, g, 0, s);
}

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
// fused acting step for 16 environments per workgroup (Curious_GAN.py:190-195)
// ------------------------------------------------------------------------------------------
struct NetP { const float* wf; const float* wb; };
struct ActArgs {
  Geo ga, gg, gd;
  NetP actor, gen, disc;
  EnvP env;
  int n_env;
  float* state;          // [n,7] fp32, in: s, out: s'
  const float* noise;    // [n,2] standard normal or null
  const float* var;      // device scalar
  float ita;
  float* replay; int capacity, dims, ld_replay; long long pointer; const long long* pointer_dev;
  double* reward_out; uint8_t* done_out;
};

__global__ __launch_bounds__(THREADS) void act_kernel(const ActArgs A) {
  constexpr int SETF = X0_FLOATS + 2 * WIDE_FLOATS + T3_FLOATS + O_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[PART_FLOATS + SETF + 5 * ROWS * 16];
  float* part = smem;
  float* b = smem + PART_FLOATS;
  Tiles t;
  t.part = part;
  t.x0 = b; b += X0_FLOATS; t.h1 = b; b += WIDE_FLOATS; t.h2 = b; b += WIDE_FLOATS; t.h3 = b; b += T3_FLOATS; t.out = b; b += O_FLOATS;
  float* sbuf = b; b += ROWS * 16;   // s
  float* abuf = b; b += ROWS * 16;   // noisy clipped action (fp32 view)
  float* gbuf = b; b += ROWS * 16;   // g^
  float* nbuf = b; b += ROWS * 16;   // s'
  float* rbuf = b;                   // r, curious_r
  const int row0 = blockIdx.x * ROWS;
  const int S = A.ga.S, Ad = A.ga.A;
  const ActsG none{};
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int gr = row0 + r;
  const bool valid = gr < A.n_env;

  Pre<NT_H, 2> p_a; p_a.issue(A.actor.wf + A.ga.offf[0]);
  sbuf[r * 16 + c] = (valid && c < S) ? A.state[(size_t)gr * S + c] : 0.f;
  __syncthreads();
  load_x0(t.x0, row0, A.n_env, MatIn{sbuf, 16, S, 1}, MatIn{nullptr, 0, 0, 0}, MatIn{nullptr, 0, 0, 0}, nullptr);
  __syncthreads();
  Pre<NT_H, 2> p_g;
  net_forward<NET_ACTOR, false>(A.actor.wf, A.ga, t, none, p_a, [&] { p_g.issue(A.gen.wf + A.gg.offf[0]); });   // a = actor.choose_action(s)
  // a = clip(normal(a, var), -1, 1) evaluated in float64 like NumPy (Curious_GAN.py:191)
  __shared__ double a64[ROWS * 2];
  if (c < Ad) {
    double av = (double)t.out[r * LD_O + c];
    if (A.noise && valid) av += (double)A.var[0] * (double)A.noise[(size_t)gr * Ad + c];
    av = clipd(av, -1.0, 1.0);
    if (c < 2) a64[r * 2 + c] = av;
    abuf[r * 16 + c] = (float)av;
  } else {
    abuf[r * 16 + c] = 0.f;
  }
  __syncthreads();
  load_x0(t.x0, row0, A.n_env, MatIn{sbuf, 16, S, 1}, MatIn{abuf, 16, Ad, 1}, MatIn{nullptr, 0, 0, 0}, nullptr);
  __syncthreads();
  Pre<NT_H, 2> p_d;
  net_forward<NET_GEN, false>(A.gen.wf, A.gg, t, none, p_g, [&] { p_d.issue(A.disc.wf + A.gd.offf[0]); });   // g = generator.predict(s, a)
  gbuf[r * 16 + c] = c < S ? t.out[r * LD_O + c] : 0.f;
  if (threadIdx.x < ROWS) {                                              // s_, r, done = env.step(a)
    const int i = row0 + threadIdx.x;
    if (i < A.n_env) {
      double s7[7]; uint8_t d;
      const double rew = env_step_one(A.env, i, a64[threadIdx.x * 2], a64[threadIdx.x * 2 + 1], s7, &d);
#pragma unroll
      for (int k = 0; k < 7; ++k) nbuf[threadIdx.x * 16 + k] = (float)s7[k];
      rbuf[threadIdx.x * 2] = (float)rew;
      if (A.reward_out) A.reward_out[i] = rew;
      if (A.done_out) A.done_out[i] = d;
    }
  }
  __syncthreads();
  load_x0(t.x0, row0, A.n_env, MatIn{sbuf, 16, S, 1}, MatIn{gbuf, 16, S, 1}, MatIn{abuf, 16, Ad, 1}, nullptr);
  __syncthreads();
  net_forward<NET_DISC, false>(A.disc.wf, A.gd, t, none, p_d, NoNext{});
  if (threadIdx.x < ROWS)                                                // curious_r = r + ITA*(1 - D_fake)
    rbuf[threadIdx.x * 2 + 1] = rbuf[threadIdx.x * 2] + A.ita * (1.f - sigmoid_f(t.out[threadIdx.x * LD_O]));
  __syncthreads();
  // M.store_transition(s, s_, g, a, r, curious_r)  (Curious_GAN.py:161-165,195)
  const long long ptr0 = A.pointer_dev ? A.pointer_dev[0] : A.pointer;
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
    if (cc < S) A.state[(size_t)i * S + cc] = nbuf[rr * 16 + cc];
  }
}

inline EnvP EP(const rlgan_env_t& e) { return EnvP{e.theta, e.ends, e.goal, e.grab, e.got}; }
inline bool env_ok(const rlgan_env_t& e) { return e.theta && e.ends && e.goal && e.grab && e.got; }
inline int launched() { return (int)hipGetLastError(); }

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

int rlgan_act_rows(int s_dim, int a_dim, rlgan_net_t actor_eval, rlgan_net_t gen, rlgan_net_t disc,
                   rlgan_env_t env, int n_env, float* state_f32, const float* noise, const float* var,
                   float ita, float* replay, int capacity, int dims, int ld_replay, int64_t pointer,
                   const int64_t* pointer_dev, double* reward_out, uint8_t* done_out, void* stream) {
  ActArgs A;
  if (!make_geo(NET_ACTOR, s_dim, a_dim, A.ga) || !make_geo(NET_GEN, s_dim, a_dim, A.gg) ||
      !make_geo(NET_DISC, s_dim, a_dim, A.gd) || s_dim != 7 || a_dim != 2) return RLGAN_E_DIMS;
  if (!env_ok(env) || n_env <= 0 || !actor_eval.wf || !gen.wf || !disc.wf || !state_f32 || !replay ||
      (noise && !var) || capacity <= 0 || pointer < 0) return RLGAN_E_ARG;
  if (dims != 3 * s_dim + a_dim + 2 || ld_replay < dims || dims > 32 || n_env > capacity) return RLGAN_E_ALIGN;
  A.actor = NetP{actor_eval.wf, nullptr}; A.gen = NetP{gen.wf, nullptr}; A.disc = NetP{disc.wf, nullptr};
  A.env = EP(env); A.n_env = n_env; A.state = state_f32; A.noise = noise; A.var = var; A.ita = ita;
  A.replay = replay; A.capacity = capacity; A.dims = dims; A.ld_replay = ld_replay; A.pointer = pointer; A.pointer_dev = (const long long*)pointer_dev;
  A.reward_out = reward_out; A.done_out = done_out;
  hipLaunchKernelGGL(act_kernel, dim3(cdiv(n_env, ROWS)), dim3(THREADS), 0, (hipStream_t)stream, A);
  return launched();
}

}  // extern "C"
