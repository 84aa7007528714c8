// The 2-link arm environment on the device (arm_env.py:43-115): float64, one thread per environment.
// Every function body turns floating-point contraction off so that the arithmetic keeps the
// reference's operation order (no fused multiply-adds), whatever the translation unit's default.
#pragma once
#include "rlgan_device.h"

namespace rlgan {
#ifdef __HIPCC__

constexpr double ENV_DT = 0.1;          // arm_env.py:22
constexpr double ENV_ARM_L = 100.0;     // arm_env.py:23-24
constexpr double ENV_VIEW = 400.0;      // arm_env.py:26
constexpr double ENV_CENTER = 200.0;    // arm_env.py:41
constexpr double ENV_POINT_L = 15.0;    // arm_env.py:29
constexpr int ENV_GRAB_T = 50;          // arm_env.py:102
constexpr double ENV_TWO_PI = 6.283185307179586;   // np.pi * 2

struct EnvP { double* theta; double* ends; double* goal; int32_t* grab; uint8_t* got; };

// Python's float modulo (sign of the divisor), what `arm_info[:, 1] %= np.pi * 2` does (arm_env.py:47)
__device__ __forceinline__ double pymod(double a, double b) {
#pragma clang fp contract(off)
  double m = fmod(a, b);
  if (m != 0.0) { if ((b < 0.0) != (m < 0.0)) m += b; }
  else m = copysign(0.0, b);
  return m;
}
__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// _get_state (arm_env.py:91-99)
__device__ __forceinline__ void env_state(const double* e, const double* g, int grab, double* s) {
#pragma clang fp contract(off)
  s[0] = grab > 0 ? 1.0 : 0.0;
  s[1] = (e[0] - g[0]) / 200.0; s[2] = (e[1] - g[1]) / 200.0;
  s[3] = (e[2] - g[0]) / 200.0; s[4] = (e[3] - g[1]) / 200.0;
  s[5] = (ENV_CENTER - g[0]) / 200.0; s[6] = (ENV_CENTER - g[1]) / 200.0;
}
__device__ __forceinline__ void env_ends(const double* th, double* e) {
#pragma clang fp contract(off)
  const double dx1 = ENV_ARM_L * cos(th[0]), dy1 = ENV_ARM_L * sin(th[0]);
  const double dx2 = ENV_ARM_L * cos(th[1]), dy2 = ENV_ARM_L * sin(th[1]);
  e[0] = ENV_CENTER + dx1; e[1] = ENV_CENTER + dy1;
  e[2] = e[0] + dx2;       e[3] = e[1] + dy2;
}

// step + _r_func (arm_env.py:43-59,101-115) for environment i; returns the reward
__device__ __forceinline__ double env_step_one(const EnvP& E, int i, double a0, double a1, double* s, uint8_t* done) {
#pragma clang fp contract(off)
  double th[2] = {E.theta[2 * i], E.theta[2 * i + 1]};
  double e[4];
  const double g[2] = {E.goal[2 * i], E.goal[2 * i + 1]};
  int grab = E.grab[i];
  bool got = E.got[i] != 0;
  th[0] = pymod(th[0] + clipd(a0, -1.0, 1.0) * ENV_DT, ENV_TWO_PI);
  th[1] = pymod(th[1] + clipd(a1, -1.0, 1.0) * ENV_DT, ENV_TWO_PI);
  env_ends(th, e);
  env_state(e, g, grab, s);                       // in_point is read BEFORE the counter moves
  const double dx = e[2] - g[0], dy = e[3] - g[1];
  const double dist = sqrt(dx * dx + dy * dy);
  double r = 0.0;
  if (dist < ENV_POINT_L && !got) {
    grab += 1;
    if (grab > ENV_GRAB_T) { r += 10.0; got = true; }
  } else if (dist > ENV_POINT_L) {
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
#pragma clang fp contract(off)
  E.got[i] = 0; E.grab[i] = 0;
  double e[4] = {E.ends[4 * i], E.ends[4 * i + 1], E.ends[4 * i + 2], E.ends[4 * i + 3]};
  double g[2];
  if (mode_hard) {
    // assignment into the integer goal array truncates toward zero (arm_env.py:39,66-67); angles and
    // link ends are left as they are
    g[0] = trunc(clipd(u0 * ENV_VIEW, 100.0, 300.0));
    g[1] = trunc(clipd(u1 * ENV_VIEW, 100.0, 300.0));
  } else {
    double th[2] = {u0 * ENV_TWO_PI, u1 * ENV_TWO_PI};
    E.theta[2 * i] = th[0]; E.theta[2 * i + 1] = th[1];
    env_ends(th, e);
    E.ends[4 * i] = e[0]; E.ends[4 * i + 1] = e[1]; E.ends[4 * i + 2] = e[2]; E.ends[4 * i + 3] = e[3];
    g[0] = 250.0; g[1] = 303.0;
  }
  E.goal[2 * i] = g[0]; E.goal[2 * i + 1] = g[1];
  env_state(e, g, 0, s);
}

#endif  // __HIPCC__
}  // namespace rlgan
