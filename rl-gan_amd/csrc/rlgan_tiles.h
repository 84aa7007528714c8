// Pieces shared by the learner and the rollout translation units: argument records, the LDS carve of
// one network's tiles, small workgroup reductions and the end-of-launch ticket.
#pragma once
#include "rlgan_device.h"
#include "rlgan_host.h"

namespace rlgan {
#ifdef __HIPCC__

struct NetP { const float* wf; const float* wb; };
struct Mat { const float* p; int ld; const long long* idx; };    // idx: optional row indirection (see MatIn)
struct Acts8 { float* p[8]; };

__device__ __forceinline__ MatIn gin(Mat m, int w) { return MatIn{m.p, m.ld, w, 0, m.idx}; }
__device__ __forceinline__ MatIn lin(const float* p, int ld, int w) { return MatIn{p, ld, w, 1, nullptr}; }
__device__ __forceinline__ MatIn nil() { return MatIn{nullptr, 0, 0, 0, nullptr}; }
__device__ __forceinline__ float mat_at(const Mat& m, int gr, int c) {
  const size_t row = m.idx ? (size_t)m.idx[gr] : (size_t)gr;
  return m.p[row * m.ld + c];
}

// The minibatch fields a tile works on, fetched once: up to four column blocks side by side in an LDS
// tile [16][STAGE_LD], zeros for rows at or beyond n_rows.  A tile that runs several network passes on
// the same rows reads them from here (lin(stage + column, STAGE_LD, width)) instead of paying a chain of
// two global round trips (row index, then row) at the start of every pass.  No barrier inside.
constexpr int STAGE_LD = 32, STAGE_FLOATS = ROWS * STAGE_LD;
template <int NW = 4>
__device__ __forceinline__ void stage_rows(float* stage, int row0, int n_rows, MatIn b0, MatIn b1, MatIn b2, MatIn b3) {
  constexpr int NTHR = NW * 64;
  const int e0 = b0.w, e1 = e0 + b1.w, e2 = e1 + b2.w, e3 = e2 + b3.w;
  const bool all_idx = b0.idx && (!b1.w || b1.idx) && (!b2.w || b2.idx) && (!b3.w || b3.idx);
  const bool no_idx = !b0.idx && !b1.idx && !b2.idx && !b3.idx;
  const bool global = !b0.local && !b1.local && !b2.local && !b3.local;
  if (global && (all_idx || no_idx)) {
    // The usual case, without a branch around any load: which block a lane reads is a matter of selects, so the
    // index loads of all its elements go out together, then the row loads -- two memory round trips in all.
    // (One branch per block made it eight, each waited for where the branches join.)
    constexpr int PER = STAGE_FLOATS / NTHR;
    long long row[PER];
    const float* src[PER];
    bool ok[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = threadIdx.x + i * NTHR, r = e / STAGE_LD, c = e % STAGE_LD, gr = row0 + r;
      ok[i] = gr < n_rows && c < e3;
      const int which = ok[i] ? (c >= e0) + (c >= e1) + (c >= e2) : 0;
      const float* p = which == 0 ? b0.p : which == 1 ? b1.p : which == 2 ? b2.p : b3.p;
      const long long* ix = which == 0 ? b0.idx : which == 1 ? b1.idx : which == 2 ? b2.idx : b3.idx;
      const int ld = which == 0 ? b0.ld : which == 1 ? b1.ld : which == 2 ? b2.ld : b3.ld;
      const int cc = ok[i] ? c - (which == 0 ? 0 : which == 1 ? e0 : which == 2 ? e1 : e2) : 0;
      const int g0 = ok[i] ? gr : 0;                          // row 0 exists
      row[i] = all_idx ? ix[g0] : (long long)g0;
      src[i] = p + cc;
      row[i] *= ld;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const float v = src[i][row[i]];
      stage[threadIdx.x + i * NTHR] = ok[i] ? v : 0.f;
    }
    return;
  }
  for (int e = threadIdx.x; e < STAGE_FLOATS; e += NTHR) {
    const int r = e / STAGE_LD, c = e % STAGE_LD;
    const int gr = row0 + r;
    float v = 0.f;
    if (gr < n_rows && c < e3) {
      if (c < e0) v = matin_at(b0, gr, r, c);
      else if (c < e1) v = matin_at(b1, gr, r, c - e0);
      else if (c < e2) v = matin_at(b2, gr, r, c - e1);
      else v = matin_at(b3, gr, r, c - e2);
    }
    stage[e] = v;
  }
}

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

// the same for a network whose third layer is 10 wide (actor, critic, discriminator): h3 is a 16-column tile
constexpr int SET10_FLOATS = X0_FLOATS + 2 * WIDE_FLOATS + 2 * O_FLOATS;
__device__ __forceinline__ TileSet carve_set10(float* base) {
  TileSet s;
  s.x0 = base; base += X0_FLOATS;
  s.h1 = base; base += WIDE_FLOATS;
  s.h2 = base; base += WIDE_FLOATS;
  s.h3 = base; base += O_FLOATS;
  s.out = base;
  return s;
}

// sum of v over the first 16 threads' values, returned to thread 0 (deterministic order)
__device__ __forceinline__ float sum16_lds(float* scratch16, float v) {
  if (threadIdx.x < ROWS) scratch16[threadIdx.x] = v;
  lds_barrier();
  float s = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) s += scratch16[i];
  }
  lds_barrier();
  return s;
}

// a [16][16] "dz4" tile whose column 0 comes from v16 (LDS), zeros elsewhere (threads beyond the tile's 256 do nothing)
__device__ __forceinline__ void fill_out_col0(float* out, const float* v16) {
  if (threadIdx.x >= 256) return;
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  out[r * LD_O + c] = (c == 0) ? v16[r] : 0.f;
}

// End-of-launch ticket.  Results that the last workgroup of a launch gathers from all others are
// written with publish() (a store performed at device scope, past this chip's non-coherent L2s) and
// read back with wave_sum_parts(); no cache-wide release fence is involved, which on this part would
// write back the whole L2 once per workgroup.
__device__ __forceinline__ void publish(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every workgroup calls this once, after its last read of anything the final step will overwrite and
// after thread 0 has publish()ed the workgroup's results.  Returns true (to all threads) in exactly
// one workgroup, the last to arrive; that workgroup also re-arms the ticket for the next launch.
// `flag` is one int of LDS.
__device__ __forceinline__ bool last_workgroup(int32_t* ticket, int n_workgroups, int* flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's published stores have been performed
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == n_workgroups - 1;
    if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}

// scale * sum(parts[0..n)) of publish()ed values by the first wave of the workgroup: lanes stride in
// index order, then a fixed-order lane reduction (the order rlgan_sum_parts uses); valid in thread 0
__device__ __forceinline__ float wave_sum_parts(const float* parts, int n, float scale) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += __hip_atomic_load(parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  return s * scale;
}

inline Mat M(rlgan_mat_t m, const int64_t* idx) { return Mat{m.p, m.ld, (const long long*)idx}; }
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

#endif  // __HIPCC__
}  // namespace rlgan
