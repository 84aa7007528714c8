// Device-side building blocks shared by the learner and acting kernels (gfx950 only).
//
// Tiling model: a workgroup of 256 threads (4 waves, one per SIMD) owns 16 consecutive
// minibatch rows and walks a whole network for them.  One dense layer for a 16-row tile is
//   Z[16, N] = X[16, K+1] * [W ; b]                 (bias folded in as the row of a constant-1 column)
// done with v_mfma_f32_16x16x4_f32 (exact f32 FMA chains).  The 16-wide column tiles of a layer are
// dealt to the four waves; X lives in LDS (it is shared by the waves); the weights are streamed
// straight from L2 into registers as ready-made B fragments (each wave reads a private share of them
// exactly once, so staging them in LDS would only add a round trip).  K is walked in groups of 16:
// one 16-byte load per lane brings the B operands of four consecutive MFMAs of a tile, one
// ds_read_b128 their A operands; GD groups are kept in flight and the first GD groups of the NEXT
// layer are requested before the current layer's epilogue, so the L2/Infinity-Cache latency of the
// weight stream hides behind arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rlgan {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_LINEAR = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_LRELU = 3, ACT_TANH = 4 };
// 0-3: the networks of the arm variant.  4, 5: actor and critic of the pixel variant on the 256 features of the state
// encoder (Conv_model.py:621-651,711-744): the same 200-200-10 body, a first layer of 17 k-groups instead of one.
enum { NET_ACTOR = 0, NET_CRITIC = 1, NET_GEN = 2, NET_DISC = 3, NET_PACTOR = 4, NET_PCRITIC = 5 };
constexpr int PIX_FEATURES = 256;   // Conv_model.py:24: the encoders' 256-unit layer
constexpr int PIX_N3 = 12;          // the 10-unit layer kept 12 wide (16-byte rows of the layer-wise parameter blocks)

constexpr int HID = 200;        // DDPG.py:82,159 ; Curious_GAN.py:59,121
constexpr int NT_H = 13;        // 16-wide column tiles of a hidden layer
constexpr int NP_H = 208;       // padded hidden width (global work arrays)
// LDS row strides: multiples of 4 floats (the A operand is read 16 bytes per lane) chosen so that the
// 16 rows a ds_read_b128 lane group touches fall on 16 disjoint 4-bank slots of the 64 banks
constexpr int LD_W = 212;       // wide activation tile (208 columns)
constexpr int LD_X0 = 36;       // network-input tile (32 columns)
constexpr int XW_COLS = 272;    // input columns of the pixel networks: 256 features, up to 15 action columns, the constant 1
constexpr int LD_XW = 276;      // their LDS row stride
constexpr int LD_O = 20;        // 16-column tile
constexpr int LDP = 20;         // row stride of a 16x16 partial-sum tile
constexpr int LD_3W = 68;       // 64-column tile
constexpr int X0_COLS = 32;     // global leading dimension of x0 work arrays
constexpr int ROWS = 16;        // minibatch rows per workgroup
constexpr int THREADS = 256;
constexpr float LRELU_ALPHA = 0.2f;  // Curious_GAN.py:56

// ---------------------------------------------------------------- geometry (host + device)
struct Geo {
  int S, A, K0, N3, NOUT, NT3;
  int K[4], N[4];
  int ngf[4], ntf[4], offf[4];   // forward fragments: k-groups (16 k each), column tiles, float offset in wf
  int ngb[4], ntb[4], offb[4];   // backward fragments (W^T)
  int kp[4], np[4], goff[4];     // padded gradient image [kp][np] per layer
  int n_params, wf_floats, wb_floats, gpad_floats, ld3;
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__host__ __device__ constexpr int first_groups(int kind) { return kind >= NET_PACTOR ? XW_COLS / 16 : kind == NET_DISC ? 2 : 1; }
__host__ __device__ constexpr int x0_cols(int kind) { return kind >= NET_PACTOR ? XW_COLS : 32; }    // global x0 work array
__host__ __device__ constexpr int x0_ld(int kind) { return kind >= NET_PACTOR ? LD_XW : LD_X0; }      // LDS input tile

__host__ __device__ inline bool make_geo(int kind, int S, int A, Geo& g) {
  if (S < 1 || A < 1) return false;
  g.S = S; g.A = A;
  switch (kind) {
    case NET_ACTOR:  g.K0 = S;         g.N3 = 10; g.NOUT = A; break;
    case NET_CRITIC: g.K0 = S + A;     g.N3 = 10; g.NOUT = 1; break;
    case NET_GEN:    g.K0 = S + A;     g.N3 = 50; g.NOUT = S; break;
    case NET_DISC:   g.K0 = 2 * S + A; g.N3 = 10; g.NOUT = 1; break;
    // pixel variant: S = 256 features; A = the action columns, a multiple of 4 (the layer-wise parameter blocks keep
    // 16-byte rows: their padding columns hold zero weights); the critic's q is column 0 of a 4-wide output
    case NET_PACTOR:  g.K0 = S;     g.N3 = PIX_N3; g.NOUT = A; break;
    case NET_PCRITIC: g.K0 = S + A; g.N3 = PIX_N3; g.NOUT = 4; break;
    default: return false;
  }
  if (kind >= NET_PACTOR && (S != PIX_FEATURES || A < 4 || A > 12 || A % 4 != 0)) return false;
  // first layer: inputs plus the constant-1 column in one k-group of 16, two for the discriminator
  // (its 2S + A = 16 inputs fill a group by themselves)
  const int ng1 = first_groups(kind);
  if (g.K0 + 1 > 16 * ng1 || (kind < NET_PACTOR && g.K0 > 16) || g.NOUT > 15) return false;
  g.NT3 = cdiv(g.N3, 16);
  g.K[0] = g.K0; g.N[0] = HID; g.K[1] = HID; g.N[1] = HID; g.K[2] = HID; g.N[2] = g.N3; g.K[3] = g.N3; g.N[3] = g.NOUT;
  int of = 0, ob = 0, og = 0, np_ = 0;
  for (int l = 0; l < 4; ++l) {
    g.ngf[l] = l == 0 ? ng1 : cdiv(g.K[l] + 1, 16); g.ntf[l] = cdiv(g.N[l], 16); g.offf[l] = of; of += g.ngf[l] * g.ntf[l] * 256;
    g.ngb[l] = cdiv(g.N[l], 16);                  g.ntb[l] = cdiv(g.K[l], 16); g.offb[l] = ob; ob += g.ngb[l] * g.ntb[l] * 256;
    g.kp[l] = 16 * cdiv(g.K[l] + 1, 16); g.np[l] = 16 * g.ntf[l]; g.goff[l] = og; og += g.kp[l] * g.np[l];
    np_ += (g.K[l] + 1) * g.N[l];
  }
  g.n_params = np_; g.wf_floats = of; g.wb_floats = ob; g.gpad_floats = og; g.ld3 = g.kp[3];
  return true;
}

// Fragment image of an operand matrix B[k][n] (k-groups of 16, column tiles of 16): the four MFMAs
// m = 0..3 of a group take k = 16*group + 4*(lane/16) + m, so lane `lane` of tile `t` holds its four
// B values of a group in one float4 at ((group*NT + t)*64 + lane)*4.
__host__ __device__ inline int frag_offset(int nt, int k, int n) {
  const int kg = k >> 4, q = (k & 15) >> 2, m = k & 3, t = n >> 4, lane = (q << 4) | (n & 15);
  return ((kg * nt + t) * 64 + lane) * 4 + m;
}
// float offset of element (k, n) of layer l in the forward-fragment image (k == K[l] is the bias row)
__host__ __device__ inline int wf_offset(const Geo& g, int l, int k, int n) { return g.offf[l] + frag_offset(g.ntf[l], k, n); }
// float offset of W[k][n] of layer l in the backward-fragment image (operand element (row n, col k))
__host__ __device__ inline int wb_offset(const Geo& g, int l, int k, int n) { return g.offb[l] + frag_offset(g.ntb[l], n, k); }

// Work arrays (activations and their gradients kept for the weight-gradient kernel) hold the batch in
// tiles of 16 rows; inside a tile of `ld` columns the order is [4 blocks of 4 rows][ld columns][4 rows].
// An MFMA lane (q = lane / 16, j = lane % 16) needs rows 4q..4q+3 of column 16*t + j: one 16-byte
// access, and the sixteen lanes of a q touch 256 consecutive bytes.  (With the columns outermost,
// [ld columns][16 rows], the same lanes touch sixteen different 64-byte pieces and the load unit
// needs almost twice as long per instruction.)
__host__ __device__ inline int work_off(int ld, int r, int c) { return (r >> 2) * (ld * 4) + c * 4 + (r & 3); }

#ifdef __HIPCC__
// ---------------------------------------------------------------- activations
// tf.nn.elu: z for z > 0, exp(z) - 1 otherwise -- computed the way TensorFlow's functor does, as an exponential and a
// subtraction (Eigen: `(features < 0).select(features.exp() - 1, features)`), with the hardware's exp2: absolute error
// below 1e-7 everywhere (exp2 is good to an ulp of a result <= 1; the rounding of z * log2(e) adds at most 2e-8), where
// expm1f, which keeps the RELATIVE error small for |z| << 1 as well, costs four times the instructions.  On the first
// layer of the universe encoder read from the uint8 ring the activation was a fifth of the layer's time.
__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.f; }

template <int ACT> __device__ __forceinline__ float act_f(float z) {
  if (ACT == ACT_RELU)  return fmaxf(z, 0.f);
  if (ACT == ACT_RELU6) return __builtin_amdgcn_fmed3f(z, 0.f, 6.f);   // one instruction; fminf(fmaxf()) adds a canonicalising max
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
template <> struct NetT<NET_ACTOR>  { static constexpr int A1 = ACT_RELU6, A2 = ACT_RELU6, A3 = ACT_RELU,  A4 = ACT_TANH,   NT3 = 1, L3 = 20; };
template <> struct NetT<NET_CRITIC> { static constexpr int A1 = ACT_RELU6, A2 = ACT_RELU6, A3 = ACT_RELU,  A4 = ACT_LINEAR, NT3 = 1, L3 = 20; };
template <> struct NetT<NET_GEN>    { static constexpr int A1 = ACT_RELU,  A2 = ACT_RELU,  A3 = ACT_RELU,  A4 = ACT_LINEAR, NT3 = 4, L3 = 68; };
template <> struct NetT<NET_DISC>   { static constexpr int A1 = ACT_LRELU, A2 = ACT_LRELU, A3 = ACT_LRELU, A4 = ACT_LINEAR, NT3 = 1, L3 = 20; };
template <> struct NetT<NET_PACTOR>  { static constexpr int A1 = ACT_RELU6, A2 = ACT_RELU6, A3 = ACT_RELU6, A4 = ACT_TANH,   NT3 = 1, L3 = 20; };
template <> struct NetT<NET_PCRITIC> { static constexpr int A1 = ACT_RELU6, A2 = ACT_RELU6, A3 = ACT_RELU6, A4 = ACT_LINEAR, NT3 = 1, L3 = 20; };

// ---------------------------------------------------------------- LDS carve (floats)

// Partial-sum tiles of a workgroup of NW waves: one slot per wave for a chained layer plus one for the shared tile's
// k-group, then the four sharers' partial sums of a shared tile.  Eight-wave workgroups pack the 16x16 slots (row
// stride 16) so that two of them fit a compute unit's LDS beside two sets of activation tiles.
template <int NW> struct PartL {
  static constexpr int LD = NW == 4 ? LDP : 16;        // row stride of a slot
  static constexpr int WS = ROWS * LD;                 // floats per slot
  static constexpr int CHAIN_SLOTS = NW + 1;
  static constexpr int SPART_OFF = CHAIN_SLOTS * WS;
  static constexpr int FLOATS = (CHAIN_SLOTS + 4) * WS;
};
constexpr int CHAIN_SLOTS = PartL<4>::CHAIN_SLOTS;   // the four-wave workgroups' figures under their old names
constexpr int SPART_OFF = PartL<4>::SPART_OFF;
constexpr int PART_FLOATS = PartL<4>::FLOATS;
constexpr int WIDE_FLOATS = ROWS * LD_W;
constexpr int T3_FLOATS = ROWS * LD_3W;              // h3 tile (16 or 64 columns)
constexpr int X0_FLOATS = ROWS * LD_X0;
constexpr int XW_FLOATS = ROWS * LD_XW;
constexpr int O_FLOATS = ROWS * LD_O;
#ifndef RLGAN_GD
#define RLGAN_GD 2
#endif

constexpr int GD = RLGAN_GD;                         // k-groups of operands kept in flight per wave

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, not for its
// outstanding global loads and stores (the activations written for the weight-gradient kernel are
// not read again inside the launch, and weight loads requested ahead of time must stay in flight).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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

// Weight fragments are fetched with buffer loads: the address is a scalar base (resource), a per-lane
// byte offset that never changes, and a scalar offset per (k-group, tile) -- no vector ALU work per
// load, which matters because vector ALU instructions do not overlap with MFMAs on the same SIMD.
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct FragSrc {
  __amdgpu_buffer_rsrc_t rs;
  int voff;                                            // this lane's byte offset inside a fragment block
  __device__ __forceinline__ void set(const float* base, int first_block) {
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    voff = (first_block * 64 + (int)(threadIdx.x & 63)) * 16;
  }
  // the 16-byte fragment of this lane in block `block` (blocks of 64 lanes x 16 bytes) past first_block
  __device__ __forceinline__ f32x4 load(int block) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, block * 1024, 0));
  }
  // general form: lane byte offset `lane_bytes` fixed at set-up, uniform byte offset per load
  __device__ __forceinline__ void set_bytes(const float* base, int lane_bytes) {
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    voff = lane_bytes;
  }
  __device__ __forceinline__ f32x4 load_at(int uniform_bytes) const {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, uniform_bytes, 0));
  }
};

// ---------------------------------------------------------------- weight fragments requested ahead
// Column tiles of a layer are dealt to the four waves round-robin: the wave with rotated index w owns
// tiles w, w+4, w+8, ...  (NT%4 waves own one tile more than the others; with fewer than four tiles
// some waves own none and only wait at the layer's barrier).  The rotation differs per workgroup so
// that the busier waves of workgroups sharing a compute unit land on different SIMDs.
// A wave walks its tiles ONE AFTER THE OTHER, each over the whole K range, so that the epilogue of a tile
// (activation, LDS and global stores: vector-ALU and memory instructions) is issued between the MFMAs of
// the next tile and costs no time of its own; only the last tile's epilogue is exposed.  The wave's work
// is therefore a stream of steps, step s = k-group s % NG of its tile s / NG, and the B operands of the
// steps are fetched through a ring of RING fragments, the refill for step s + DS requested at step s.
// Pre holds the first DS fragments of the stream: issue() only starts the loads; nothing waits until
// dense_tiles consumes them.
// A wide layer with 4n+1 column tiles (13) would leave one wave a third more MFMAs than the others, and
// the layer ends when that wave does.  There the last tile is SHARED instead: the other tiles are dealt
// whole (three each), and of the last tile every wave computes the k-groups w, w+4, w+8, (w+12) as further
// steps of its stream; the four partial sums meet in LDS (`spart`, slot = rotated wave index, added in slot
// order) and all threads of the workgroup apply the epilogue to that tile, one element each, behind one more barrier.
// (depths 4 / 5 / 6 / 7 / 8 / 12 measured on the headline step, same box: 0.1726 / 0.1708 / 0.1693 / 0.1701 / 0.1704 / 0.1732 ms)
#ifndef RLGAN_RING
#define RLGAN_RING 6
#endif
constexpr int RING = RLGAN_RING;

template <int NT, int NG, int NW = 4>
struct Pre {
  static_assert(NW == 4 || NW == 8, "workgroups of four or eight waves");
  static constexpr bool SPLIT = (NT % 4 == 1) && NT > 4 && NG >= 8;
  static constexpr int NTF = SPLIT ? NT - 1 : NT;      // tiles dealt whole, round-robin: wave w owns w, w + NW, ...
  static constexpr int TMAX = (NTF + NW - 1) / NW, TMIN = NTF / NW, NHEAVY = NTF % NW;
  // the shared tile's k-groups go to the waves with TMIN whole tiles (all of them when the whole tiles deal evenly):
  // sharer sw = w - NHEAVY takes groups sw, sw + NSH, ...
  static constexpr int NSH = SPLIT ? NW - NHEAVY : 1;
  static_assert(!SPLIT || NSH == 4, "four sharers: four partial-sum slots, added in slot order");
  static constexpr int NS = SPLIT ? (NG + NSH - 1) / NSH : 0;   // k-groups of the shared tile per sharer (at most)
  // steps of a wave with T whole tiles, and the fragments it keeps in flight
  static constexpr int steps(int T, bool share) { return T * NG + (share ? NS : 0); }
  // (four-wave workgroups keep one depth for all their waves, that of the waves with the fewest steps)
  static constexpr int depth(int T, bool share) {
    const int n = NW == 4 ? steps(TMIN > 0 ? TMIN : 1, SPLIT) : steps(T, share);
    return n < RING ? n : RING;
  }
  static constexpr bool HSHARE = SPLIT && NHEAVY == 0;          // do the waves with TMAX tiles share as well?
  static constexpr int DSH = depth(TMAX, HSHARE), DSL = depth(TMIN, SPLIT);
  static constexpr int DSMAX = DSH > DSL ? (DSH > 1 ? DSH : 1) : (DSL > 1 ? DSL : 1);
  f32x4 b[DSMAX];
  FragSrc src;                                         // fragment blocks counted from this wave's tile 0, group 0
  int w;                                               // rotated wave index

  // fragment block of step s of a wave that owns T whole tiles (steps at or beyond T*NG: the shared tile's k-groups)
  template <int T>
  __device__ __forceinline__ int block(int s) const {
    if (s < T * NG) return (s % NG) * NT + NW * (s / NG);
    return min(w - NHEAVY + NSH * (s - T * NG), NG - 1) * NT + (NT - 1) - w;
  }
  template <int T, bool SHARE>
  __device__ __forceinline__ void load_first() {
#pragma unroll
    for (int s = 0; s < depth(T, SHARE); ++s) b[s] = src.load(block<T>(s));
  }
  __device__ __forceinline__ void issue(const float* __restrict__ wp) {
    w = (wave_id() + (int)blockIdx.x) & (NW - 1);
    src.set(wp, w);
    if constexpr (NHEAVY == 0) load_first<TMAX, SPLIT>();
    else if (w < NHEAVY) load_first<TMAX, false>();
    else if constexpr (TMIN > 0) load_first<TMIN, SPLIT>();
  }
};

// the first layer's fragments of a network of kind KIND
template <int KIND, int NW = 4> using FwdPre = Pre<NT_H, first_groups(KIND), NW>;

struct NoPre { __device__ __forceinline__ float operator()(int, int) const { return 0.f; } };
struct NoNext { __device__ __forceinline__ void operator()() const {} };

// The share of one wave in a dense layer: T column tiles, each over the whole K range, one after the other.
//   AREG: the layer has one k-group whose A operand (all 16 rows, the 16 k) the wave already holds in `areg`;
//         otherwise the A operands come from LDS, one float4 at xq + 16*group.
// Two accumulators per tile take the MFMAs in turn (a dependent MFMA would wait 40 cycles instead of 32).
template <int NT, int NG, int NW, int T, bool SHARE, bool AREG, class P, class F, class X>
__device__ __forceinline__ void dense_tiles(Pre<NT, NG, NW>& p, const float* xq, const f32x4 areg, int r0, int c0, P pre, F f,
                                            float* gdst, X next, float* spart) {
  using Q = Pre<NT, NG, NW>;
  static_assert(!AREG || NG == 1, "a register operand covers one k-group");
  static_assert(!SHARE || Q::SPLIT, "only a split layer has a shared tile");
  constexpr int NSW = SHARE ? Q::NS : 0, NSH = Q::NSH;          // this wave's steps on the shared tile; the sharers
  constexpr int DS = Q::depth(T, SHARE), SW = T * NG, S = SW + NSW;
  const int sw = p.w - Q::NHEAVY;                                // sharer index (meaningful when SHARE)
  // epilogue of a finished tile, cut into pieces that are issued among the MFMAs of the steps that follow
  constexpr int PIECES = (NG >= 7 || NSW >= 4) ? 6 : 1;
  f32x4 ap0{0.f, 0.f, 0.f, 0.f}, ap1{0.f, 0.f, 0.f, 0.f}, zp{0.f, 0.f, 0.f, 0.f}, vp{0.f, 0.f, 0.f, 0.f};
  float hvp[4] = {0.f, 0.f, 0.f, 0.f};
  int colp = 0;
  auto piece = [&](int k) {
    // k = 0 .. PIECES-1 of the previous tile's epilogue; piece 0 adds the two accumulators (by then the tile's last
    // MFMA has long left the pipe)
    if (k == 0) zp = ap0 + ap1;
    if constexpr (PIECES == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vp[r] = f(r0 + r, colp, zp[r], hvp[r]);
      if (gdst) *(f32x4*)(gdst + work_off(NT * 16, r0, colp)) = vp;
    } else {
      if (k >= 1 && k <= 4) vp[k - 1] = f(r0 + k - 1, colp, zp[k - 1], hvp[k - 1]);
      if (k == 5 && gdst) *(f32x4*)(gdst + work_off(NT * 16, r0, colp)) = vp;
    }
  };
  // A operands are read two steps ahead of their use (an LDS read takes about as long as four MFMAs)
  auto a_at = [&](int s) {
    return *(const f32x4*)(xq + 16 * (s < SW ? s % NG : min(sw + NSH * (s - SW), NG - 1)));
  };
#ifndef RLGAN_AD
#define RLGAN_AD 2
#endif
  constexpr int AD = RLGAN_AD;                            // steps an A operand is read ahead of its use (1, 2, 3, 4 measured: the step time does not move)
  f32x4 ar[AD + 1];
  if constexpr (!AREG) {
#pragma unroll
    for (int i = 0; i < AD; ++i) if (i < S) ar[i] = a_at(i);
  }
  f32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};
  f32x4 s0{0.f, 0.f, 0.f, 0.f}, s1{0.f, 0.f, 0.f, 0.f};      // the shared tile's partial sums
  float hv[4] = {0.f, 0.f, 0.f, 0.f};
  STAMP(2);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const bool whole = s < SW;
    const int i = whole ? s / NG : T, g = whole ? s % NG : s - SW;
    if (whole && g == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = pre(r0 + r, 16 * (p.w + NW * i) + c0);
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (!AREG) {
      if (s + AD < S) ar[(s + AD) % (AD + 1)] = a_at(s + AD);
    }
    const f32x4 av = AREG ? areg : ar[s % (AD + 1)];
    const f32x4 bv = p.b[s % DS];
    if (whole) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
    } else if (NSH * g + NSH - 1 < NG || sw + NSH * g < NG) {   // only the last k-group of the shared tile depends on the wave
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], s1, 0, 0, 0);
    }
    if (s + DS < S) p.b[s % DS] = p.src.load(p.template block<T>(s + DS));
    // epilogue pieces of the tile finished before this one
    if (i > 0) {
      if constexpr (PIECES == 1) {
        if (g == 0) piece(0);
      } else if (whole) {
        // from the second step on: the first piece reads the accumulators, and the tile's last MFMA has left the
        // pipe only a step after it was issued
        // All six pieces in ONE step: a lone vector instruction between two MFMAs costs the pipe about 6 cycles, one of
        // a run about 2.5 (tools/valu_mix.hip), so the epilogue is issued as a block inside the next tile's second step
        // instead of a piece per step (headline step 0.1711 -> 0.1703 ms on one box, A B A B).
        if (g == 1) { piece(0); piece(1); piece(2); piece(3); piece(4); piece(5); }
      } else {
        // the shared tile's second step takes them all
        if (g == 1) { piece(0); piece(1); piece(2); piece(3); piece(4); piece(5); }
      }
    }
    if (whole && g == NG - 1) {
      // this tile is complete: hand it to the steps that follow
      ap0 = acc0; ap1 = acc1; colp = 16 * (p.w + NW * i) + c0;
#pragma unroll
      for (int r = 0; r < 4; ++r) hvp[r] = hv[r];
    }
    // keep the refill DS steps ahead of its use and the epilogue pieces inside their step: the scheduler would
    // otherwise sink the loads next to their consumers to save registers, and the wave would wait out a full L2
    // round trip per step
    __builtin_amdgcn_sched_barrier(0);
  }
  STAMP(3);
  next();
  __builtin_amdgcn_sched_barrier(0);   // the next layer's requests stay ahead of what is left of this layer's epilogue
  if constexpr (SHARE) {
    static_assert(Q::NS >= 4 && T >= 1, "the shared tile's steps carry the last whole tile's epilogue");
    using PL = PartL<NW>;
    const f32x4 sz = s0 + s1;
    float* pw = spart + sw * PL::WS + r0 * PL::LD + c0;
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r * PL::LD] = sz[r];
  } else {
    // the last tile's epilogue has nothing left to hide behind
#pragma unroll
    for (int k = 0; k < PIECES; ++k) piece(k);
  }
}

template <int NT, int NW, class P, class F, class X>
__device__ __forceinline__ void dense16_rega(Pre<NT, 1, NW>& p, const f32x4 a, P pre, F f, float* gdst, X next) {
  using Q = Pre<NT, 1, NW>;
  const int lane = threadIdx.x & 63;
  const int r0 = (lane >> 4) * 4, c0 = lane & 15;
  if constexpr (Q::NHEAVY == 0) dense_tiles<NT, 1, NW, Q::TMAX, false, true>(p, nullptr, a, r0, c0, pre, f, gdst, next, nullptr);
  else if (p.w < Q::NHEAVY) dense_tiles<NT, 1, NW, Q::TMAX, false, true>(p, nullptr, a, r0, c0, pre, f, gdst, next, nullptr);
  else if constexpr (Q::TMIN > 0) dense_tiles<NT, 1, NW, Q::TMIN, false, true>(p, nullptr, a, r0, c0, pre, f, gdst, next, nullptr);
  else next();
  lds_barrier();
}

// ---------------------------------------------------------------- chaining into a one-tile layer
// A layer that follows a wide one and has a single column tile (N <= 16) needs all of the wide
// layer's output as its K range.  Instead of a stage of its own it is split over K along the tiles of
// the wide layer: right after its epilogue a wave multiplies the output tiles it has just written to
// LDS (reading them back as A operands needs no workgroup barrier: a wave's LDS operations execute in
// order) with the matching k-groups of the narrow layer's weights, and leaves a 16x16 partial sum in
// `part`; the workgroup barrier that ends the wide layer then covers both, and whoever needs the
// narrow layer's output adds the four partial sums in wave order.
struct NoChain {
  static constexpr bool active = false;
  template <int T> __device__ __forceinline__ void run(int, const float*, int) const {}
  __device__ __forceinline__ void run_none(int) const {}
};
// LAST: the wide layer's last tile is shared between the waves (Pre::SPLIT), so nobody owns it; the
// narrow layer's last k-group is then added separately once that tile is complete (`wlast` holds its
// fragments): run_last() into a fifth slot, or operand-swapped in registers where the consumer wants that.
template <int TMAX, bool LAST = false, int NW = 4>
struct Chain {
  static constexpr bool active = true;
  using PL = PartL<NW>;
  f32x4 wc[TMAX];                                      // narrow layer's fragments of the k-groups w, w+NW, ...
  f32x4 wlast;                                         // ... and of its last k-group (LAST only)
  float* part;                                         // [NW + 1][16][PL::LD]
  // wn: fragment image of the narrow layer (one column tile); ng: its number of k-groups
  //     nt, tile: the narrow "layer" may also be ONE column tile of a layer with nt tiles per k-group (the action columns
  //     of the pixel critic's input gradient)
  __device__ __forceinline__ void issue(const float* __restrict__ wn, int ng, int w, int nt = 1, int tile = 0) {
    FragSrc src;
    src.set(wn, tile);
#pragma unroll
    for (int i = 0; i < TMAX; ++i) wc[i] = src.load(min(w + NW * i, ng - 1) * nt);
    if constexpr (LAST) wlast = src.load((ng - 1) * nt);
  }
  // the last k-group's product from the completed tile at column 16*(ng-1) of xs, into the slot after the waves' (one wave calls this)
  __device__ __forceinline__ void run_last(const float* xs, int ldx, int ng) {
    const int lane = threadIdx.x & 63;
    const int r0 = (lane >> 4) * 4, c0 = lane & 15;
    const f32x4 a = *(const f32x4*)(xs + c0 * ldx + r0 + 16 * (ng - 1));
    f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], wlast[m], acc, 0, 0, 0);
    float* pw = part + NW * PL::WS + r0 * PL::LD + c0;
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r * PL::LD] = acc[r];
  }
  // a wave without a tile of the wide layer leaves zeros in its slot
  __device__ __forceinline__ void run_none(int w) {
    const int lane = threadIdx.x & 63;
    const int r0 = (lane >> 4) * 4, c0 = lane & 15;
    float* pw = part + w * PL::WS + r0 * PL::LD + c0;
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r * PL::LD] = 0.f;
  }
  // xs: the wide layer's output tile in LDS (row stride ldx); this wave owns its column tiles w, w+NW, ...
  // Order of additions: a four-wave workgroup's wave w owns tiles w, w+4, w+8 and leaves (t_w then t_{w+8} on one
  // accumulator) + (t_{w+4} on another).  In an eight-wave workgroup wave w < 4 owns exactly w and w+8 and wave w+4 owns
  // w+4: each chains its tiles on ONE accumulator and chain_sum adds the pair first, so both forms give the same bits.
  template <int T>
  __device__ __forceinline__ void run(int w, const float* xs, int ldx) {
    const int lane = threadIdx.x & 63;
    const int r0 = (lane >> 4) * 4, c0 = lane & 15;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 a[T];
#pragma unroll
    for (int i = 0; i < T; ++i) a[i] = *(const f32x4*)(xs + c0 * ldx + r0 + 16 * (w + NW * i));
#pragma unroll
    for (int i = 0; i < T; ++i) {
      constexpr int two = NW == 4 ? 1 : 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[i & two] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][m], wc[i][m], acc[i & two], 0, 0, 0);
    }
    const f32x4 z = NW == 4 ? acc[0] + acc[1] : acc[0];
    float* pw = part + w * PL::WS + r0 * PL::LD + c0;      // slot = tile owner, whatever wave that is in this workgroup
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r * PL::LD] = z[r];
  }
};
// sum of the waves' partial tiles at (row, col .. col+3): four waves in wave order; eight waves as the pairs (w, w+4)
// first, then those four in order (see Chain::run)
template <int NW = 4>
__device__ __forceinline__ f32x4 chain_sum4(const float* part, int row, int col) {
  using PL = PartL<NW>;
  const float* q = part + row * PL::LD + col;
  constexpr int WS = PL::WS;
  if constexpr (NW == 4) {
    return ((*(const f32x4*)q + *(const f32x4*)(q + WS)) + *(const f32x4*)(q + 2 * WS)) + *(const f32x4*)(q + 3 * WS);
  } else {
    const f32x4 z0 = *(const f32x4*)q + *(const f32x4*)(q + 4 * WS), z1 = *(const f32x4*)(q + WS) + *(const f32x4*)(q + 5 * WS);
    const f32x4 z2 = *(const f32x4*)(q + 2 * WS) + *(const f32x4*)(q + 6 * WS), z3 = *(const f32x4*)(q + 3 * WS) + *(const f32x4*)(q + 7 * WS);
    return ((z0 + z1) + z2) + z3;
  }
}
template <int NW = 4>
__device__ __forceinline__ float chain_sum1(const float* part, int row, int col) {
  using PL = PartL<NW>;
  const float* q = part + row * PL::LD + col;
  constexpr int WS = PL::WS;
  if constexpr (NW == 4) return ((q[0] + q[WS]) + q[2 * WS]) + q[3 * WS];
  else return (((q[0] + q[4 * WS]) + (q[WS] + q[5 * WS])) + (q[2 * WS] + q[6 * WS])) + (q[3 * WS] + q[7 * WS]);
}
template <int NW = 4>
__device__ __forceinline__ float chain_sum1_last(const float* part, int row, int col) {   // with run_last()'s slot
  using PL = PartL<NW>;
  return chain_sum1<NW>(part, row, col) + part[NW * PL::WS + row * PL::LD + col];
}

// One dense layer of a 16-row tile, Z[16, NT*16] = X[16, NG*16] * fragments.
//   p     the layer's weight fragments, already requested with p.issue()
//   xs    the input tile in LDS (row stride ldx, at least NG*16 valid columns)
//   pre   pre(row, col): a value the epilogue needs per output element (read before the MFMAs so its
//         LDS latency hides behind them); NoPre when none is needed
//   f     f(row, col, z, pre_value) -> value: the epilogue, once per output element (it writes LDS)
//   gdst  when non-null, the returned values are also stored to this tile of a global work array of
//         NT*16 columns (see work_off; one 16-byte store per lane and column)
//   next  called by every wave between its last MFMA and its epilogue: the place to issue() the next
//         layer's Pre
//   chain / ys, ldy: optional chained one-tile layer (see Chain); ys is the LDS tile f writes
// Ends with a workgroup barrier on LDS (what the epilogues wrote is visible to every wave after it).
template <int NT, int NG, int NW, class P, class F, class X, class CH>
__device__ __forceinline__ void dense16(Pre<NT, NG, NW>& p, const float* xs, int ldx, P pre, F f, float* gdst, X next,
                                        CH& chain, const float* ys, int ldy, float* spart = nullptr) {
  using Q = Pre<NT, NG, NW>;
  const int lane = threadIdx.x & 63;
  const int r0 = (lane >> 4) * 4, c0 = lane & 15;
  const float* xq = xs + c0 * ldx + r0;                 // A operands of a group: one float4 at xq + 16*group
  const f32x4 none{0.f, 0.f, 0.f, 0.f};
  STAMP(1);
  if constexpr (Q::NHEAVY == 0) {
    dense_tiles<NT, NG, NW, Q::TMAX, Q::SPLIT, false>(p, xq, none, r0, c0, pre, f, gdst, next, spart);
    if constexpr (CH::active) chain.template run<Q::TMAX>(p.w, ys, ldy);
  } else {
    if (p.w < Q::NHEAVY) {
      dense_tiles<NT, NG, NW, Q::TMAX, false, false>(p, xq, none, r0, c0, pre, f, gdst, next, spart);
      if constexpr (CH::active) chain.template run<Q::TMAX>(p.w, ys, ldy);
    } else if constexpr (Q::TMIN > 0) {
      dense_tiles<NT, NG, NW, Q::TMIN, Q::SPLIT, false>(p, xq, none, r0, c0, pre, f, gdst, next, spart);
      if constexpr (CH::active) chain.template run<Q::TMIN>(p.w, ys, ldy);
    } else {
      static_assert(!Q::SPLIT, "a shared tile needs sharers with a whole tile each");
      next();
      if constexpr (CH::active) chain.run_none(p.w);
    }
  }
  STAMP(4);
  lds_barrier();
  if constexpr (Q::SPLIT) {
    // the shared tile: four partial sums in slot order, epilogue, one element per thread of the first four waves
    using PL = PartL<NW>;
    if (NW == 4 || threadIdx.x < 256) {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15, col = 16 * (NT - 1) + c;
      const float* q = spart + r * PL::LD + c;
      constexpr int WS = PL::WS;
      const float z = ((q[0] + q[WS]) + q[2 * WS]) + q[3 * WS];
      const float v = f(r, col, z, pre(r, col));
      if (gdst) gdst[work_off(NT * 16, r, col)] = v;
    }
    lds_barrier();
  }
  STAMP(5);
}
template <int NT, int NG, int NW, class P, class F, class X>
__device__ __forceinline__ void dense16(Pre<NT, NG, NW>& p, const float* xs, int ldx, P pre, F f, float* gdst, X next,
                                        float* spart = nullptr) {
  NoChain nc;
  dense16(p, xs, ldx, pre, f, gdst, next, nc, nullptr, 0, spart);
}

// the fragments of a one-tile layer (NG k-groups), loaded by EVERY wave: for layers so small that each
// wave computes them for itself rather than wait for another wave's result
template <int NG>
struct PreAll {
  f32x4 b[NG];
  __device__ __forceinline__ void issue(const float* __restrict__ wp) {
    FragSrc src;
    src.set(wp, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) b[g] = src.load(g);
  }
};

// LDS tiles of one network for one 16-row tile
struct Tiles {
  float* x0;   // [16][LD_X0]  inputs, constant-1 column at K0, zeros after
  float* h1;   // [16][LD_W]
  float* h2;   // [16][LD_W]
  float* h3;   // [16][LD_3W or LD_O]
  float* out;  // [16][LD_O]
  float* part; // PART_FLOATS: partial sums of chained layers
};

// global work arrays of one network, already offset to this 16-row tile (or all null); layout inside
// a tile: work_off
struct ActsG {
  float* x0; float* h1; float* h2; float* h3; float* dz1; float* dz2; float* dz3; float* dz4;
};

// what the caller of net_backward requests ahead for its first stage: the fragments of W4^T
template <int KIND, int NW> struct BwdFirst { using type = PreAll<1>; };
template <int NW> struct BwdFirst<NET_GEN, NW> { using type = Pre<4, 1, NW>; };
template <int KIND, int NW = 4> using BwdPre = typename BwdFirst<KIND, NW>::type;

// forward pass of a whole network for the tile in t.x0; activations stay in t.h1..t.out and, when
// STORE, are also written to the global work arrays for the weight-gradient pass.  `p1` holds the
// first layer's fragments (issue()d by the caller, ideally well before); `next` is called late in the
// pass so the caller can request the fragments of whatever runs after this network.
//   networks with a 10-wide third layer:  L1 | L2 + L3 partial sums (chained) | one wave: L3 total, L4
//   generator (50-wide third layer):      L1 | L2 | L3 + L4 partial sums (chained) | L4 total
template <int KIND, bool STORE, int NW = 4, class X>
__device__ __forceinline__ void net_forward(const float* __restrict__ wf, const Geo& g, const Tiles& t, const ActsG& ws,
                                            FwdPre<KIND, NW>& p1, X next) {
  using N = NetT<KIND>;
  using PL = PartL<NW>;
  const bool tile_thread = NW == 4 || threadIdx.x < 256;      // one thread per element of a 16x16 tile
  constexpr int L3 = N::L3, NT3 = N::NT3;
  float* h1 = t.h1; float* h2 = t.h2; float* h3 = t.h3; float* out = t.out;
  const int lane = threadIdx.x & 63;
  const int n3 = g.N3, nout = g.NOUT;
  Pre<NT_H, NT_H, NW> p2;
  dense16(p1, t.x0, x0_ld(KIND), NoPre{}, [=](int r, int c, float z, float) {
    const float v = c < HID ? act_f<N::A1>(z) : (c == HID ? 1.f : 0.f);
    h1[r * LD_W + c] = v;
    return v;
  }, STORE ? ws.h1 : nullptr, [&] { p2.issue(wf + g.offf[1]); }, t.part + PL::SPART_OFF);   // the pixel networks' 17 k-groups share the last tile
  auto f2 = [=](int r, int c, float z, float) {
    const float v = c < HID ? act_f<N::A2>(z) : (c == HID ? 1.f : 0.f);
    h2[r * LD_W + c] = v;
    return v;
  };
  if constexpr (NT3 == 1) {
    using Q2 = Pre<NT_H, NT_H, NW>;
    Chain<Q2::TMAX, Q2::SPLIT, NW> ch;
    ch.part = t.part;
    PreAll<1> p4;
    dense16(p2, h1, LD_W, NoPre{}, f2, STORE ? ws.h2 : nullptr,
            [&] { ch.issue(wf + g.offf[2], NT_H, p2.w); p4.issue(wf + g.offf[3]); }, ch, h2, LD_W, t.part + PL::SPART_OFF);
    next();
    if (wave_id() == 0) {
      // third layer: total of the partial sums, activation; the lane holds row l&15, columns 4q..4q+3,
      // which is the A operand of the fourth layer
      const int row = lane & 15, k0 = (lane >> 4) * 4;
      f32x4 s = chain_sum4<NW>(t.part, row, k0);
      if constexpr (Q2::SPLIT) {
        // the last k-group (the shared tile of h2, complete since the layer's second barrier): operands
        // swapped, so the product arrives in this lane's (row, four columns) layout
        const f32x4 hb = *(const f32x4*)(h2 + row * LD_W + 16 * (NT_H - 1) + k0);
        f32x4 e{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) e = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.wlast[m], hb[m], e, 0, 0, 0);
        s += e;
      }
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = k0 + r < n3 ? act_f<N::A3>(s[r]) : (k0 + r == n3 ? 1.f : 0.f);
      *(f32x4*)(h3 + row * L3 + k0) = v;
      if (STORE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ws.h3[work_off(16, row, k0 + r)] = v[r];
      }
      f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[m], p4.b[0][m], acc, 0, 0, 0);
      const int r0 = (lane >> 4) * 4, c0 = lane & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(r0 + r) * LD_O + c0] = c0 < nout ? act_f<N::A4>(acc[r]) : 0.f;
    }
    lds_barrier();
  } else {
    Pre<NT3, NT_H, NW> p3;
    dense16(p2, h1, LD_W, NoPre{}, f2, STORE ? ws.h2 : nullptr, [&] { p3.issue(wf + g.offf[2]); }, t.part + PL::SPART_OFF);
    Chain<1, false, NW> ch;
    ch.part = t.part;
    dense16(p3, h2, LD_W, NoPre{}, [=](int r, int c, float z, float) {
      const float v = c < n3 ? act_f<N::A3>(z) : (c == n3 ? 1.f : 0.f);
      h3[r * L3 + c] = v;
      return v;
    }, STORE ? ws.h3 : nullptr, [&] { ch.issue(wf + g.offf[3], NT3, p3.w); }, ch, h3, L3);
    next();
    if (tile_thread) {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
      const float z = chain_sum1<NW>(t.part, r, c);
      out[r * LD_O + c] = c < nout ? act_f<N::A4>(z) : 0.f;
    }
    lds_barrier();
  }
}

// backward pass: t.out must hold dL/dz4 ([16][LD_O], zeros beyond NOUT), t.h1..h3 the forward
// activations.  dz1 and dz2 overwrite h1 and h2 in place and, when STORE, all dz are written to the
// global work arrays.  With NEED_DX0 the gradient w.r.t. the network inputs is left in t.x0 (columns
// 0..15, stride LD_X0).  `p4` = fragments of W4^T requested by the caller, `next` as in net_forward.
//   10-wide third layer: every wave computes dz3 for itself (four MFMAs with swapped operands leave it
//   in A-operand layout in registers), then its tiles of dz2 | dz1 (+ chained dX0 partial sums) | dX0 total
template <int KIND, bool NEED_DX0, bool STORE, int NW = 4, class X>
__device__ __forceinline__ void net_backward(const float* __restrict__ wb, const Geo& g, const Tiles& t, const ActsG& ws,
                                             BwdPre<KIND, NW>& p4, X next) {
  using N = NetT<KIND>;
  using PL = PartL<NW>;
  const bool tile_thread = NW == 4 || threadIdx.x < 256;      // one thread per element of a 16x16 tile
  constexpr int L3 = N::L3, NT3 = N::NT3;
  float* h1 = t.h1; float* h2 = t.h2; float* h3 = t.h3; float* x0 = t.x0;
  const int lane = threadIdx.x & 63;
  const int r0 = (lane >> 4) * 4, c0 = lane & 15;
  if (STORE && tile_thread) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    ws.dz4[work_off(16, r, c)] = t.out[r * LD_O + c];
  }
  const int n3 = g.N3;
  Pre<NT_H, NT3, NW> p3;
  p3.issue(wb + g.offb[2]);
  Pre<NT_H, NT_H, NW> p2;
  auto f2 = [=](int r, int c, float z, float h) {
    const float v = c < HID ? z * act_d<N::A2>(h) : 0.f;
    h2[r * LD_W + c] = v;
    return v;
  };
  auto g2 = [=](int r, int c) { return h2[r * LD_W + c]; };
  if constexpr (NT3 == 1) {
    // dH3^T = W4 * dZ4^T: the operands trade places, so the lane gets row l&15, columns 4q..4q+3 of dH3
    const f32x4 a4 = *(const f32x4*)(t.out + c0 * LD_O + r0);
    const f32x4 hv = *(const f32x4*)(h3 + c0 * L3 + r0);
    f32x4 d{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) d = __builtin_amdgcn_mfma_f32_16x16x4f32(p4.b[0][m], a4[m], d, 0, 0, 0);
    f32x4 dz3;
#pragma unroll
    for (int r = 0; r < 4; ++r) dz3[r] = r0 + r < n3 ? d[r] * act_d<N::A3>(hv[r]) : 0.f;
    if (STORE && wave_id() == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ws.dz3[work_off(16, c0, r0 + r)] = dz3[r];
    }
    // dz2 tiles of this wave: one k-group, its A operand is dz3 in registers
    dense16_rega(p3, dz3, g2, f2, STORE ? ws.dz2 : nullptr, [&] { p2.issue(wb + g.offb[1]); });
  } else {
    dense16(p4, t.out, LD_O, [=](int r, int c) { return h3[r * L3 + c]; }, [=](int r, int c, float z, float h) {
      const float v = c < n3 ? z * act_d<N::A3>(h) : 0.f;
      h3[r * L3 + c] = v;
      return v;
    }, STORE ? ws.dz3 : nullptr, NoNext{});
    dense16(p3, h3, L3, g2, f2, STORE ? ws.dz2 : nullptr, [&] { p2.issue(wb + g.offb[1]); });
  }
  auto f1 = [=](int r, int c, float z, float h) {
    const float v = c < HID ? z * act_d<N::A1>(h) : 0.f;
    h1[r * LD_W + c] = v;
    return v;
  };
  auto g1 = [=](int r, int c) { return h1[r * LD_W + c]; };
  if constexpr (NEED_DX0) {
    using Q2 = Pre<NT_H, NT_H, NW>;
    Chain<Q2::TMAX, Q2::SPLIT, NW> ch;
    ch.part = t.part;
    // the arm networks have one tile of inputs; of the pixel critic's 17 only the last, which holds the action, is wanted
    const int xt = KIND >= NET_PACTOR ? g.ntb[0] - 1 : 0;
    dense16(p2, h2, LD_W, g1, f1, STORE ? ws.dz1 : nullptr, [&] { ch.issue(wb + g.offb[0], NT_H, p2.w, g.ntb[0], xt); }, ch, h1, LD_W,
            t.part + PL::SPART_OFF);
    next();
    if constexpr (Q2::SPLIT) {
      if (wave_id() == 0) ch.run_last(h1, LD_W, NT_H);      // dz1's shared tile times the last k-group of W1^T
      lds_barrier();
    }
    if (tile_thread) {
      const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
      x0[r * x0_ld(KIND) + 16 * xt + c] = Q2::SPLIT ? chain_sum1_last<NW>(t.part, r, c) : chain_sum1<NW>(t.part, r, c);
    }
    lds_barrier();
  } else {
    dense16(p2, h2, LD_W, g1, f1, STORE ? ws.dz1 : nullptr, next, t.part + PL::SPART_OFF);
  }
}

// fill the input tile: up to three column blocks, then the constant-1 column, then zeros; rows at or
// beyond n_rows are zero.  Optionally mirrors the tile to the global x0 work array.
// local: an LDS tile indexed by tile row; idx (global blocks only, optional): row gr comes from row
// idx[gr] of the matrix, which is how a minibatch is read straight out of the replay ring
struct MatIn { const float* p; int ld; int w; int local; const long long* idx; };
__device__ __forceinline__ float matin_at(const MatIn& b, int gr, int r, int c) {
  if (b.local) return b.p[r * b.ld + c];
  const size_t row = b.idx ? (size_t)b.idx[gr] : (size_t)gr;
  return b.p[row * b.ld + c];
}
template <int NW = 4>
__device__ __forceinline__ void load_x0(float* x0, int row0, int n_rows, MatIn b0, MatIn b1, MatIn b2,
                                        float* x0_global) {
  const int k0 = b0.w + b1.w + b2.w;
  for (int e = threadIdx.x; e < ROWS * X0_COLS; e += NW * 64) {
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
    if (x0_global) x0_global[work_off(X0_COLS, r, c)] = v;
  }
}

// Input tile of the pixel networks, [16][LD_XW]: 256 features (rows of a dense matrix), then `aw` action columns (from
// a matrix in memory or from an LDS tile of stride lda), the constant 1, zeros up to XW_COLS; rows at or beyond
// n_rows are zero.  Optionally mirrored to the global x0 work array ([4 row blocks][XW_COLS][4 rows] per tile, after a
// workgroup barrier of its own).
//   a: aw columns are read (rows of a matrix in memory, or of an LDS tile when a_local), columns aw .. ap-1 are zero
//   (the parameter blocks' padding), column ap holds the constant 1
template <int NW = 4>
__device__ __forceinline__ void load_xw(float* x0, int row0, int n_rows, const float* __restrict__ phis, int ldp,
                                        const float* a, int lda, int a_local, int aw, int ap, float* x0_global) {
  constexpr int NTHR = NW * 64;
  for (int e = threadIdx.x; e < ROWS * (PIX_FEATURES / 4); e += NTHR) {
    const int r = e / (PIX_FEATURES / 4), c4 = e - r * (PIX_FEATURES / 4), gr = row0 + r;
    f32x4 v{0.f, 0.f, 0.f, 0.f};
    if (gr < n_rows) v = *(const f32x4*)(phis + (size_t)gr * ldp + 4 * c4);
    *(f32x4*)(x0 + r * LD_XW + 4 * c4) = v;
  }
  if (NW == 4 || threadIdx.x < 256) {                     // one thread per element of the 16x16 tail
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15, gr = row0 + r;
    float v = 0.f;
    if (c < aw && gr < n_rows) v = a_local ? a[r * lda + c] : a[(size_t)gr * lda + c];
    if (c == ap) v = 1.f;
    x0[r * LD_XW + PIX_FEATURES + c] = v;
  }
  if (x0_global) {
    lds_barrier();
    for (int e = threadIdx.x; e < 4 * XW_COLS; e += NTHR) {
      const int b = e / XW_COLS, c = e - b * XW_COLS;
      const float* xp = x0 + (4 * b) * LD_XW + c;
      *(f32x4*)(x0_global + work_off(XW_COLS, 4 * b, c)) = f32x4{xp[0], xp[LD_XW], xp[2 * LD_XW], xp[3 * LD_XW]};
    }
  }
}

__device__ __forceinline__ ActsG acts_at(const float* const* p8, int row0, int ld3, int x0c = X0_COLS) {
  ActsG a;
  a.x0 = (float*)p8[0] + (size_t)row0 * x0c;
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
