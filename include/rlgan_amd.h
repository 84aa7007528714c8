/*
 * rlgan_amd.h -- C ABI of librlgan_amd.so: the MI355X (gfx950) hot path of RL-GAN.
 *
 * The reference (hithmh/RL-GAN, directory GAN-RL/) is pure Python + TensorFlow 1.x and has no
 * FFI layer of its own; every entry point below replaces one group of `sess.run` / NumPy calls
 * and cites it.  All pointers marked "device" are HIP device pointers owned by the caller (the
 * Python layer allocates them as torch tensors); the library only enqueues work on `stream`
 * (a hipStream_t passed as void*) and never allocates, frees or synchronises.
 *
 * Return value of every function: 0 = ok, < 0 = argument error (RLGAN_E_*), > 0 = hipError_t.
 * No C++ exception crosses this boundary.
 *
 * Network kinds and tensor layouts
 * --------------------------------
 * A network is held three ways on the device:
 *   canon : flat fp32 vector, tensors in TensorFlow variable-creation order, each row-major
 *           [in, out] (DDPG.py:78-95,153-172; Curious_GAN.py:46-82,112-135);
 *   wf    : "forward fragments" -- per layer the matrix [W ; b] ((K+1) x N, bias as last row) cut
 *           into MFMA 16x16x4 B-operand fragments, K in groups of 16, N in tiles of 16: the float4
 *           at float index ((group*NT + tile)*64 + lane)*4 holds the elements
 *           (k = 16*group + 4*(lane/16) + m, n = 16*tile + lane%16), m = 0..3; zero outside;
 *   wb    : "backward fragments" -- the same for W^T (N x K, no bias) used for dX = dZ * W^T.
 * `rlgan_build_maps` gives, per canonical parameter, its float offset in wf, wb and in the padded
 * gradient image the weight-gradient kernel writes.
 */
#ifndef RLGAN_AMD_H
#define RLGAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLGAN_ABI_VERSION 1

enum { RLGAN_NET_ACTOR = 0, RLGAN_NET_CRITIC = 1, RLGAN_NET_GENERATOR = 2, RLGAN_NET_DISCRIMINATOR = 3 };

enum {
  RLGAN_E_ARG = -1,      /* null pointer / negative size */
  RLGAN_E_DIMS = -2,     /* state/action dims outside what the kernels are built for */
  RLGAN_E_KIND = -3,     /* unknown network kind */
  RLGAN_E_NOTFULL = -4,  /* replay sample before the ring is full (Curious_GAN.py:167 assert) */
  RLGAN_E_ALIGN = -5,    /* row count / leading dimension not as documented */
  RLGAN_E_SCRATCH = -6   /* size beyond the fixed device scratch of an allocation-free entry point */
};

/* sizes of the three device images of one network */
typedef struct {
  int32_t n_params;   /* canon length */
  int32_t wf_floats;  /* forward-fragment image length */
  int32_t wb_floats;  /* backward-fragment image length */
  int32_t gpad_floats;/* padded gradient image length (one batch chunk) */
  int32_t k0;         /* first-layer input width (sum of input blocks, without the bias row) */
  int32_t n3;         /* width of the third layer (10 or 50) */
  int32_t n_out;      /* output width */
  int32_t ld3;        /* leading dimension of the h3 / dz3 work arrays (16 or 64) */
} rlgan_geom_t;

/* device images of one network as the kernels consume them */
typedef struct {
  const float* wf;    /* device, forward fragments (required) */
  const float* wb;    /* device, backward fragments (NULL for forward-only use) */
} rlgan_net_t;

/* per-network activation / pre-activation-gradient work arrays written by a *_rows kernel and
 * consumed by rlgan_weight_grads.  `rows` rows (a multiple of 16) of `ld` columns each, stored as
 * [rows/16][4][ld][4]: inside every 16-row tile, four blocks of four rows, each block holding for
 * every column its four rows (element (r, c) of a tile at (r/4)*4*ld + 4*c + r%4).  ld = 32 for x0,
 * 208 for h1/h2/dz1/dz2, geom.ld3 for h3/dz3, 16 for dz4.  All device. */
typedef struct {
  float* x0;  float* h1;  float* h2;  float* h3;
  float* dz1; float* dz2; float* dz3; float* dz4;
} rlgan_acts_t;

/* a [rows, width] fp32 matrix on the device with an arbitrary row stride (in floats) */
typedef struct { const float* p; int32_t ld; } rlgan_mat_t;

const char* rlgan_error_string(int code);
int rlgan_abi_version(void);

/* ---- geometry / layout helpers (host-side, no GPU work) ------------------------------- */
int rlgan_geometry(int kind, int s_dim, int a_dim, rlgan_geom_t* out);
/* host arrays of n_params int32 each; map_b may be NULL.  Offsets of -1 mean "not present". */
int rlgan_build_maps(int kind, int s_dim, int a_dim, int32_t* map_f, int32_t* map_b, int32_t* map_g);
/* rows per batch chunk and chunk count the weight-gradient kernel will use for `rows` rows */
int rlgan_wgrad_chunks(int rows, int32_t* chunk_rows, int32_t* n_chunks);

/* scatter canon -> wf (and wb when non-NULL); entries whose offset is -1 are skipped.  wf/wb must have been
 * zero-filled once. */
int rlgan_pack_weights(const float* canon, int n_params, const int32_t* map_f, float* wf,
                       const int32_t* map_b, float* wb, void* stream);

/* ---- forward passes (acting path) ------------------------------------------------------
 * Actor.choose_action DDPG.py:103-105; Generator.predict Curious_GAN.py:150-153;
 * Discriminator.determine Curious_GAN.py:87-92; batched forms follow Conv_model.py:296,562.
 * Inputs by kind: actor (s); critic (s, a); generator (s, a); discriminator (s, x, a).
 * out_mode: 0 = network output, 1 = sigmoid(output), 2 = 1 - sigmoid(output) ("determine"). */
int rlgan_mlp_forward(int kind, int s_dim, int a_dim, rlgan_net_t net,
                      rlgan_mat_t s, rlgan_mat_t x, rlgan_mat_t a,
                      float* out, int ld_out, int out_mode, int n_rows, void* stream);

/* ---- the four update steps: row passes ---------------------------------------------------
 * Each processes 16-row tiles of the minibatch: forward, loss, backward-to-inputs, and leaves the
 * per-layer inputs and pre-activation gradients in `acts` for rlgan_weight_grads.
 * `inv_global_batch` = 1 / (batch rows summed over all data-parallel ranks).
 * `loss_parts` receives one partial SUM per 16-row tile (n_tiles = ceil(n_rows/16)).
 * `row_idx` (device int64[n_rows], may be NULL): when given, minibatch row i of every input matrix of
 * the step is read from row row_idx[i] of that matrix -- the matrices then point into the replay ring
 * and the fancy-index copy of Memory.sample (Curious_GAN.py:169) needs no pass of its own.  Indices
 * must lie inside the ring; they are not checked.
 * `counter` (device int32, may be NULL) is incremented once by the launch: the t_replace_counter of
 * Actor/Critic (DDPG.py:64,101,178) or the step count Adam's bias correction uses.  Doing it here
 * lets the optimizer launch that follows read the new value.                                   */

/* Critic.learn DDPG.py:141-148,174-175: y = r + gamma*q'(s_, a'(s_)); L = mean (y-q(s,a))^2.
 * When `actor_eval.wf` is non-NULL the gradient is continued into the actor eval net with TanhGrad
 * taken at the fed action (minimize() without var_list, DDPG.py:148) and `acts_actor` is filled. */
typedef struct {
  int32_t s_dim, a_dim, n_rows;
  rlgan_net_t critic_eval, critic_target, actor_target, actor_eval;
  rlgan_mat_t s, a, r, s_next;
  float gamma, inv_global_batch;
  rlgan_acts_t acts_critic, acts_actor;
  float* loss_parts;
  const int64_t* row_idx;
  int32_t* counter;
} rlgan_critic_step_t;
int rlgan_critic_step_rows(const rlgan_critic_step_t* step, void* stream);

/* Actor.learn DDPG.py:97-98,107-113 with a_grads = d(sum q)/da (DDPG.py:151): batch-SUM gradient. */
typedef struct {
  int32_t s_dim, a_dim, n_rows;
  rlgan_net_t actor_eval, critic_eval;
  rlgan_mat_t s;
  rlgan_acts_t acts_actor;
  const int64_t* row_idx;
  int32_t* counter;
} rlgan_actor_step_t;
int rlgan_actor_step_rows(const rlgan_actor_step_t* step, void* stream);

/* Discriminator.learn Curious_GAN.py:33-45,84-85.  acts_disc has 2*rows16 rows: fake rows first
 * (x = stored G sample), then real rows (x = s_next).  loss_parts: fake+real CE sum per tile. */
typedef struct {
  int32_t s_dim, a_dim, n_rows;
  rlgan_net_t disc;
  rlgan_mat_t s, g_stored, a, s_next;
  float label_real, inv_global_batch;
  rlgan_acts_t acts_disc;
  float* loss_parts;
  const int64_t* row_idx;
  int32_t* counter;
} rlgan_disc_step_t;
int rlgan_disc_step_rows(const rlgan_disc_step_t* step, void* stream);

/* Generator.learn Curious_GAN.py:137-145: L = mean CE(D_logit(s, a, G(s,a)), 1), dX through D. */
typedef struct {
  int32_t s_dim, a_dim, n_rows;
  rlgan_net_t gen, disc;
  rlgan_mat_t s, a;
  float inv_global_batch;
  rlgan_acts_t acts_gen;
  float* loss_parts;
  const int64_t* row_idx;
  int32_t* counter;
} rlgan_gen_step_t;
int rlgan_gen_step_rows(const rlgan_gen_step_t* step, void* stream);

/* The reference runs critic.learn, actor.learn, discriminator.learn, generator.learn in that order
 * (Curious_GAN.py:208-211); critic->actor and D->G are independent chains, so {critic, D} and then
 * {actor, G} can share a launch each (one tile of either step per compute unit at a time).
 * Same arguments and results as the single-step calls; both steps must have the same n_rows. */
int rlgan_phase1_rows(const rlgan_critic_step_t* critic, const rlgan_disc_step_t* disc, void* stream);
int rlgan_phase2_rows(const rlgan_actor_step_t* actor, const rlgan_gen_step_t* gen, void* stream);

/* Discriminator.eval + Generator.eval Curious_GAN.py:94-95,147-148 in one launch (two workgroups
 * per 16-row tile: D loss = fake + real pass, G loss = G then D).  loss_parts: [2][n_tiles] -- D loss
 * sums then G loss sums.  Inputs: the matrices s, g_stored, a, s_next (through row_idx when given), or --
 * when x0_disc and x0_gen are both non-NULL -- the input tiles (`acts.x0`) that rlgan_disc_step_rows
 * and rlgan_gen_step_rows left for the same minibatch, which costs no gather and does not touch the
 * replay ring (so an acting step may store into the ring in the same launch).  When loss_out is non-NULL the last workgroup to finish also writes
 * loss_out[0..1] = loss_scale * totals (fixed summation order); that needs `ticket`, one device int32
 * that is zero before the first launch using it (every launch leaves it zero again). */
typedef struct {
  int32_t s_dim, a_dim, n_rows;
  rlgan_net_t gen, disc;
  rlgan_mat_t s, g_stored, a, s_next;
  float label_real, loss_scale;
  float* loss_parts;
  float* loss_out;
  const int64_t* row_idx;
  const float* x0_disc; const float* x0_gen;
} rlgan_eval_t;
int rlgan_gan_eval_rows(const rlgan_eval_t* eval, int32_t* ticket, void* stream);

/* ---- weight gradients, reduction, optimizers ------------------------------------------------
 * Each call handles up to three (weight gradients) / four (reduce, apply) networks in one launch. */

/* dW_l = X_l^T dZ_l (bias gradient = the row of the constant-one input column) for the four layers
 * of a network, batch rows split into n_chunks chunks (rlgan_wgrad_chunks); chunk c writes its
 * partial image at gslabs + c*gpad_floats.  `rows` = row count of the acts arrays (multiple of 16). */
typedef struct { int32_t kind; int32_t rows; rlgan_acts_t acts; float* gslabs; } rlgan_wgrad_job_t;
int rlgan_weight_grads(const rlgan_wgrad_job_t* jobs, int n_jobs, int s_dim, int a_dim, void* stream);

/* grad[p] = sum_c gslabs[c*gpad_floats + map_g[p]] in chunk order (bitwise reproducible);
 * optional: loss_out[0] = loss_scale * sum(loss_parts[0..n_parts)), counter[0] += 1. */
typedef struct {
  const float* gslabs; int32_t n_chunks; int32_t gpad_floats; const int32_t* map_g; int32_t n_params;
  float* grad; const float* loss_parts; int32_t n_parts; float loss_scale; float* loss_out; int32_t* counter;
} rlgan_reduce_job_t;
int rlgan_grad_reduce(const rlgan_reduce_job_t* jobs, int n_jobs, void* stream);

/* One optimizer step per job, writing canon and its fragment images (wf may be NULL for a layer that is used
 * from canon alone; map offsets of -1 are skipped).
 *  opt 0: tf.train.RMSPropOptimizer (DDPG.py:112,148): ms <- decay*ms+(1-decay)*g^2 (ms = slot0, starts
 *         at 1, decay = beta1), w <- w - lr*g/sqrt(ms+eps);
 *  opt 1: tf.train.AdamOptimizer (Curious_GAN.py:45,142): m = slot0, v = slot1, lr_t =
 *         lr*sqrt(1-b2^t)/(1-b1^t) with t = step[0] (already incremented for this step, e.g. by
 *         rlgan_grad_reduce), eps outside the sqrt.
 * When target_canon is non-NULL the job also updates that target network from the new weights:
 * target <- tau*w + (1-tau)*target when (counter[0]-1) % period == 0 (counter NULL: always); the
 * reference only has the hard copy, tau = 1 (DDPG.py:99-101,176-178); target_wf may be NULL when wf is.
 * When `reduce` is non-NULL the reduction it describes (see rlgan_grad_reduce; its `counter` must be
 * NULL, its n_params equal) runs in the same launch and feeds the optimizer directly, `grad` is then
 * ignored and reduce->grad, if non-NULL, receives the gradient: the one-process form, where nothing
 * has to happen between reduction and optimizer. */
typedef struct {
  int32_t opt; int32_t n_params;
  float* canon; const float* grad; float* slot0; float* slot1;
  double lr, beta1, beta2; float eps;
  const int32_t* step;
  const int32_t* map_f; float* wf; const int32_t* map_b; float* wb;
  float* target_canon; float* target_wf; float tau; const int32_t* counter; int32_t period;
  const rlgan_reduce_job_t* reduce;   /* host pointer, may be NULL */
} rlgan_apply_job_t;
int rlgan_apply(const rlgan_apply_job_t* jobs, int n_jobs, void* stream);

/* out[k] = scale * sum(parts[k*n_parts .. (k+1)*n_parts)), k < n_sums, in a fixed order: turns the
 * per-tile loss sums of a *_rows call into the loss value (tf.reduce_mean, Curious_GAN.py:33-40). */
int rlgan_sum_parts(const float* parts, int n_parts, int n_sums, float scale, float* out, void* stream);

/* target <- tau*eval + (1-tau)*target when (counter[0]-1) % period == 0 (counter NULL: always).  map_f and target_wf
 * may both be NULL for parameters that are used from their canonical vector alone (no fragment image to follow). */
int rlgan_target_update(float* target_canon, const float* eval_canon, int n_params, float tau,
                        const int32_t* counter, int period, const int32_t* map_f, float* target_wf,
                        void* stream);

/* ---- environment ---------------------------------------------------------------------------
 * ArmEnv.step / _get_state / _r_func arm_env.py:43-59,91-115, one thread per environment, float64.
 * State arrays (device, struct of arrays, n_env each): theta[2n] (th1,th2 interleaved), ends[4n]
 * (x1,y1,x2,y2), goal[2n] (integer-valued doubles), grab[n] int32, got[n] uint8.
 * action: [n,2] fp32 or fp64 (action_f64 != 0).  Outputs: state_out [n,7] double, state_out_f32
 * [n,7] float (either may be NULL), reward[n] double, done[n] uint8. */
typedef struct { double* theta; double* ends; double* goal; int32_t* grab; uint8_t* got; } rlgan_env_t;

int rlgan_env_step(rlgan_env_t env, int n_env, const void* action, int action_f64, int ld_action,
                   double* state_out, float* state_out_f32, double* reward, uint8_t* done, void* stream);

/* ArmEnv.reset arm_env.py:61-78 for the environments with mask[i] != 0 (mask NULL: all).
 * u: [n,2] uniform [0,1) doubles (the reference's np.random.rand(2)).  mode_hard: 1 = 'hard'. */
int rlgan_env_reset(rlgan_env_t env, int n_env, int mode_hard, const uint8_t* mask, const double* u,
                    double* state_out, float* state_out_f32, void* stream);

/* Episode cut of the rollout loop (Curious_GAN.py:180,222-234): environment i is reset when
 * done[i] != 0 or ep_t[i] >= max_ep_steps-1 (then ep_t[i] = 0 and its new state goes to
 * state_f32[i,:]), otherwise ep_t[i] += 1.  n_resets (optional) counts resets.  counters
 * (optional, device int64[2]) is advanced by the same launch: [0] += n_env (the replay pointer the
 * acting step reads), [1] += 1 (vector steps taken), which lets a captured graph replay the loop. */
int rlgan_env_autoreset(rlgan_env_t env, int n_env, int mode_hard, const uint8_t* done, int32_t* ep_t,
                        int max_ep_steps, const double* u, float* state_f32, int32_t* n_resets,
                        int64_t* counters, void* stream);

/* ---- replay ring ----------------------------------------------------------------------------
 * Memory.store_transition Curious_GAN.py:161-165 / DDPG.py:187-191 for n rows at once:
 * data[(pointer+i) % capacity, :] = rows[i, :]; pointer is held by the caller.  The ring is fp32
 * with row stride ld_data >= dims floats (32 makes every row one aligned 128-byte segment). */
int rlgan_replay_store(float* data, int capacity, int dims, int ld_data, int64_t pointer, const float* rows,
                       int ld_rows, int n_rows, void* stream);
/* Memory.sample's fancy-index copy Curious_GAN.py:169: out[i,:] = data[idx[i],:]. */
int rlgan_replay_gather(const float* data, int capacity, int dims, int ld_data, const int64_t* idx, int n,
                        float* out, int ld_out, void* stream);

/* np.random.choice(n_pop, size=n) on the legacy MT19937 stream (Curious_GAN.py:168): masked
 * rejection over 32-bit outputs, bit-exact.  state: device uint32[625] = key[624] + position. */
int rlgan_mt19937_seed(uint32_t* state, uint32_t seed, void* stream);
int rlgan_mt19937_choice(uint32_t* state, int64_t n_pop, int n, int64_t* out_idx, void* stream);

/* Philox4x32-10 draws for order-independent exploration noise and goal placement (no reference
 * counterpart: the reference draws both from the global MT19937, Curious_GAN.py:191, arm_env.py:66).
 * element i of `out` pairs: counter (index0+i), stream id `stream_id`, key `seed`.  When `step` (device
 * int64) is non-NULL, step[0]*step_mul is added to index0 on the device. */
int rlgan_philox_normal(float* out, int n_pairs, uint64_t seed, uint32_t stream_id, uint64_t index0,
                        const int64_t* step, int64_t step_mul, void* stream);
int rlgan_philox_uniform(double* out, int n_pairs, uint64_t seed, uint32_t stream_id, uint64_t index0,
                         const int64_t* step, int64_t step_mul, void* stream);

/* ---- uint8 frame replay of the pixel variants ------------------------------------------------------
 * The reference keeps Python lists of float32 frame stacks (Conv_model.py:102-157) that its environment
 * wrapper produced as uint8 * (1/255) (env_wrapper.py:44,58).  The ring holds the uint8 stacks
 * ([slots][frame_bytes], frame_bytes a multiple of 4); the gather applies the scale, the same float32
 * product bit for bit.
 *   store : ring[slot_idx[i], :] = frames[i, :]       (slot rule of the reference: host side)
 *   gather: out[i, :] = float(ring[idx[i], :]) * scale, out float32 [n][frame_bytes], 16-byte aligned;
 *           an index outside [0, slots) gives a zero frame. */
int rlgan_frames_store(uint8_t* ring, int64_t slots, int32_t frame_bytes, const uint8_t* frames, int32_t n,
                       const int64_t* slot_idx, void* stream);
int rlgan_frames_gather(const uint8_t* ring, int64_t slots, int32_t frame_bytes, const int64_t* idx, int32_t n, float scale,
                        float* out, void* stream);
/* One channel of the gathered frames, the `b_s_[:, :, :, [-1]]` the pixel loop hands to the discriminator
 * (Curious_GAN_mario.py:92): out[i, p] = float(ring[idx[i], p * channels + channel]) * scale, out float32
 * [n][pixels], frames of pixels * channels bytes; the same values as that slice of rlgan_frames_gather's output. */
int rlgan_frames_gather_channel(const uint8_t* ring, int64_t slots, int32_t pixels, int32_t channels, int32_t channel,
                                const int64_t* idx, int32_t n, float scale, float* out, void* stream);

/* ---- fused acting path for n_env environments (Curious_GAN.py:190-198,222-234) -------------------
 * a = clip(actor(s) + var*z, -1, 1); g = G(s, a); (s', r, done) = env.step(a);
 * curious_r = r + ita*(1 - D(s, a, g)); row = [s, s', g, a, r, curious_r] -> replay ring at
 * (pointer + env index) % capacity; then the episode cut: environment i is reset when done or
 * ep_t[i] >= max_ep_steps-1 (ep_t[i] = 0), else ep_t[i] += 1; state_f32[i,:] receives the state the
 * next step starts from.
 *  z       : `noise` [n,2] standard normal fp32 when non-NULL; else, when rng_seed != 0, Philox stream 0
 *            at counter step*n_env + i (what rlgan_philox_normal gives for the same key); else 0.
 *  reset   : uniforms from `reset_u` [n,2] float64 when non-NULL, else Philox stream 1 at counter
 *            n_env + step*n_env + i.  ep_t NULL: no episode cut.
 *  counters: device int64[2] = {replay pointer, vector steps}; when non-NULL they replace `pointer` and
 *            `step` and the last workgroup of the launch advances them by n_env and 1, which lets a
 *            captured graph replay the loop.
 *  var     : device float64 scalar.  When var_decay > 0 the last workgroup sets
 *            var = max(var*var_decay, var_min) if the pointer after this step exceeds `capacity`
 *            (`if M.pointer > MEMORY_CAPACITY`, Curious_GAN.py:197-198).
 *  counters_out / var_out: when counters_out is non-NULL the launch leaves `counters` and `var` as it
 *            read them and writes the advanced values (and the decayed or unchanged noise scale, var_out
 *            required when var is given) to these other buffers instead; the caller swaps the two pairs
 *            between launches (a captured loop holds two graphs, one per direction).  Nothing then waits
 *            for the last workgroup.
 * `ticket` (see rlgan_gan_eval_rows) is required when counters_out is NULL and counters is non-NULL or
 * var_decay > 0. */
typedef struct {
  int32_t s_dim, a_dim, n_env;
  rlgan_net_t actor_eval, gen, disc;
  rlgan_env_t env;
  float* state_f32;
  const float* noise;
  double* var;
  float ita;
  float* replay; int32_t capacity, dims, ld_replay;
  int64_t pointer, step;
  int64_t* counters;
  double* reward_out; uint8_t* done_out;
  uint64_t rng_seed;
  int32_t* ep_t; int32_t max_ep_steps, mode_hard; const double* reset_u; int32_t* n_resets;
  double var_decay, var_min;
  int64_t* counters_out; double* var_out;
} rlgan_act_t;
int rlgan_act_rows(const rlgan_act_t* act, int32_t* ticket, void* stream);

/* np.random.choice(n_pop, size=n) as one extra workgroup of another launch (see rlgan_mt19937_choice). */
typedef struct { uint32_t* mt_state; int64_t n_pop; int32_t n; int64_t* out_idx; } rlgan_sample_t;

/* One launch for everything that only READS the networks between two updates: the loss evaluation of
 * the update just done, the acting step that follows it, and the index draw for the next minibatch
 * (any of the three may be NULL).  With equal tile counts every compute unit holds one acting, one D-loss
 * and one G-loss tile. */
int rlgan_rollout_eval_rows(const rlgan_act_t* act, const rlgan_eval_t* eval, const rlgan_sample_t* sample,
                            int32_t* ticket, void* stream);

/* ---- pixel encoder layers ------------------------------------------------------------------------
 * tf.layers.conv2d(..., padding='valid', activation=tf.nn.elu) on NHWC tensors, the building block of
 * the reference's feature extractors (Conv_model.py:15-24,36-56; Conv_DDPG.py:83-86): square kernel
 * k x k, stride `stride`, c_in and c_out multiples of 4, c_out <= 64.
 * Parameters in canonical order: the HWIO kernel [k, k, c_in, c_out] row-major (TensorFlow's layout),
 * then bias[c_out]: n_params = (k*k*c_in + 1) * c_out.  On the device the layer is kept like a dense
 * one: `canon` plus a fragment image `wfrag` of [W ; b] (wfrag_floats floats, zero-filled once) that
 * rlgan_pack_weights / rlgan_apply fill through the int32 map of rlgan_conv_build_map. */
typedef struct { int32_t h, w, c_in, c_out, k, stride; } rlgan_conv_t;
typedef struct { int32_t h_out, w_out, k_len, n_params, wfrag_floats, slab_floats, wdx_floats; } rlgan_conv_geom_t;
int rlgan_conv_geometry(const rlgan_conv_t* conv, rlgan_conv_geom_t* out);
int rlgan_conv_build_map(const rlgan_conv_t* conv, int32_t* map_f);   /* host array of n_params int32 */
/* Second image for the input gradient (wdx_floats floats, zero-filled once; 0: not available for this
 * layer): the kernel regrouped by stride phase as MFMA operands.  map_dx: host array of n_params int32,
 * -1 for the bias; fill the image with rlgan_pack_weights(canon, n, map_f, wfrag, map_dx, wdx). */
int rlgan_conv_build_dx_map(const rlgan_conv_t* conv, int32_t* map_dx);

/* y[n, h_out, w_out, c_out] = act(conv(x[n, h, w, c_in]) + b); act_elu != 0: ELU, else linear. */
int rlgan_conv_forward(const rlgan_conv_t* conv, const float* x, const float* wfrag, float* y, int n_images,
                       int act_elu, void* stream);
/* First layer straight from the uint8 frame ring ([slots][h*w*c_in], see rlgan_frames_*): image i of the
 * batch is ring slot idx[i] (idx NULL: slot i, n_images <= slots), an input value is float(byte) * scale, so
 * the float32 frames of the reference's pixel replay (Conv_model.py:145-157, env_wrapper.py:44,58) are never
 * written to memory.  Results are bit-equal to rlgan_frames_gather followed by the float32 entry point.
 * An index outside [0, slots) reads as a zero image.  The first layer needs no input gradient. */
int rlgan_conv_forward_u8(const rlgan_conv_t* conv, const uint8_t* ring, int64_t slots, const int64_t* idx, float scale,
                          const float* wfrag, float* y, int n_images, int act_elu, void* stream);
int rlgan_conv_backward_weights_u8(const rlgan_conv_t* conv, const uint8_t* ring, int64_t slots, const int64_t* idx, float scale,
                                   const float* dz, const float* y_elu, float* slabs, int n_slabs, float* grad, int n_images,
                                   void* stream);

/* dz = dy * ELU'(y) elementwise, ELU' taken from the layer OUTPUT y (y > 0 ? 1 : y + 1). */
int rlgan_elu_backward(const float* dy, const float* y, float* dz, int64_t n, void* stream);
/* dx[n, h, w, c_in] from dz[n, h_out, w_out, c_out].  With wdx (see rlgan_conv_build_dx_map): one
 * implicit GEMM per stride phase on the matrix cores; with wdx NULL: a gather from the canonical kernel
 * w_canon (device, [k*k*c_in, c_out]) on the vector ALUs, for layers wdx does not cover.
 * A layer with c_in == 4 (an encoder's first layer) and w_canon given runs a kernel of its own instead, one
 * thread per input pixel, whatever wdx is.
 * y_elu (optional, needs wdx or c_in == 4): the layer's output; `dz` is then dy and is multiplied by ELU'(y)
 * as it is read, which saves the separate rlgan_elu_backward pass. */
int rlgan_conv_backward_input(const rlgan_conv_t* conv, const float* dz, const float* y_elu, const float* w_canon, const float* wdx,
                              float* dx, int n_images, void* stream);
/* The same with one more factor: when x_elu (this layer's input, i.e. the ELU output of the layer below, [n, h, w, c_in])
 * is given, dx leaves multiplied by ELU'(x_elu) -- it is then the dZ of the layer below, whose own backward calls take
 * it with y_elu = NULL and no longer read their layer output (for a 42x42x4 -> 20x20x32 first layer that output is
 * 210 MB per 4096 frames, read by the weight gradient at the memory roof).  dx * ELU'(x) is the product those calls
 * formed while reading: the same bits.  x_elu = NULL: rlgan_conv_backward_input. */
int rlgan_conv_backward_input_dz(const rlgan_conv_t* conv, const float* dz, const float* y_elu, const float* w_canon, const float* wdx,
                                 float* dx, const float* x_elu, int n_images, void* stream);
/* grad (canonical order, kernel then bias) = sum over all output pixels.  `slabs`: device scratch of
 * n_slabs * slab_floats floats (one partial image per workgroup, added in a fixed order).  y_elu as above. */
int rlgan_conv_backward_weights(const rlgan_conv_t* conv, const float* x, const float* dz, const float* y_elu, float* slabs,
                                int n_slabs, float* grad, int n_images, void* stream);

/* ---- fully connected head of the pixel encoders (Conv_model.py:24 `fc`, :52-56 `dense1` + elu) ------
 * wb: canonical parameters on the device, the [k, n] kernel (row-major, TF layout) followed by the
 * bias row [n].  x: [m, k] with leading dimension ldx; y, dz: [m, n].  Bases 16-byte aligned, leading
 * dimensions multiples of four.
 *   forward:          y = act(x * W + b), act: 0 linear, 1 ELU, 2 ReLU, 3 relu6, 4 tanh
 *   backward_input:   dx[m, k] = dz * W^T           (dz already times ELU'(y): rlgan_elu_backward)
 *   backward_weights: gwb[k+1, n] = [x 1]^T * dz    (kernel gradient, then the bias gradient row)
 * The sum over the long dimension can be cut into n_slabs ranges (more workgroups when the output alone
 * has few tiles); the partial products then go to `slabs`, device scratch of n_slabs * rows * n floats
 * (rows = m forward, k+1 for the weights; m * k floats per slab for the input gradient's _split form),
 * and are added in slab order, so the result does not depend on timing.  n_slabs <= 1: no scratch.  rlgan_dense_slabs suggests a count for a product with an
 * m_out x n_out result and k_red terms per element. */
int rlgan_dense_slabs(int m_out, int n_out, int k_red);
int rlgan_dense_forward(const float* x, int ldx, const float* wb, float* y, int ldy, int m, int k, int n, int act,
                        float* slabs, int n_slabs, void* stream);
int rlgan_dense_backward_input(const float* dz, int lddz, const float* wb, float* dx, int lddx, int m, int k, int n, void* stream);
int rlgan_dense_backward_input_split(const float* dz, int lddz, const float* wb, float* dx, int lddx, int m, int k, int n,
                                     float* slabs, int n_slabs, void* stream);
int rlgan_dense_backward_weights(const float* x, int ldx, const float* dz, int lddz, float* gwb, int m, int k, int n,
                                 float* slabs, int n_slabs, void* stream);
/* The kernel gradient alone, gw[k, n] = x^T * dz (slabs of k * n floats): for a caller that takes the bias gradient from
 * elsewhere, like the transposed convolutions below (their bias is added after the placement, Conv_model.py:457-473, so
 * its gradient is a column sum of dY, not of this product's dz).  With k a multiple of 64 the ones row of
 * rlgan_dense_backward_weights would cost a row of 64-row tiles of its own. */
int rlgan_dense_backward_kernel(const float* x, int ldx, const float* dz, int lddz, float* gw, int m, int k, int n,
                                float* slabs, int n_slabs, void* stream);

/* ---- transposed convolutions of the pixel generator's decoder --------------------------------------
 * tf.layers.conv2d_transpose(net, c_out, k, strides=stride) on NHWC tensors, padding 'valid', the decoder
 * stack of the reference's full-prediction generator, 1x1x200 -> 3x3x128 -> 8x8x64 -> 19x19x32 -> 42x42x1
 * with ReLU between the layers (Conv_model.py:457-473).  Parameters in canonical order: TensorFlow's kernel
 * [k, k, c_out, c_in] row-major, then bias[c_out].  c_in and k*k*c_out must be multiples of 4.
 * A layer runs as   cols = x2d * Wmat   (rlgan_dense_forward, x2d = [n*h*w][c_in], Wmat = [c_in + 1][kk] with a
 * zero last row, kk = k*k*c_out; filled from canon through rlgan_deconv_build_map and rlgan_pack_weights),
 * then rlgan_deconv_place.  Backward: rlgan_deconv_gather (dcols from dY), rlgan_dense_backward_input (dX),
 * rlgan_dense_backward_kernel + rlgan_gather_map (kernel gradient in canonical order), rlgan_column_sums
 * (bias gradient). */
typedef struct { int32_t h, w, c_in, c_out, k, stride; } rlgan_deconv_t;
typedef struct { int32_t h_out, w_out, kk, n_params, wmat_floats, colsum_floats; } rlgan_deconv_geom_t;
int rlgan_deconv_geometry(const rlgan_deconv_t* deconv, rlgan_deconv_geom_t* out);
/* host array of n_params int32: offset of every canonical parameter in Wmat, -1 for the bias */
int rlgan_deconv_build_map(const rlgan_deconv_t* deconv, int32_t* map_w);
/* y[n, oy, ox, co] = act(bias[co] + sum of cols[(n, iy, ix)][(ky, kx, co)] over oy = iy*stride + ky, ox = ix*stride + kx),
 * taps added in ascending (ky, kx) order; act_relu 0/1.  n_images <= 65535. */
int rlgan_deconv_place(const rlgan_deconv_t* deconv, const float* cols, const float* bias, float* y, int n_images,
                       int act_relu, void* stream);
/* dcols[(n, iy, ix)][(ky, kx, co)] = dy[n, iy*stride + ky, ix*stride + kx, co] (* ReLU'(y_relu) when y_relu != NULL) */
int rlgan_deconv_gather(const rlgan_deconv_t* deconv, const float* dy, const float* y_relu, float* dcols, int n_images,
                        void* stream);
/* The two products of the backward pass that read dcols, reading dz = dy * ReLU'(y) (rlgan_relu_backward; dy itself for
 * a layer without activation) in place of it: element [(n, iy, ix)][(ky, kx, co)] of dcols IS dz[n, iy*stride + ky,
 * ix*stride + kx, co], so rlgan_deconv_gather and its matrix (151 MB for 3x3x128 -> 8x8x64 at 4096 images) are not
 * needed (Conv_model.py:463, the backward pass TensorFlow derives for it).  Same tiles, products and order as
 * rlgan_dense_backward_input_split / rlgan_dense_backward_kernel on dcols: the same bits.  For whole tiles only:
 * rlgan_deconv_view_ok says whether n_images * h * w, c_in, k*k*c_out and c_out are all multiples of 64 (1) or the
 * cols form has to be used (0); the two products return RLGAN_E_DIMS in that case.
 * dx[n*h*w][c_in] = dcols * Wmat[0:c_in]^T;  gw[c_in][kk] = x2d^T * dcols;  slabs / n_slabs as in the dense products. */
int rlgan_deconv_view_ok(const rlgan_deconv_t* deconv, int n_images);
int rlgan_deconv_backward_input_view(const rlgan_deconv_t* deconv, const float* dz, const float* wmat, float* dx, int n_images,
                                     float* slabs, int n_slabs, void* stream);
int rlgan_deconv_backward_kernel_view(const rlgan_deconv_t* deconv, const float* x, const float* dz, float* gw, int n_images,
                                      float* slabs, int n_slabs, void* stream);
/* out[c] = sum over rows of dy[r][c] (* ReLU'(y_relu[r][c]) when y_relu != NULL), n_cols <= 256, in a fixed order
 * (bias gradient of a layer); partial: scratch of 512 * n_cols floats. */
int rlgan_column_sums(const float* dy, const float* y_relu, int64_t rows, int32_t n_cols, float* partial, float* out,
                      void* stream);
/* rlgan_column_sums without the `partial` argument, for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * partial sums go to a fixed 512 KiB array inside the library (512 partial rows x 256 columns, one copy per GPU,
 * held until the process ends), so n_cols <= 256 or RLGAN_E_SCRATCH; `rows` is unlimited.  Same kernels, same
 * order of additions and same bits as rlgan_column_sums.  The call allocates and frees nothing.  All calls on one
 * GPU share that array: two calls whose kernels may overlap on one GPU (different streams) race; calls on one
 * stream, or ordered by events, do not. */
int rlgan_column_sums_noalloc(const float* dy, const float* y_relu, int64_t rows, int32_t n_cols, float* out, void* stream);
/* Set-up for rlgan_column_sums_noalloc on the CURRENT device: looks up the device address of the fixed scratch array
 * (one symbol lookup, which may load the library's code object on that GPU).  Call it once per GPU, outside HIP graph
 * capture, before the first allocation-free call that is recorded into a graph; a call that finds the address missing
 * still looks it up itself (for the device that owns `stream`), which is fine outside capture only.  No counterpart in
 * the reference.  Allocates nothing. */
int rlgan_noalloc_init(void);
/* dst[i] = map[i] >= 0 ? src[map[i]] : 0   (a gradient image back into canonical parameter order) */
int rlgan_gather_map(const float* src, const int32_t* map, int32_t n, float* dst, void* stream);

/* ---- actor and critic of the pixel variant as row passes (Conv_model.py:569-784) ------------------------------------
 * Network kinds 4 (actor) and 5 (critic) of rlgan_geometry / rlgan_build_maps / rlgan_weight_grads: s_dim = 256, the
 * encoder's features; a_dim = the action columns padded to a multiple of 4 (4 .. 12).  Canonical order = the four
 * layers' blocks one after the other, each the [K, N] kernel then the bias: actor 256x200, 200x200, 200x12, 12xa_pad;
 * critic (256 + a_pad)x200 (rows: features, then action columns), 200x200, 200x12, 12x4 (q is output column 0); the
 * padding columns hold zero weights and receive zero gradients.  relu6 after the first three layers, tanh on the
 * actor's output.  The x0 work array of these kinds is 272 columns wide.
 *
 * rlgan_pixel_critic_step_rows (Conv_model.py:610-616,653-663): a' = actor_target(phis'), q' = critic_target(phis', a'),
 * q = critic_eval(phis, a), loss = mean((q - (r + gamma q'))^2), backward through critic_eval; loss_parts[tile] =
 * the tile's sum of squared errors (the loss is their sum / n_rows).  a: [n_rows][>= a_cols], its first a_cols
 * columns are the action; r: [n_rows].  counter (optional) += 1.
 * rlgan_pixel_actor_step_rows (Conv_model.py:619,746-754,778-784): a = actor_eval(phis), dq/da through critic_eval with
 * d(sum_b q_b) = 1, backward through actor_eval.
 * Afterwards: rlgan_weight_grads on the filled work arrays, then rlgan_apply. */
typedef struct {
  int32_t a_pad, a_cols, n_rows;
  rlgan_net_t critic_eval, critic_target, actor_target;
  rlgan_mat_t phis, phis_next, a;
  const float* r;
  float gamma;
  rlgan_acts_t acts_critic;
  float* loss_parts;
  int32_t* counter;
} rlgan_pixel_critic_step_t;
typedef struct {
  int32_t a_pad, n_rows;
  rlgan_net_t actor_eval, critic_eval;
  rlgan_mat_t phis;
  rlgan_acts_t acts_actor;
  int32_t* counter;
} rlgan_pixel_actor_step_t;
int rlgan_pixel_critic_step_rows(const rlgan_pixel_critic_step_t* step, void* stream);
int rlgan_pixel_actor_step_rows(const rlgan_pixel_actor_step_t* step, void* stream);

/* ---- the same layers without the cols matrix in memory (dec_deconv3 and dec_deconv4, Conv_model.py:468-473) ----
 * Two stride-2 shapes run as fused kernels: 8x8x64 -> 19x19x32 with a 5x5 kernel (kind 1) and 19x19x32 -> 42x42x1
 * with a 6x6 kernel (kind 2); 3x3x128 -> 8x8x64 with a 4x4 kernel (dec_deconv2, Conv_model.py:463) has a fused FORWARD
 * pass only (kind 3: wff alone, map_fb may be NULL, no workspace, rlgan_deconv_fused_backward refuses it with
 * RLGAN_E_DIMS; its backward pass is the cols form, so the layer keeps Wmat as well);
 * rlgan_deconv_fused_geometry reports kind 0 for every other shape, which keeps the cols form above.  The kernel is read from two fragment images filled from the canonical vector through
 * rlgan_deconv_fused_build_maps + rlgan_pack_weights (or by rlgan_apply, map_f/wf and map_b/wb): wff for the
 * forward pass, wfb for the input gradient; both must start as zeros (their padding is never written).
 * Results: the values of the cols form up to the order of the float32 additions inside a product; taps are added
 * in ascending (ky, kx) order in both forms, and every reduction has a fixed order (the same bits every run). */
typedef struct { int32_t kind, wff_floats, wfb_floats, kernel_floats, max_chunks, bias_part_floats_per_image; } rlgan_deconv_fused_geom_t;
int rlgan_deconv_fused_geometry(const rlgan_deconv_t* deconv, rlgan_deconv_fused_geom_t* out);
/* floats of workspace rlgan_deconv_fused_backward needs for n_images (0: shape not fused) */
int64_t rlgan_deconv_fused_workspace(const rlgan_deconv_t* deconv, int n_images);
/* host arrays of n_params int32 each: offset of every canonical parameter in wff / wfb, -1 for the bias */
int rlgan_deconv_fused_build_maps(const rlgan_deconv_t* deconv, int32_t* map_ff, int32_t* map_fb);
/* y[n, oy, ox, co] = act(bias[co] + sum over taps and input channels), x = [n, h, w, c_in]; act_relu 0/1 */
int rlgan_deconv_fused_forward(const rlgan_deconv_t* deconv, const float* x, const float* wff, const float* bias, float* y,
                               int n_images, int act_relu, void* stream);
/* grad = [kernel gradient in canonical order, bias gradient] of dZ = dy (* ReLU'(y_relu) when y_relu != NULL);
 * dx (may be NULL) = input gradient [n, h, w, c_in].  workspace: rlgan_deconv_fused_workspace floats. */
int rlgan_deconv_fused_backward(const rlgan_deconv_t* deconv, const float* x, const float* y_relu, const float* dy,
                                const float* wfb, float* dx, float* workspace, float* grad, int n_images, void* stream);
/* The same with dx_relu: 1 = dx leaves multiplied by ReLU'(x) (x is the ReLU output of the layer below), i.e. as that
 * layer's dZ; its backward call then takes y_relu = NULL and does not read its output again (kind 2 holds x in LDS
 * anyway, kind 1 reads it beside its stores).  dx must be given.  dx * ReLU'(x) is the product the layer below formed
 * while reading: the same bits. */
int rlgan_deconv_fused_backward_dz(const rlgan_deconv_t* deconv, const float* x, const float* y_relu, const float* dy,
                                   const float* wfb, float* dx, int dx_relu, float* workspace, float* grad, int n_images,
                                   void* stream);

/* ---- glue of the pixel generator's training step (Conv_model.py:437-456,522-541) ------------------------
 * dz = dy * ReLU'(y)   (tf.nn.relu after the generator's first layer, Conv_model.py:455) */
int rlgan_relu_backward(const float* dy, const float* y, float* dz, int64_t n, void* stream);
/* dst[r][0..n_copy) = src[r][0..n_copy), dst[r][n_copy..n_copy+n_zero) = 0 for r < rows: concatenation of
 * [features, action] rows (Conv_model.py:452) with zero columns up to a 16-byte row, and column slices of such rows on
 * the way back (n_zero = 0). */
int rlgan_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t rows, int n_copy, int n_zero, void* stream);
/* The concatenation itself in one launch: dst[r] = [s0[r][0..w0) | s1[r][0..w1) | s2[r][0..w2) | n_zero zeros] for r < rows
 * (`tf.concat([phis, a], 1)` Conv_model.py:452, `tf.concat([phi1, phi2], 1)` :806, the discriminator's
 * [phis, features, a] rows :183-199).  A block of width 0 is skipped (its pointer may be null). */
int rlgan_concat_cols(const float* s0, int ld0, int w0, const float* s1, int ld1, int w1, const float* s2, int ld2, int w2,
                      float* dst, int ld_dst, int64_t rows, int n_zero, void* stream);
/* One-channel images [n][h][w] <-> their 2x2 blocks as four channels [n][h/2][w/2][4], channel 2*dy + dx =
 * image[2Y + dy][2X + dx] (h, w even; both buffers 16-byte aligned).  to_blocks != 0: src is the image.  A 3x3
 * stride-2 convolution of the image (`universe_feature_abstraction(single_S_, time_step=1)`, Conv_model.py:36-49,
 * Curious_GAN_mario.py:313-314) is a 2x2 stride-1 convolution of the blocks, whose kernel holds the nine taps and seven
 * zeros: the form the single-frame encoder's first layer runs in; the other direction carries its input gradient
 * back to the image. */
int rlgan_blocks2x2(const float* src, float* dst, int64_t n_images, int h, int w, int to_blocks, void* stream);
/* tf.reduce_mean(tf.squared_difference(fake, real)) (Conv_model.py:527): parts[b] = sum of (fake-real)^2 over
 * workgroup b's 4096 elements (n_parts = ceil(n / 4096); the loss is rlgan_sum_parts(parts, n_parts, 1, 1/n)),
 * dfake = 2 (fake - real) / n when non-NULL. */
int rlgan_mse(const float* fake, const float* real, int64_t n, float* parts, float* dfake, void* stream);
/* dst[r][0..n) += src[r][0..n) for r < rows (the two uses of phis in the pixel discriminator, Conv_model.py:183-191) */
int rlgan_add_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t rows, int n, void* stream);
/* D_loss_real / D_loss_fake of the pixel discriminator (tf.nn.sigmoid_cross_entropy_with_logits + reduce_mean,
 * Conv_model.py:207-214) on column 0 of logits[n_real + n_fake][ld], real rows first (label 1), fake rows after
 * (label 0): loss[0] = mean over the real rows, loss[1] = mean over the fake rows; when dlogits is non-NULL,
 * dlogits[r][0] = (sigmoid - label) / (rows of r's half) and its other ld - 1 columns are zeroed. */
int rlgan_sigmoid_ce(const float* logits, int ld, int n_real, int n_fake, float* loss, float* dlogits, void* stream);
/* out[r] = 1 - sigmoid(logits[r][0]), r < n: the intrinsic reward `1 - D_fake` (Conv_model.py:294,298) */
int rlgan_one_minus_sigmoid(const float* logits, int ld, int n, float* out, void* stream);

/* ---- inverse model of the pixel variant (StateActionPredictor, Conv_model.py:787-881) ------------------
 * tf.layers.batch_normalization over the feature axis (Conv_model.py:838,844,850; defaults: epsilon 1e-3, trained
 * gamma and beta) fused with the relu6 that follows it (act 3; 0: none):
 *   training != 0: y = act(gamma * (z - mean) / sqrt(var + eps) + beta) with the mean and biased variance of
 *                  the m rows, which are written to save_mean / save_inv_std (1 / sqrt(var + eps)) for the backward;
 *   training == 0: the moving statistics, which the reference leaves at their initial 0 and 1 (its update ops are
 *                  never run): y = act(gamma * z / sqrt(1 + eps) + beta); save_* may be NULL. */
int rlgan_batchnorm_forward(const float* z, int ldz, const float* gamma, const float* beta, int m, int n, float eps, int training,
                            int act, float* y, int ldy, float* save_mean, float* save_inv_std, float* scratch, void* stream);
/* Gradient of the training-mode form: dy is the gradient of the activated output y (needed when act == 3);
 * writes dz [m][n], dgamma[n], dbeta[n].  Column sums in a fixed order. */
int rlgan_batchnorm_backward(const float* dy, int lddy, const float* y, int ldy, const float* z, int ldz, const float* gamma,
                             const float* save_mean, const float* save_inv_std, int m, int n, int act, float* dz, int lddz,
                             float* dgamma, float* dbeta, float* scratch, void* stream);
/* Row chunks the two calls above cut m rows into.  1 (m <= 256): one launch each, scratch unused (may be NULL).  Otherwise
 * each call takes per-chunk partial results through `scratch`, device memory of 2 * chunks * n floats, joined in chunk
 * order (chunk means and squared deviations by the pairwise update formula), and save_mean / save_inv_std are required
 * in both modes. */
int rlgan_batchnorm_chunks(int m);
/* tf.nn.sparse_softmax_cross_entropy_with_logits(logits, tf.argmax(a, 1)) per row (Conv_model.py:801,818):
 * loss_rows[r] (the loss is their mean: rlgan_sum_parts(loss_rows, m, 1, 1/m)); when dlogits is non-NULL,
 * dlogits[r][c] = (softmax - onehot) / m for c < n_classes and 0 for the padding columns up to ld. */
int rlgan_softmax_ce(const float* logits, int ld, int n_classes, const float* a, int ld_a, int m, float* loss_rows, float* dlogits,
                     void* stream);

/* dz = dy * act'(y) from the activated output y: act 3 = relu6 (mask 0 < y < 6), act 4 = tanh (1 - y^2); the
 * activations of the pixel variant's actor and critic (Conv_model.py:634,641,648,721,728,735,742). */
int rlgan_act_backward(const float* dy, const float* y, float* dz, int64_t n, int act, void* stream);

/* TD error of the pixel critic (Conv_model.py:609-613): d[r] = q[r][0] - (reward[r] + gamma * q_next[r][0]), r < m;
 * parts[b] = sum of d^2 over rows 256 b .. 256 b + 255 (the loss is rlgan_sum_parts(parts, ceil(m / 256), 1, 1 / m));
 * when dq is non-NULL, dq[r][0] = 2 d[r] / m and dq[r][1 .. ld) = 0.  q, q_next, dq: leading dimension ld. */
int rlgan_td_error(const float* q, const float* q_next, int ld, const float* reward, float gamma, int m, float* parts, float* dq,
                   void* stream);

/* State update of tf.contrib.rnn.BasicLSTMCell (the `lstm_unit` of the pixel actor, Conv_model.py:67-99) from the gate
 * pre-activations gates[m][4 * units] = [x, h] * kernel + bias (rlgan_dense_forward), TensorFlow's gate order i, j, f, o:
 * c_out = c_in * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j), h_out = tanh(c_out) * sigmoid(o); forget_bias is 1 in
 * the reference (the cell's default).  c_in, c_out, h_out: dense [m][units]. */
int rlgan_lstm_gates(const float* gates, const float* c_in, int m, int units, float forget_bias, float* c_out, float* h_out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLGAN_AMD_H */
