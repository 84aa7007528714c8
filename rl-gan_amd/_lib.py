"""ctypes binding of librlgan_amd.so (the C ABI declared in include/rlgan_amd.h).

The library is the product: there is no Python or CPU fallback.  If it is missing the import
fails with instructions to build it; if a call returns a non-zero status a RuntimeError carrying
the library's own error string is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librlgan_amd.so")

NET_ACTOR, NET_CRITIC, NET_GENERATOR, NET_DISCRIMINATOR = 0, 1, 2, 3
NET_PIXEL_ACTOR, NET_PIXEL_CRITIC = 4, 5                         # actor / critic of the pixel variant on 256 encoder features
KIND_NAMES = {NET_ACTOR: "actor", NET_CRITIC: "critic", NET_GENERATOR: "generator", NET_DISCRIMINATOR: "discriminator",
              NET_PIXEL_ACTOR: "pixel actor", NET_PIXEL_CRITIC: "pixel critic"}


class Geom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_params", "wf_floats", "wb_floats", "gpad_floats", "k0", "n3", "n_out", "ld3")]


class Net(C.Structure):
    _fields_ = [("wf", C.c_void_p), ("wb", C.c_void_p)]


class Acts(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x0", "h1", "h2", "h3", "dz1", "dz2", "dz3", "dz4")]


class Mat(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ld", C.c_int32)]


class Env(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("theta", "ends", "goal", "grab", "got")]


_i32, _fp = C.c_int32, C.c_void_p


class CriticStep(C.Structure):
    _fields_ = [("s_dim", _i32), ("a_dim", _i32), ("n_rows", _i32),
                ("critic_eval", Net), ("critic_target", Net), ("actor_target", Net), ("actor_eval", Net),
                ("s", Mat), ("a", Mat), ("r", Mat), ("s_next", Mat),
                ("gamma", C.c_float), ("inv_global_batch", C.c_float),
                ("acts_critic", Acts), ("acts_actor", Acts), ("loss_parts", _fp), ("row_idx", _fp),
                ("counter", _fp)]


class ActorStep(C.Structure):
    _fields_ = [("s_dim", _i32), ("a_dim", _i32), ("n_rows", _i32), ("actor_eval", Net), ("critic_eval", Net),
                ("s", Mat), ("acts_actor", Acts), ("row_idx", _fp), ("counter", _fp)]


class DiscStep(C.Structure):
    _fields_ = [("s_dim", _i32), ("a_dim", _i32), ("n_rows", _i32), ("disc", Net),
                ("s", Mat), ("g_stored", Mat), ("a", Mat), ("s_next", Mat),
                ("label_real", C.c_float), ("inv_global_batch", C.c_float), ("acts_disc", Acts), ("loss_parts", _fp),
                ("row_idx", _fp), ("counter", _fp)]


class GenStep(C.Structure):
    _fields_ = [("s_dim", _i32), ("a_dim", _i32), ("n_rows", _i32), ("gen", Net), ("disc", Net),
                ("s", Mat), ("a", Mat), ("inv_global_batch", C.c_float), ("acts_gen", Acts), ("loss_parts", _fp),
                ("row_idx", _fp), ("counter", _fp)]


class PixelCriticStep(C.Structure):
    _fields_ = [("a_pad", _i32), ("a_cols", _i32), ("n_rows", _i32),
                ("critic_eval", Net), ("critic_target", Net), ("actor_target", Net),
                ("phis", Mat), ("phis_next", Mat), ("a", Mat), ("r", _fp), ("gamma", C.c_float),
                ("acts_critic", Acts), ("loss_parts", _fp), ("counter", _fp)]


class PixelActorStep(C.Structure):
    _fields_ = [("a_pad", _i32), ("n_rows", _i32), ("actor_eval", Net), ("critic_eval", Net), ("phis", Mat),
                ("acts_actor", Acts), ("counter", _fp)]


class Eval(C.Structure):
    _fields_ = [("s_dim", _i32), ("a_dim", _i32), ("n_rows", _i32), ("gen", Net), ("disc", Net),
                ("s", Mat), ("g_stored", Mat), ("a", Mat), ("s_next", Mat),
                ("label_real", C.c_float), ("loss_scale", C.c_float), ("loss_parts", _fp), ("loss_out", _fp),
                ("row_idx", _fp), ("x0_disc", _fp), ("x0_gen", _fp)]


class Act(C.Structure):
    _fields_ = [("s_dim", _i32), ("a_dim", _i32), ("n_env", _i32), ("actor_eval", Net), ("gen", Net), ("disc", Net),
                ("env", Env), ("state_f32", _fp), ("noise", _fp), ("var", _fp), ("ita", C.c_float),
                ("replay", _fp), ("capacity", _i32), ("dims", _i32), ("ld_replay", _i32),
                ("pointer", C.c_int64), ("step", C.c_int64), ("counters", _fp),
                ("reward_out", _fp), ("done_out", _fp), ("rng_seed", C.c_uint64),
                ("ep_t", _fp), ("max_ep_steps", _i32), ("mode_hard", _i32), ("reset_u", _fp), ("n_resets", _fp),
                ("var_decay", C.c_double), ("var_min", C.c_double), ("counters_out", _fp), ("var_out", _fp)]


class Conv(C.Structure):
    _fields_ = [(n, _i32) for n in ("h", "w", "c_in", "c_out", "k", "stride")]


class ConvGeom(C.Structure):
    _fields_ = [(n, _i32) for n in ("h_out", "w_out", "k_len", "n_params", "wfrag_floats", "slab_floats", "wdx_floats")]


class Deconv(C.Structure):
    _fields_ = [(n, _i32) for n in ("h", "w", "c_in", "c_out", "k", "stride")]


class DeconvGeom(C.Structure):
    _fields_ = [(n, _i32) for n in ("h_out", "w_out", "kk", "n_params", "wmat_floats", "colsum_floats")]


class DeconvFusedGeom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "wff_floats", "wfb_floats", "kernel_floats", "max_chunks", "bias_part_floats_per_image")]


class Sample(C.Structure):
    _fields_ = [("mt_state", _fp), ("n_pop", C.c_int64), ("n", _i32), ("out_idx", _fp)]


class WgradJob(C.Structure):
    _fields_ = [("kind", _i32), ("rows", _i32), ("acts", Acts), ("gslabs", _fp)]


class ReduceJob(C.Structure):
    _fields_ = [("gslabs", _fp), ("n_chunks", _i32), ("gpad_floats", _i32), ("map_g", _fp), ("n_params", _i32),
                ("grad", _fp), ("loss_parts", _fp), ("n_parts", _i32), ("loss_scale", C.c_float),
                ("loss_out", _fp), ("counter", _fp)]


class ApplyJob(C.Structure):
    _fields_ = [("opt", _i32), ("n_params", _i32), ("canon", _fp), ("grad", _fp), ("slot0", _fp), ("slot1", _fp),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_float),
                ("step", _fp), ("map_f", _fp), ("wf", _fp), ("map_b", _fp), ("wb", _fp),
                ("target_canon", _fp), ("target_wf", _fp), ("tau", C.c_float), ("counter", _fp), ("period", _i32),
                ("reduce", C.POINTER(ReduceJob))]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `bash rl-gan_amd/csrc/build.sh` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the RL-GAN hot path.")
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _f, _d, _i64, _u32, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64, C.c_uint32, C.c_uint64

# name -> argtypes; every function returns int except the two noted below.
PROTOTYPES = {
    "rlgan_abi_version": [],
    "rlgan_geometry": [_i, _i, _i, C.POINTER(Geom)],
    "rlgan_build_maps": [_i, _i, _i, _vp, _vp, _vp],
    "rlgan_wgrad_chunks": [_i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "rlgan_pack_weights": [_vp, _i, _vp, _vp, _vp, _vp, _vp],
    "rlgan_mlp_forward": [_i, _i, _i, Net, Mat, Mat, Mat, _vp, _i, _i, _i, _vp],
    "rlgan_critic_step_rows": [C.POINTER(CriticStep), _vp],
    "rlgan_actor_step_rows": [C.POINTER(ActorStep), _vp],
    "rlgan_disc_step_rows": [C.POINTER(DiscStep), _vp],
    "rlgan_gen_step_rows": [C.POINTER(GenStep), _vp],
    "rlgan_pixel_critic_step_rows": [C.POINTER(PixelCriticStep), _vp],
    "rlgan_pixel_actor_step_rows": [C.POINTER(PixelActorStep), _vp],
    "rlgan_phase1_rows": [C.POINTER(CriticStep), C.POINTER(DiscStep), _vp],
    "rlgan_phase2_rows": [C.POINTER(ActorStep), C.POINTER(GenStep), _vp],
    "rlgan_gan_eval_rows": [C.POINTER(Eval), _vp, _vp],
    "rlgan_rollout_eval_rows": [C.POINTER(Act), C.POINTER(Eval), C.POINTER(Sample), _vp, _vp],
    "rlgan_weight_grads": [C.POINTER(WgradJob), _i, _i, _i, _vp],
    "rlgan_grad_reduce": [C.POINTER(ReduceJob), _i, _vp],
    "rlgan_apply": [C.POINTER(ApplyJob), _i, _vp],
    "rlgan_sum_parts": [_vp, _i, _i, _f, _vp, _vp],
    "rlgan_target_update": [_vp, _vp, _i, _f, _vp, _i, _vp, _vp, _vp],
    "rlgan_env_step": [Env, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "rlgan_env_reset": [Env, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "rlgan_env_autoreset": [Env, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "rlgan_replay_store": [_vp, _i, _i, _i, _i64, _vp, _i, _i, _vp],
    "rlgan_replay_gather": [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp],
    "rlgan_mt19937_seed": [_vp, _u32, _vp],
    "rlgan_mt19937_choice": [_vp, _i64, _i, _vp, _vp],
    "rlgan_philox_normal": [_vp, _i, _u64, _u32, _u64, _vp, _i64, _vp],
    "rlgan_philox_uniform": [_vp, _i, _u64, _u32, _u64, _vp, _i64, _vp],
    "rlgan_act_rows": [C.POINTER(Act), _vp, _vp],
    "rlgan_conv_geometry": [C.POINTER(Conv), C.POINTER(ConvGeom)],
    "rlgan_conv_build_map": [C.POINTER(Conv), _vp],
    "rlgan_conv_build_dx_map": [C.POINTER(Conv), _vp],
    "rlgan_conv_forward": [C.POINTER(Conv), _vp, _vp, _vp, _i, _i, _vp],
    "rlgan_elu_backward": [_vp, _vp, _vp, _i64, _vp],
    "rlgan_conv_backward_input": [C.POINTER(Conv), _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "rlgan_conv_backward_input_dz": [C.POINTER(Conv), _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "rlgan_conv_backward_weights": [C.POINTER(Conv), _vp, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "rlgan_deconv_geometry": [C.POINTER(Deconv), C.POINTER(DeconvGeom)],
    "rlgan_deconv_build_map": [C.POINTER(Deconv), _vp],
    "rlgan_deconv_place": [C.POINTER(Deconv), _vp, _vp, _vp, _i, _i, _vp],
    "rlgan_deconv_gather": [C.POINTER(Deconv), _vp, _vp, _vp, _i, _vp],
    "rlgan_deconv_view_ok": [C.POINTER(Deconv), _i],
    "rlgan_deconv_backward_input_view": [C.POINTER(Deconv), _vp, _vp, _vp, _i, _vp, _i, _vp],
    "rlgan_deconv_backward_kernel_view": [C.POINTER(Deconv), _vp, _vp, _vp, _i, _vp, _i, _vp],
    "rlgan_deconv_fused_geometry": [C.POINTER(Deconv), C.POINTER(DeconvFusedGeom)],
    "rlgan_deconv_fused_build_maps": [C.POINTER(Deconv), _vp, _vp],
    "rlgan_deconv_fused_forward": [C.POINTER(Deconv), _vp, _vp, _vp, _vp, _i, _i, _vp],
    "rlgan_deconv_fused_backward": [C.POINTER(Deconv), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "rlgan_deconv_fused_backward_dz": [C.POINTER(Deconv), _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "rlgan_column_sums": [_vp, _vp, C.c_int64, _i, _vp, _vp, _vp],
    "rlgan_column_sums_noalloc": [_vp, _vp, C.c_int64, _i, _vp, _vp],
    "rlgan_noalloc_init": [],
    "rlgan_gather_map": [_vp, _vp, _i, _vp, _vp],
    "rlgan_relu_backward": [_vp, _vp, _vp, C.c_int64, _vp],
    "rlgan_copy_cols": [_vp, _i, _vp, _i, C.c_int64, _i, _i, _vp],
    "rlgan_concat_cols": [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _i, C.c_int64, _i, _vp],
    "rlgan_blocks2x2": [_vp, _vp, C.c_int64, _i, _i, _i, _vp],
    "rlgan_add_cols": [_vp, _i, _vp, _i, C.c_int64, _i, _vp],
    "rlgan_sigmoid_ce": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "rlgan_one_minus_sigmoid": [_vp, _i, _i, _vp, _vp],
    "rlgan_batchnorm_forward": [_vp, _i, _vp, _vp, _i, _i, _f, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "rlgan_batchnorm_chunks": [_i],
    "rlgan_batchnorm_backward": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "rlgan_act_backward": [_vp, _vp, _vp, C.c_int64, _i, _vp],
    "rlgan_td_error": [_vp, _vp, _i, _vp, _f, _i, _vp, _vp, _vp],
    "rlgan_lstm_gates": [_vp, _vp, _i, _i, _f, _vp, _vp, _vp],
    "rlgan_softmax_ce": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp],
    "rlgan_mse": [_vp, _vp, C.c_int64, _vp, _vp, _vp],
    "rlgan_conv_forward_u8": [C.POINTER(Conv), _vp, C.c_int64, _vp, _f, _vp, _vp, _i, _i, _vp],
    "rlgan_conv_backward_weights_u8": [C.POINTER(Conv), _vp, C.c_int64, _vp, _f, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "rlgan_frames_store": [_vp, C.c_int64, _i, _vp, _i, _vp, _vp],
    "rlgan_frames_gather": [_vp, C.c_int64, _i, _vp, _i, _f, _vp, _vp],
    "rlgan_frames_gather_channel": [_vp, C.c_int64, _i, _i, _i, _vp, _i, _f, _vp, _vp],
    "rlgan_dense_slabs": [_i, _i, _i],
    "rlgan_dense_forward": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "rlgan_dense_backward_input": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "rlgan_dense_backward_input_split": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "rlgan_dense_backward_weights": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "rlgan_dense_backward_kernel": [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _i, _vp],
}

for _name, _args in PROTOTYPES.items():
    _fn = getattr(lib, _name)          # AttributeError here = the .so is older than this binding
    _fn.argtypes = _args
    _fn.restype = C.c_int
lib.rlgan_deconv_fused_workspace.argtypes = [C.POINTER(Deconv), C.c_int]
lib.rlgan_deconv_fused_workspace.restype = C.c_int64
lib.rlgan_error_string.argtypes = [C.c_int]
lib.rlgan_error_string.restype = C.c_char_p


class RlganError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.rlgan_error_string(rc).decode()
        raise RlganError(f"{what or 'librlgan_amd'} failed with status {rc}: {msg}")


def geometry(kind: int, s_dim: int, a_dim: int) -> Geom:
    g = Geom()
    check(lib.rlgan_geometry(kind, s_dim, a_dim, C.byref(g)), "rlgan_geometry")
    return g


def wgrad_chunks(rows: int):
    cr, nc = C.c_int32(), C.c_int32()
    check(lib.rlgan_wgrad_chunks(rows, C.byref(cr), C.byref(nc)), "rlgan_wgrad_chunks")
    return cr.value, nc.value
