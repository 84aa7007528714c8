"""Parameters and minibatches for tests of the pixel variant's whole learning iteration: one set of float32-exact
arrays shaped like the reference's variables, handed to the float64 restatement and to the device classes alike."""
import numpy as np


def f32(v):
    return np.asarray(v).astype(np.float32).astype(np.float64)


def pixel_params(rng, a_dim):
    """Every variable of the pixel learner, spread widely enough that no unit sits at an activation's kink."""
    from oracle import deconv as odc
    from test_oracle_pixel_ddpg import params as ddpg_params
    u = lambda lim, shape: f32(rng.uniform(-lim, lim, shape))
    feat = 9 * 9 * 32                                       # 42 -> 20 -> 9
    P = {}
    P["st_params"] = [(u(0.2, (3, 3, 4, 32)), u(0.1, 32)), (u(0.1, (3, 3, 32, 32)), u(0.1, 32))]
    P["st_dense"] = (u(0.03, (feat, 256)), f32(np.full(256, 0.001)))
    P["enc_params"] = [(u(0.3, (3, 3, 1, 32)), u(0.1, 32)), (u(0.1, (3, 3, 32, 32)), u(0.1, 32))]
    P["enc_dense"] = (u(0.03, (feat, 256)), f32(np.full(256, 0.001)))
    P["gen"] = dict(w1_s=u(0.1, (256, 200)), w1_a=u(0.2, (a_dim, 200)), b1=f32(np.full(200, 0.001)),
                    decoder=[(f32(w), f32(b)) for w, b in odc.init_decoder(rng, 200, bias=0.0)])
    P["disc"] = dict(w1_G=u(0.1, (256, 200)), w1_s=u(0.1, (256, 200)), w1_a=u(0.2, (a_dim, 200)), b1=f32(np.full(200, 0.001)),
                     w3=u(0.2, (200, 10)), b3=f32(np.full(10, 0.001)), w2=u(0.5, (10, 1)), b2=f32(np.full(1, 0.001)))
    P["a_pred"] = dict(w1_s=u(0.1, (256, 500)), w1_s_=u(0.1, (256, 500)), b1=f32(np.full(500, 0.001)), g1=u(0.5, 500) + 1.0,
                       be1=u(0.5, 500) + 1.0, w2=u(0.1, (500, 100)), b2=f32(np.full(100, 0.001)), g2=u(0.5, 100) + 1.0,
                       be2=u(0.5, 100) + 1.0, w3=u(0.2, (100, a_dim)), b3=f32(np.full(a_dim, 0.001)), g3=u(0.5, a_dim) + 1.0,
                       be3=u(0.5, a_dim))
    for name in ("", "_t"):
        A, C = ddpg_params(rng, 256, a_dim, spread=0.5)
        P["actor" + name], P["critic" + name] = {k: f32(v) for k, v in A.items()}, {k: f32(v) for k, v in C.items()}
    return P


def pixel_transitions(rng, n, a_dim):
    """n transitions as the replay holds them: uint8 frame stacks s and s', one-hot actions, and the two reward
    columns (r, curious_r)."""
    s = rng.randint(0, 256, (n, 42, 42, 4)).astype(np.uint8)
    s_next = rng.randint(0, 256, (n, 42, 42, 4)).astype(np.uint8)
    a = np.eye(a_dim)[rng.randint(0, a_dim, n)]
    rewards = f32(rng.uniform(0, 1, (n, 2)))
    return s, s_next, a, rewards


def oracle_iteration(P, **kw):
    from oracle.pixel_iteration import PixelIteration
    return PixelIteration(P["st_params"], P["st_dense"], P["enc_params"], P["enc_dense"], P["gen"], P["disc"], P["a_pred"],
                          P["actor"], P["actor_t"], P["critic"], P["critic_t"], **kw)
