"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): actor and critic of the pixel variant.

Restates, in float64 NumPy, `Actor` and `Critic` of Conv_model.py:569-784 as Curious_GAN_mario.py:319-325 wires
them with USE_BATCH_NORM = False (Curious_GAN_mario.py:384) and USE_LSTM = False (Conv_model.py:667,701-706):

    a = tanh(dense(relu6(dense10(relu6(dense200(relu6(dense200(phis))))))))                    Conv_model.py:711-744
    q = dense1(relu6(dense10(relu6(dense200(relu6(phis w1_s + a w1_a + b1))))))                Conv_model.py:621-651
    critic: loss = mean((R + gamma q_target(phis', a_target(phis')) - q(phis, a))^2), RMSProp(lr) on the critic's
            eval variables only (var_list = e_params, Conv_model.py:610-616)
    actor:  policy_grads = sum_b da/dtheta * dq/da, RMSProp(-lr) on the actor's eval dense variables
            (Conv_model.py:619,778-784; the LSTM variables under the same scope get no gradient and are skipped)

phis and phis' are the state encoder's features of s and s'; neither step trains the encoder (it is in neither
variable list).  Parity unpinned against TensorFlow (not installed; no fixture in the reference): checked against
torch autograd in tests/test_oracle_pixel_ddpg.py."""
import numpy as np


def relu6(x):
    return np.clip(x, 0, 6)


def _mask(z):
    return (z > 0) & (z < 6)


def actor_forward(phis, A):
    z1 = phis @ A["w1"] + A["b1"]; h1 = relu6(z1)
    z2 = h1 @ A["w2"] + A["b2"]; h2 = relu6(z2)
    z3 = h2 @ A["w3"] + A["b3"]; h3 = relu6(z3)
    a = np.tanh(h3 @ A["wa"] + A["ba"])
    return a, (phis, z1, h1, z2, h2, z3, h3, a)


def actor_backward(da, A, cache):
    phis, z1, h1, z2, h2, z3, h3, a = cache
    dz4 = da * (1.0 - a * a)
    G = {"wa": h3.T @ dz4, "ba": dz4.sum(0)}
    dz3 = (dz4 @ A["wa"].T) * _mask(z3)
    G["w3"], G["b3"] = h2.T @ dz3, dz3.sum(0)
    dz2 = (dz3 @ A["w3"].T) * _mask(z2)
    G["w2"], G["b2"] = h1.T @ dz2, dz2.sum(0)
    dz1 = (dz2 @ A["w2"].T) * _mask(z1)
    G["w1"], G["b1"] = phis.T @ dz1, dz1.sum(0)
    return G


def critic_forward(phis, a, C):
    z1 = phis @ C["w1_s"] + a @ C["w1_a"] + C["b1"]; h1 = relu6(z1)
    z2 = h1 @ C["w2"] + C["b2"]; h2 = relu6(z2)
    z3 = h2 @ C["w3"] + C["b3"]; h3 = relu6(z3)
    return h3 @ C["wq"] + C["bq"], (phis, a, z1, h1, z2, h2, z3, h3)


def critic_backward(dq, C, cache):
    """Returns (da, grads)."""
    phis, a, z1, h1, z2, h2, z3, h3 = cache
    G = {"wq": h3.T @ dq, "bq": dq.sum(0)}
    dz3 = (dq @ C["wq"].T) * _mask(z3)
    G["w3"], G["b3"] = h2.T @ dz3, dz3.sum(0)
    dz2 = (dz3 @ C["w3"].T) * _mask(z2)
    G["w2"], G["b2"] = h1.T @ dz2, dz2.sum(0)
    dz1 = (dz2 @ C["w2"].T) * _mask(z1)
    G.update(w1_s=phis.T @ dz1, w1_a=a.T @ dz1, b1=dz1.sum(0))
    return dz1 @ C["w1_a"].T, G


def critic_loss_and_grads(phis, a, r, phis_next, C_eval, C_target, A_target, gamma):
    """Conv_model.py:599-616,653-660.  r: [n, 1]."""
    a_next, _ = actor_forward(phis_next, A_target)
    q_next, _ = critic_forward(phis_next, a_next, C_target)
    target = r + gamma * q_next
    q, cache = critic_forward(phis, a, C_eval)
    loss = np.mean((target - q) ** 2)
    _, G = critic_backward(2.0 * (q - target) / q.shape[0], C_eval, cache)
    return loss, G


def actor_policy_grads(phis, A_eval, C_eval):
    """Conv_model.py:619,778-780: gradients of the actions weighted by dq/da, summed over the batch."""
    a, acache = actor_forward(phis, A_eval)
    q, ccache = critic_forward(phis, a, C_eval)
    da, _ = critic_backward(np.ones_like(q), C_eval, ccache)
    return a, da, actor_backward(da, A_eval, acache)


def lstm_step(x, c, h, kernel, bias, forget_bias=1.0):
    """One step of tf.contrib.rnn.BasicLSTMCell (Conv_model.py:80-91): gates = [x, h] kernel + bias in the order
    i, j, f, o; c' = c sigmoid(f + forget_bias) + sigmoid(i) tanh(j); h' = tanh(c') sigmoid(o).  Returns (c', h')."""
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    i, j, f, o = np.split(np.concatenate([x, h], 1) @ kernel + bias, 4, axis=1)
    c2 = c * sig(f + forget_bias) + sig(i) * np.tanh(j)
    return c2, np.tanh(c2) * sig(o)
