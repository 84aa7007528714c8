"""The composed restatement of the pixel variant's learning iteration (oracle/pixel_iteration.py,
Curious_GAN_mario.py:86-98) against its own step functions chained by hand: which variables each step moves, which
optimizer owns which slots, and what each step sees of the steps before it."""
import numpy as np

from pixel_util import oracle_iteration, pixel_params, pixel_transitions


def _same(x, y):
    if isinstance(x, dict):
        return all(_same(x[k], y[k]) for k in x)
    if isinstance(x, (list, tuple)):
        return all(_same(a, b) for a, b in zip(x, y))
    return np.array_equal(x, y)


def test_iteration_is_the_chain_of_its_steps():
    from oracle import inverse_model as om, pixel_ddpg as od, pixel_gan as op
    from oracle.optim import Adam, RMSProp
    rng = np.random.RandomState(0)
    a_dim, n = 6, 2
    P = pixel_params(rng, a_dim)
    s8, sn8, a, rew = pixel_transitions(rng, n, a_dim)
    s, s_next, r = s8 / 255.0, sn8 / 255.0, rew[:, 1:2]
    it = oracle_iteration(P, period_c=2, period_a=3)
    out = it.iteration(s, a, r, s_next)
    lay = op.oc.universe_head(2)
    last = s_next[:, :, :, -1:]

    # generator step: on the initial encoders; nothing but the generator moved before the discriminator's step
    phis0, _ = op.encoder(s, P["st_params"], P["st_dense"], lay)
    G = P["gen"]
    loss, _, g = op.generator_loss_and_grads(phis0, a, last, G["w1_s"], G["w1_a"], G["b1"], G["decoder"], P["enc_params"], P["enc_dense"])
    assert out["G_loss"] == loss
    adam = lambda w, dw, lr=1e-4: Adam(w.size, lr).step(w.ravel(), np.asarray(dw).ravel()).reshape(w.shape)
    assert np.array_equal(it.gen["w1_s"], adam(G["w1_s"], g["w1_s"]))
    assert np.array_equal(it.gen["decoder"][2][0], adam(G["decoder"][2][0], g["decoder"][2][0]))

    # discriminator step: the image comes from the UPDATED generator and the features of s from the initial encoder
    assert np.array_equal(out["image"], it.predict_batch(phis0, a))
    lr_, lf_, gd = op.discriminator_loss_and_grads(s, a, last, out["image"], P["disc"], P["st_params"], P["st_dense"], P["enc_params"],
                                                   P["enc_dense"])
    assert (out["D_loss_real"], out["D_loss_fake"]) == (lr_, lf_)
    assert np.array_equal(it.disc["w3"], adam(P["disc"]["w3"], gd["w3"]))
    # the frame encoder is trained by the discriminator's optimizer alone
    w_ob = adam(P["enc_params"][0][0], gd["obs_encoder"][0][0])
    assert np.array_equal(it.enc_params[0][0], w_ob) and w_ob.shape == (3, 3, 1, 32)
    assert np.array_equal(it.enc_dense[1], adam(P["enc_dense"][1], gd["obs_encoder"][2][1]))

    # the state encoder: D's Adam first, then A_pred's Adam from slots of its own (both at t = 1), the second on the
    # encoder the first left
    st_after_d = [(adam(w, dw), adam(b, db)) for (w, b), (dw, db) in zip(P["st_params"], gd["state_encoder"][:2])]
    sd_after_d = (adam(P["st_dense"][0], gd["state_encoder"][2][0]), adam(P["st_dense"][1], gd["state_encoder"][2][1]))
    la, ga = om.loss_and_grads(s, s_next, a, P["a_pred"], st_after_d, sd_after_d)
    assert out["A_pred_loss"] == la
    assert np.array_equal(it.st_params[1][0], adam(st_after_d[1][0], ga["state_encoder"][1][0]))
    assert np.array_equal(it.st_dense[0], adam(sd_after_d[0], ga["state_encoder"][2][0]))
    assert np.array_equal(it.a_pred["g2"], adam(P["a_pred"]["g2"], ga["g2"]))
    d_slot, p_slot = it.opt_d.slots["st.conv0.w"], it.opt_a_pred.slots["st.conv0.w"]
    assert d_slot is not p_slot and d_slot.t == p_slot.t == 1 and not np.array_equal(d_slot.m, p_slot.m)
    assert "ob.conv0.w" in it.opt_d.slots and "ob.conv0.w" not in it.opt_a_pred.slots and "ob.conv0.w" not in it.opt_g.slots

    # critic and actor: features from the encoder as both steps left it; the actor sees the critic of this iteration
    phis, _ = op.encoder(s, it.st_params, it.st_dense, lay)
    phis_n, _ = op.encoder(s_next, it.st_params, it.st_dense, lay)
    lc, gc = od.critic_loss_and_grads(phis, a, r, phis_n, P["critic"], P["critic_t"], P["actor_t"], 0.99)
    assert out["critic_loss"] == lc
    rms = lambda w, dw, lr: RMSProp(w.size, lr).step(w.ravel(), np.asarray(dw).ravel()).reshape(w.shape)
    c_new = {k: rms(P["critic"][k], gc[k], 1e-4) for k in P["critic"]}
    assert _same(it.critic, c_new) and _same(it.critic_t, c_new)              # counter 0: the copy follows the first step
    _, _, gp = od.actor_policy_grads(phis, P["actor"], c_new)
    assert _same(it.actor, {k: rms(P["actor"][k], gp[k], -1e-4) for k in P["actor"]}) and _same(it.actor_t, it.actor)

    # second iteration: counters 1 and 1 -- neither target follows (periods 2 and 3); third: the critic's does
    it.iteration(s, a, r, s_next)
    assert _same(it.critic_t, c_new) and not _same(it.critic, c_new) and not _same(it.actor_t, it.actor)
    assert d_slot.t == p_slot.t == 2
    it.iteration(s, a, r, s_next)
    assert _same(it.critic_t, it.critic) and not _same(it.actor_t, it.actor)
    assert (it.counter_c, it.counter_a) == (3, 3)
