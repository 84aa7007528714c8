"""The pixel variant's whole learning iteration on the device (`PixelGanTrainer`, the calls of
Curious_GAN_mario.py:86-98 on one minibatch from the uint8 replay) against the composed float64 restatement
(oracle/pixel_iteration.py), free-running for three iterations from the same variables: every step's loss, and at
the end every variable of the seven networks, the targets and the counters."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

A_DIM = 6


def _device_learner(P, n_slots, batch, period_c, period_a):
    import rl_gan_amd as R
    st = R.ConvEncoder(42, 42, 4, R.universe_head(2), params=P["st_params"], dense=256, dense_params=P["st_dense"])
    ob = R.single_frame_encoder(42, 2, params=P["enc_params"], dense_params=P["enc_dense"])
    g = P["gen"]
    G = R.PixelGenerator(256, A_DIM, 1e-4, ob, w1_s=g["w1_s"], w1_a=g["w1_a"], b1=g["b1"], decoder_params=g["decoder"])
    D = R.PixelDiscriminator(A_DIM, 1e-4, st, ob, params=P["disc"])
    a_pred = R.StateActionPredictor(A_DIM, 1e-4, st, params=P["a_pred"])
    actor = R.PixelActor(A_DIM, 1e-4, period_a, st, params=P["actor"], target_params=P["actor_t"])
    critic = R.PixelCritic(A_DIM, 1e-4, 0.99, period_c, actor, st, params=P["critic"], target_params=P["critic_t"])
    frames = R.FrameMemory(n_slots, (42, 42, 4), rng=R.MT19937Device(1))
    actions = R.DeviceMemory(n_slots + 1, A_DIM + 2)
    return R.PixelGanTrainer(frames, actions, G, D, batch, a_pred=a_pred, actor=actor, critic=critic)


def _kb(layer, k=None, n=None):
    """(kernel, bias) of a dense layer cut back to the reference's k x n."""
    kern, bias = layer.kernel_bias()
    k, n = k or kern.shape[0], n or kern.shape[1]
    assert not kern[k:].any() and not kern[:, n:].any() and not bias[n:].any()
    return kern[:k, :n], bias[:n]


def _encoder_vars(enc, prefix, single=False):
    out = {}
    for i, L in enumerate(enc.layers):
        kern, bias = L.kernel_bias()
        if single and i == 0:                                # run as a 2x2 kernel over 2x2 blocks: the seven extra entries stay zero
            assert kern.shape == (3, 3, 1, 32) and not L.canon.cpu().numpy()[:16 * 32].reshape(2, 2, 4, 32)[~L._real].any()
        out[f"{prefix}.conv{i}.w"], out[f"{prefix}.conv{i}.b"] = kern, bias
    out[prefix + ".dense.w"], out[prefix + ".dense.b"] = enc.head.kernel_bias()
    return out


def _device_vars(tr):
    """Every variable of the learner by the names the restatement's optimizers use for their slots."""
    V = {}
    V.update(_encoder_vars(tr.D.state_encoder, "st"))
    V.update(_encoder_vars(tr.D.obs_encoder, "ob", single=True))
    kern, bias = tr.G.l1.kernel_bias()
    assert not kern[256 + A_DIM:].any()
    V["G.w1_s"], V["G.w1_a"], V["G.b1"] = kern[:256], kern[256:256 + A_DIM], bias
    for i, L in enumerate(tr.G.decoder.layers):
        V[f"G.dec{i}.w"], V[f"G.dec{i}.b"] = L.kernel_bias()
    kern, bias = tr.D.l1.kernel_bias()
    assert not kern[512 + A_DIM:].any()
    V["D.w1_s"], V["D.w1_G"], V["D.w1_a"], V["D.b1"] = kern[:256], kern[256:512], kern[512:512 + A_DIM], bias
    V["D.w3"], V["D.b3"] = _kb(tr.D.l3, 200, 10)
    V["D.w2"], V["D.b2"] = _kb(tr.D.l2, 10, 1)
    M = tr.a_pred
    kern, bias = M.l1.kernel_bias()
    V["A_pred.w1_s"], V["A_pred.w1_s_"], V["A_pred.b1"] = kern[:256], kern[256:], bias
    V["A_pred.w2"], V["A_pred.b2"] = M.l2.kernel_bias()
    V["A_pred.w3"], V["A_pred.b3"] = _kb(M.l3, 100, A_DIM)
    for i, bn in ((1, M.bn1), (2, M.bn2), (3, M.bn3)):
        c = bn.canon.cpu().numpy()
        n = A_DIM if i == 3 else bn.n
        V[f"A_pred.g{i}"], V[f"A_pred.be{i}"] = c[:bn.n][:n], c[bn.n:][:n]
    for tag, net in (("A", tr.actor.eval_net), ("At", tr.actor.target_net)):
        for L, (kn, bn_), (k, n) in zip(net.layers, (("w1", "b1"), ("w2", "b2"), ("w3", "b3"), ("wa", "ba")),
                                        ((256, 200), (200, 200), (200, 10), (10, A_DIM))):
            V[f"{tag}.{kn}"], V[f"{tag}.{bn_}"] = _kb(L, k, n)
    for tag, net in (("C", tr.critic.eval_net), ("Ct", tr.critic.target_net)):
        kern, bias = net.layers[0].kernel_bias()
        assert not kern[256 + A_DIM:].any()
        V[f"{tag}.w1_s"], V[f"{tag}.w1_a"], V[f"{tag}.b1"] = kern[:256], kern[256:256 + A_DIM], bias
        for L, (kn, bn_), (k, n) in zip(net.layers[1:], (("w2", "b2"), ("w3", "b3"), ("wq", "bq")), ((200, 200), (200, 10), (10, 1))):
            V[f"{tag}.{kn}"], V[f"{tag}.{bn_}"] = _kb(L, k, n)
    return V


def _oracle_vars(it):
    V = {}
    for prefix, params, dense in (("st", it.st_params, it.st_dense), ("ob", it.enc_params, it.enc_dense)):
        for i, (w, b) in enumerate(params):
            V[f"{prefix}.conv{i}.w"], V[f"{prefix}.conv{i}.b"] = w, b
        V[prefix + ".dense.w"], V[prefix + ".dense.b"] = dense
    for k in ("w1_s", "w1_a", "b1"):
        V["G." + k] = it.gen[k]
    for i, (w, b) in enumerate(it.gen["decoder"]):
        V[f"G.dec{i}.w"], V[f"G.dec{i}.b"] = w, b
    for tag, d in (("D", it.disc), ("A_pred", it.a_pred), ("A", it.actor), ("At", it.actor_t), ("C", it.critic), ("Ct", it.critic_t)):
        for k, v in d.items():
            V[f"{tag}.{k}"] = v
    return V


def _check_variables(tr, it, steps, drift):
    """Every variable against the restatement's.  Adam-trained ones: plus what gradient errors within 1e-5 may do to an
    entry over the steps of every optimizer that trains it (oracle/optim.py, Adam.allow; the state encoder has two
    optimizers).  drift: further allowance per variable, as a fraction of the distance an optimizer can move an entry in
    `steps` steps of 1e-4 (0: none, the plain bar)."""
    from gpu_util import close
    got, ref = _device_vars(tr), _oracle_vars(it)
    assert set(got) == set(ref)
    opts = (it.opt_g, it.opt_d, it.opt_a_pred)
    for name in sorted(ref):
        allow = sum((o.slots[name].allow.reshape(ref[name].shape) for o in opts if name in o.slots), np.zeros(ref[name].shape))
        if name in ("A_pred.b1", "A_pred.b2", "A_pred.b3"):
            # A bias in front of a batch normalisation: the batch mean takes it out again, so its gradient is exactly zero
            # and no output depends on it.  In float64 the sums leave 1e-18, far below Adam's epsilon: no step.  In float32
            # they leave 1e-9..1e-8, the size of epsilon: a fraction of lr per step, in a direction the rounding decides
            # (measured: 1.05e-4 after three steps; TensorFlow's float32 graph is in the same position).
            allow = allow + steps * 1e-4
        g, what = got[name].reshape(ref[name].shape), f"{name} after {steps} iteration(s)"
        if drift == 0.0:
            close(g, ref[name], atol=allow, what=what)
            continue
        # with drift: at most one entry in 10,000 may be further out than the allowance (an entry whose gradient has become
        # rounding noise on one side takes steps of its own), and none further than Adam can carry it: 2 lr per step
        err = np.abs(g - ref[name])
        out = err > allow + drift * steps * 1e-4 + 1e-5 * (np.abs(ref[name]) + np.abs(ref[name]).max())
        assert out.sum() <= max(1, out.size // 10000), (what, int(out.sum()), out.size, float(err.max()))
        close(g, ref[name], atol=allow + 2 * steps * 1e-4, what=what)
    return got


def test_three_iterations_track_the_restatement():
    from pixel_util import oracle_iteration, pixel_params, pixel_transitions
    rng = np.random.RandomState(21)
    n_slots, batch = 12, 6
    P = pixel_params(rng, A_DIM)
    s8, sn8, a, rew = pixel_transitions(rng, n_slots, A_DIM)
    tr = _device_learner(P, n_slots, batch, period_c=2, period_a=3)
    tr.frames.store_transitions(s8, sn8)
    tr.actions.store_rows(torch.from_numpy(np.concatenate([a, rew], 1).astype(np.float32)).cuda())
    it = oracle_iteration(P, period_c=2, period_a=3)
    worst = {}
    for k in range(3):
        idx = rng.randint(0, n_slots, batch)                 # with repeats, like np.random.choice(len, n)
        tr.idx.copy_(torch.from_numpy(idx))
        tr.step()
        want = it.iteration(s8[idx] / 255.0, a[idx], rew[idx, 1:2], sn8[idx] / 255.0)
        d_real, d_fake = tr.D.losses.tolist()
        got = {"G_loss": tr.G.loss_out.item(), "D_loss_real": d_real, "D_loss_fake": d_fake, "A_pred_loss": tr.a_pred.loss_out.item(),
               "critic_loss": tr.critic.loss_out.item()}
        for name, v in got.items():
            err = abs(v - want[name]) / abs(want[name])
            worst[name] = max(worst.get(name, 0.0), err)
            # The generator's and the discriminator's step of iteration 0 start from identical variables: the bar of the
            # single-step tests.  Everything after them starts from variables that already differ where Adam took a step on
            # a gradient entry that is rounding noise (about lr = 1e-4 either way, see oracle/optim.py) -- the inverse model
            # and the critic of iteration 0 read the state encoder after such steps -- so those losses agree to the size of
            # these steps' effect (measured: critic loss of iteration 0 1.5e-5, a squared difference of two q values).
            strict = k == 0 and name in ("G_loss", "D_loss_real", "D_loss_fake")
            assert err <= (1e-5 if strict else 1e-4), (k, name, v, want[name])
        if k == 0:                                           # one iteration from identical variables: no allowance for drift
            _check_variables(tr, it, 1, drift=0.0)
    print("worst relative loss errors over three iterations:", {k: f"{v:.1e}" for k, v in worst.items()})
    # Later iterations: the two sides no longer hold the same variables.  Where a gradient entry of the discriminator's step
    # is rounding noise, Adam moved that entry of the shared encoders by about lr in a direction the rounding decided; the
    # generator's, the inverse model's, the critic's and the actor's losses do depend on those entries, so their gradients
    # differ by a few 1e-4 of their scale from then on (measured: 8.4e-6 on an entry of the generator's b1 after three
    # steps of 1e-4, 3 % of what it moved; 3.6e-5 on one of the 230,400 entries of the decoder's first kernel).  Every
    # variable must still have moved like the restatement's to within 10 % of the distance three steps can cover; an optimizer reading another one's slots, a step on stale features or a
    # missed target copy is off by whole steps.
    got = _check_variables(tr, it, 3, drift=0.1)
    assert int(tr.G.step.item()) == int(tr.D.step.item()) == int(tr.a_pred.step.item()) == 3
    assert int(tr.critic.t_replace_counter.item()) == int(tr.actor.t_replace_counter.item()) == 3
    # counters 0, 1, 2 against periods 2 and 3: the critic's target was copied after the third step, the actor's was not
    assert np.array_equal(got["Ct.w2"], got["C.w2"]) and not np.array_equal(got["At.w2"], got["A.w2"])
