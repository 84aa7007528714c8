"""The pixel variant's steps at the bench's batch of 4096 through a property that does not depend on the size: every
loss of these steps is a mean (or, for the actor, a sum) over the minibatch rows, so the gradients of one step on 4096
transitions equal the mean (sum) of the gradients of eight steps on 512 of them each, taken from the same parameters.

The small steps are the ones the float64 restatement vouches for (tests of a few to a few hundred rows); at 4096 rows
other code runs -- weight-gradient slabs by the thousand and their folding, several batch chunks per workgroup in the
fused decoder kernels, 256 row tiles per network pass, split reductions in the matrix products -- and this ties its
results to theirs.  Steps are taken with a learning rate of 0, which leaves the gradient buffers filled and the
parameters where they were.  (The inverse model normalises over the batch and has no such property; its chunked
statistics are checked against the restatement at 65,536 rows in test_gpu_inverse_model.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, PARTS, A_DIM = 4096, 8, 6


def _grads(layers):
    return [L.grad.cpu().numpy().astype(np.float64) for L in layers]


def _check(full, parts, scale, what):
    from gpu_util import close
    for i, (f, ps) in enumerate(zip(full, parts)):
        close(f, scale * np.sum(ps, axis=0), what=f"{what}, trained tensor {i}")


def test_gan_steps_at_4096_rows_are_the_mean_of_eight_steps_at_512():
    import rl_gan_amd as R
    rng = np.random.RandomState(5)
    st = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
    ob = R.single_frame_encoder(42, 2, rng=rng)
    G = R.PixelGenerator(256, A_DIM, 1e-4, ob, rng=rng)
    D = R.PixelDiscriminator(A_DIM, 1e-4, st, ob, rng=rng)
    g = torch.Generator(device="cuda").manual_seed(3)
    s = torch.randint(0, 256, (N, 42, 42, 4), dtype=torch.uint8, device="cuda", generator=g).float() * (1.0 / 255)
    last = torch.rand((N, 42, 42, 1), device="cuda", generator=g)
    a = torch.nn.functional.one_hot(torch.randint(0, A_DIM, (N,), device="cuda", generator=g), A_DIM).float()
    phis = st.forward(s).clone()
    chunks = [slice(i * N // PARTS, (i + 1) * N // PARTS) for i in range(PARTS)]

    # generator step (Conv_model.py:522-541): mean squared feature error over all rows and features
    g_layers = [o.layer for o in G._opt]
    before = [L.canon.clone() for L in g_layers]
    loss_full = float(G.learn(phis, a, last, lr=0.0).item())
    full = _grads(g_layers)
    img_full = G._img.clone()
    parts, losses = [], []
    for c in chunks:
        losses.append(float(G.learn(phis[c].contiguous(), a[c].contiguous(), last[c].contiguous(), lr=0.0).item()))
        parts.append(_grads(g_layers))
        assert torch.equal(G._img, img_full[c])               # a row's image does not depend on its batch
    _check(full, list(zip(*parts)), 1.0 / PARTS, "generator")
    assert abs(loss_full - np.mean(losses)) <= 1e-5 * loss_full
    assert all(torch.equal(b, L.canon) for b, L in zip(before, g_layers))

    # discriminator step (Conv_model.py:207-223,270-281): two mean cross-entropies; trains both encoders as well
    img = G.predict_batch(phis, a).clone()
    d_layers = [o.layer for o in D._opt]
    l_full = D.learn(img, last, s, a, lr=0.0).cpu().numpy().astype(np.float64)
    full = _grads(d_layers)
    parts, losses = [], []
    for c in chunks:
        losses.append(D.learn(img[c].contiguous(), last[c].contiguous(), s[c].contiguous(), a[c].contiguous(), lr=0.0).cpu().numpy().astype(np.float64))
        parts.append(_grads(d_layers))
    _check(full, list(zip(*parts)), 1.0 / PARTS, "discriminator")
    np.testing.assert_allclose(l_full, np.mean(losses, axis=0), rtol=1e-5)


def test_critic_and_actor_steps_at_4096_rows_against_eight_steps_at_512():
    import rl_gan_amd as R
    rng = np.random.RandomState(6)

    class Enc:
        features = 256
    actor = R.PixelActor(A_DIM, 0.0, 1100, Enc(), rng=rng)            # learning rate 0: RMSProp leaves the parameters alone
    critic = R.PixelCritic(A_DIM, 0.0, 0.99, 1000, actor, Enc(), rng=rng)
    assert actor.row_passes and critic.row_passes
    g = torch.Generator(device="cuda").manual_seed(4)
    phis = torch.rand((N, 256), device="cuda", generator=g) * 2
    phis_n = torch.rand((N, 256), device="cuda", generator=g) * 2
    a = torch.nn.functional.one_hot(torch.randint(0, A_DIM, (N,), device="cuda", generator=g), A_DIM).float()
    r = torch.rand((N, 1), device="cuda", generator=g)
    chunks = [slice(i * N // PARTS, (i + 1) * N // PARTS) for i in range(PARTS)]
    c_layers, a_layers = critic.eval_net.layers, actor.eval_net.layers
    # the first call of each step copies the eval network over its target (counter 0, Conv_model.py:661-663,752-753): made
    # here, so that every measured step below sees the same targets
    critic.learn(None, a[:16].contiguous(), r[:16].contiguous(), None, phis=phis[:16].contiguous(), phis_next=phis_n[:16].contiguous())
    actor.learn(None, critic, phis=phis[:16].contiguous())
    before = [L.canon.clone() for L in c_layers + a_layers]

    loss_full = float(critic.learn(None, a, r, None, phis=phis, phis_next=phis_n).item())
    full = _grads(c_layers)
    parts, losses = [], []
    for c in chunks:
        losses.append(float(critic.learn(None, a[c].contiguous(), r[c].contiguous(), None, phis=phis[c].contiguous(),
                                         phis_next=phis_n[c].contiguous()).item()))
        parts.append(_grads(c_layers))
    _check(full, list(zip(*parts)), 1.0 / PARTS, "critic (mean squared TD error)")
    assert abs(loss_full - np.mean(losses)) <= 1e-5 * loss_full

    actor.learn(None, critic, phis=phis)
    full = _grads(a_layers)
    parts = []
    for c in chunks:
        actor.learn(None, critic, phis=phis[c].contiguous())
        parts.append(_grads(a_layers))
    _check(full, list(zip(*parts)), 1.0, "actor (policy gradient, a sum over the rows)")
    assert all(torch.equal(b, L.canon) for b, L in zip(before, c_layers + a_layers))


def test_whole_iteration_at_4096_rows_is_reproducible_bit_for_bit():
    """Two learners built from the same seeds, two iterations each at the bench's batch: every variable, optimizer slot and
    loss identical to the last bit.  No kernel of the iteration may depend on the order in which workgroups finish
    (sums over slabs, chunks and tiles are added in a fixed order), which is also what keeps data-parallel replicas equal."""
    import rl_gan_amd as R

    def build():
        rng = np.random.RandomState(0)
        st = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
        ob = R.single_frame_encoder(42, 2, rng=rng)
        G = R.PixelGenerator(256, A_DIM, 1e-4, ob, rng=rng)
        D = R.PixelDiscriminator(A_DIM, 1e-4, st, ob, rng=rng)
        a_pred = R.StateActionPredictor(A_DIM, 1e-4, st, rng=rng)
        actor = R.PixelActor(A_DIM, 1e-4, 1100, st, rng=rng)
        critic = R.PixelCritic(A_DIM, 1e-4, 0.99, 1000, actor, st, rng=rng)
        mem = R.FrameMemory(5000, (42, 42, 4), rng=R.MT19937Device(1))
        g = torch.Generator(device="cuda").manual_seed(0)
        for _ in range(2):
            f = torch.randint(0, 256, (2500, 42, 42, 4), dtype=torch.uint8, device="cuda", generator=g)
            mem.store_transitions(f, f.flip(0))
        scal = R.DeviceMemory(5001, A_DIM + 2)
        scal.store_rows(torch.rand(5000, A_DIM + 2, device="cuda", generator=g) * 2 - 1)
        return R.PixelGanTrainer(mem, scal, G, D, N, a_pred=a_pred, actor=actor, critic=critic)

    def state(tr):
        out = [o.layer.canon for o in tr.G._opt + tr.D._opt + tr.a_pred._opt] + [o.m for o in tr.G._opt + tr.D._opt + tr.a_pred._opt]
        out += [o.v for o in tr.D._opt]
        for net in (tr.actor, tr.critic):
            out += [o.layer.canon for o in net._opt] + [o.ms for o in net._opt] + [o.target.canon for o in net._opt]
        return out + [tr.G.loss_out, tr.D.losses, tr.a_pred.loss_out, tr.critic.loss_out]

    runs = []
    for _ in range(2):
        tr = build()
        for _ in range(2):
            tr.sample()
            tr.step()
        torch.cuda.synchronize()
        runs.append([t.clone() for t in state(tr)] + [tr.idx.clone()])
        del tr
        torch.cuda.empty_cache()
    assert len(runs[0]) == len(runs[1])
    for i, (x, y) in enumerate(zip(*runs)):
        assert torch.equal(x, y), f"state tensor {i} differs between two identical runs"
    assert all(torch.isfinite(t).all() for t in runs[0][:-1])
