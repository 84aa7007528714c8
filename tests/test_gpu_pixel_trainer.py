"""The GAN half of the pixel training iteration (Curious_GAN_mario.py:85-93) as one recorded HIP graph against
the same calls made one by one: identical bits after several iterations, indices drawn like the reference's
replay does (Conv_model.py:147)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(batch, seed=0, full=False):
    import rl_gan_amd as R
    rng = np.random.RandomState(seed)
    a_dim, cap = 6, 40
    st = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
    ob = R.single_frame_encoder(42, 2, rng=rng)
    G = R.PixelGenerator(256, a_dim, 1e-4, ob, rng=rng)
    D = R.PixelDiscriminator(a_dim, 1e-4, st, ob, rng=rng)
    frames = R.FrameMemory(cap, (42, 42, 4), rng=R.MT19937Device(11))
    actions = R.DeviceMemory(cap + 1, a_dim + 2 if full else a_dim)
    f = rng.randint(0, 256, (cap, 42, 42, 4)).astype(np.uint8)
    frames.store_transitions(f, np.roll(f, 1, axis=0))
    actions.store_rows(torch.from_numpy(rng.uniform(-1, 1, (cap, actions.dims)).astype(np.float32)).cuda())
    if not full:
        return R.PixelGanTrainer(frames, actions, G, D, batch)
    a_pred = R.StateActionPredictor(a_dim, 1e-4, st, rng=rng)
    actor = R.PixelActor(a_dim, 1e-4, 3, st, rng=rng)
    critic = R.PixelCritic(a_dim, 1e-4, 0.99, 2, actor, st, rng=rng)
    return R.PixelGanTrainer(frames, actions, G, D, batch, a_pred=a_pred, actor=actor, critic=critic)


@pytest.mark.parametrize("full", [False, True])
def test_graph_replay_equals_eager_calls(full):
    """full: with the inverse model's, the critic's and the actor's steps after the GAN half (Curious_GAN_mario.py:96-98)."""
    eager, graphed = _make(8, full=full), _make(8, full=full)
    graphed.capture()
    # capture ran two iterations on the zero index array before recording; bring the eager twin to the same state
    for _ in range(2):
        eager.step()
    want_idx = np.random.RandomState(11)
    for it in range(3):
        for t in (eager, graphed):
            idx = t.sample()
        np.testing.assert_array_equal(idx.cpu().numpy(), want_idx.choice(40, size=8))
        eager.step(); graphed.step()
        assert torch.equal(eager.G.loss_out, graphed.G.loss_out) and torch.equal(eager.D.losses, graphed.D.losses)
    for a, b in zip(eager.G._opt + eager.D._opt, graphed.G._opt + graphed.D._opt):
        assert torch.equal(a.layer.canon, b.layer.canon) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
    if full:
        for t in (eager, graphed):
            assert int(t.a_pred.step.item()) == 5 and int(t.critic.t_replace_counter.item()) == 5 and int(t.actor.t_replace_counter.item()) == 5
        assert torch.equal(eager.a_pred.loss_out, graphed.a_pred.loss_out) and torch.equal(eager.critic.loss_out, graphed.critic.loss_out)
        assert np.isfinite(graphed.a_pred.loss_out.item()) and np.isfinite(graphed.critic.loss_out.item())
        pairs = zip(eager.a_pred._opt, graphed.a_pred._opt)
        for a, b in pairs:
            assert torch.equal(a.layer.canon, b.layer.canon) and torch.equal(a.m, b.m)
        for e, g in ((eager.actor, graphed.actor), (eager.critic, graphed.critic)):
            for a, b in zip(e._opt, g._opt):
                assert torch.equal(a.layer.canon, b.layer.canon) and torch.equal(a.ms, b.ms) and torch.equal(a.target.canon, b.target.canon)
        # five calls: the critic (period 2) copied after calls 0, 2, 4, the actor (period 3) after calls 0 and 3
        assert torch.equal(graphed.critic._opt[0].layer.canon, graphed.critic._opt[0].target.canon)
        assert not torch.equal(graphed.actor._opt[0].layer.canon, graphed.actor._opt[0].target.canon)
    assert int(graphed.G.step.item()) == 5 and int(graphed.D.step.item()) == 5
    assert np.isfinite(graphed.G.loss_out.item()) and np.isfinite(graphed.D.losses.cpu().numpy()).all()
