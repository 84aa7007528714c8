"""The GAN half of the pixel training iteration (Curious_GAN_mario.py:85-93) as one recorded HIP graph against
the same calls made one by one: identical bits after several iterations, indices drawn like the reference's
replay does (Conv_model.py:147)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(batch, seed=0):
    import rl_gan_amd as R
    rng = np.random.RandomState(seed)
    a_dim, cap = 6, 40
    st = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
    ob = R.single_frame_encoder(42, 2, rng=rng)
    G = R.PixelGenerator(256, a_dim, 1e-4, ob, rng=rng)
    D = R.PixelDiscriminator(a_dim, 1e-4, st, ob, rng=rng)
    frames = R.FrameMemory(cap, (42, 42, 4), rng=R.MT19937Device(11))
    actions = R.DeviceMemory(cap + 1, a_dim)
    f = rng.randint(0, 256, (cap, 42, 42, 4)).astype(np.uint8)
    frames.store_transitions(f, np.roll(f, 1, axis=0))
    actions.store_rows(torch.from_numpy(rng.uniform(-1, 1, (cap, a_dim)).astype(np.float32)).cuda())
    return R.PixelGanTrainer(frames, actions, G, D, batch)


def test_graph_replay_equals_eager_calls():
    eager, graphed = _make(8), _make(8)
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
    assert int(graphed.G.step.item()) == 5 and int(graphed.D.step.item()) == 5
    assert np.isfinite(graphed.G.loss_out.item()) and np.isfinite(graphed.D.losses.cpu().numpy()).all()
