"""The pixel variant's training loop (Curious_GAN_mario.py:19-165) on the device classes, driven by a small synthetic
environment with the wrapper's interface (gym and the emulator are not available): bookkeeping of the reference
(replay gate, episode ends, exploration decay, learning-rate steps) and run-to-run determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class ToyEnv:
    """Frame stacks of random bytes scaled by 1/255 (env_wrapper.py:44,58); reward 1 when the first action is taken."""

    def __init__(self, seed, shape=(42, 42, 4), life_ends_at=None):
        self.rng, self.shape, self.t, self.life_ends_at = np.random.RandomState(seed), shape, 0, life_ends_at

    def _frames(self):
        return self.rng.randint(0, 256, self.shape).astype(np.uint8).astype(np.float32) * (1.0 / 255)

    def reset(self):
        self.t = 0
        return self._frames()

    def _step(self, a):
        self.t += 1
        life = 0 if self.life_ends_at is not None and self.t >= self.life_ends_at else 1
        return self._frames(), float(np.argmax(a) == 0), False, {"life": life, "time": 400 - self.t}, None


def _run(seed=1):
    import rl_gan_amd as R
    np.random.seed(seed)
    cfg = R.PixelSettings(MEMORY_CAPACITY=12, BATCH_SIZE=8, MAX_EP_STEPS=10, LR_DECAY_LIST=[20, 25], SEED=seed)
    agent = R.PixelAgent(cfg)
    envs = iter(range(100))
    hist = R.train_pixel(agent, lambda: ToyEnv(next(envs), life_ends_at=7 if agent.total_steps >= 20 else None), episodes=3, log=None)
    return agent, hist


def test_loop_bookkeeping_and_determinism():
    agent, hist = _run()
    # episodes 0 and 1 run their 10 steps; episode 2 ends when the toy environment reports life == 0 at its step 7
    assert [h["steps"] for h in hist] == [10, 10, 7]
    assert agent.M.pointer == 27 and agent.total_steps == 27 and agent.M.len == 13
    np.testing.assert_allclose([h["Explore"] for h in hist], [2 * .99, 2 * .99 ** 2, 2 * .99 ** 3])
    learned = 27 - 12                                       # every transition from the 13th on (pointer > MEMORY_CAPACITY after its store)
    for counter in (agent.generator.step, agent.discriminator.step, agent.A_pred.step, agent.critic.t_replace_counter,
                    agent.actor.t_replace_counter):
        assert int(counter.item()) == learned
    assert agent.lr_pointer == 2                            # total_steps passed 20 and 25
    for h in hist[1:]:
        assert np.isfinite([h["D_loss"], h["G_loss"], h["A_pred_loss"], h["Curious_R"]]).all() and h["D_loss"] > 0
    assert hist[0]["D_loss"] == 0 and hist[2]["Prediction_error"] == 0
    again, hist2 = _run()
    assert hist == hist2
    for a, b in zip(agent.generator._opt + agent.discriminator._opt + agent.A_pred._opt, again.generator._opt + again.discriminator._opt + again.A_pred._opt):
        assert torch.equal(a.layer.canon, b.layer.canon)
    for a, b in zip(agent.actor._opt + agent.critic._opt, again.actor._opt + again.critic._opt):
        assert torch.equal(a.layer.canon, b.layer.canon) and torch.equal(a.target.canon, b.target.canon)


def test_transfer_sparse_action():
    import rl_gan_amd as R
    np.testing.assert_array_equal(R.transfer_sparse_action(np.array([0, 0, 1, 0])), [0.1, 0.1, 1.0, 0.1])


def test_checkpoint_continues_bit_for_bit(tmp_path):
    """A second agent restored from a checkpoint learns exactly like the one that kept going: weights, both kinds of
    optimizer slots, counters, exploration scale, learning-rate pointer and NumPy's stream are all in the file."""
    import rl_gan_amd as R
    agent, _ = _run(seed=4)
    path = str(tmp_path / "pixel.npz")
    agent.save_checkpoint(path)
    other = R.PixelAgent(R.PixelSettings(MEMORY_CAPACITY=12, BATCH_SIZE=8, MAX_EP_STEPS=10, LR_DECAY_LIST=[20, 25], SEED=99))
    np.random.seed(123)
    other.load_checkpoint(path)
    assert other.var == agent.var and other.total_steps == agent.total_steps and other.lr_pointer == agent.lr_pointer
    other.M = agent.M                                       # the replay is not part of the checkpoint (nor of the reference's)
    st = np.random.get_state()
    la = [agent.learn_minibatch() for _ in range(3)]
    np.random.set_state(st)                                 # the same index draws for the restored agent
    agent2_losses = [other.learn_minibatch() for _ in range(3)]
    assert la == agent2_losses
    for (ka, ta), (kb, tb) in zip(agent._state_tensors().items(), other._state_tensors().items()):
        assert ka == kb and torch.equal(ta, tb), ka


def test_checkpoint_keeps_the_single_frame_kernel_in_the_reference_shape(tmp_path):
    """The single-frame encoder's first layer runs on 2x2 blocks (17 f numbers, seven structural zeros per filter); the file
    holds it, and its Adam slots, as the reference's [3, 3, 1, f] kernel + bias (10 f numbers), and loading leaves zeros
    in the seven entries whatever the file held around them."""
    import rl_gan_amd as R
    agent, _ = _run(seed=4)
    path = str(tmp_path / "pixel.npz")
    agent.save_checkpoint(path)
    L = agent.obs_encoder.layers[0]
    f = L.c_out
    names = agent._block_layers()
    assert "single_feature_abstraction/0" in names and sum(k.endswith("/m") for k in names) >= 1
    with np.load(path) as z:
        kern, bias = L.kernel_bias()
        np.testing.assert_array_equal(z["single_feature_abstraction/0"], np.concatenate([kern.ravel(), bias]))
        for k in names:
            assert z[k].shape == (10 * f,), k
    other = R.PixelAgent(R.PixelSettings(MEMORY_CAPACITY=12, BATCH_SIZE=8, MAX_EP_STEPS=10, LR_DECAY_LIST=[20, 25], SEED=99))
    other.load_checkpoint(path)
    outside = (other.obs_encoder.layers[0].mask == 0)
    for k, t in other._state_tensors().items():
        if k in names:
            assert not t[outside].any(), k
            assert torch.equal(t, agent._state_tensors()[k]), k
