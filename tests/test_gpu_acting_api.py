"""The fused acting step, the reference-shaped class API and the vectorised learner against the
oracle.  Reference: Curious_GAN.py:186-213,318-363."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_acting_step_matches_oracle_sequence():
    """actor -> noise -> G -> env.step -> D -> intrinsic reward -> replay row, 100 envs x 5 steps
    with injected noise (Curious_GAN.py:190-195)."""
    import rl_gan_amd as R
    from gpu_util import Pair, close
    from rl_gan_amd import _lib
    from rl_gan_amd.core import _stream
    from oracle.arm_env import VecArmEnv as OEnv
    n, cap = 100, 256
    p = Pair(spread=False)
    rng = np.random.RandomState(4)
    env, oenv = R.VecArmEnv(n, "hard"), OEnv(n, "hard")
    u = rng.rand(n, 2)
    s = oenv.reset(np.ones(n, bool), u)
    env.reset(torch.from_numpy(u).cuda())
    mem = R.DeviceMemory(cap, 25, rng=R.MT19937Device(1))
    var = torch.tensor([2.0], dtype=torch.float64, device="cuda")
    rows_ref = []
    for t in range(5):
        z = rng.randn(n, 2).astype(np.float32)
        a32 = p.o.choose_action(s.astype(np.float32)).astype(np.float32)           # TF returns float32
        a = np.clip(a32.astype(np.float64) + 2.0 * z.astype(np.float64), -1, 1)    # np.random.normal(a, var) in float64
        g = p.o.predict(s.astype(np.float32), a.astype(np.float32))
        s_, r, done = oenv.step(a)
        cr = p.o.intrinsic_reward(r, s.astype(np.float32), a.astype(np.float32), g.astype(np.float32))
        rows_ref.append(np.hstack([s, s_, g, a, r[:, None], cr]))
        zt = torch.from_numpy(z).cuda()
        act = _lib.Act(7, 2, n, p.A.e.net(), p.G.net.net(), p.D.net.net(), env.handle(), env.state32.data_ptr(),
                       zt.data_ptr(), var.data_ptr(), 0.01, mem._data.data_ptr(), cap, 25, mem.ld, mem.pointer, t,
                       None, env.reward.data_ptr(), env.done.data_ptr(), 0, None, 0, 1, None, None, 0.0, 0.0)
        _lib.check(_lib.lib.rlgan_act_rows(act, None, _stream()))
        mem.pointer += n
        np.testing.assert_array_equal(env.reward.cpu().numpy(), r)
        np.testing.assert_allclose(env.state32.cpu().numpy(), s_.astype(np.float32), rtol=0, atol=2e-6)
        s = s_
    data = mem.data
    for t in range(5):
        slots = (np.arange(n) + t * n) % cap
        if t >= 3:       # rows of steps 3 and 4 are still in the ring (cap 256 holds the last 256 rows)
            close(data[slots], rows_ref[t], what=f"replay rows of step {t}")


def test_reference_shaped_api_one_training_iteration():
    """Build the object graph exactly as Curious_GAN.py:341-363 does and run the loop body of
    Curious_GAN.py:190-213 once the memory is full, against the oracle."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import nets as onets
    from oracle.agent import CuriousAgent
    from oracle.arm_env import ScalarArmEnv

    np.random.seed(1)
    R.set_random_seed(1)
    MEMORY_CAPACITY, BATCH_SIZE, ITA = 64, 16, 0.01
    env = R.ArmEnv(mode='hard')
    STATE_DIM, ACTION_DIM, ACTION_BOUND = env.state_dim, env.action_dim, env.action_bound
    S = R.placeholder(None, shape=[None, STATE_DIM], name='s')
    R_ = R.placeholder(None, [None, 1], name='r')
    S_ = R.placeholder(None, shape=[None, STATE_DIM], name='s_')
    G = R.placeholder(None, shape=[None, STATE_DIM], name='G')
    sess = R.Session()
    actor = R.Actor(sess, ACTION_DIM, ACTION_BOUND[1], 1e-4, 1100, S=S, S_=S_)
    critic = R.Critic(sess, STATE_DIM, ACTION_DIM, 1e-4, 0.9, 1000, actor.a, actor.a_, S=S, S_=S_, R=R_)
    actor.add_grad_to_graph(critic.a_grads)
    generator = R.Generator(Learning_rate=1e-4, a=actor.a, S=S, action_dim=ACTION_DIM, sess=sess, state_dim=STATE_DIM)
    discriminator = R.Discriminator(G=generator.G, Learning_rate=1e-4, S=S, S_=S_, a=actor.a, action_dim=ACTION_DIM,
                                    sess=sess, state_dim=STATE_DIM)
    generator.model_loss(discriminator.D_logit_fake)
    M = R.Memory(MEMORY_CAPACITY, dims=3 * STATE_DIM + ACTION_DIM + 2)
    assert len(actor.e_params) == 8 and actor.e_params[0] == 'Actor/eval_net/l1/kernel'
    assert discriminator.params[0] == 'Discriminator/eval_net/l1/w1_G'

    # oracle twin with the same weights and the same NumPy stream
    o = CuriousAgent(seed=0)
    o.actor_e = onets.unflatten(o.actor_spec, actor.state.e.get_flat().astype(np.float64))
    o.actor_t = onets.unflatten(o.actor_spec, actor.state.t.get_flat().astype(np.float64))
    o.critic_e = onets.unflatten(o.critic_spec, critic.state.e.get_flat().astype(np.float64))
    o.critic_t = onets.unflatten(o.critic_spec, critic.state.t.get_flat().astype(np.float64))
    o.gen = onets.unflatten(o.gen_spec, generator.state.net.get_flat().astype(np.float64))
    o.disc = onets.unflatten(o.disc_spec, discriminator.state.net.get_flat().astype(np.float64))
    assert not np.array_equal(actor.state.e.get_flat(), actor.state.t.get_flat())     # independent target init
    np.testing.assert_allclose(actor.state.e.get_tensors()["l1/bias"], 0.001)
    np.testing.assert_allclose(actor.state.e.get_tensors()["a/a/bias"], 0.0)
    np.testing.assert_allclose(critic.state.e.get_tensors()["q/dense/bias"], 0.01)

    var = 2.0
    s = env.reset()
    oenv = ScalarArmEnv('hard', rng=np.random.RandomState(1))
    so = oenv.reset()
    np.testing.assert_allclose(s, so, atol=1e-12)
    for t in range(MEMORY_CAPACITY + 2):
        a = actor.choose_action(s)
        assert a.shape == (2,) and a.dtype == np.float32
        a = np.clip(np.random.normal(a, var), *ACTION_BOUND)
        g = generator.predict(s, a)
        s_, r, done = env.step(a)
        det = discriminator.determine(s, a, g)
        assert det.shape == (1, 1)
        curious_r = r + ITA * det[0]
        # oracle on the same transition
        close(a if t else a, a, what="noop")
        close(g, o.predict(s[None].astype(np.float32), a[None].astype(np.float32))[0], what="predict")
        close(det, o.determine(s[None].astype(np.float32), a[None].astype(np.float32), g[None]), what="determine")
        so_, ro, do = oenv.step(a)
        np.testing.assert_allclose(s_, so_, atol=1e-12)
        assert r == ro and done == do
        M.store_transition(s, s_, g, a, r, curious_r)
        if M.pointer > MEMORY_CAPACITY:
            var = max([var * .9999, 0.1])
            b_M = M.sample(BATCH_SIZE)
            assert b_M.shape == (BATCH_SIZE, 25) and b_M.dtype == np.float64
            b_s = b_M[:, :STATE_DIM]
            b_s_ = b_M[:, STATE_DIM:2 * STATE_DIM]
            b_g = b_M[:, STATE_DIM * 2:STATE_DIM * 3]
            b_a = b_M[:, -2 - ACTION_DIM:-2]
            b_curious_r = b_M[:, -1:]
            critic.learn(b_s, b_a, b_curious_r, b_s_)
            actor.learn(b_s)
            discriminator.learn(b_g, b_s_, b_s, b_a)
            generator.learn(b_s, b_a)
            D_loss = discriminator.eval(b_g, b_s_, b_s, b_a)
            G_loss = generator.eval(b_s, b_a)
            ref = o.update(b_M)
            close(D_loss, ref["D_loss"], what="D_loss")
            close(G_loss, ref["G_loss"], what="G_loss")
            close(critic.last_loss(), ref["critic_loss"], what="critic loss")
            close(actor.state.e.get_flat(), o.flat("actor_e"), what="actor weights")
            close(critic.state.t.get_flat(), o.flat("critic_t"), what="critic target")
            close(generator.state.net.get_flat(), o.flat("gen"), what="gen weights")
            close(discriminator.state.net.get_flat(), o.flat("disc"), what="disc weights")
        s = s_
    assert actor.t_replace_counter == 2 and critic.t_replace_counter == 2


def test_vectorised_learner_runs_and_is_deterministic():
    import rl_gan_amd as R
    cfg = R.LearnerConfig(n_env=256, batch_size=256, memory_capacity=1024, seed=3)
    runs = []
    for _ in range(2):
        L = R.CuriousLearner(cfg)
        L.fill()
        assert L.memory.pointer == 1280 and not np.isnan(L.memory.data).any()
        assert L.ctr.tolist() == [1280, 5]
        for _ in range(3):
            L.step()
        torch.cuda.synchronize()
        runs.append((L.losses(), L.actor.e.get_flat(), L.disc.net.get_flat(), L.memory.data.copy(), float(L.var.item())))
        assert L.updates == 3 and all(np.isfinite(v) for v in L.losses().values())
        # one decay per acting step taken with a full ring: the last step of fill() and the three after it
        assert abs(runs[-1][4] - 2.0 * 0.9999 ** 4) < 1e-12
    assert runs[0][0] == runs[1][0]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    np.testing.assert_array_equal(runs[0][2], runs[1][2])
    np.testing.assert_array_equal(runs[0][3], runs[1][3])
    d = runs[0][3]
    assert set(np.unique(d[:, 0])) <= {0.0, 1.0} and (np.abs(d[:, 21:23]) <= 1).all()


@pytest.mark.parametrize("captured", [False, True], ids=["eager", "graph"])
def test_vectorised_learner_tracks_the_oracle_loop(captured):
    """fill() + iterations of the device loop against the NumPy restatement of the same loop
    (oracle/loop.py): ring rows, index draws, goals, episode clocks, noise scale, losses, weights.
    Short episodes (max_ep_steps = 4) so that resets happen inside the window.  `graph`: the same with the
    iteration recorded as HIP graphs after fill() (the launch form bench.py times) -- capture() itself runs two
    iterations, the restatement follows them, and both directions of the counter hand-over are then replayed twice."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle_util import load_weights
    from oracle.agent import CuriousAgent, Hyper
    from oracle.loop import VecLoop
    cfg = R.LearnerConfig(n_env=48, batch_size=40, memory_capacity=128, seed=5, max_ep_steps=4, replace_iter_a=2, replace_iter_c=3)
    L = R.CuriousLearner(cfg)
    ag = CuriousAgent(seed=0, hyper=Hyper(replace_iter_a=2, replace_iter_c=3))
    load_weights(ag, {"actor_e": L.actor.e.get_flat(), "actor_t": L.actor.t.get_flat(), "critic_e": L.critic.e.get_flat(),
                      "critic_t": L.critic.t.get_flat(), "gen": L.gen.net.get_flat(), "disc": L.disc.net.get_flat()})
    O = VecLoop(ag, cfg.n_env, cfg.batch_size, cfg.memory_capacity, cfg.mode, L.philox_seed, cfg.seed, max_ep_steps=cfg.max_ep_steps)
    np.testing.assert_allclose(L.env.state32.cpu().numpy(), O.state32, rtol=0, atol=1e-7)
    L.fill()
    while O.pointer <= O.cap:
        O.act()
    if captured:
        L.capture()                  # two launched iterations, then the recording (which runs nothing)
        for _ in range(2):
            O.update()
            O.act()
        assert L._graph is not None and L.updates == 2
    for it in range(4 if captured else 3):
        assert L.memory.pointer == O.pointer and L.ctr.tolist() == [O.pointer, O.step_count]
        np.testing.assert_array_equal(L.idx.cpu().numpy(), O.idx)                       # bit-exact draws
        close(L.memory._data[:, :25].cpu().numpy(), O.ring, what=f"ring before update {it}")
        np.testing.assert_array_equal(L.env.goal.cpu().numpy().reshape(-1, 2), O.env.goal)
        np.testing.assert_array_equal(L.ep_t.cpu().numpy(), O.ep_t)
        assert int(L.n_resets.item()) == O.n_resets and abs(float(L.var.item()) - O.var) < 1e-15
        ref = O.update()
        L.step()                     # update on the same minibatch, its losses, the next acting step, the next draw
        O.act()
        l = L.losses()
        close(l["critic"], ref["critic_loss"], what=f"critic loss {it}")
        close(l["D_loss"], ref["D_loss"], what=f"D_loss {it}")
        close(l["G_loss"], ref["G_loss"], what=f"G_loss {it}")
    assert O.n_resets > 0
    close(L.actor.e.get_flat(), ag.flat("actor_e"), what="actor weights")
    close(L.critic.e.get_flat(), ag.flat("critic_e"), what="critic weights")
    if not captured:
        np.testing.assert_array_equal(L.actor.t.get_flat(), L.actor.e.get_flat())   # copied on calls 0 and 2
    else:
        close(L.actor.t.get_flat(), ag.flat("actor_t"), what="actor target")
        close(L.critic.t.get_flat(), ag.flat("critic_t"), what="critic target")
    close(L.gen.net.get_flat(), ag.flat("gen"), what="generator weights", atol=0.25e-4 * L.updates)   # Adam: sign-of-noise entries
    close(L.disc.net.get_flat(), ag.flat("disc"), what="discriminator weights", atol=0.25e-4 * L.updates)


def test_graph_replay_equals_eager():
    """The launch form bench.py times -- two recorded HIP graphs replayed in turn, one per direction of the hand-over of
    the acting step's counters and noise scale -- against the same iterations launched kernel by kernel: two learners
    from one seed, seven iterations after the recording (each direction replayed at least three times), everything the
    loop owns bit for bit: replay ring, index draw, counters, noise scale, environments, losses, the six networks and
    their optimizer slots."""
    import rl_gan_amd as R
    cfg = R.LearnerConfig(n_env=64, batch_size=48, memory_capacity=256, seed=9, max_ep_steps=5, replace_iter_a=3, replace_iter_c=2)
    runs = []
    for captured in (False, True):
        L = R.CuriousLearner(cfg)
        L.fill()
        if captured:
            L.capture()
            assert L._graph is not None and all(g is not None for g in L._graph)
        else:
            for _ in range(2):       # capture() runs two iterations before it records
                L.step()
        snaps = []
        for it in range(7):
            L.step()
            torch.cuda.synchronize()
            snaps.append(dict(
                ring=L.memory._data.cpu().numpy().copy(), idx=L.idx.cpu().numpy().copy(), ctr=L.ctr.tolist(), cur=L._cur,
                ctr2=L._ctr2.cpu().numpy().copy(), var2=L._var2.cpu().numpy().copy(), var=float(L.var.item()),
                state=L.env.state32.cpu().numpy().copy(), goal=L.env.goal.cpu().numpy().copy(), ep_t=L.ep_t.cpu().numpy().copy(),
                resets=int(L.n_resets.item()), losses=L.losses(),
                nets=[n.get_flat().copy() for n in (L.actor.e, L.actor.t, L.critic.e, L.critic.t, L.gen.net, L.disc.net)],
                slots=[t.cpu().numpy().copy() for t in (L.actor.ms, L.critic.ms, L.critic.ms_actor, L.gen.m, L.gen.v, L.disc.m, L.disc.v)],
                steps=[int(t.item()) for t in (L.actor.counter, L.critic.counter, L.gen.step, L.disc.step)]))
        assert L.updates == 9 and L.memory.pointer == L.ctr.tolist()[0]
        runs.append(snaps)
    for it, (e, g) in enumerate(zip(*runs)):
        for key in ("ctr", "cur", "var", "resets", "steps", "losses"):
            assert e[key] == g[key], (it, key, e[key], g[key])
        for key in ("ring", "idx", "ctr2", "var2", "state", "goal", "ep_t"):
            np.testing.assert_array_equal(e[key], g[key], err_msg=f"iteration {it}: {key}")
        for k, (a, b) in enumerate(zip(e["nets"], g["nets"])):
            np.testing.assert_array_equal(a, b, err_msg=f"iteration {it}: network {k}")
        for k, (a, b) in enumerate(zip(e["slots"], g["slots"])):
            np.testing.assert_array_equal(a, b, err_msg=f"iteration {it}: slot {k}")
    assert runs[0][-1]["resets"] > 0 and all(np.isfinite(v) for v in runs[0][-1]["losses"].values())


def test_update_without_acting_draws_fresh_indices():
    """`CuriousLearner.update()` samples before it learns, like the reference (Curious_GAN.py:199): consecutive calls
    train on consecutive draws of the replay's MT19937 stream, never on the same minibatch twice."""
    import rl_gan_amd as R
    cfg = R.LearnerConfig(n_env=32, batch_size=24, memory_capacity=64, seed=11)
    L = R.CuriousLearner(cfg)
    L.fill()
    draws = [L.idx.cpu().numpy().copy()]
    for _ in range(3):
        L.update()
        draws.append(L.idx.cpu().numpy().copy())
    L.update(redraw=False)
    np.testing.assert_array_equal(L.idx.cpu().numpy(), draws[-1])
    rs = np.random.RandomState(cfg.seed)
    for _ in range(64):                                   # find the draw the fill ended on, then follow the stream
        if np.array_equal(rs.choice(cfg.memory_capacity, size=cfg.batch_size), draws[0]):
            break
    else:
        raise AssertionError("the last draw of fill() is not on the NumPy stream")
    for d in draws[1:]:
        np.testing.assert_array_equal(d, rs.choice(cfg.memory_capacity, size=cfg.batch_size))
    assert all(np.isfinite(v) for v in L.losses().values())


def test_reference_training_loops_and_checkpoint_round_trip(tmp_path):
    """`train` / `train_DDPG` (Curious_GAN.py:172-307) on the drop-in classes for two short episodes, and
    the variables saved and restored under their TensorFlow names (Curious_GAN.py:247-251)."""
    import rl_gan_amd as R
    cfg = R.Settings(MAX_EP_STEPS=30, MEMORY_CAPACITY=40, BATCH_SIZE=16, SEED=1)
    ag = R.Agent(cfg)
    hist = R.train(ag, episodes=2, log=None)
    assert len(hist) == 2 and all(np.isfinite(list(h.values())).all() for h in hist)
    assert ag.M.pointer == sum(h["steps"] for h in hist) > cfg.MEMORY_CAPACITY
    assert hist[1]["D_loss"] > 0 and hist[1]["G_loss"] > 0 and ag.var < 2.0
    assert ag.actor.t_replace_counter == ag.M.pointer - cfg.MEMORY_CAPACITY
    names = ag.variables()
    assert "Actor/eval_net/l1/kernel" in names and "Discriminator/eval_net/l1/w1_G" in names and len(names) == 2 * 8 + 2 * 9 + 9 + 10
    path = str(tmp_path / "Curious_GAN.npz")
    ag.save_checkpoint(path)
    other = R.Agent(R.Settings(SEED=7))
    assert not np.array_equal(other.actor.state.e.get_flat(), ag.actor.state.e.get_flat())
    other.load_checkpoint(path)
    for a, b in ((other.actor.state.e, ag.actor.state.e), (other.critic.state.t, ag.critic.state.t),
                 (other.generator.state.net, ag.generator.state.net), (other.discriminator.state.net, ag.discriminator.state.net)):
        np.testing.assert_array_equal(a.get_flat(), b.get_flat())
    s = np.linspace(-1, 1, 7)
    np.testing.assert_array_equal(other.actor.choose_action(s), ag.actor.choose_action(s))     # fragment images followed
    # optimizer slots, Adam steps, replace counters, exploration scale and the NumPy stream came along:
    # the same batch moves both agents to bit-identical weights
    assert other.var == ag.var and other.actor.t_replace_counter == ag.actor.t_replace_counter
    assert np.random.get_state()[2] == int(np.load(path)["np_random.pos_gauss"][0])
    b = ag.M.data[:16]
    for x in (ag, other):
        bs, bs_, bg, ba, br = b[:, :7], b[:, 7:14], b[:, 14:21], b[:, 21:23], b[:, -1:]
        x.critic.learn(bs, ba, br, bs_); x.actor.learn(bs)
        x.discriminator.learn(bg, bs_, bs, ba); x.generator.learn(bs, ba)
    for a, b2 in ((other.actor.state.e, ag.actor.state.e), (other.critic.state.e, ag.critic.state.e),
                  (other.generator.state.net, ag.generator.state.net), (other.discriminator.state.net, ag.discriminator.state.net)):
        np.testing.assert_array_equal(a.get_flat(), b2.get_flat())
    h2 = R.train_DDPG(R.Agent(cfg), episodes=2, log=None)
    assert len(h2) == 2 and all(np.isfinite(h["R"]) for h in h2)


def test_minibatch_slices_are_served_from_the_device_copy():
    """Column slices of the array `Memory.sample` returns reach the kernels without a second upload: the training
    loop is bit-identical to one that hands over copies of the slices (which take the upload path), and an array
    that is not such a slice is never mistaken for one."""
    import rl_gan_amd as R
    from rl_gan_amd.replay import device_view
    finals = []
    for copies in (False, True):
        R.set_random_seed(3); np.random.seed(3)
        ag = R.Agent(R.Settings(MEMORY_CAPACITY=60, BATCH_SIZE=16, MAX_EP_STEPS=40, SEED=3))
        R.train(ag, episodes=2, log=None)                                   # fills the ring, a few updates
        for _ in range(5):
            b_M = ag.M.sample(16)
            b_s, b_s_, b_g, b_a, b_r = b_M[:, :7], b_M[:, 7:14], b_M[:, 14:21], b_M[:, -4:-2], b_M[:, -1:]
            if not copies:
                for v, c0 in ((b_s, 0), (b_s_, 7), (b_g, 14), (b_a, 21), (b_r, 24)):
                    d = device_view(v)
                    assert d is not None and d.shape == v.shape
                    np.testing.assert_array_equal(d.cpu().numpy().astype(np.float64), v)
                assert device_view(b_M[:8, :7]) is None and device_view(b_M[:, :7].copy()) is None and device_view(b_M[:, ::2]) is None
                with pytest.raises(ValueError):
                    b_M[0, 0] = 1.0                                          # read-only: the device rows cannot go stale
            else:
                b_s, b_s_, b_g, b_a, b_r = (v.copy() for v in (b_s, b_s_, b_g, b_a, b_r))
                assert all(device_view(v) is None for v in (b_s, b_s_, b_g, b_a, b_r))
            ag.critic.learn(b_s, b_a, b_r, b_s_)
            ag.actor.learn(b_s)
            ag.discriminator.learn(b_g, b_s_, b_s, b_a)
            ag.generator.learn(b_s, b_a)
            d_loss, g_loss = ag.discriminator.eval(b_g, b_s_, b_s, b_a), ag.generator.eval(b_s, b_a)
        finals.append((d_loss, g_loss, {k: v.copy() for k, v in ag.variables().items()}))
    assert finals[0][0] == finals[1][0] and finals[0][1] == finals[1][1]
    for k, v in finals[0][2].items():
        np.testing.assert_array_equal(v, finals[1][2][k], err_msg=k)
