"""The reference's own training loops, one transition at a time, on the drop-in classes: `train`
(Curious_GAN.py:172-251) and `train_DDPG` (Curious_GAN.py:253-307), with the object graph built the
way `__main__` builds it (Curious_GAN.py:318-370).  This is the compatibility path (every call moves
NumPy arrays over PCIe); `CuriousLearner` is the fast one.  Plotting, rendering and TensorFlow's Saver
are not part of it; `save_checkpoint` / `load_checkpoint` store the same variables by their
TensorFlow names in an .npz file.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np

from .arm_env import ArmEnv
from .curious_gan import Discriminator, Generator, Memory
from .ddpg import Actor, Critic
from .session import Session, placeholder, set_random_seed


@dataclass
class Settings:
    # names and defaults of Curious_GAN.py:319-339
    MAX_EPISODES: int = 1000
    MAX_EP_STEPS: int = 200
    LR_A: float = 1e-4
    LR_C: float = 1e-4
    LR_D: float = 1e-4
    LR_G: float = 1e-4
    GAMMA: float = 0.9
    REPLACE_ITER_A: int = 1100
    REPLACE_ITER_C: int = 1000
    MEMORY_CAPACITY: int = 5000
    BATCH_SIZE: int = 16
    VAR_MIN: float = 0.1
    ITA: float = 0.01
    MODE: str = 'hard'
    SEED: int = 1


class Agent:
    """actor, critic, generator, discriminator, M wired as in Curious_GAN.py:341-362."""

    def __init__(self, cfg: Optional[Settings] = None):
        self.cfg = cfg = cfg or Settings()
        np.random.seed(cfg.SEED)                     # Curious_GAN.py:320
        set_random_seed(cfg.SEED)                    # Curious_GAN.py:321
        self.env = ArmEnv(mode=cfg.MODE)
        sd, ad, bound = self.env.state_dim, self.env.action_dim, self.env.action_bound
        S = placeholder(None, shape=[None, sd], name='s')
        R = placeholder(None, [None, 1], name='r')
        S_ = placeholder(None, shape=[None, sd], name='s_')
        self.sess = Session()
        self.actor = Actor(self.sess, ad, bound[1], cfg.LR_A, cfg.REPLACE_ITER_A, S=S, S_=S_)
        self.critic = Critic(self.sess, sd, ad, cfg.LR_C, cfg.GAMMA, cfg.REPLACE_ITER_C, self.actor.a, self.actor.a_, S=S, S_=S_, R=R)
        self.actor.add_grad_to_graph(self.critic.a_grads)
        self.generator = Generator(Learning_rate=cfg.LR_G, a=self.actor.a, S=S, action_dim=ad, sess=self.sess, state_dim=sd)
        self.discriminator = Discriminator(G=self.generator.G, Learning_rate=cfg.LR_D, S=S, S_=S_, a=self.actor.a,
                                           action_dim=ad, sess=self.sess, state_dim=sd)
        self.generator.model_loss(self.discriminator.D_logit_fake)
        self.M = Memory(cfg.MEMORY_CAPACITY, dims=3 * sd + ad + 2)
        self.var = 2.0                               # Curious_GAN.py:173

    # ---- checkpoints keyed by TensorFlow variable names (Curious_GAN.py:247-251,372-377) -------
    def variables(self) -> Dict[str, np.ndarray]:
        out: Dict[str, np.ndarray] = {}
        for scope, net in (("Actor/eval_net", self.actor.state.e), ("Actor/target_net", self.actor.state.t),
                           ("Critic/eval_net", self.critic.state.e), ("Critic/target_net", self.critic.state.t),
                           ("Generator/eval_net", self.generator.state.net), ("Discriminator/eval_net", self.discriminator.state.net)):
            for name, value in net.get_tensors().items():
                out[f"{scope}/{name}"] = value
        return out

    def _slot_tables(self):
        """(optimizer scope, variable scope, network, flat slot tensor, TF slot suffix): what a
        `tf.train.Saver()` over all global variables also holds besides the weights, i.e. the RMSProp
        accumulators of `C_train` (critic and, through the missing var_list, actor variables, DDPG.py:148)
        and `A_train` (DDPG.py:112) and the Adam moments of the D and G optimizers (Curious_GAN.py:45,142)."""
        a, c, g, d = self.actor.state, self.critic.state, self.generator.state, self.discriminator.state
        c.ensure_actor_slots(a.e.n_params)
        return (("C_train", "Critic/eval_net", c.e, c.ms, "RMSProp"), ("C_train", "Actor/eval_net", a.e, c.ms_actor, "RMSProp"),
                ("A_train", "Actor/eval_net", a.e, a.ms, "RMSProp"),
                ("D_train", "Discriminator/eval_net", d.net, d.m, "Adam"), ("D_train", "Discriminator/eval_net", d.net, d.v, "Adam_1"),
                ("G_train", "Generator/eval_net", g.net, g.m, "Adam"), ("G_train", "Generator/eval_net", g.net, g.v, "Adam_1"))

    def training_state(self) -> Dict[str, np.ndarray]:
        """Everything besides the weights that the next update depends on: optimizer slots per variable,
        Adam step counts, target-replacement counters, the exploration scale and NumPy's global stream
        (which the reference's Saver cannot hold: it restarts `var` and the stream on restore).  The replay
        memory and the environment are not part of it, as in the reference."""
        out: Dict[str, np.ndarray] = {}
        for opt, scope, net, slot, suffix in self._slot_tables():
            flat, o = slot.detach().cpu().numpy(), 0
            for name, shape in net.table:
                k = int(np.prod(shape))
                out[f"{opt}/{scope}/{name}/{suffix}"] = flat[o:o + k].reshape(shape).copy()
                o += k
        out["D_train/step"] = self.discriminator.state.step.cpu().numpy()
        out["G_train/step"] = self.generator.state.step.cpu().numpy()
        out["Actor/t_replace_counter"] = self.actor.state.counter.cpu().numpy()
        out["Critic/t_replace_counter"] = self.critic.state.counter.cpu().numpy()
        out["var"] = np.array(self.var, np.float64)
        kind, key, pos, has_gauss, cached = np.random.get_state()
        assert kind == "MT19937"
        out["np_random/key"] = np.asarray(key, np.uint32)
        out["np_random/pos_gauss"] = np.array([pos, has_gauss], np.int64)
        out["np_random/cached_gaussian"] = np.array(cached, np.float64)
        return out

    def _load_training_state(self, flat: Dict[str, np.ndarray]) -> None:
        import torch
        for opt, scope, net, slot, suffix in self._slot_tables():
            vec = np.concatenate([np.asarray(flat[f"{opt}/{scope}/{name}/{suffix}"], np.float32).reshape(-1) for name, _ in net.table])
            slot.copy_(torch.from_numpy(vec))
        for key, t in (("D_train/step", self.discriminator.state.step), ("G_train/step", self.generator.state.step),
                       ("Actor/t_replace_counter", self.actor.state.counter), ("Critic/t_replace_counter", self.critic.state.counter)):
            t.copy_(torch.from_numpy(np.asarray(flat[key], np.int32)))
        self.var = float(flat["var"])
        pos, has_gauss = (int(v) for v in flat["np_random/pos_gauss"])
        np.random.set_state(("MT19937", np.asarray(flat["np_random/key"], np.uint32), pos, has_gauss, float(flat["np_random/cached_gaussian"])))

    def save_checkpoint(self, path: str, training_state: bool = True) -> None:
        """Weights under their TensorFlow variable names; with training_state also the optimizer slots,
        counters, exploration scale and RNG stream, so that the next `learn` calls continue bit for bit."""
        items = dict(self.variables())
        if training_state:
            items.update(self.training_state())
        np.savez(path, **{k.replace("/", "."): v for k, v in items.items()})

    def load_checkpoint(self, path: str) -> None:
        data = np.load(path)                         # arrays only: nothing in the file is executed
        flat = {k.replace(".", "/"): data[k] for k in data.files}
        for scope, net in (("Actor/eval_net", self.actor.state.e), ("Actor/target_net", self.actor.state.t),
                           ("Critic/eval_net", self.critic.state.e), ("Critic/target_net", self.critic.state.t),
                           ("Generator/eval_net", self.generator.state.net), ("Discriminator/eval_net", self.discriminator.state.net)):
            net.set_tensors({name: flat[f"{scope}/{name}"] for name, _ in net.table})
        if "var" in flat:
            self._load_training_state(flat)


def train(agent: Agent, episodes: Optional[int] = None, log: Optional[Callable[[str], None]] = print) -> List[dict]:
    """Curious_GAN.py:172-245: returns one record per episode (reward, curious reward, mean losses)."""
    cfg, env, M = agent.cfg, agent.env, agent.M
    actor, critic, generator, discriminator = agent.actor, agent.critic, agent.generator, agent.discriminator
    sd, ad, bound = env.state_dim, env.action_dim, env.action_bound
    history = []
    for ep in range(cfg.MAX_EPISODES if episodes is None else episodes):
        s = env.reset()
        ep_reward = ep_curious_reward = D_loss = G_loss = l_2_loss = 0.0
        done, t = False, 0
        for t in range(cfg.MAX_EP_STEPS):
            a = actor.choose_action(s)
            a = np.clip(np.random.normal(a, agent.var), *bound)
            g = generator.predict(s, a)
            s_, r, done = env.step(a)
            curious_r = r + cfg.ITA * discriminator.determine(s, a, g)[0]
            M.store_transition(s, s_, g, a, r, curious_r)
            l_2_loss += float(np.sum(np.square(s_ - g)))
            if M.pointer > cfg.MEMORY_CAPACITY:
                agent.var = max([agent.var * .9999, cfg.VAR_MIN])
                b_M = M.sample(cfg.BATCH_SIZE)
                b_s, b_s_, b_g = b_M[:, :sd], b_M[:, sd:2 * sd], b_M[:, sd * 2:sd * 3]
                b_a, b_curious_r = b_M[:, -2 - ad:-2], b_M[:, -1:]
                critic.learn(b_s, b_a, b_curious_r, b_s_)
                actor.learn(b_s)
                discriminator.learn(b_g, b_s_, b_s, b_a)
                generator.learn(b_s, b_a)
                D_loss += discriminator.eval(b_g, b_s_, b_s, b_a)
                G_loss += generator.eval(b_s, b_a)
            s = s_
            ep_reward += r
            ep_curious_reward += float(np.ravel(curious_r)[0])
            if t == cfg.MAX_EP_STEPS - 1 or done:
                break
        n = max(t, 1)
        rec = dict(ep=ep, done=bool(done), steps=t + 1, R=ep_reward, Curious_R=ep_curious_reward, D_loss=D_loss / n,
                   G_loss=G_loss / n, Prediction_error=l_2_loss / n, Explore=agent.var)
        history.append(rec)
        if log:
            log('Ep: %i | %s | R: %i | Curious_R: %f | D_loss: %f | G_loss: %f | Prediction_error: %f | Explore: %.2f' % (
                ep, 'done' if done else '----', int(ep_reward), ep_curious_reward, rec["D_loss"], rec["G_loss"],
                rec["Prediction_error"], agent.var))
    return history


def train_DDPG(agent: Agent, episodes: Optional[int] = None, log: Optional[Callable[[str], None]] = print) -> List[dict]:
    """Curious_GAN.py:253-307: plain DDPG on the extrinsic reward (column -2 of the 6-field rows)."""
    cfg, env, M = agent.cfg, agent.env, agent.M
    actor, critic = agent.actor, agent.critic
    sd, ad, bound = env.state_dim, env.action_dim, env.action_bound
    history = []
    for ep in range(cfg.MAX_EPISODES if episodes is None else episodes):
        s = env.reset()
        ep_reward, done, t = 0.0, False, 0
        for t in range(cfg.MAX_EP_STEPS):
            a = actor.choose_action(s)
            a = np.clip(np.random.normal(a, agent.var), *bound)
            s_, r, done = env.step(a)
            M.store_transition(s, s_, s_, a, r, r)
            if M.pointer > cfg.MEMORY_CAPACITY:
                agent.var = max([agent.var * .9999, cfg.VAR_MIN])
                b_M = M.sample(cfg.BATCH_SIZE)
                b_s, b_s_ = b_M[:, :sd], b_M[:, sd:2 * sd]
                b_a, b_r = b_M[:, -2 - ad:-2], b_M[:, -2:-1]
                critic.learn(b_s, b_a, b_r, b_s_)
                actor.learn(b_s)
            s = s_
            ep_reward += r
            if t == cfg.MAX_EP_STEPS - 1 or done:
                break
        history.append(dict(ep=ep, done=bool(done), steps=t + 1, R=ep_reward, Explore=agent.var))
        if log:
            log('Ep: %i | %s | R: %i | Explore: %.2f' % (ep, 'done' if done else '----', int(ep_reward), agent.var))
    return history
