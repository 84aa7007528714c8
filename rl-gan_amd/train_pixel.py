"""The reference's training loop of the pixel variant (`train`, Curious_GAN_mario.py:19-165) on the device classes, with
the object graph of `build_model` (Curious_GAN_mario.py:283-358) in its default configuration: 'full_prediction' mode,
no batch norm in the encoders / actor / critic / generator / discriminator, no LSTM in the actor's input, no GAN
pre-training (Curious_GAN_mario.py:375,384-385,397-398).

The environment is the caller's: anything with the wrapper's interface, `reset() -> s` and
`_step(a) -> (s_, r, done, info, _)` with `info['life']` and `info['time']` (env_wrapper.py:48-62,
Curious_GAN_mario.py:66,135), frame stacks as uint8 * 1/255 float32 [h, w, TIME_STEP].  gym and the Mario emulator
are not part of this package.  Acting goes one transition at a time through NumPy like the reference's; learning
keeps the sampled minibatch on the device."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch

from .conv import ConvEncoder, universe_head
from .inverse_model import StateActionPredictor
from .pixel_ddpg import PixelActor, PixelCritic
from .pixel_gan import PixelDiscriminator, PixelGenerator, single_frame_encoder
from .replay import PixelMemory


@dataclass
class PixelSettings:
    # names and defaults of Curious_GAN_mario.py:370-399
    fshape: tuple = (42, 42)
    TIME_STEP: int = 4
    ACTION_DIM: int = 6
    MAX_EPISODES: int = 100
    MAX_EP_STEPS: int = 4000
    LR_DECAY_LIST: List[int] = field(default_factory=lambda: [50000, 100000])
    LR_D_list: List[float] = field(default_factory=lambda: [1e-4, 1e-6, 1e-6])
    LR_G_list: List[float] = field(default_factory=lambda: [1e-4, 1e-6, 1e-6])
    LR_A_pred_list: List[float] = field(default_factory=lambda: [1e-4, 1e-4, 1e-4])
    LR_A: float = 1e-4
    LR_C: float = 1e-4
    GAMMA: float = 0.99
    REPLACE_ITER_A: int = 1100
    REPLACE_ITER_C: int = 1000
    MEMORY_CAPACITY: int = 5000
    BATCH_SIZE: int = 32
    VAR_MIN: float = 1.0
    ITA: float = 1.0
    SEED: int = 1


def transfer_sparse_action(a: np.ndarray) -> np.ndarray:
    """Curious_GAN_mario.py:267-270."""
    dense_a = np.ones(a.shape) * 0.1
    dense_a[np.argmax(a)] = 1
    return dense_a


class PixelAgent:
    """`build_model()` (Curious_GAN_mario.py:283-358): one state encoder shared by the inverse model, the actor, the
    critic, the generator and the discriminator, one single-frame encoder shared by the generator's loss and the
    discriminator."""

    def __init__(self, cfg: Optional[PixelSettings] = None):
        self.cfg = cfg = cfg or PixelSettings()
        rng = np.random.RandomState(cfg.SEED)
        h, w = cfg.fshape
        self.state_encoder = ConvEncoder(h, w, cfg.TIME_STEP, universe_head(2), rng=rng, dense=256, bias_init=0.001)
        self.obs_encoder = single_frame_encoder(h, 2, rng=rng)
        ad = cfg.ACTION_DIM
        self.A_pred = StateActionPredictor(ad, cfg.LR_A_pred_list[0], self.state_encoder, rng=rng)
        self.actor = PixelActor(ad, cfg.LR_A, cfg.REPLACE_ITER_A, self.state_encoder, rng=rng)
        self.critic = PixelCritic(ad, cfg.LR_C, cfg.GAMMA, cfg.REPLACE_ITER_C, self.actor, self.state_encoder, rng=rng)
        self.M = PixelMemory(cfg.MEMORY_CAPACITY, (h, w, cfg.TIME_STEP), a_dim=ad, lstm_units=self.state_encoder.features)
        self.generator = PixelGenerator(self.state_encoder.features, ad, cfg.LR_G_list[0], self.obs_encoder, rng=rng,
                                        state_encoder=self.state_encoder)
        self.discriminator = PixelDiscriminator(ad, cfg.LR_D_list[0], self.state_encoder, self.obs_encoder, rng=rng)
        self.var = 2.0                                           # Curious_GAN_mario.py:21
        self.total_steps = 0
        self.lr_pointer = 0

    # ---- checkpoints ---------------------------------------------------------------------------------
    def _state_tensors(self) -> dict:
        """Every device tensor that defines the training state, under a stable name: the parameters of each layer in
        the layout the layer keeps them in (kernel then bias, widths padded as the classes document), the optimizer
        slots of each optimizer (the encoders appear once as parameters and once per optimizer that trains them, like
        TensorFlow's slot variables), the step and target-replacement counters."""
        out = {}
        for name, enc in (("feature_abstraction", self.state_encoder), ("single_feature_abstraction", self.obs_encoder)):
            for i, L in enumerate(enc.layers + [enc.head]):
                out[f"{name}/{i}"] = L.canon
        nets = dict(A_pred=self.A_pred, Generator=self.generator, Discriminator=self.discriminator)
        for name, net in nets.items():
            for i, o in enumerate(net._opt):
                own = not any(o.layer is L for enc in (self.state_encoder, self.obs_encoder) for L in enc.layers + [enc.head])
                if own:
                    out[f"{name}/{i}"] = o.layer.canon
                out[f"{name}/Adam/{i}/m"], out[f"{name}/Adam/{i}/v"] = o.m, o.v
            out[f"{name}/step"] = net.step
        for name, net in (("Actor", self.actor), ("Critic", self.critic)):
            for i, o in enumerate(net._opt):
                out[f"{name}/eval_net/{i}"], out[f"{name}/target_net/{i}"] = o.layer.canon, o.target.canon
                out[f"{name}/RMSProp/{i}"] = o.ms
            out[f"{name}/t_replace_counter"] = net.t_replace_counter
        for name, unit in (("Actor/eval_net/lstm", self.actor.LSTM_unit), ("Actor/target_net/lstm", self.actor.LSTM_unit_)):
            out[name] = unit.cell.canon
        return out

    def _block_layers(self) -> dict:
        """Names of `_state_tensors` that belong to a first layer kept as a 2x2-block kernel (`SingleFrameConv`: 17 f numbers,
        seven structural zeros per filter) -> that layer.  In a checkpoint these are written in the reference's own layout,
        kernel [3, 3, 1, f] then bias, so the file does not depend on how the layer is run; loading puts zeros in the
        seven entries outside the 3x3 window, for the parameters and for every optimizer slot."""
        from .conv import SingleFrameConv
        out = {}
        for name, enc in (("feature_abstraction", self.state_encoder), ("single_feature_abstraction", self.obs_encoder)):
            for i, L in enumerate(enc.layers):
                if isinstance(L, SingleFrameConv):
                    out[f"{name}/{i}"] = L
        for name, net in dict(A_pred=self.A_pred, Generator=self.generator, Discriminator=self.discriminator).items():
            for i, o in enumerate(net._opt):
                if isinstance(o.layer, SingleFrameConv):
                    out[f"{name}/Adam/{i}/m"] = out[f"{name}/Adam/{i}/v"] = o.layer
        return out

    def save_checkpoint(self, path: str) -> None:
        """What the reference's Saver keeps (Curious_GAN_mario.py:167-172) plus what it leaves out and a bit-exact
        continuation needs: optimizer slots, counters, the exploration scale, the learning-rate pointer and NumPy's
        stream.  The replay is not saved, like the reference's."""
        st = np.random.get_state()
        blocked = self._block_layers()
        np.savez(path, **{k: (blocked[k].to_frame_layout(v) if k in blocked else v.cpu().numpy()) for k, v in self._state_tensors().items()},
                 **{"host/var": self.var, "host/total_steps": self.total_steps, "host/lr_pointer": self.lr_pointer,
                    "host/np_keys": st[1], "host/np_pos": st[2], "host/np_gauss": np.array([st[3], st[4]])})

    def load_checkpoint(self, path: str) -> None:
        with np.load(path) as f:
            tensors = self._state_tensors()
            missing = [k for k in tensors if k not in f.files]
            if missing:
                raise KeyError(f"checkpoint lacks {missing[:3]}...")
            blocked = self._block_layers()
            for k, t in tensors.items():
                a = f[k]
                if k in blocked:                                   # kept in the reference's [3, 3, 1, f] + bias layout in the file
                    if a.shape != (10 * blocked[k].c_out,):
                        raise ValueError(f"{k}: checkpoint has shape {a.shape}, the reference layout is {(10 * blocked[k].c_out,)}")
                    a = blocked[k].from_frame_layout(a).numpy()
                if a.shape != tuple(t.shape):
                    raise ValueError(f"{k}: checkpoint has shape {a.shape}, the model {tuple(t.shape)}")
                t.copy_(torch.from_numpy(a))
            self.var, self.total_steps = float(f["host/var"]), int(f["host/total_steps"])
            self.lr_pointer = int(f["host/lr_pointer"])
            g = f["host/np_gauss"]
            np.random.set_state(("MT19937", f["host/np_keys"], int(f["host/np_pos"]), int(g[0]), float(g[1])))
        # the convolutions and the decoder keep fragment images of their parameters: rebuild them from the loaded values
        for enc in (self.state_encoder, self.obs_encoder):
            for L in enc.layers:
                L.pack()
        for L in self.generator.decoder.layers:
            L.pack()
        for net in (self.actor, self.critic):                     # the row-pass kernels read fragment images as well
            if getattr(net, "row_passes", False):
                net._rn.repack()
                net._rn_t.repack()

    def learn_minibatch(self) -> dict:
        """Curious_GAN_mario.py:82-103 on one sampled minibatch; returns the three evaluated losses."""
        cfg, G, D = self.cfg, self.generator, self.discriminator
        if self.total_steps == cfg.LR_DECAY_LIST[min(self.lr_pointer, len(cfg.LR_DECAY_LIST) - 1)]:
            self.lr_pointer = min(self.lr_pointer + 1, len(cfg.LR_D_list))
        lp = min(self.lr_pointer, len(cfg.LR_D_list) - 1)       # the reference would index past the list at its last decay
        b = self.M.sample_device(cfg.BATCH_SIZE)
        b_s, b_s_, b_a = b["s"], b["s_"], b["a"].contiguous()
        last = b_s_[..., -1:].contiguous()                       # b_s_[:, :, :, [-1]]
        phis = self.state_encoder.forward(b_s)
        G.learn(phis, b_a, last, cfg.LR_G_list[lp])
        b_g = G.predict_batch(phis, b_a)
        D.learn(b_g, last, b_s, b_a, cfg.LR_D_list[lp], phis=phis)
        self.A_pred.learn(b_s, b_s_, b_a, cfg.LR_A_pred_list[lp])
        phis_next = self.state_encoder.forward(b_s_)
        phis = self.state_encoder.forward(b_s)
        self.critic.learn(None, b_a, b["curious_r"].contiguous(), None, phis=phis, phis_next=phis_next)
        self.actor.learn(None, self.critic, phis=phis)
        d_loss = D.eval(b_g, last, b_s, b_a)
        g_loss = G.eval(self.state_encoder.forward(b_s), b_a, last)
        return dict(D_loss=d_loss, G_loss=g_loss, A_pred_loss=self.A_pred.eval(b_s, b_s_, b_a))


def train_pixel(agent: PixelAgent, make_env: Callable[[], object], episodes: Optional[int] = None,
                log: Optional[Callable[[str], None]] = print) -> List[dict]:
    """Curious_GAN_mario.py:27-165 (ASYN_TRAIN_GAN False): one record per episode."""
    cfg, M = agent.cfg, agent.M
    actor, generator, discriminator = agent.actor, agent.generator, agent.discriminator
    history = []
    for ep in range(cfg.MAX_EPISODES if episodes is None else episodes):
        ep_reward = ep_curious_reward = D_loss = G_loss = l_2_loss = 0.0
        a_pred_loss, done, t = 0.0, False, 0
        lstm_state = actor.LSTM_unit.get_initial_state()
        env = make_env()
        s = env.reset()
        for t in range(cfg.MAX_EP_STEPS):
            a = actor.choose_action(s, lstm_state, agent.var)
            dense_a = transfer_sparse_action(a)
            s_, r, done, info, _ = env._step(a)
            g = generator.predict(s, a)
            curious_r = cfg.ITA * discriminator.determine(s, a, g)[0]
            lstm_state_ = actor.LSTM_unit.get_state(s, lstm_state)
            l_2_loss += discriminator.observe_and_compare(s_, g)
            M.store_transition(s, g, lstm_state, a, r, curious_r, s_, lstm_state_, dense_a)
            if M.pointer > cfg.MEMORY_CAPACITY:
                losses = agent.learn_minibatch()
                D_loss += losses["D_loss"]; G_loss += losses["G_loss"]; a_pred_loss = losses["A_pred_loss"]
            s, lstm_state = s_, lstm_state_
            ep_reward += r
            ep_curious_reward += float(np.ravel(curious_r)[0])
            agent.total_steps += 1
            if t == cfg.MAX_EP_STEPS - 1 or done or info['life'] == 0 or info['time'] <= 1:
                t = t + 1
                agent.var = max([agent.var * .99, cfg.VAR_MIN])
                if hasattr(env, "close"):
                    env.close()
                break
        rec = dict(ep=ep, done=bool(done), steps=t, R=ep_reward, Curious_R=ep_curious_reward, D_loss=D_loss / max(t, 1),
                   G_loss=G_loss / max(t, 1), A_pred_loss=a_pred_loss, Prediction_error=l_2_loss / max(t, 1), Explore=agent.var)
        history.append(rec)
        if log:
            log('Ep: %i | %s | R: %i | Curious_R: %f | D_loss: %f | G_loss: %f | Prediction_error: %f | Explore: %.2f' % (
                ep, '| done' if done else '| ----', int(ep_reward), ep_curious_reward, rec["D_loss"], rec["G_loss"],
                rec["Prediction_error"], agent.var))
    return history
