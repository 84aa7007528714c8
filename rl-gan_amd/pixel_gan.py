"""Generator of the reference's pixel variant in its default 'full_prediction' mode
(`Conv_model.Generator`, Conv_model.py:343-376,437-475,522-541; wiring Curious_GAN_mario.py:328-353):

    code  = relu([phis, a] * [w1_s; w1_a] + b1)                     200 units          Conv_model.py:444-455
    image = decoder(code)                                           42 x 42 x 1        Conv_model.py:457-473
    fake_observation = single-frame encoder(image)                  256 features       Curious_GAN_mario.py:340-343
    G_loss = mean((fake_observation - observation)^2),  observation = the same encoder on the last frame of s'
    train_op = Adam(lr).minimize(G_loss, var_list = Generator variables)               Conv_model.py:527-531

The single-frame encoder belongs to the discriminator's variable list in this mode (Conv_model.py:192-194); the
generator's step differentiates through it and leaves it unchanged.  Batch norm off (Curious_GAN_mario.py:384).
The 1-channel image is padded to the 4 channels the convolution kernels take; the encoder's first kernel has
zero rows for the three padding channels, which zero inputs keep at zero."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .conv import ConvEncoder, DenseLayer
from .core import _stream, device, launch_apply, row_ld
from .deconv import Decoder

N_L1 = 200                                                      # Conv_model.py:445


def single_frame_encoder(hw: int = 42, n_convs: int = 2, rng: Optional[np.random.RandomState] = None, params=None,
                         dense_params=None, bias_init: float = 0.0) -> ConvEncoder:
    """`universe_feature_abstraction(single_S_, time_step=1, nConvs=2)` (Curious_GAN_mario.py:313-314): an encoder of
    [n, hw, hw, 1] frames.  params: [(kernel [3,3,1,32], bias), (kernel [3,3,32,32], bias), ...]."""
    from .conv import universe_head
    rng = rng or np.random.RandomState()
    if params is None:                                      # tf.layers.conv2d defaults: glorot-uniform kernels, zero biases
        params, c = [], 1
        for f, k, _ in universe_head(n_convs):
            lim = np.sqrt(6.0 / (k * k * c + k * k * f))
            params.append((rng.uniform(-lim, lim, (k, k, c, f)), np.full(f, bias_init)))
            c = f
    return ConvEncoder(hw, hw, 1, universe_head(n_convs), rng=rng, bias_init=0.001, params=params, dense=256, dense_params=dense_params)


class _Adam:
    """tf.train.AdamOptimizer slots of one layer (Conv_model.py:531)."""

    def __init__(self, layer, map_f=None, image=None):
        self.layer, self.map_f, self.image = layer, map_f, image
        self.m = torch.zeros_like(layer.canon)
        self.v = torch.zeros_like(layer.canon)

    def job(self, step: torch.Tensor, lr: float) -> _lib.ApplyJob:
        L = self.layer
        return _lib.ApplyJob(1, L.n_params, L.canon.data_ptr(), L.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                             lr, 0.9, 0.999, 1e-8, step.data_ptr(), None if self.map_f is None else self.map_f.data_ptr(),
                             None if self.image is None else self.image.data_ptr(), None, None, None, None, 1.0, None, 0, None)


class PixelGenerator:
    def __init__(self, h_dim: int, a_dim: int, learning_rate: float, obs_encoder: ConvEncoder,
                 rng: Optional[np.random.RandomState] = None, w1_s=None, w1_a=None, b1=None, decoder_params=None,
                 state_encoder: Optional[ConvEncoder] = None):
        """state_encoder: the encoder of the frame stacks, only needed by `predict` (which takes frames like the
        reference's; the batched calls take its features)."""
        rng = rng or np.random.RandomState()
        self.state_encoder = state_encoder
        self.h_dim, self.a_dim, self.lr = h_dim, a_dim, learning_rate
        self.k_pad = (h_dim + a_dim + 3) // 4 * 4
        if w1_s is None:                                        # xavier_initializer, biases 0.001 (Conv_model.py:439-448)
            lim_s, lim_a = np.sqrt(6.0 / (h_dim + N_L1)), np.sqrt(6.0 / (a_dim + N_L1))
            w1_s, w1_a = rng.uniform(-lim_s, lim_s, (h_dim, N_L1)), rng.uniform(-lim_a, lim_a, (a_dim, N_L1))
            b1 = np.full(N_L1, 0.001)
        kernel = np.zeros((self.k_pad, N_L1))
        kernel[:h_dim], kernel[h_dim:h_dim + a_dim] = w1_s, w1_a
        self.l1 = DenseLayer(self.k_pad, N_L1, kernel=kernel, bias=np.asarray(b1).reshape(-1))
        self.decoder = Decoder(N_L1, rng=rng, params=decoder_params)
        self.obs_encoder = obs_encoder
        self.step = torch.zeros(1, dtype=torch.int32, device=device())
        self._opt: List[_Adam] = [_Adam(self.l1)] + [_adam_for(L) for L in self.decoder.layers]
        self._x = self._h1 = self._img = None
        self._parts: Optional[torch.Tensor] = None
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=device())

    # ---- pieces ------------------------------------------------------------------------------------
    def _input(self, phis: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        n = phis.shape[0]
        x = torch.empty((n, self.k_pad), dtype=torch.float32, device=phis.device)
        check(lib.rlgan_concat_cols(phis.data_ptr(), row_ld(phis), self.h_dim, a.data_ptr(), row_ld(a), self.a_dim, None, 0, 0,
                                    x.data_ptr(), self.k_pad, n, self.k_pad - self.h_dim - self.a_dim, _stream()), "rlgan_concat_cols")
        return x

    def predict_batch(self, phis: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        """[n, 42, 42, 1] predicted next frames (Conv_model.py:562-567)."""
        self._x = self._input(phis.contiguous(), a.contiguous())
        self._h1 = self.l1.forward(self._x, act="relu")
        self._img = self.decoder.forward(self._h1)
        return self._img

    def predict(self, s, a) -> np.ndarray:
        """The predicted next frame [42, 42, 1] for one state [h, w, c] and action (Conv_model.py:553-560)."""
        assert self.state_encoder is not None, "predict() takes frames: construct the generator with state_encoder="
        st = torch.as_tensor(np.asarray(s, dtype=np.float32)[None], device=device())
        at = torch.as_tensor(np.asarray(a, dtype=np.float32)[None], device=device())
        return self.predict_batch(self.state_encoder.forward(st), at).cpu().numpy()[0]

    def features(self, frames1: torch.Tensor) -> torch.Tensor:
        """Single-frame encoder on [n, h, w, 1] frames (Curious_GAN_mario.py:313-314,340-343)."""
        return self.obs_encoder.forward(frames1.contiguous())

    def _mse(self, fake: torch.Tensor, real: torch.Tensor, want_grad: bool):
        n = fake.numel()
        n_parts = (n + 4095) // 4096
        if self._parts is None or self._parts.numel() < n_parts:
            self._parts = torch.empty(n_parts, dtype=torch.float32, device=fake.device)
        d = torch.empty_like(fake) if want_grad else None
        check(lib.rlgan_mse(fake.data_ptr(), real.data_ptr(), n, self._parts.data_ptr(), None if d is None else d.data_ptr(), _stream()),
              "rlgan_mse")
        check(lib.rlgan_sum_parts(self._parts.data_ptr(), n_parts, 1, 1.0 / n, self.loss_out.data_ptr(), _stream()), "rlgan_sum_parts")
        return d

    # ---- reference API -----------------------------------------------------------------------------
    def eval(self, phis: torch.Tensor, a: torch.Tensor, single_s_next: torch.Tensor) -> float:
        """G_loss (Conv_model.py:543-551)."""
        obs = self.features(single_s_next).clone()
        self._mse(self.features(self.predict_batch(phis, a)), obs, False)
        return float(self.loss_out.item())

    def learn(self, phis: torch.Tensor, a: torch.Tensor, single_s_next: torch.Tensor, lr: Optional[float] = None,
              obs: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One Adam step on the generator's variables (Conv_model.py:533-541); returns the loss before the step
        (device scalar).  The encoder's variables are not in the var_list: only the input gradient is taken through it.
        obs: the single-frame encoder's features of single_s_next when the caller already holds them."""
        if obs is None:
            obs = self.features(single_s_next).clone()
        fake = self.features(self.predict_batch(phis, a))      # the encoder now holds the activations of the fake pass
        dfake = self._mse(fake, obs, True)
        dimg = self.obs_encoder.backward(dfake, need_dx=True, need_dw=False)
        dcode = self.decoder.backward(dimg)
        self.l1.backward(self._x, self._h1, dcode.contiguous(), need_dx=False, act="relu")
        self.step.add_(1)
        _adam_jobs(self._opt, self.step, self.lr if lr is None else lr)
        return self.loss_out


N_L3 = 10                                                       # Conv_model.py:256


def _adam_for(layer) -> _Adam:
    """Adam slots for a dense layer (used from canon alone), a decoder layer or a convolution (canon + images)."""
    if hasattr(layer, "wfrag"):
        o = _Adam(layer, layer.map_f, layer.wfrag)
        o.map_b, o.image_b = layer.map_dx, layer.wdx
        return o
    if hasattr(layer, "wmat"):
        return _Adam(layer, layer.map_w, layer.wmat)
    return _Adam(layer)


def _adam_jobs(opts, step: torch.Tensor, lr: float):
    jobs = []
    for o in opts:
        j = o.job(step, lr)
        if getattr(o, "image_b", None) is not None:
            j.map_b, j.wb = o.map_b.data_ptr(), o.image_b.data_ptr()
        jobs.append(j)
    for i in range(0, len(jobs), 4):
        launch_apply(jobs[i:i + 4])


class PixelDiscriminator:
    """`Discriminator` of the pixel variant in 'full_prediction' mode (Conv_model.py:159-340, wiring
    Curious_GAN_mario.py:340-346):

        D(h, a, g) = sigmoid(dense1(relu(dense10(relu(h w1_s + g w1_G + a w1_a + b1)))))          Conv_model.py:225-268
        D_real on (phis(s), a, observation = E(single s')), D_fake on (phis(s), a, fake_observation = E(image))
        D_loss = mean CE(real, 1) + mean CE(fake, 0); Adam on the discriminator's, the state encoder's and the
        single-frame encoder's variables (Conv_model.py:192-194,207-223).

    The generated image enters `learn` as data (Conv_model.py:275-276), so nothing flows to the generator.  The
    10- and 1-unit layers are kept 12 and 4 wide (16-byte rows); the extra units have zero weights and receive
    zero gradients."""

    def __init__(self, a_dim: int, learning_rate: float, state_encoder: ConvEncoder, obs_encoder: ConvEncoder,
                 rng: Optional[np.random.RandomState] = None, params: Optional[dict] = None):
        rng = rng or np.random.RandomState()
        self.h_dim, self.obs_dim, self.a_dim, self.lr = state_encoder.features, obs_encoder.features, a_dim, learning_rate
        k_in = self.h_dim + self.obs_dim + a_dim
        self.k_pad = (k_in + 3) // 4 * 4
        if params is None:                                      # xavier_initializer, biases 0.001 (Conv_model.py:231-232)
            xav = lambda i, o: rng.uniform(-np.sqrt(6.0 / (i + o)), np.sqrt(6.0 / (i + o)), (i, o))
            params = dict(w1_G=xav(self.obs_dim, N_L1), w1_s=xav(self.h_dim, N_L1), w1_a=xav(a_dim, N_L1), b1=np.full(N_L1, 0.001),
                          w3=xav(N_L1, N_L3), b3=np.full(N_L3, 0.001), w2=xav(N_L3, 1), b2=np.full(1, 0.001))
        k1 = np.zeros((self.k_pad, N_L1))
        k1[:self.h_dim], k1[self.h_dim:self.h_dim + self.obs_dim] = params["w1_s"], params["w1_G"]
        k1[self.h_dim + self.obs_dim:k_in] = params["w1_a"]
        k3, b3 = np.zeros((N_L1, 12)), np.zeros(12)
        k3[:, :N_L3], b3[:N_L3] = params["w3"], np.asarray(params["b3"]).reshape(-1)
        k2, b2 = np.zeros((12, 4)), np.zeros(4)
        k2[:N_L3, :1], b2[:1] = params["w2"], np.asarray(params["b2"]).reshape(-1)
        self.l1 = DenseLayer(self.k_pad, N_L1, kernel=k1, bias=np.asarray(params["b1"]).reshape(-1))
        self.l3 = DenseLayer(N_L1, 12, kernel=k3, bias=b3)
        self.l2 = DenseLayer(12, 4, kernel=k2, bias=b2)
        self.state_encoder, self.obs_encoder = state_encoder, obs_encoder
        self.step = torch.zeros(1, dtype=torch.int32, device=device())
        self.losses = torch.zeros(2, dtype=torch.float32, device=device())     # D_loss_real, D_loss_fake
        trained = [self.l1, self.l3, self.l2] + state_encoder.layers + [state_encoder.head] + obs_encoder.layers + [obs_encoder.head]
        self._opt = [_adam_for(L) for L in trained]

    def _rows(self, phis: torch.Tensor, feats: torch.Tensor, a: torch.Tensor, copies: int) -> torch.Tensor:
        """[copies * n, k_pad] rows [phis, feats, a, 0]; phis and a repeat for every copy, feats is [copies * n, obs_dim]."""
        n = phis.shape[0]
        x = torch.empty((copies * n, self.k_pad), dtype=torch.float32, device=phis.device)
        pad = self.k_pad - self.h_dim - self.obs_dim - self.a_dim
        ldf = row_ld(feats)
        for c in range(copies):                                  # one launch per copy: [phis | this copy's features | a | 0]
            check(lib.rlgan_concat_cols(phis.data_ptr(), row_ld(phis), self.h_dim, feats.data_ptr() + 4 * c * n * ldf, ldf, self.obs_dim,
                                        a.data_ptr(), row_ld(a), self.a_dim, x.data_ptr() + 4 * c * n * self.k_pad, self.k_pad, n, pad,
                                        _stream()), "rlgan_concat_cols")
        return x

    def _net(self, x: torch.Tensor):
        h1 = self.l1.forward(x, act="relu")
        h3 = self.l3.forward(h1, act="relu")
        return h1, h3, self.l2.forward(h3, act=False)

    def determine_batch(self, s, a: torch.Tensor, g_image: torch.Tensor) -> torch.Tensor:
        """1 - D_fake for a batch, the intrinsic reward (Conv_model.py:296-303): [n, 1]."""
        phis = self.state_encoder.forward(s)
        fobs = self.obs_encoder.forward(g_image.contiguous())
        _, _, logits = self._net(self._rows(phis, fobs, a.contiguous(), 1))
        out = torch.empty((logits.shape[0], 1), dtype=torch.float32, device=logits.device)
        check(lib.rlgan_one_minus_sigmoid(logits.data_ptr(), 4, logits.shape[0], out.data_ptr(), _stream()), "rlgan_one_minus_sigmoid")
        return out

    def determine(self, s, a, g) -> np.ndarray:
        """1 - D_fake as an array of shape (1, 1) for one state [h, w, c], action and generated frame [42, 42, 1]
        (Conv_model.py:283-294)."""
        dev = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32)[None], device=device())
        return self.determine_batch(dev(s), dev(a), dev(g)).cpu().numpy()

    def observe_and_compare(self, s_, g) -> float:
        """The hidden-state mode's squared feature error; 0 in 'full_prediction' mode, the one built here
        (Conv_model.py:305-313)."""
        return 0.

    def _forward_pair(self, s, a, single_s_next, g_image, phis=None, obs_set=None):
        n = a.shape[0]
        if phis is None:                                         # else: the state encoder still holds this batch's activations
            phis = self.state_encoder.forward(s)
        if obs_set is not None:                                  # both halves are already encoded: real rows first, fake rows after
            feats = self.obs_encoder.use(obs_set)
            assert feats.shape[0] == 2 * n
        else:
            frames = torch.cat([single_s_next, g_image], 0).contiguous()
            feats = self.obs_encoder.forward(frames)
        x = self._rows(phis, feats, a.contiguous(), 2)
        h1, h3, logits = self._net(x)
        return n, x, h1, h3, logits

    def eval(self, g_image, single_s_next, s, a) -> float:
        """D_loss (Conv_model.py:328-340)."""
        n, _, _, _, logits = self._forward_pair(s, a, single_s_next, g_image)
        check(lib.rlgan_sigmoid_ce(logits.data_ptr(), 4, n, n, self.losses.data_ptr(), None, _stream()), "rlgan_sigmoid_ce")
        real, fake = self.losses.tolist()
        return real + fake

    def learn(self, g_image, single_s_next, s, a, lr: Optional[float] = None, phis: Optional[torch.Tensor] = None,
              obs_set=None) -> torch.Tensor:
        """One Adam step (Conv_model.py:270-281); returns (D_loss_real, D_loss_fake) before the step (device).
        phis: features of `s` from a forward pass of the state encoder that is still its latest one (saves the pass).
        obs_set: an activation set of the single-frame encoder (`ConvEncoder.new_activation_set(2n)`) whose rows 0..n hold
        the pass over single_s_next and rows n..2n the pass over g_image, both with the encoder's current variables."""
        n, x, h1, h3, logits = self._forward_pair(s, a, single_s_next, g_image, phis, obs_set)
        dlogits = torch.empty_like(logits)
        check(lib.rlgan_sigmoid_ce(logits.data_ptr(), 4, n, n, self.losses.data_ptr(), dlogits.data_ptr(), _stream()), "rlgan_sigmoid_ce")
        dh3 = self.l2.backward(h3, logits, dlogits, act=False)
        dh1 = self.l3.backward(h1, h3, dh3, act="relu")
        dx = self.l1.backward(x, h1, dh1, act="relu")
        dfeats = torch.empty((2 * n, self.obs_dim), dtype=torch.float32, device=dx.device)
        check(lib.rlgan_copy_cols(dx.data_ptr() + 4 * self.h_dim, self.k_pad, dfeats.data_ptr(), self.obs_dim, 2 * n, self.obs_dim, 0,
                                  _stream()), "rlgan_copy_cols")
        dphis = torch.empty((n, self.h_dim), dtype=torch.float32, device=dx.device)
        check(lib.rlgan_copy_cols(dx.data_ptr(), self.k_pad, dphis.data_ptr(), self.h_dim, n, self.h_dim, 0, _stream()), "rlgan_copy_cols")
        check(lib.rlgan_add_cols(dx.data_ptr() + 4 * n * self.k_pad, self.k_pad, dphis.data_ptr(), self.h_dim, n, self.h_dim, _stream()),
              "rlgan_add_cols")
        self.obs_encoder.backward(dfeats)
        self.state_encoder.backward(dphis)
        self.step.add_(1)
        _adam_jobs(self._opt, self.step, self.lr if lr is None else lr)
        return self.losses


class PixelGanTrainer:
    """The GAN half of the pixel variant's training iteration on one minibatch (Curious_GAN_mario.py:85-93):

        b_s, b_a, b_s_ = M.sample(BATCH_SIZE)
        generator.learn(b_s, b_a, lr, b_s_);  b_g = generator.predict_batch(b_s, b_a)
        discriminator.learn(b_g, b_s_[:, :, :, [-1]], b_s, b_a, lr)

    with the frames in a uint8 `FrameMemory` (read in place by the state encoder), the actions in a float
    `DeviceMemory` with the same slots, and the minibatch named by an index array on the device.  `capture()`
    records the iteration as one HIP graph: at the reference's batch of 32 the ~100 launches cost more than
    their kernels."""

    def __init__(self, frames, actions, generator: PixelGenerator, discriminator: PixelDiscriminator, batch: int,
                 a_pred=None, actor=None, critic=None):
        """With a_pred / critic / actor the iteration goes on as the reference's does (Curious_GAN_mario.py:96-98):

            A_pred.learn(b_s, b_s_, b_a);  critic.learn(b_s, b_a, b_curious_r, b_s_);  actor.learn(b_s)

        The rows of `actions` are then [a (a_dim), r, curious_r]; the critic's reward is the last column."""
        self.frames, self.actions, self.G, self.D, self.batch = frames, actions, generator, discriminator, int(batch)
        self.a_pred, self.actor, self.critic = a_pred, actor, critic
        assert (actor is None) == (critic is None)
        assert critic is None or actions.dims >= discriminator.a_dim + 2
        self.idx = torch.zeros(self.batch, dtype=torch.int64, device=device())
        self._idx2 = torch.zeros(2 * self.batch, dtype=torch.int64, device=device())
        self._ob_set = None
        self._graph = None

    def sample(self) -> torch.Tensor:
        """np.random.choice(len, n) of the reference's replay (Conv_model.py:147) into the step's index array."""
        return self.frames.sample_indices(self.batch, out=self.idx)

    def _body(self) -> None:
        F, G, D = self.frames, self.G, self.D
        ref = F.ref(self.idx, "s")
        phis = D.state_encoder.forward(ref)
        rows = self.actions.gather(self.idx)
        a = rows[:, :self.D.a_dim].contiguous()
        last = F.gather_channel(self.idx, -1, "s_")             # b_s_[:, :, :, [-1]], Curious_GAN_mario.py:92
        # The generator's step does not train the single-frame encoder, so its features of the real frames serve the
        # discriminator's step as well: they are encoded once, into the first half of the discriminator's batch.
        ob = D.obs_encoder
        if G.obs_encoder is not ob:                              # two encoders: every step encodes for itself
            G.learn(phis, a, last)
            D.learn(G.predict_batch(phis, a), last, ref, a, phis=phis)
        else:
            if self._ob_set is None:
                self._ob_set = ob.new_activation_set(2 * self.batch)
            obs = ob.forward_rows(last, self._ob_set, 0)
            G.learn(phis, a, last, obs=obs)
            img = G.predict_batch(phis, a)
            ob.forward_rows(img, self._ob_set, self.batch)
            D.learn(img, last, ref, a, phis=phis, obs_set=self._ob_set)
        if self.a_pred is None and self.critic is None:
            return
        if self.a_pred is not None:
            # s and s' of the minibatch as one batch of 2n frames read from the uint8 ring in place
            self.a_pred.learn(F.ref_pair(self.idx, out=self._idx2), None, a)
        if self.critic is not None:
            r = rows[:, self.D.a_dim + 1:self.D.a_dim + 2].contiguous()
            # the encoder has been trained by the two steps above: its features again, read from the uint8 ring in place
            # (bit-equal to encoding the gathered float32 frames)
            # (s and s' as one batch of 2n frames: half the launches)
            both = self.critic.state_encoder.forward(F.ref_pair(self.idx, out=self._idx2))
            phis, phis_next = both[:self.batch], both[self.batch:]
            self.critic.learn(None, a, r, None, phis=phis, phis_next=phis_next)
            self.actor.learn(None, self.critic, phis=phis)

    def capture(self) -> None:
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(2):                                   # every buffer and kernel attribute exists before recording
                self._body()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self._graph = g

    def step(self) -> None:
        """One iteration on the indices currently in `idx`; losses in `G.loss_out` and `D.losses`."""
        if self._graph is not None:
            self._graph.replay()
        else:
            self._body()
