"""rl-gan_amd: MI355X-native hot path of RL-GAN (DDPG + Curious_GAN on the 2-link arm).

Python mirrors the reference's classes (`Actor`, `Critic`, `Generator`, `Discriminator`, `Memory`,
`ArmEnv`); all arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
librlgan_amd.so (include/rlgan_amd.h).  Importing this package loads that library and fails if it
has not been built.
"""
from . import _lib                                   # noqa: F401  (loads librlgan_amd.so or raises)
from .session import Session, placeholder, set_random_seed, TensorHandle, Placeholder
from .ddpg import Actor, Critic
from .ddpg import Memory as DDPGMemory
from .curious_gan import Generator, Discriminator, Memory
from .arm_env import ArmEnv, VecArmEnv
from .replay import DeviceMemory, FrameMemory, MT19937Device, PixelMemory
from .engine import StepEngine, GradExchange, ActorState, CriticState, AdamState
from .core import DeviceNet, Workspace, mlp_forward
from .learner import CuriousLearner, LearnerConfig
from .conv import ConvEncoder, ConvLayer, DenseLayer, FrameRef, DOOM_HEAD, universe_head
from .deconv import Decoder, DeconvLayer, DECODER
from .pixel_gan import PixelDiscriminator, PixelGanTrainer, PixelGenerator, single_frame_encoder
from .inverse_model import BatchNorm, StateActionPredictor
from .pixel_ddpg import LSTMUnit, PixelActor, PixelCritic
from .train_pixel import PixelAgent, PixelSettings, train_pixel, transfer_sparse_action
from .train import Agent, Settings, train, train_DDPG

__all__ = ["Session", "placeholder", "set_random_seed", "Actor", "Critic", "Generator", "Discriminator", "Memory",
           "DDPGMemory", "ArmEnv", "VecArmEnv", "DeviceMemory", "FrameMemory", "PixelMemory", "PixelAgent", "PixelSettings", "train_pixel", "transfer_sparse_action", "FrameRef", "Decoder", "DeconvLayer", "DECODER", "PixelGenerator", "PixelDiscriminator", "PixelGanTrainer", "single_frame_encoder", "StateActionPredictor", "PixelActor", "LSTMUnit", "PixelCritic", "BatchNorm", "MT19937Device", "StepEngine", "GradExchange",
           "CuriousLearner", "LearnerConfig", "DeviceNet", "mlp_forward", "ConvEncoder", "ConvLayer", "DenseLayer", "DOOM_HEAD",
           "universe_head", "Agent", "Settings", "train", "train_DDPG"]
