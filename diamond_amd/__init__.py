"""diamond_amd -- MI355X-native implementation of DIAMOND's imagined-rollout hot path.

Public surface mirrors the reference's Python API (Agent / WorldModelEnv / DiffusionSampler /
Denoiser / ActorCritic / RewEndModel and their config dataclasses); the arithmetic lives in
diamond_amd/libdiamond_hip.so (hand-written HIP for gfx950, C ABI in include/diamond_hip.h).
"""
from .actor_critic import ActorCritic, ActorCriticConfig, ActorCriticLossConfig, compute_lambda_returns
from .agent import Agent, AgentConfig, default_agent_config
from .denoiser import Denoiser, DenoiserConfig, SigmaDistributionConfig
from .diffusion_sampler import DiffusionSampler, DiffusionSamplerConfig, build_sigmas
from .inner_model import InnerModel, InnerModelConfig
from .rew_end_model import RewEndModel, RewEndModelConfig
from .world_model_env import WorldModelEnv, WorldModelEnvConfig

__all__ = [
    "ActorCritic", "ActorCriticConfig", "ActorCriticLossConfig", "Agent", "AgentConfig", "Denoiser", "DenoiserConfig",
    "DiffusionSampler", "DiffusionSamplerConfig", "InnerModel", "InnerModelConfig", "RewEndModel", "RewEndModelConfig",
    "SigmaDistributionConfig", "WorldModelEnv", "WorldModelEnvConfig", "build_sigmas", "compute_lambda_returns",
    "default_agent_config",
]
