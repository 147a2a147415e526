"""Agent container (reference agent.py:15-62): same children names, config propagation,
`setup_training` and prefix-split `load`."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from pathlib import Path

import torch
from torch import nn

from .actor_critic import ActorCritic, ActorCriticConfig, ActorCriticLossConfig
from .denoiser import Denoiser, DenoiserConfig, SigmaDistributionConfig
from .rew_end_model import RewEndModel, RewEndModelConfig


def extract_state_dict(state_dict, module_name: str) -> OrderedDict:
    return OrderedDict({k.split(".", 1)[1]: v for k, v in state_dict.items() if k.startswith(module_name)})


@dataclass
class AgentConfig:
    denoiser: DenoiserConfig
    rew_end_model: RewEndModelConfig
    actor_critic: ActorCriticConfig
    num_actions: int

    def __post_init__(self) -> None:
        self.denoiser.inner_model.num_actions = self.num_actions
        self.rew_end_model.num_actions = self.num_actions
        self.actor_critic.num_actions = self.num_actions


class Agent(nn.Module):
    def __init__(self, cfg: AgentConfig) -> None:
        super().__init__()
        self.denoiser = Denoiser(cfg.denoiser)
        self.rew_end_model = RewEndModel(cfg.rew_end_model)
        self.actor_critic = ActorCritic(cfg.actor_critic)

    @property
    def device(self):
        return self.denoiser.device

    def setup_training(self, sigma_distribution_cfg: SigmaDistributionConfig, actor_critic_loss_cfg: ActorCriticLossConfig,
                       rl_env) -> None:
        self.denoiser.setup_training(sigma_distribution_cfg)
        self.actor_critic.setup_training(rl_env, actor_critic_loss_cfg)

    def load(self, path_to_ckpt: Path, load_denoiser: bool = True, load_rew_end_model: bool = True,
             load_actor_critic: bool = True) -> None:
        sd = torch.load(Path(path_to_ckpt), map_location=self.device)
        parts = {k: extract_state_dict(sd, k) for k in ("denoiser", "rew_end_model", "actor_critic")}
        if load_denoiser:
            self.denoiser.load_state_dict(parts["denoiser"])
        if load_rew_end_model:
            self.rew_end_model.load_state_dict(parts["rew_end_model"])
        if load_actor_critic:
            self.actor_critic.load_state_dict(parts["actor_critic"])


def default_agent_config(num_actions: int = 4, img_size: int = 64, denoiser_attn_depths=(0, 0, 0, 0)) -> AgentConfig:
    """Values of the reference's config/agent/default.yaml:1-32 (hydra is not needed)."""
    from .inner_model import InnerModelConfig

    return AgentConfig(
        denoiser=DenoiserConfig(
            inner_model=InnerModelConfig(img_channels=3, num_steps_conditioning=4, cond_channels=256, depths=[2, 2, 2, 2],
                                         channels=[64, 64, 64, 64], attn_depths=list(denoiser_attn_depths)),
            sigma_data=0.5, sigma_offset_noise=0.3),
        rew_end_model=RewEndModelConfig(lstm_dim=512, img_channels=3, img_size=img_size, cond_channels=128,
                                        depths=[2, 2, 2, 2], channels=[32, 32, 32, 32], attn_depths=[0, 0, 0, 0]),
        actor_critic=ActorCriticConfig(lstm_dim=512, img_channels=3, img_size=img_size, channels=[32, 32, 64, 64],
                                       down=[1, 1, 1, 1]),
        num_actions=num_actions)
