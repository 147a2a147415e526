"""Network configurations WIDER than (and off the channel grid of) the reference's one published configuration
(config/agent/default.yaml: 64 / 32 channels everywhere): the reference's constructors take any `channels` / `depths` lists
(models/blocks.py:183-222, rew_end_model.py:93-133, actor_critic.py:101-113).  Shared by tests/golden/make_golden.py --wide (which
runs the reference on them) and tests/test_wide_configs.py.  Small on purpose: three levels at 32x32, so that the SIMT interpreter
runs a forward + backward in seconds; the widths are chosen to hit every host-side fallback -- more than 256 input channels in one
convolution (two sources, and one source on its own), weight-gradient shapes the kernel has no instance for (160 -> 160, 640 -> 320,
96 -> 160, 64 -> 3 ...), GroupNorm backward over 96 / 160 / 320 channels, attention at 160 and 320 channels."""

WEIGHT_SEED = 3
DENOISER = dict(img_channels=3, num_steps_conditioning=4, cond_channels=256, depths=[1, 1, 1], channels=[64, 160, 320],
                attn_depths=[0, 1, 1], num_actions=4)
REW_END = dict(lstm_dim=512, img_channels=3, img_size=32, cond_channels=128, depths=[1, 1, 1], channels=[64, 96, 320],
               attn_depths=[0, 0, 1], num_actions=4)
ACTOR_CRITIC = dict(lstm_dim=512, img_channels=3, img_size=32, channels=[32, 64, 96, 160], down=[1, 1, 1, 1], num_actions=4)
SIZE = 32
SIGMA_DIST = dict(loc=-0.4, scale=1.2, sigma_min=2e-3, sigma_max=20)
GRAD_STRIDE = 97  # gradients of more than 4096 elements are stored as every 97th element (plus every tensor's norm)


def sample_grad(g):
    return g if g.numel() <= 4096 else g.flatten()[::GRAD_STRIDE].clone()


def agent_config(AgentConfig, DenoiserConfig, InnerModelConfig, RewEndModelConfig, ActorCriticConfig):
    """The three networks above as one AgentConfig, from the config classes of either package (the reference's or diamond_amd's)"""
    strip = lambda d: {k: v for k, v in d.items() if k != "num_actions"}
    return AgentConfig(denoiser=DenoiserConfig(inner_model=InnerModelConfig(**strip(DENOISER)), sigma_data=0.5, sigma_offset_noise=0.3),
                       rew_end_model=RewEndModelConfig(**strip(REW_END)), actor_critic=ActorCriticConfig(**strip(ACTOR_CRITIC)), num_actions=4)
