"""Import the upstream DIAMOND reference (read-only, /root/reference) with stub modules.

Only used by `make_golden.py` (fixture generation, build container only), by the optional
cross-checks in tests that skip when /root/reference is absent (it does not exist on the
GPU box), and by `oracle/reference_window.py` (bench.py's CPU baseline on the reference's
own bytecode, oracle/_ref).  Nothing in the product path imports this.

The stubs cover third-party packages that are not installable here:
omegaconf, wandb, gymnasium, ale_py, cv2, torcheval.
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isdir(REF_SRC)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(src=None):
    """Register stubs and put the reference `src/` on sys.path. Idempotent.  `src`: where the reference's modules are
    (default /root/reference/src; oracle/reference_window.py passes oracle/_ref/src, their bytecode, on the GPU box)."""
    global REF_SRC
    if src is not None:
        REF_SRC = src
    if "agent" in sys.modules and (getattr(sys.modules["agent"], "__file__", None) or "").startswith(REF_SRC):
        return
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    _mod("omegaconf", OmegaConf=_Any(), DictConfig=dict)
    _mod("wandb")
    _mod("ale_py")
    _mod("cv2")

    class _Env:
        pass

    class _Wrapper:
        pass

    class _RCA:
        def __init__(self, *a, **k):
            pass

    g = _mod("gymnasium", Env=_Env, Wrapper=_Wrapper, make=_Any())
    g.vector = _mod("gymnasium.vector", AsyncVectorEnv=_Any)
    g.core = _mod("gymnasium.core", Env=_Env, Wrapper=_Wrapper, WrapperActType=object, WrapperObsType=object)
    g.spaces = _mod("gymnasium.spaces", Box=_Any)
    g.utils = _mod("gymnasium.utils", RecordConstructorArgs=_RCA)
    te = _mod("torcheval")
    te.metrics = _mod("torcheval.metrics")

    def multiclass_confusion_matrix(input, target, num_classes):
        """torcheval (absent here) semantics as published: [i, j] = #samples of true class i predicted argmax j."""
        import torch

        pred = input.argmax(dim=1) if input.ndim == 2 else input
        return torch.bincount(target.long() * num_classes + pred, minlength=num_classes ** 2).reshape(num_classes, num_classes)

    te.metrics.functional = _mod("torcheval.metrics.functional", multiclass_confusion_matrix=multiclass_confusion_matrix)
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)


def default_agent_config(num_actions=4, img_size=64, denoiser_attn_depths=(0, 0, 0, 0)):
    """AgentConfig with the values of config/agent/default.yaml:1-32 (hydra is absent)."""
    install()
    from agent import AgentConfig
    from models.actor_critic import ActorCriticConfig
    from models.diffusion import DenoiserConfig, InnerModelConfig
    from models.rew_end_model import RewEndModelConfig

    return AgentConfig(
        denoiser=DenoiserConfig(
            inner_model=InnerModelConfig(
                img_channels=3, num_steps_conditioning=4, cond_channels=256,
                depths=[2, 2, 2, 2], channels=[64, 64, 64, 64], attn_depths=list(denoiser_attn_depths),
            ),
            sigma_data=0.5, sigma_offset_noise=0.3,
        ),
        rew_end_model=RewEndModelConfig(
            lstm_dim=512, img_channels=3, img_size=img_size, cond_channels=128,
            depths=[2, 2, 2, 2], channels=[32, 32, 32, 32], attn_depths=[0, 0, 0, 0],
        ),
        actor_critic=ActorCriticConfig(
            lstm_dim=512, img_channels=3, img_size=img_size, channels=[32, 32, 64, 64], down=[1, 1, 1, 1],
        ),
        num_actions=num_actions,
    )
