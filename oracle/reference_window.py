"""One whole imagination window of configs[0] on the REFERENCE ITSELF (CPU), timed.  *** TEST INFRASTRUCTURE ONLY ***

The reference's own `Agent`, `WorldModelEnv`, `make_env_loop` and `ActorCritic.forward()` + `loss.backward()`
(/root/reference/src: agent.py, envs/world_model_env.py:45-139, coroutines/env_loop.py:12-74, models/actor_critic.py:75-98,
trainer.py:363-366) on the synthetic weights / initial conditions bench.py uses, imported either from the bytecode that
`oracle/make_ref.py` compiled into `oracle/_ref/src` (this is what travels to the GPU box) or, in the build container, from
/root/reference/src directly.  Used by bench.py's `cpu_baseline` leg (`kind: "reference"`) and by tests/; never by the product.
"""
import importlib.util
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_BYTECODE = os.path.join(HERE, "_ref", "src")
REF_SOURCE = "/root/reference/src"


def reference_location():
    """(path, what) of an importable reference, or (None, why not)."""
    man = os.path.join(HERE, "_ref", "MANIFEST.json")
    if os.path.isdir(REF_BYTECODE) and os.path.exists(man):
        magic = json.load(open(man)).get("magic")
        if magic == importlib.util.MAGIC_NUMBER.hex():
            return REF_BYTECODE, "oracle/_ref (bytecode of the reference, oracle/make_ref.py)"
        why = f"oracle/_ref was compiled by another interpreter (magic {magic})"
    else:
        why = "oracle/_ref not built (python oracle/make_ref.py, build container)"
    if os.path.isdir(REF_SOURCE):
        return REF_SOURCE, "/root/reference/src"
    return None, why


def _install(path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    gold = os.path.join(ROOT, "tests", "golden")
    if gold not in sys.path:
        sys.path.insert(0, gold)
    import _refimport as R

    R.install(path)
    return R


class _Loader:
    """What WorldModelEnv needs from a DataLoader (world_model_env.py:38,115-122): .batch_sampler.batch_size and batches."""

    class _BS:
        def __init__(self, b):
            self.batch_size = b

    def __init__(self, batch, seed, size):
        self.batch_sampler = self._BS(batch)
        self._args = (seed, batch, 4, size, size)

    def __iter__(self):
        from data import Batch
        from diamond_amd.testing import initial_condition_batches

        seed, b, t, h, w = self._args
        for obs, act in initial_condition_batches(seed, b, t, h=h, w=w):
            yield Batch(obs=obs, act=act, rew=None, end=None, trunc=None, mask_padding=None, info=None, segment_ids=None)


def run_window(img_size=64, threads=None, batch=16, horizon=15, windows=1, attn_depths=(0, 0, 0, 0)):
    """Builds the reference agent + imagination env and times `windows` complete windows, the first one including the reset
    (pool preload + reward/end burn-in), like bench.py's port baseline.  Returns a dict for the bench line."""
    import torch

    path, what = reference_location()
    if path is None:
        raise RuntimeError(what)
    R = _install(path)
    if threads:
        torch.set_num_threads(threads)
    from agent import Agent
    from envs import WorldModelEnv, WorldModelEnvConfig
    from models.actor_critic import ActorCriticLossConfig
    from models.diffusion import DiffusionSamplerConfig, SigmaDistributionConfig

    from diamond_amd.testing import fill_module_

    agent = Agent(R.default_agent_config(num_actions=4, img_size=img_size, denoiser_attn_depths=tuple(attn_depths)))
    fill_module_(agent, 0)
    torch.manual_seed(1)
    t0 = time.perf_counter()
    env = WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(batch, 5, img_size),
                        WorldModelEnvConfig(horizon=horizon, num_batches_to_preload=1,
                                            diffusion_sampler=DiffusionSamplerConfig(num_steps_denoising=3)))
    agent.setup_training(SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                         ActorCriticLossConfig(backup_every=horizon, gamma=0.985, lambda_=0.95, weight_value_loss=1.0,
                                               weight_entropy_loss=0.001), env)
    ac = agent.actor_critic
    per = []
    for _ in range(windows):
        t1 = time.perf_counter()
        loss, _ = ac()
        loss.backward()
        ac.zero_grad()
        per.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    return {"value": batch * horizon * windows / dt, "unit": "imagined frames/s", "cores": torch.get_num_threads(), "kind": "reference",
            "sample": f"the reference's own code ({what}): {windows} whole window(s), B={batch}, reset + {horizon} imagined "
                      f"steps (3 Euler denoise + rew/end + actor-critic) + AC backward, {img_size}x{img_size}, torch-CPU fp32 "
                      f"(unbiased synthetic end-logits: includes mid-window resets / burn-in), {dt:.1f}s",
            "seconds_per_window": per, "loss": float(loss.detach())}


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--windows", type=int, default=1)
    ap.add_argument("--horizon", type=int, default=15)
    ap.add_argument("--attn-depths", type=str, default="0,0,0,0")
    a = ap.parse_args()
    sys.dont_write_bytecode = True
    print(json.dumps(run_window(a.img_size, a.threads or None, a.batch, a.horizon, a.windows, tuple(int(v) for v in a.attn_depths.split(",")))))
