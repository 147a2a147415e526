"""Same-process, same-box A/B of the env loop under episodes that END (measurement scaffolding, not product).

Arms: this tree's slots loop (round 6: a step's deaths resolved on the device, DIAMOND_ENV_LOOP=slots), the reference's sequential
order of calls (DIAMOND_ENV_LOOP=sequential) and -- if its files are there -- the ROUND-4 loop + env (the round-5 pipelined loop,
host-planned resets + speculation, was removed after profiles/r06c_ab_env_loop.txt: the slots loop beat it in every regime but p = 0.5)
(diamond_amd/ablate/r04_{env_loop,world_model_env}.py = `git show d83241c:diamond_amd/...`, git-ignored, loaded beside the
current modules).  Regimes: end probability p per env-step through the synthetic reward/end head, from synchronised episodes
("lockstep") or from episode lengths spread over the horizon ("steady": the state the reference's training loop converges
to, batch / horizon truncations at every step).  One agent, one process; per (arm, regime): 1 warm-up + 3 timed windows,
arms interleaved per regime, the whole table twice.  Prints one JSON object.

    python tools/ab_env_loop.py > gpurun_out/r05/ab_env_loop.json
"""
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(f"diamond_amd.{name}", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"diamond_amd.{name}"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    import diamond_amd as D
    from bench import _Loader, build_agent, measured_windows, set_end_rate
    from diamond_amd import env_loop as EL

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    agent = build_agent(dev, 64, 0)
    ac = agent.actor_critic
    ac.loss_cfg = D.ActorCriticLossConfig(backup_every=15, gamma=0.985, lambda_=0.95, weight_value_loss=1.0, weight_entropy_loss=0.001)
    opt = torch.optim.AdamW(ac.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.0)
    cfg = D.WorldModelEnvConfig(horizon=15, num_batches_to_preload=2, diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3))
    abl = os.path.join(ROOT, "diamond_amd", "ablate")
    legacy = os.path.exists(os.path.join(abl, "r04_env_loop.py"))
    if legacy:
        L_env = _load("r04_world_model_env", os.path.join(abl, "r04_world_model_env.py"))
        L_loop = _load("r04_env_loop", os.path.join(abl, "r04_env_loop.py"))

    arms = {}

    def add(name, env_cls, make_loop, environ):
        saved = {k: os.environ.get(k) for k in environ}
        os.environ.update({k: v for k, v in environ.items() if v is not None})
        env = env_cls(agent.denoiser, agent.rew_end_model, _Loader(256, 100, 64), cfg)
        loop = make_loop(env, ac, expo_fn=lambda l: None)
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        arms[name] = (env, loop, environ)

    add("slots loop (round 6: deaths resolved on the device)", D.WorldModelEnv, EL.make_env_loop, {"DIAMOND_ENV_LOOP": "slots"})
    add("sequential (reference order of calls)", D.WorldModelEnv, EL.make_env_loop, {"DIAMOND_ENV_LOOP": "sequential"})
    if legacy:
        add("round-4 loop", L_env.WorldModelEnv, L_loop.make_env_loop, {})

    def window_of(name):
        env, loop, environ = arms[name]

        def window():
            os.environ.update(environ)
            ac.env_loop = loop
            loss, _ = ac()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ac.parameters(), 100.0)
            opt.step()
            opt.zero_grad(set_to_none=False)
            for k in environ:
                os.environ.pop(k, None)
        return window

    regimes = [("no ends", None, False), ("lockstep p=0.003", 0.003, False), ("lockstep p=0.01", 0.01, False),
               ("steady p=0", 1e-9, True), ("steady p=0.003", 0.003, True), ("steady p=0.01", 0.01, True), ("lockstep p=0.5", 0.5, False)]
    for name in arms:  # first windows: resets, caches
        w = window_of(name)
        w(), w()
    torch.cuda.synchronize()
    out = {"rounds": []}
    for rnd in range(2):
        table = {}
        for rname, p, stagger in regimes:
            set_end_rate(agent, p)
            for name in arms:
                env = arms[name][0]
                ep = (torch.arange(256) % 15) if stagger else torch.zeros(256, dtype=torch.long)
                if hasattr(env, "reset_statistics"):
                    env.reset_statistics()
                if hasattr(env, "set_episode_lengths"):
                    env.set_episode_lengths(ep)
                else:
                    env.ep_len = ep.to(dev)
                w = window_of(name)
                before = dict(getattr(env, "stats", {}))
                dt, ms = measured_windows(w, 3, 1)
                st = {k: v - before.get(k, 0) for k, v in getattr(env, "stats", {}).items()}
                table.setdefault(rname, {})[name] = {"fps": round(256 * 15 / dt), "step_ms": ms, "stats": st}
                print(f"[{rnd}] {rname:18s} {name:46s} {256 * 15 / dt:8.0f} fps  {ms}  {st}", file=sys.stderr, flush=True)
        out["rounds"].append(table)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
