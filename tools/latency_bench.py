"""B=1 latency of the interactive world-model env (reference play.py:105-109, game/play_env.py:113-124): ms per
imagined frame with the diffusion sampler launched eagerly vs replayed as a captured hipGraph.

    python tools/latency_bench.py [frames]       -> one JSON line
Each frame = WorldModelEnv.step(act) (3 Euler denoising steps + reward/end model + bookkeeping) followed by a host
read of the reward, like the play loop does (`rew.item()`, play_env.py:127)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import diamond_amd as D
from bench import _Loader, build_agent


def run(graph: bool, frames: int):
    dev = torch.device("cuda:0")
    agent = build_agent(dev, 64, 0)
    env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(1, 7, 64),
                          D.WorldModelEnvConfig(horizon=1000, num_batches_to_preload=1,
                                                diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)),
                          return_denoising_trajectory=True, graph_sampler=graph)
    env.reset()
    act = torch.zeros(1, dtype=torch.long, device=dev)
    for _ in range(12):  # warm-up: caches, and in graph mode one capture per ring head
        env.step(act)
    torch.cuda.synchronize()
    t_s, t0 = 0.0, time.perf_counter()
    for i in range(frames):
        ts = time.perf_counter()
        env.predict_next_obs()
        torch.cuda.synchronize()
        t_s += time.perf_counter() - ts
    sampler_ms = 1e3 * t_s / frames
    t0 = time.perf_counter()
    for i in range(frames):
        obs, rew, end, trunc, info = env.step(act)
        float(rew.item())
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / frames, sampler_ms


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    eager, eager_s = run(False, n)
    graph, graph_s = run(True, n)
    print(json.dumps({"metric": "ms per imagined frame, B=1, 64x64, 3 Euler denoise steps (play.py latency mode)",
                      "frames": n, "eager_ms_per_frame": eager, "graph_ms_per_frame": graph, "speedup": eager / graph,
                      "sampler_only_eager_ms": eager_s, "sampler_only_graph_ms": graph_s,
                      "fps_eager": 1e3 / eager, "fps_graph": 1e3 / graph}))
