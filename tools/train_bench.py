"""Throughput of the denoiser TRAINING step: Denoiser.forward(batch) + loss.backward() + AdamW at the
reference's training shape (config/trainer.yaml: batch 32, segments of 4 conditioning + 1 predicted frame), 64x64.

    python tools/train_bench.py [batch] [steps]      -> one JSON line"""
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import diamond_amd as D
from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
agent = D.Agent(D.default_agent_config())
fill_module_(agent, 0)
den = agent.denoiser.to(dev).train()
den.setup_training(D.SigmaDistributionConfig(loc=-0.4, scale=1.2, sigma_min=2e-3, sigma_max=20))
opt = torch.optim.AdamW(den.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
t = 5
batch = SimpleNamespace(obs=synthetic_frames(g, b, t, 3, 64, 64).to(dev), act=synthetic_actions(g, 4, b, t).to(dev),
                        mask_padding=torch.ones(b, t, dtype=torch.bool, device=dev))


def step():
    loss, _ = den(batch)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(den.parameters(), 1.0)
    opt.step()
    opt.zero_grad()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"metric": "denoiser training step (forward + backward + clip + AdamW), 64x64, 1 predicted frame per segment",
                  "batch": b, "ms_per_step": 1e3 * dt, "frames_per_s": b / dt, "loss": float(loss.detach()),
                  "algorithmic_tflops": 3 * 6.0909e9 * b / dt / 1e12}))
