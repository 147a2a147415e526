"""Workload for the rocprofv3 --pmc passes: a calibration copy of known size, then denoiser
forwards at the bench shape (B=256, 64x64).  See tools/pmc_collect.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_agent
from diamond_amd.testing import synthetic_actions, synthetic_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
agent = build_agent(dev, 64, 0)
g = torch.Generator().manual_seed(0)
obs = synthetic_frames(g, B, 12, 64, 64).to(dev)
act = synthetic_actions(g, 4, B, 4).to(dev)
x = torch.randn(B, 3, 64, 64, device=dev)
# calibration: float4 copy kernel reading and writing exactly 1 GiB each (dmd_nchw_to_nhwc with C == CPad == 4
# is a plain strided gather; use torch's copy kernel instead: contiguous -> contiguous clone)
cal = torch.randn(256 * 1024 * 1024, device=dev)
for _ in range(2):
    cal2 = cal.clone()
torch.cuda.synchronize()
for _ in range(2):
    agent.denoiser.denoise(x, 1.0, obs, act)
torch.cuda.synchronize()
print("done")
