"""Per-shape timing of dmd_linear (development aid): the GEMM shapes of the imagined-rollout path at batch 256."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from diamond_amd import engine as E

SHAPES = [  # name, M, N, K
    ("FiLM table (denoiser)", 256, 7168, 256), ("FiLM table (rew/end)", 256, 1280, 128), ("cond_proj", 256, 256, 256),
    ("rew/end LSTM x-gates", 256, 2048, 2048), ("LSTM h-gates", 256, 2048, 512), ("rew/end head 0", 256, 512, 512),
    ("AC LSTM x-gates", 256, 2048, 1024), ("heads", 256, 5, 512), ("AC bwd dx", 256, 1024, 2048), ("AC bwd dhx", 256, 512, 2048),
    ("AC bwd dW_ih", 2048, 1024, 256), ("AC bwd dW_hh", 2048, 512, 256), ("AC bwd dh from heads", 256, 512, 16),
]
for name, m, n, k in SHAPES:
    a = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda")
    b = torch.randn(n, device="cuda")
    for _ in range(3):
        E.linear(a, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        E.linear(a, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:28s} M{m} N{n} K{k}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF/s", flush=True)
