"""Times dmd_attention (the long-sequence split-fp16 kernel) at the shapes of configs[4]: 8 images, 64 channels = 8 heads, T tokens.
    [DIAMOND_LIB=...] python tools/attention_bench.py [T ...]   -> us per launch
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diamond_amd import engine as E  # noqa: E402


def main():
    ts = [int(a) for a in sys.argv[1:]] or [4096, 1024]
    for t in ts:
        n, c = 8, 64
        g = torch.Generator().manual_seed(t)
        qkv = E.Act(torch.randn(n, t, 1, 3 * c, generator=g).cuda().view(n, int(t ** 0.5), int(t ** 0.5), 3 * c).contiguous())
        for _ in range(5):
            out = E.attention(qkv, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            out = E.attention(qkv, c)
        e1.record()
        torch.cuda.synchronize()
        print(f"T={t}: {1e3 * e0.elapsed_time(e1) / reps:.1f} us per launch (N={n}, C={c}); checksum {float(out.double().sum()):.6f}")


if __name__ == "__main__":
    main()
