"""Throughput of the denoiser TRAINING step: `python bench.py --config train` (kept as an alias).
    python tools/train_bench.py [batch] [steps]      -> one JSON line"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = sys.argv[1] if len(sys.argv) > 1 else "32"
n = sys.argv[2] if len(sys.argv) > 2 else "20"
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "train", "--batch", b, "--steps", n]))
