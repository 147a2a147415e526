"""B=1 latency of the interactive world-model env: `python bench.py --config latency` (kept as an alias).
    python tools/latency_bench.py [frames]       -> one JSON line"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "200"
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "latency", "--steps", n]))
