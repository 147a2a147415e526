"""Per-stage wall timing of the imagined-rollout path on one GPU (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t00 = time.perf_counter()
import torch
print(f"import torch {time.perf_counter()-t00:.1f}s", flush=True)
import diamond_amd as D
from diamond_amd import engine as E
from bench import build_agent, _Loader

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
def sync(): torch.cuda.synchronize()
def timed(name, fn, n=1):
    sync(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    sync(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.1f} ms", flush=True)
    return r
agent = build_agent(dev, 64, 0)
from diamond_amd.testing import synthetic_frames, synthetic_actions
g = torch.Generator().manual_seed(0)
obs = synthetic_frames(g, B, 4, 3, 64, 64).to(dev); act = synthetic_actions(g, 4, B, 4).to(dev)
x = torch.randn(B, 3, 64, 64, device=dev)
den = agent.denoiser
timed("denoise first (incl. weight packing)", lambda: den.denoise(x, 1.0, obs.reshape(B, 12, 64, 64), act))
timed("denoise", lambda: den.denoise(x, 1.0, obs.reshape(B, 12, 64, 64), act), 3)
# host-only launch cost: enqueue without sync
t0 = time.perf_counter(); den.denoise(x, 1.0, obs.reshape(B, 12, 64, 64), act); t1 = time.perf_counter(); sync()
print(f"denoise host enqueue time: {(t1-t0)*1e3:.1f} ms", flush=True)
nv.PROFILER = E.LaunchProfiler()
den.denoise(x, 1.0, obs.reshape(B, 12, 64, 64), act)
for k, v in sorted(nv.PROFILER.summary().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k}: {v['launches']} launches {v['ms']:.2f} ms  {v['flops']/v['ms']/1e9:.1f} TF/s  {v['bytes']/v['ms']/1e6:.0f} GB/s", flush=True)
nv.PROFILER = None
sampler = D.DiffusionSampler(den, D.DiffusionSamplerConfig(num_steps_denoising=3))
timed("sample (3 Euler)", lambda: sampler.sample(obs, act), 2)
rem = agent.rew_end_model
nx = x.clamp(-1, 1)
timed("rew_end first", lambda: rem.predict_rew_end(obs[:, -1:], act[:, -1:], nx.unsqueeze(1)))
timed("rew_end", lambda: rem.predict_rew_end(obs[:, -1:], act[:, -1:], nx.unsqueeze(1)), 3)
ac = agent.actor_critic
timed("ac fwd first", lambda: ac.predict_act_value(nx, None))
timed("ac fwd", lambda: ac.predict_act_value(nx, None), 3)
def fb():
    o = ac.predict_act_value(nx, None); (o.logits_act.sum() + o.val.sum()).backward()
timed("ac fwd+bwd", fb, 3)
env = D.WorldModelEnv(den, rem, _Loader(B, 100, 64), D.WorldModelEnvConfig(horizon=15, num_batches_to_preload=2,
      diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))
timed("env.reset (pool preload)", lambda: env.reset())
a = torch.zeros(B, dtype=torch.long, device=dev)
timed("env.step", lambda: env.step(a), 3)
