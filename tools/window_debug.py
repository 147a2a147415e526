"""Run bench.py's window step by step with timestamps + a watchdog stack dump (development aid)."""
import faulthandler, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.dump_traceback_later(45, repeat=True, file=sys.stdout)
import torch
import diamond_amd as D
from bench import build_agent, _Loader
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
T0 = time.perf_counter()
def mark(s):
    torch.cuda.synchronize(); print(f"[{time.perf_counter()-T0:7.2f}s] {s}", flush=True)
agent = build_agent(dev, 64, 0); mark("agent built")
env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(B, 100, 64),
                      D.WorldModelEnvConfig(horizon=15, num_batches_to_preload=2,
                                            diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))
agent.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                     D.ActorCriticLossConfig(backup_every=15, gamma=0.985, lambda_=0.95, weight_value_loss=1.0,
                                             weight_entropy_loss=0.001), env)
ac = agent.actor_critic
opt = torch.optim.AdamW(ac.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.0)
mark("setup done")
for w in range(3):
    loss, metrics = ac(); mark(f"window {w}: forward done, loss {float(loss):.4f}")
    loss.backward(); mark(f"window {w}: backward done")
    torch.nn.utils.clip_grad_norm_(ac.parameters(), 100.0); mark("clip")
    opt.step(); opt.zero_grad(set_to_none=False); mark("opt step")
