"""Where a configs[1] window's wall time goes, phase by phase (development aid; adds a device sync between phases):
rollout (15 env steps incl. reset / burn-in) | actor-critic loss | backward | clip + AdamW.  Next to each phase's wall time:
the host time until its last launch was issued (host-bound phases have the two nearly equal)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import diamond_amd as D
import bench

dev = torch.device("cuda:0")
agent = bench.build_agent(dev, 64, 0)
env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, bench._Loader(256, 100, 64),
                      D.WorldModelEnvConfig(horizon=15, num_batches_to_preload=2,
                                            diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3, order=1)))
agent.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                     D.ActorCriticLossConfig(backup_every=15, gamma=0.985, lambda_=0.95, weight_value_loss=1.0, weight_entropy_loss=0.001), env)
ac = agent.actor_critic
opt = torch.optim.AdamW(ac.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.0)
from diamond_amd.actor_critic import actor_critic_loss

def phase(fn):
    t0 = time.perf_counter(); out = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return out, 1e3 * (t1 - t0), 1e3 * (t2 - t0)

tot = {}
for w in range(4):
    c = ac.loss_cfg
    roll, h, t = phase(lambda: ac.env_loop.send(c.backup_every))
    tot.setdefault("rollout", []).append((h, t))
    _, act, rew, end, trunc, logits_act, val, val_bootstrap, _ = roll
    (loss, _), h, t = phase(lambda: actor_critic_loss(logits_act, val, act, rew, end, trunc, val_bootstrap, c))
    tot.setdefault("loss", []).append((h, t))
    _, h, t = phase(lambda: loss.backward())
    tot.setdefault("backward", []).append((h, t))
    def optim():
        torch.nn.utils.clip_grad_norm_(ac.parameters(), 100.0); opt.step(); opt.zero_grad(set_to_none=False)
    _, h, t = phase(optim)
    tot.setdefault("clip+adamw", []).append((h, t))
for k, v in tot.items():
    v = v[1:]
    print(f"{k:12s} host-issue {sum(a for a, _ in v) / len(v):7.2f} ms   wall {sum(b for _, b in v) / len(v):7.2f} ms")
