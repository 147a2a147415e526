"""Where does one WorldModelEnv.step / one window go?  (development aid, syncs between pieces)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diamond_amd as D
from bench import build_agent, _Loader
B = 256
dev = torch.device("cuda:0")
agent = build_agent(dev, 64, 0)
env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(B, 100, 64),
                      D.WorldModelEnvConfig(horizon=15, num_batches_to_preload=2,
                                            diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))
agent.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                     D.ActorCriticLossConfig(backup_every=15, gamma=0.985, lambda_=0.95, weight_value_loss=1.0,
                                             weight_entropy_loss=0.001), env)
ac = agent.actor_critic
acc = {}
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)
wrap(env, "predict_next_obs", "sampler")
wrap(env, "predict_rew_end", "rew_end")
wrap(env, "reset_dead", "reset_dead")
wrap(env, "step", "env.step total")
wrap(ac, "predict_act_value", "ac fwd")
opt = torch.optim.AdamW(ac.parameters(), lr=1e-4)
for w in range(4):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _ = ac()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(ac.parameters(), 100.0); opt.step(); opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"window {w}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f}  clip+opt {1e3*(t3-t2):.1f} | " +
          "  ".join(f"{k} {1e3*v:.1f}" for k, v in acc.items()), flush=True)
