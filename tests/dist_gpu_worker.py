"""Worker of tests/test_gpu_dist.py, launched with `python -m torch.distributed.run --nproc-per-node 1`: everything an
N-GPU run of the imagined-rollout path does with RCCL, at world size 1 on the one GPU of the test box --
init_process_group("nccl"), the flat parameter broadcast, the flat-bucket gradient all-reduce over the actor-critic AND over
the denoiser / reward-end parameters (the reference DDP-wraps all three, trainer.py:110 / utils.py:105-106), and
torch's own DistributedDataParallel around ActorCritic (forward = one imagined window, backward through the hooks).
Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def trainer_body(dev):
    """The loop body of the reference's Trainer.train_component("actor_critic") (trainer.py:363-382) for two optimiser steps, model =
    DDP(agent.actor_critic) as utils.py:105-106 wraps it, grad_acc_steps = 1, max_grad_norm = 100 (config/trainer.yaml)."""
    import diamond_amd as D
    from bench import _Loader, build_agent

    agent = build_agent(dev, 64, 0)
    env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(8, 100, 64),
                          D.WorldModelEnvConfig(horizon=5, num_batches_to_preload=2,
                                                diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))
    agent.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                         D.ActorCriticLossConfig(backup_every=5, gamma=0.985, lambda_=0.95, weight_value_loss=1.0, weight_entropy_loss=0.001), env)
    model = torch.nn.parallel.DistributedDataParallel(agent.actor_critic)
    try:  # the reference's own optimizer factory (utils.py:129-165), from its bytecode where that travelled
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import reference_window as RW

        path, _ = RW.reference_location()
        RW._install(path)
        from utils import configure_opt

        opt, opt_from = configure_opt(agent.actor_critic, lr=1e-4, weight_decay=0.0, eps=1e-8), "reference configure_opt"
    except Exception as e:  # noqa: BLE001
        opt, opt_from = torch.optim.AdamW(agent.actor_critic.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.0), f"torch AdamW ({e!r})"
    before = torch.cat([p.detach().reshape(-1) for p in agent.actor_critic.parameters()]).clone()
    model.train()
    opt.zero_grad()
    losses, norms, nkeys = [], [], 0
    for i in range(2):
        loss, metrics = model()
        loss.backward()
        grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 100.0)
        metrics["grad_norm_before_clip"] = grad_norm
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
        norms.append(float(grad_norm))
        nkeys = len(metrics)
    torch.cuda.synchronize()
    after = torch.cat([p.detach().reshape(-1) for p in agent.actor_critic.parameters()])
    return {"steps": 2, "losses": losses, "losses_finite": all(l == l and abs(l) < 1e30 for l in losses), "grad_norms": norms,
            "params_moved": bool((after != before).any()), "optimizer": opt_from, "metrics_keys": nkeys,
            "grads_zeroed": all(p.grad is None or not bool(p.grad.any()) for p in agent.actor_critic.parameters())}


def main():
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    if "--trainer-body" in sys.argv:
        out = trainer_body(dev)
        dist.barrier()
        dist.destroy_process_group()
        print(json.dumps(out))
        return
    import diamond_amd as D
    from bench import _Loader, build_agent
    from diamond_amd.dist import GradAllReducer, broadcast_parameters, parameter_checksum

    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    agent = build_agent(dev, 64, 0)
    cs0 = parameter_checksum(agent)
    versions = [p._version for p in agent.parameters()]
    broadcast_parameters(agent, src=0)
    out["broadcast_keeps_values"] = parameter_checksum(agent) == cs0
    out["broadcast_bumps_versions"] = all(p._version > v for p, v in zip(agent.parameters(), versions))

    # world-model parameters: a known gradient pattern through the flat bucket and RCCL
    wm = list(agent.denoiser.parameters()) + list(agent.rew_end_model.parameters())
    red_wm = GradAllReducer(wm)
    for i, p in enumerate(wm):
        p.grad.fill_(float(i % 13) - 6.0)
    flat = red_wm.all_reduce_mean()
    torch.cuda.synchronize()
    out["wm_bucket_mb"] = flat.numel() * 4 / 2 ** 20
    out["wm_allreduce_ok"] = all(bool((p.grad == float(i % 13) - 6.0).all()) for i, p in enumerate(wm))

    def make_env():
        return D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(8, 100, 64),
                               D.WorldModelEnvConfig(horizon=5, num_batches_to_preload=2,
                                                     diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)))

    ac = agent.actor_critic

    def setup(env):
        # (ActorCritic.setup_training is once-only like the reference's: a fresh env loop per run is attached by hand)
        ac.env_loop, ac.loss_cfg = None, None
        ac.setup_training(env, D.ActorCriticLossConfig(backup_every=5, gamma=0.985, lambda_=0.95, weight_value_loss=1.0,
                                                       weight_entropy_loss=0.001))

    # reference run: plain module + GradAllReducer (what bench.py does)
    setup(make_env())
    torch.manual_seed(7)
    red = GradAllReducer(list(ac.parameters()))
    loss, _ = ac()
    loss.backward()
    g_ref = red.all_reduce_mean().clone()
    named_ref = {n: p.grad.detach().clone() for n, p in ac.named_parameters()}
    ac.zero_grad(set_to_none=True)
    # the early slice (LSTM + heads) all-reduced on a side stream from INSIDE backward(), the encoder's slice afterwards: same gradients
    setup(make_env())
    torch.manual_seed(7)
    red_e = GradAllReducer(list(ac.parameters()), early=[p for n, p in ac.named_parameters() if not n.startswith("encoder.")])
    loss_e, _ = ac()
    loss_e.backward()
    out["early_slice_launched_inside_backward"] = red_e.early_launches == 1 and red_e._early_work is not None
    red_e.all_reduce_mean()
    torch.cuda.synchronize()
    out["early_slice_mb"] = red_e.early_numel * 4 / 2 ** 20
    out["early_grad_rel_diff"] = max(float((p.grad - named_ref[n]).abs().max() / named_ref[n].abs().max().clamp_min(1e-30)) for n, p in ac.named_parameters())
    red_e.close()
    ac.zero_grad(set_to_none=True)
    # the reference's own wrapper: DistributedDataParallel(actor_critic) (utils.py:105-106), same seed, fresh env
    setup(make_env())
    torch.manual_seed(7)
    ddp = torch.nn.parallel.DistributedDataParallel(ac)  # exactly the reference's `DDP(module)` (utils.py:105)
    loss2, _ = ddp()
    loss2.backward()
    torch.cuda.synchronize()
    g_ddp = torch.cat([p.grad.reshape(-1) for p in ac.parameters()])
    out["loss_equal"] = float(loss) == float(loss2)
    out["ddp_grad_rel_diff"] = float((g_ddp - g_ref).abs().max() / g_ref.abs().max())
    out["grads_finite"] = bool(torch.isfinite(g_ddp).all())
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
