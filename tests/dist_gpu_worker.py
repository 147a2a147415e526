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


def main():
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
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
