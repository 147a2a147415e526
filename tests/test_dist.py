"""world_size-2 and -4 gloo tests of the data-parallel path (CPU): parameter broadcast (and that it invalidates the
packed-weight caches), the flat-bucket gradient all-reduce -- the mean of the ranks' actor-critic gradients equals the
gradient of the mean loss over the concatenated batch --, bucket views surviving zero_grad(set_to_none=True), and
the replica checksum bench.py asserts."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import diamond_amd as D
    from diamond_amd.dist import GradAllReducer, broadcast_parameters
    from diamond_amd.testing import fill_module_, synthetic_frames

    from diamond_amd.dist import parameter_checksum

    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))  # (the ranks share the host's cores: no oversubscription)
    torch.manual_seed(rank)  # different default init per rank: broadcast must fix it
    ac = D.ActorCritic(D.default_agent_config().actor_critic)
    if rank == 0:
        fill_module_(ac, 5)
    versions = [p._version for p in ac.parameters()]
    broadcast_parameters(ac)
    # the broadcast must bump Tensor._version: packed kernel-layout copies are cached per parameter version
    bumped = all(p._version > v for p, v in zip(ac.parameters(), versions))
    cs = torch.tensor([parameter_checksum(ac)], dtype=torch.float64)
    all_cs = [torch.zeros_like(cs) for _ in range(world)]
    dist.all_gather(all_cs, cs)
    same_params = all(float(c) == float(all_cs[0]) for c in all_cs)
    # The product's kernels need a GPU; this CPU test exercises the host-side collective logic only, so the
    # gradients come from the CPU oracle (test infrastructure) evaluated on the module's own Parameters.
    from oracle import diamond_oracle as O

    def predict(model, x):
        z = torch.zeros(x.size(0), model.lstm_dim)
        logits, val, _ = O.ac_predict(dict(model.named_parameters()), O.ActorCriticSpec(), x, z, z)
        return logits, val

    red = GradAllReducer(list(ac.parameters()))
    g = torch.Generator().manual_seed(40)
    per = 2 if world == 2 else 1  # (frames per rank: the oracle's backward on 8 cores shared by 4 ranks is the test's run time)
    obs_all = synthetic_frames(g, per * world, 3, 64, 64)
    obs = obs_all[rank * per:(rank + 1) * per]
    # a first backward, then zero_grad(set_to_none=True): the reducer must re-attach its bucket views
    logits, val = predict(ac, obs)
    (logits.square().mean() + val.mean()).backward()
    ac.zero_grad(set_to_none=True)
    logits, val = predict(ac, obs)
    (logits.square().mean() + val.mean()).backward()  # autograd allocates fresh .grad tensors outside the bucket
    flat = red.all_reduce_mean().clone()
    views_ok = all(p.grad.data_ptr() >= red.bucket.data_ptr() and
                   p.grad.data_ptr() < red.bucket.data_ptr() + red.bucket.numel() * 4 for p in ac.parameters())
    # the early slice (LSTM + heads, final before the encoder's backward is over) all-reduced from INSIDE backward() by the
    # post-accumulate hooks, the rest afterwards: the same mean gradients as the one blocking call above
    ac4 = D.ActorCritic(D.default_agent_config().actor_critic)
    fill_module_(ac4, 5)
    red4 = GradAllReducer(list(ac4.parameters()), early=[p_ for n_, p_ in ac4.named_parameters() if not n_.startswith("encoder.")])
    early_ok = 0 < red4.num_early < len(red4.params) and red4.early_launches == 0
    for rep in range(2 if world == 2 else 1):  # (twice at world 2: the hooks re-arm, zero_grad(set_to_none=False) keeps the bucket views)
        logits, val = predict(ac4, obs)
        (logits.square().mean() + val.mean()).backward()
        early_ok = early_ok and red4.early_launches == rep + 1 and red4._early_work is not None
        red4.all_reduce_mean()
        early_ok = early_ok and red4._early_work is None and all(
            bool(torch.allclose(p4.grad, p1.grad, rtol=1e-6, atol=1e-9)) for (n4, p4), (n1, p1) in zip(ac4.named_parameters(), ac.named_parameters()))
        if rep == 0:
            ac4.zero_grad(set_to_none=False)
    # the world-model parameters go through the same flat bucket (the reference DDP-wraps all three sub-models,
    # trainer.py:110): rank-dependent gradient pattern -> mean over the ranks; and torch's own DistributedDataParallel
    # constructor (utils.py:105-106: parameter verification + broadcast from rank 0) accepts the module
    den = D.Denoiser(D.default_agent_config().denoiser)
    rem = D.RewEndModel(D.default_agent_config().rew_end_model)
    wm = list(den.parameters()) + list(rem.parameters())
    red_wm = GradAllReducer(wm)
    for i, p in enumerate(wm):
        p.grad.fill_((rank + 1) * (float(i % 11) - 5.0))
    red_wm.all_reduce_mean()
    mean_scale = sum(r + 1 for r in range(world)) / world
    wm_ok = all(bool(torch.allclose(p.grad, torch.full_like(p.grad, mean_scale * (float(i % 11) - 5.0)))) for i, p in enumerate(wm))
    torch.manual_seed(100 + rank)
    ac3 = D.ActorCritic(D.default_agent_config().actor_critic)  # different values per rank
    ddp = torch.nn.parallel.DistributedDataParallel(ac3)
    cs3 = torch.tensor([parameter_checksum(ddp.module)], dtype=torch.float64)
    all_cs3 = [torch.zeros_like(cs3) for _ in range(world)]
    dist.all_gather(all_cs3, cs3)
    ddp_ok = all(float(c) == float(all_cs3[0]) for c in all_cs3)
    if rank == 0:
        # single-process reference over the whole batch
        ac2 = D.ActorCritic(D.default_agent_config().actor_critic)
        fill_module_(ac2, 5)
        logits, val = predict(ac2, obs_all)
        (logits.square().mean() + val.mean()).backward()
        ref = torch.cat([p.grad.reshape(-1) for p in ac2.parameters()])
        q.put(float((flat - ref).abs().max() / ref.abs().max()))
        q.put(bool(views_ok and bumped and same_params and wm_ok and ddp_ok and early_ok))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4])
def test_grad_allreduce_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17 * world) % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time

    got, deadline = [], time.time() + 240
    while len(got) < 2:
        try:
            got.append(q.get(timeout=2))
        except queue.Empty:
            assert time.time() < deadline, "timed out"
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: " + str([p.exitcode for p in procs])
    err, ok = got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5 and ok


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without WORLD_SIZE in the environment must start its own N ranks (it used to assert): the
    command it would run, and that `--gpus 1` does not take that route."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = d["self_launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "2", "--steps", "2", "--warmup", "1"]  # the ranks get the caller's flags (minus --dry-launch)
    assert d["env"]["MASTER_ADDR"] == "127.0.0.1" and d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher (WORLD_SIZE set) nothing is re-launched: the rank path is taken (and fails here for want of a GPU, not on an assert
    # about the launcher)
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                        capture_output=True, text=True, timeout=300)
    assert "self_launch" not in r2.stdout


def _ddp_worker(rank, world, port, q):
    """One rank of torch's own DistributedDataParallel over the package's Denoiser (what an unchanged trainer.py does at N > 1:
    utils.py:105-106, trainer.py:110) -- the product's recorded forward + hand-written backward, kernels on the SIMT interpreter."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace

    import diamond_amd as D
    from diamond_amd.inner_model import InnerModelConfig
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames
    from tests import wide_configs as W
    from tests.simt.host_harness import engine_on_interpreter

    torch.set_num_threads(2)
    cfg = dict(W.DENOISER, depths=[1, 1], channels=[64, 96], attn_depths=[0, 1])  # (two levels at 16 x 16: seconds per step)
    den = D.Denoiser(D.DenoiserConfig(inner_model=InnerModelConfig(**cfg), sigma_data=0.5, sigma_offset_noise=0.3))
    fill_module_(den, 3 + rank)  # different values per rank: DDP's constructor broadcasts rank 0's
    den.setup_training(D.SigmaDistributionConfig(**W.SIGMA_DIST))
    den.randn_fn = lambda shape: torch.randn(*shape)

    def batch_of(r):
        g = torch.Generator().manual_seed(50 + r)
        return SimpleNamespace(obs=synthetic_frames(g, 1, 5, 3, 16, 16), act=synthetic_actions(g, 4, 1, 5), mask_padding=torch.ones(1, 5, dtype=torch.bool))

    def step(model, r):
        den.zero_grad()
        torch.manual_seed(70 + r)  # (sigma and noise of rank r's batch)
        loss, _ = model(batch_of(r))
        loss.backward()
        return torch.cat([p.grad.reshape(-1) for p in den.parameters()]).clone()

    with engine_on_interpreter():
        ddp = torch.nn.parallel.DistributedDataParallel(den)
        mine = step(ddp, rank)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if rank == 0:
            same = all(torch.equal(g_, gathered[0]) for g_ in gathered)
            # without DDP (the module itself, rank 0's = everybody's weights): the mean of the ranks' own gradients
            ref = sum(step(den, r) for r in range(world)) / world
            q.put(float((mine - ref).abs().max() / ref.abs().max()))
            q.put(bool(same))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_of_the_denoiser_world2_on_the_interpreter():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    import socket

    with socket.socket() as sk:  # (a port nobody holds right now, not one derived from the pid)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time

    got, deadline = [], time.time() + 300
    while len(got) < 2:
        try:
            got.append(q.get(timeout=2))
        except queue.Empty:
            assert time.time() < deadline, "timed out"
            assert all(p.exitcode in (None, 0) for p in procs), "a rank died: " + str([p.exitcode for p in procs])
    err, same = got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same and err < 1e-6, (same, err)
