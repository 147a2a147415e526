"""bench.py -- imagined frames/s of DIAMOND's imagined-rollout hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete actor-critic BPTT window, exactly what `Trainer.train_component
("actor_critic")` does per optimiser step (reference trainer.py:363-382): ActorCritic.forward()
(15 imagined env steps: policy forward -> action sample -> 3-step Euler diffusion sampling ->
reward/end model -> bookkeeping/resets) + loss.backward() + gradient all-reduce (N > 1) +
clip_grad_norm_ + AdamW step.  Workload = BASELINE.json configs[1]: Breakout-shaped 64x64x3
frames, batch 256 per GPU, horizon 15, 3 denoising steps, fp32, synthetic weights/inputs.
value = B_global * 15 / (max-over-ranks seconds per step).

Extra objects on the JSON line: `roofline` for the dominant kernel (the 64-channel 3x3
implicit-GEMM conv, MFMA-fp32 bound), measured with HIP events in a dedicated instrumented
window after the timed region, and `cpu_baseline` = the CPU oracle timed on this box's host
cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* = fp32 vector peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (same guide); conv_f16s executes 3 f16 MACs per fp32 MAC
HBM_PEAK_GBS = 8000.0
FLOP_PER_FRAME = 19.080e9  # algorithmic, SURVEY.md §8(d): 3 x 6.0909 denoiser + 0.4477 rew/end + 0.3597 AC fwd+bwd
BYTES_PER_FRAME = 168.8e6  # algorithmic NHWC fp32 conv traffic model, SURVEY.md §8(d)


class _Loader:
    """What WorldModelEnv needs from a DataLoader: .batch_sampler.batch_size and batches with .obs (B,4,3,H,W) in
    [-1,1] and .act (B,4).  Like the reference's DataLoader (worker processes + pin_memory, trainer.py:140-167) the
    batches are ready, pinned host tensors when the env asks for them: a small synthetic set generated once and
    cycled, so the timed region contains the host->device upload but not synthetic-data generation."""

    class _BS:
        def __init__(self, b):
            self.batch_size = b

    def __init__(self, batch, seed, size, num_distinct=4):
        from diamond_amd.testing import initial_condition_batches

        self.batch_sampler = self._BS(batch)
        gen = initial_condition_batches(seed, batch, 4, h=size, w=size)
        self._batches = []
        for _ in range(num_distinct):
            obs, act = next(gen)
            if torch.cuda.is_available():
                obs, act = obs.pin_memory(), act.pin_memory()
            self._batches.append((obs, act))

    def __iter__(self):
        from types import SimpleNamespace

        i = 0
        while True:
            obs, act = self._batches[i % len(self._batches)]
            i += 1
            yield SimpleNamespace(obs=obs, act=act)


def build_agent(device, img_size, rank, attn_depths=(0, 0, 0, 0)):
    import diamond_amd as D
    from diamond_amd.testing import fill_module_

    agent = D.Agent(D.default_agent_config(num_actions=4, img_size=img_size, denoiser_attn_depths=tuple(attn_depths)))
    fill_module_(agent, 0)
    with torch.no_grad():
        # Synthetic weights would terminate ~half of the imagined episodes at every step; bias
        # the end logits (through one saturated hidden unit) so episodes end by horizon
        # truncation like a trained world model's do.  Pure workload shaping, same FLOPs.
        head = agent.rew_end_model.head
        head[0].bias[0] = 50.0
        head[2].weight[3].zero_()
        head[2].weight[4].zero_()
        head[2].weight[3, 0] = 0.2
        head[2].weight[4, 0] = -0.2
    return agent.to(device)


def usable_cores(cap=32):
    """Host threads the CPU baseline may use: scheduler affinity, cgroup CPU quota, and a cap (a
    256-thread oneDNN pool on a quota-limited container thrashes for minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_baseline_worker(img_size, threads):
    """CPU oracle (oracle/diamond_oracle.py, torch-CPU fp32) on a bounded sample of configs[0]
    (B=16, 3 of its 15 imagined steps + actor-critic loss backward).  Runs in its own process."""
    from diamond_amd.testing import fill_state_dict_, initial_condition_batches
    from oracle import diamond_oracle as O
    import diamond_amd as D

    torch.set_num_threads(threads)
    agent = D.Agent(D.default_agent_config(num_actions=4, img_size=img_size))
    sd = agent.state_dict()
    fill_state_dict_(sd, 0)
    sub = lambda p: {k[len(p) + 1:]: v.clone() for k, v in sd.items() if k.startswith(p + ".")}
    a = O.AgentSD(denoiser=sub("denoiser"), rew_end_model=sub("rew_end_model"), actor_critic=sub("actor_critic"),
                  aspec=O.ActorCriticSpec(img_size=img_size), rspec=O.RewEndSpec(img_size=img_size))
    a.actor_critic = {k: v.requires_grad_(True) for k, v in a.actor_critic.items()}
    b, t = 16, 3
    draws = O.DrawSource(torch.Generator().manual_seed(1))
    env = O.ImaginationEnv(a, initial_condition_batches(5, b, 4, h=img_size, w=img_size), b, 15, draws, 1)
    state = (env.reset(), torch.zeros(b, 512), torch.zeros(b, 512))
    t0 = time.perf_counter()
    (obs, act, rew, end, trunc, logits, val, vb), state = O.rollout(a, env, state, t, draws)
    loss, _ = O.ac_loss(logits, val, act, rew, end, trunc, vb, O.LossSpec(backup_every=t))
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": b * t / dt, "unit": "imagined frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"configs[0] shape, B={b}, {t} of 15 imagined steps (3 Euler denoise + rew/end + actor-critic) + AC "
                      f"backward, {img_size}x{img_size}, fp32 torch-CPU oracle, {dt:.1f}s"}


def cpu_baseline(img_size, timeout_s=240):
    """Time the CPU oracle in a child process (own thread pool, hard timeout: the bench line must
    never hang on the baseline)."""
    import subprocess

    threads = usable_cores()
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), "--img-size", str(img_size)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
        return json.loads(out.decode().strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        proc.kill()  # exactly the PID we started
        proc.communicate()
        return {"value": None, "unit": "imagined frames/s", "cores": threads, "kind": "port",
                "sample": f"timed out after {timeout_s}s"}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "imagined frames/s", "cores": threads, "kind": "port", "sample": f"failed: {e!r}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="imagination batch PER GPU")
    ap.add_argument("--horizon", type=int, default=15)
    ap.add_argument("--denoise-steps", type=int, default=3)
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--attn-depths", type=str, default="0,0,0,0",
                    help="denoiser attention per level; BASELINE configs[4] (256x256) uses 0,0,1,1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(args.img_size, args.cpu_baseline_worker)))
        return

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # "nccl" == RCCL on ROCm

    import diamond_amd as D
    from diamond_amd import engine as E
    from diamond_amd.dist import GradAllReducer

    torch.manual_seed(1234 + rank)
    attn = tuple(int(v) for v in args.attn_depths.split(","))
    agent = build_agent(device, args.img_size, rank, attn)
    env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(args.batch, 100 + rank, args.img_size),
                          D.WorldModelEnvConfig(horizon=args.horizon, num_batches_to_preload=2,
                                                diffusion_sampler=D.DiffusionSamplerConfig(
                                                    num_steps_denoising=args.denoise_steps)))
    agent.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                         D.ActorCriticLossConfig(backup_every=args.horizon, gamma=0.985, lambda_=0.95,
                                                 weight_value_loss=1.0, weight_entropy_loss=0.001), env)
    ac = agent.actor_critic
    opt = torch.optim.AdamW(ac.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.0)
    reducer = GradAllReducer(list(ac.parameters())) if world > 1 else None

    def window():
        loss, metrics = ac()
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_mean()
        torch.nn.utils.clip_grad_norm_(ac.parameters(), 100.0)
        opt.step()
        opt.zero_grad(set_to_none=False)
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def progress(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    progress("setup done")
    for _ in range(args.warmup):
        window()
    fence()
    progress("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        window()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    progress(f"timed region done: {elapsed:.2f}s for {args.steps} steps")
    is_cfg1 = (args.img_size, args.batch, args.horizon, args.denoise_steps, attn) == (64, 256, 15, 3, (0, 0, 0, 0))
    cfg_name = ("configs[1] (Breakout-shaped)" if world == 1 else "configs[2] (Breakout-shaped, sharded)") if is_cfg1 else \
        f"custom (attn_depths {args.attn_depths})"
    frames = args.batch * world * args.horizon * args.steps
    fps = frames / elapsed
    line = {
        "metric": "imagined frames/sec (64x64, 3 denoise steps, batch 256)", "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (world-model 3x3 convs: fp32 operands split into 2 x fp16 pieces on v_mfma_f32_32x32x16_f16, fp32 accumulate, "
                 "fp32-class accuracy; everything else incl. actor-critic fwd/bwd: exact fp32 v_mfma_f32_16x16x4_f32)", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {args.img_size}x{args.img_size}x3 frames, batch {args.batch}/GPU, "
                               f"horizon {args.horizon}, {args.denoise_steps} Euler denoise steps; step = "
                               "ActorCritic.forward()+backward+all-reduce+clip+AdamW over one 15-step imagined window",
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (batch-sharded envs, flat-bucket "
                   "RCCL all-reduce of actor-critic grads)", "actor_critic_backend": ac.backend},
        "whole_step_algorithmic": {"tflops": fps * FLOP_PER_FRAME / 1e12 / world,
                                   "frac_fp32_peak": fps * FLOP_PER_FRAME / 1e12 / world / FP32_MFMA_PEAK_TFLOPS,
                                   "hbm_gbs": fps * BYTES_PER_FRAME / 1e9 / world,
                                   "frac_hbm_peak": fps * BYTES_PER_FRAME / 1e9 / world / HBM_PEAK_GBS},
    }

    if not args.no_roofline:
        # instrumented window (not part of `value`): HIP events around every dmd_conv2d launch
        E.PROFILER = E.LaunchProfiler()
        window()
        summ = E.PROFILER.summary()
        E.PROFILER = None
        key = max(summ, key=lambda k: summ[k]["ms"])
        d = summ[key]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        split = key.startswith("conv_f16s")
        peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        pmc = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path)).get(key)
        line["roofline"] = {
            "kernel": key, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB units;
            # tools/pmc_collect.sh -> profiles/pmc_traffic.json), next to the algorithmic bytes per launch
            "traffic": None if pmc is None else pmc["hbm_bytes_per_launch"],
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], "launches": d["launches"],
            "avg_launch_ms": d["ms"] / d["launches"], "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
            "algorithmic_hbm_gbs": d["bytes"] / (d["ms"] * 1e-3) / 1e9,
            "frac_hbm_peak": d["bytes"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": ("achieved = ALGORITHMIC fp32 conv FLOPs (2 per MAC) / measured kernel time; peak = dense f16 MFMA. The kernel "
                     "splits each fp32 operand into two fp16 pieces and issues 3 f16 MFMAs per algorithmic MAC (fp32-class "
                     "accuracy), so the matrix pipe executes 3x the algorithmic rate: executed_mfma_frac below."
                     if split else "exact-fp32 MFMA kernel: peak = fp32 MFMA/vector peak"),
            "executed_mfma_frac": (3.0 if split else 1.0) * achieved / peak,
            "frac_of_fp32_direct_conv_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
            "conv_share_of_window_ms": {k: v["ms"] for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])},
        }

    progress("roofline window done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.img_size)
        progress("cpu baseline done")

    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
