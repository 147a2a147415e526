"""bench.py -- imagined frames/s of DIAMOND's imagined-rollout hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W                      (BASELINE configs[1], the headline metric)
    python bench.py --config 3 --steps 1 --warmup 1                    (configs[3]: 50-step Heun sampler)
    python bench.py --config 4 --steps 2 --warmup 1                    (configs[4]: 256x256, attention, 8 envs per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete actor-critic BPTT window, exactly what `Trainer.train_component
("actor_critic")` does per optimiser step (reference trainer.py:363-382): ActorCritic.forward()
(15 imagined env steps: policy forward -> action sample -> diffusion sampling -> reward/end model ->
bookkeeping/resets) + loss.backward() + gradient all-reduce (N > 1) + clip_grad_norm_ + AdamW step.
Default workload = BASELINE.json configs[1]: Breakout-shaped 64x64x3 frames, batch 256 per GPU, horizon 15,
3 Euler denoising steps, synthetic weights/inputs.  value = B_global * 15 / (max-over-ranks seconds per step).

Extra objects on the JSON line: `roofline` for the dominant kernel (found by measured time over EVERY C-ABI launch:
HIP events around each one in a dedicated instrumented window after the timed region), `exact_fp32` = the same window
with every convolution on the exact-fp32 MFMA kernels (configs[1] only), `also` = what else fits one GPU, measured after the
timed region (configs[3], configs[4], the B = 1 latency mode, the denoiser training step, and configs[1] WITHOUT the end-logit
bias, i.e. with mid-window resets: `--no-also` skips them), and `cpu_baseline` = the REFERENCE ITSELF (its bytecode in
oracle/_ref, see oracle/make_ref.py) and the CPU oracle port, both timed on this box's host cores on one whole window of
configs[0] at the same thread count (rank 0, N=1 only).  The roofline prices the kernel against the dense f16 MFMA peak of the
guide (2.5 PFLOP/s) as the contract asks; `roofline.power_limited_peak` is what an MFMA-only stream sustains on this chip
with random operands (profiles/r04_clock.json: 1,661 TFLOP/s at the 1.58 GHz the power management allows it).
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
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (same guide); the split kernels execute 3 f16 MACs per fp32 MAC
F16_MFMA_SUSTAINED_TFLOPS = 1661.0  # measured: back-to-back v_mfma_f32_32x32x16_f16 on random operands, all SIMDs (profiles/r04_clock.json)
HBM_PEAK_GBS = 8000.0

# BASELINE.json configs -> (img_size, batch per GPU, horizon, denoise steps, sampler order, denoiser attn_depths)
CONFIGS = {
    1: dict(img_size=64, batch=256, horizon=15, denoise_steps=3, order=1, attn_depths="0,0,0,0"),
    3: dict(img_size=64, batch=256, horizon=15, denoise_steps=50, order=2, attn_depths="0,0,0,0"),
    4: dict(img_size=256, batch=8, horizon=15, denoise_steps=3, order=1, attn_depths="0,0,1,1"),
}


def algorithmic_work(img_size, denoise_steps, order, attn):
    """(FLOP, bytes) per imagined frame, SURVEY.md §8(d) / BASELINE.md §2 (MAC = 2; NHWC fp32, every conv reads its
    input once and writes its output once)."""
    calls = denoise_steps if order == 1 else 2 * denoise_steps - 1  # Heun skips the 2nd evaluation of the last step
    s = (img_size / 64) ** 2
    if img_size == 256 and attn == (0, 0, 1, 1):
        den_flop, den_bytes = 121.555e9, 827.25 * 2 ** 20
    else:
        den_flop, den_bytes = 6.0909e9 * s, 51.757e6 * s
    rest_flop = (0.4477e9 + 0.3597e9) * s
    rest_bytes = (7.22e6 + 3 * 2.11e6) * s
    return calls * den_flop + rest_flop, calls * den_bytes + rest_bytes, calls


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


END_LOGIT_BIAS_NOTE = ("end-logits of the synthetic reward/end model are biased so that episodes end by horizon truncation "
                       "(as a trained world model's do) instead of ~50 % of the envs dying at every step: all 256 envs reset "
                       "+ burn in together at the window boundary, no mid-window resets; same kernels and FLOPs per frame")
# The headline regime since round 6: the STEADY STATE of the reference's training loop.  With horizon == backup_every == 15
# (config/trainer.yaml:68,136) every sampled `end` desynchronises its env for good, so a batch's episode lengths spread over the
# horizon and batch / horizon envs truncate at EVERY step; on top of that every env ends with probability p per step.
STEADY_STATE_END_RATE = 0.003
STEADY_STATE_NOTE = ("the steady state of the reference's training loop: episode lengths spread over the horizon (batch / horizon = 17 "
                     "truncation resets at EVERY step) and every env ends with probability {p} per step (through the synthetic reward/end "
                     "head): resets, V(final observation) and burn-in inside every step of the timed windows "
                     "(world_model_env.py:77-82, env_loop.py:45-56); --no-ends is the round-1..5 headline (nobody ends mid-window)")


def set_end_rate(agent, p=None):
    """The synthetic reward/end model's end probability per env-step, through ONE saturated hidden unit of its head (in place: the
    packed weight copies notice the version bump).  None: the round-1..4 bias (logit difference 20 +- 0.4: nobody ever ends);
    0 < p < 1: hidden unit 0 is pinned to silu(50) = 50 exactly and the two end logits are -+ log(p / (1 - p)) / 2, i.e. every env
    ends with probability p at every step, independently of the frame."""
    import math

    with torch.no_grad():
        head = agent.rew_end_model.head
        head[0].bias[0] = 50.0
        head[2].weight[3].zero_()
        head[2].weight[4].zero_()
        if p is None:
            head[2].weight[3, 0] = 0.2
            head[2].weight[4, 0] = -0.2
        else:
            head[0].weight[0].zero_()
            d = math.log(p / (1.0 - p))
            head[2].weight[3, 0] = -d / 100.0
            head[2].weight[4, 0] = d / 100.0


def build_agent(device, img_size, rank, attn_depths=(0, 0, 0, 0), bias_end_logits=True, end_rate=None):
    import diamond_amd as D
    from diamond_amd.testing import fill_module_

    agent = D.Agent(D.default_agent_config(num_actions=4, img_size=img_size, denoiser_attn_depths=tuple(attn_depths)))
    fill_module_(agent, 0)
    if not bias_end_logits:
        return agent.to(device)
    # Synthetic weights would terminate ~half of the imagined episodes at every step; bias the end logits (through one
    # saturated hidden unit) so episodes end by horizon truncation like a trained world model's do -- or at a stated rate.
    # Disclosed in config.workload (END_LOGIT_BIAS_NOTE).
    set_end_rate(agent, end_rate)
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


REFERENCE_MEASURED = {"value": 6.2, "unit": "imagined frames/s", "cores": 8, "kind": "reference",
                      "where": "the reference itself (src/ imported with stubs), configs[0], 2nd window, build container, "
                               "8 threads, torch-CPU fp32 -- BASELINE.md §3; it cannot travel to the GPU box"}


def cpu_baseline_worker(img_size, threads):
    """CPU oracle (oracle/diamond_oracle.py, torch-CPU fp32) on ONE WHOLE WINDOW of configs[0]: B=16, reset
    (pool preload + reward/end burn-in) + 15 imagined steps (3 Euler denoise + rew/end + actor-critic each) +
    actor-critic loss backward.  Runs in its own process."""
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
    b, t = 16, 15
    draws = O.DrawSource(torch.Generator().manual_seed(1))
    t0 = time.perf_counter()
    env = O.ImaginationEnv(a, initial_condition_batches(5, b, 4, h=img_size, w=img_size), b, 15, draws, 1)
    state = (env.reset(), torch.zeros(b, 512), torch.zeros(b, 512))
    (obs, act, rew, end, trunc, logits, val, vb), state = O.rollout(a, env, state, t, draws)
    loss, _ = O.ac_loss(logits, val, act, rew, end, trunc, vb, O.LossSpec(backup_every=t))
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": b * t / dt, "unit": "imagined frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"configs[0]: one whole window, B={b}, reset + {t} imagined steps (3 Euler denoise + rew/end + actor-critic) "
                      f"+ AC backward, {img_size}x{img_size}, fp32 torch-CPU oracle (unbiased synthetic end-logits: includes "
                      f"mid-window resets / burn-in), {dt:.1f}s"}


def _child_json(cmd, threads, timeout_s):
    """stdout's last line of a child process as JSON (own thread pool, hard timeout: the bench line must never hang on a baseline)"""
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", PYTHONDONTWRITEBYTECODE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
        return json.loads(out.decode().strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        proc.kill()  # exactly the PID we started
        proc.communicate()
        return {"value": None, "sample": f"timed out after {timeout_s}s"}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "sample": f"failed: {e!r}"}


def cpu_baseline_config4(timeout_s=240):
    """configs[4]'s frame on the reference itself (one env, a 2-step window at 256x256 with attention [0,0,1,1]): there is no
    CPU-runnable BASELINE config at this size, so the sample is as small as the reference's loop allows."""
    threads = usable_cores()
    ref = _child_json([sys.executable, os.path.join(ROOT, "oracle", "reference_window.py"), "--threads", str(threads), "--img-size", "256",
                       "--batch", "1", "--horizon", "2", "--attn-depths", "0,0,1,1"], threads, timeout_s)
    ref.setdefault("unit", "imagined frames/s"), ref.setdefault("cores", threads), ref.setdefault("kind", "reference")
    return ref


def cpu_baseline(img_size, timeout_s=240, allow_reference=True):
    """The CPU path beside the GPU number, on this box's host cores, each in a child process: the REFERENCE ITSELF where its
    bytecode travelled with the snapshot (oracle/_ref, built by oracle/make_ref.py in the build container: kind "reference") and
    the oracle port (oracle/diamond_oracle.py: kind "port") on the same window at the same thread count."""
    threads = usable_cores()
    port = _child_json([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads), "--img-size", str(img_size)],
                       threads, timeout_s)
    port.setdefault("unit", "imagined frames/s"), port.setdefault("cores", threads), port.setdefault("kind", "port")
    if not allow_reference:
        return dict(port, reference_unavailable="--cpu-baseline-kind port", reference_measured=REFERENCE_MEASURED)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import reference_window as RW  # test infrastructure: the locator only -- the run itself happens in the child

        where, what = RW.reference_location()
    except Exception as e:  # noqa: BLE001
        where, what = None, repr(e)
    finally:
        sys.path.pop(0)
    if where is None:
        return dict(port, reference_unavailable=what, reference_measured=REFERENCE_MEASURED)
    ref = _child_json([sys.executable, os.path.join(ROOT, "oracle", "reference_window.py"), "--threads", str(threads), "--img-size",
                       str(img_size)], threads, timeout_s)
    if ref.get("value") is None:
        return dict(port, reference_unavailable=ref.get("sample"), reference_measured=REFERENCE_MEASURED)
    ref["port"] = {k: port.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    if port.get("value"):
        ref["port_over_reference_at_equal_cores"] = port["value"] / ref["value"]
    return ref


def precision_label(E, ac_native):
    wm = ("world-model convs (3x3 stride 1 / 2, 1x1): fp32 operands split into 2 x fp16 pieces, 3 x v_mfma_f32_*_f16 per product, "
          "fp32 accumulate (fp32-class, 22-bit operands); attention core, linears: exact fp32 v_mfma_f32_16x16x4_f32"
          if E.WORLD_MODEL_PRECISION == "f16x2" else "world model: exact fp32 v_mfma_f32_16x16x4_f32")
    ac = ("actor-critic encoder forward, dgrad and weight-gradient convs: same split-fp16 form"
          if ac_native.AC_PRECISION == "f16x2" else "actor-critic encoder fwd/bwd: exact fp32 MFMA")
    short = "f32" if (E.WORLD_MODEL_PRECISION, ac_native.AC_PRECISION) == ("f32", "f32") else "f32 via split-f16 MFMA"
    return f"{short} ({wm}; {ac}; LSTM cells / heads: exact fp32 MFMA (dmd_linear))"


def latency_run(graph: bool, frames: int, warmup: int = 12):
    """ms per imagined frame of the B=1 interactive env (reference play.py:105-109, game/play_env.py:113-127): each frame =
    WorldModelEnv.step(act) (3 Euler denoising steps + reward/end model + bookkeeping) + the play loop's host read of the
    reward.  Returns (ms per frame, ms per frame of the sampler alone)."""
    import diamond_amd as D

    dev = torch.device("cuda:0")
    agent = build_agent(dev, 64, 0)
    env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(1, 7, 64),
                          D.WorldModelEnvConfig(horizon=100000, num_batches_to_preload=1,
                                                diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=3)),
                          return_denoising_trajectory=True, graph_sampler=graph)
    env.reset()
    act = torch.zeros(1, dtype=torch.long, device=dev)
    for _ in range(max(warmup, 12)):  # caches, and in graph mode one capture per ring head
        env.step(act)
    torch.cuda.synchronize()
    t_s = 0.0
    for _ in range(frames):
        ts = time.perf_counter()
        env.predict_next_obs()
        torch.cuda.synchronize()
        t_s += time.perf_counter() - ts
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        _, rew, _, _, _ = env.step(act)
        float(rew.item())
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / frames, 1e3 * t_s / frames


def latency_line(args, eager=True):
    frames = args.steps or 200
    eager, eager_s = latency_run(False, frames, args.warmup) if eager else (None, None)
    graph, graph_s = latency_run(True, frames, args.warmup)
    return {"metric": "ms per imagined frame, B=1 interactive world-model env (64x64, 3 Euler denoise steps)", "value": graph,
            "unit": "ms/frame", "n_gpus": 1, "steps": frames, "warmup": max(args.warmup, 12), "ms_per_step": graph,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32 via split-f16 MFMA", "data": "synthetic",
            "config": {"workload": "SURVEY §8 (f4): play.py's B=1 WorldModelEnv.step + host read of the reward per frame; the "
                                   "sampler's launches replayed as one hipGraph per ring head", "global_batch": 1},
            "frames_per_s": 1e3 / graph, "eager_ms_per_frame": eager, "sampler_only_graph_ms": graph_s,
            "sampler_only_eager_ms": eager_s,
            # 3 denoiser forwards + the reward/end model per frame (3 x 6.0909 + 0.4477 GFLOP): launch-latency-bound at B = 1
            "algorithmic_tflops": (3 * 6.0909e9 + 0.4477e9) / (graph * 1e-3) / 1e12,
            "frac_f16_mfma_peak": (3 * 6.0909e9 + 0.4477e9) / (graph * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS}


def train_line(args, eager=True):
    """Denoiser training step (reference trainer.py:349-388, denoiser.py:93-122): forward over a segment of
    4 conditioning + 1 predicted frame, backward, gradient clipping, AdamW -- at the reference batch of 32 (--batch)."""
    from types import SimpleNamespace

    import diamond_amd as D
    from diamond_amd.testing import fill_module_, synthetic_actions, synthetic_frames

    b, steps = args.batch or 32, args.steps or 20
    dev = torch.device("cuda:0")
    agent = D.Agent(D.default_agent_config())
    fill_module_(agent, 0)
    den = agent.denoiser.to(dev).train()
    den.setup_training(D.SigmaDistributionConfig(loc=-0.4, scale=1.2, sigma_min=2e-3, sigma_max=20))
    opt = torch.optim.AdamW(den.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(0)
    t = 5
    batch = SimpleNamespace(obs=synthetic_frames(g, b, t, 3, 64, 64).to(dev), act=synthetic_actions(g, 4, b, t).to(dev),
                            mask_padding=torch.ones(b, t, dtype=torch.bool, device=dev))

    def step():
        loss, _ = den(batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(den.parameters(), 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss.detach()  # (no reference to the autograd graph survives the step: train_graph.py)

    dt_eager = None
    if eager:
        for _ in range(max(args.warmup, 3)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt_eager = (time.perf_counter() - t0) / steps
    # the same step captured into one hipGraph and replayed on static input buffers (diamond_amd/train_graph.py)
    from diamond_amd.train_graph import GraphedTrainStep

    # fused=True: the capturable FOREACH AdamW divides every tensor by its 0-dim bias corrections with one broadcast kernel each
    # (2 x 236 launches of ~4 us per step: profiles/r04_train_kernel_stats.csv); torch's fused form is one multi-tensor kernel
    opt_g = torch.optim.AdamW(den.parameters(), lr=1e-4, capturable=True, fused=None if getattr(args, "foreach_adamw", False) else True)
    gstep = GraphedTrainStep(den, opt_g, 1.0, batch, warmup_steps=max(args.warmup, 3))
    for _ in range(2):
        gstep(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = gstep(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # per-kernel roofline of the step (one EAGER step under the launch profiler: a replayed graph issues no launches it could time)
    from diamond_amd import native as nv

    nv.PROFILER = nv.LaunchProfiler()
    try:
        step()
        summ = nv.PROFILER.summary()
    finally:
        nv.PROFILER = None
    kernels, covered = kernel_table(summ, None, None, top=6, cover=0.85)
    c_abi_launches = sum(v["launches"] for v in summ.values())
    return {"metric": "denoiser training step (forward + backward + clip + AdamW), 64x64, 1 predicted frame per segment",
            "value": 1e3 * dt, "unit": "ms/step", "n_gpus": 1, "steps": steps, "warmup": max(args.warmup, 3), "ms_per_step": 1e3 * dt,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32 via split-f16 MFMA", "data": "synthetic",
            "config": {"workload": "SURVEY §8 (f2): Denoiser.forward(batch) + loss.backward() + clip_grad_norm_ + AdamW, the whole step "
                                   "replayed as one hipGraph (GraphedTrainStep; torch.optim.AdamW(capturable=True, "
                                   f"fused={not getattr(args, 'foreach_adamw', False)}))", "global_batch": b},
            "frames_per_s": b / dt, "eager_ms_per_step": None if dt_eager is None else 1e3 * dt_eager, "loss": float(loss.detach()),
            # forward + dgrad + wgrad = 3 x the forward's 6.0909 GFLOP per frame, against the f16 MFMA peak the split kernels run on
            "algorithmic_tflops": 3 * 6.0909e9 * b / dt / 1e12, "frac_f16_mfma_peak": 3 * 6.0909e9 * b / dt / 1e12 / F16_MFMA_PEAK_TFLOPS,
            "roofline": {"kernels": kernels, "kernels_cover_launch_time": round(covered, 4), "c_abi_launches_per_eager_step": c_abi_launches,
                         "launch_time_ms_eager_step": round(sum(v["ms"] for v in summ.values()), 3)}}


def rollout_setup(device, rank, img_size, batch, horizon, denoise_steps, order, attn, use_dist=False, bias_end_logits=True, end_rate=None):
    """Agent + imagination env + optimizer of one configuration; returns (agent, actor_critic, window) with window() = one
    optimiser step of `Trainer.train_component("actor_critic")` (reference trainer.py:363-382)."""
    import diamond_amd as D
    from diamond_amd.dist import GradAllReducer, broadcast_parameters

    agent = build_agent(device, img_size, rank, attn, bias_end_logits, end_rate)
    if use_dist:
        broadcast_parameters(agent, src=0)  # what the DDP constructor does in the reference (utils.py:106)
    env = D.WorldModelEnv(agent.denoiser, agent.rew_end_model, _Loader(batch, 100 + rank, img_size),
                          D.WorldModelEnvConfig(horizon=horizon, num_batches_to_preload=2,
                                                diffusion_sampler=D.DiffusionSamplerConfig(num_steps_denoising=denoise_steps, order=order)))
    agent.setup_training(D.SigmaDistributionConfig(-0.4, 1.2, 2e-3, 20),
                         D.ActorCriticLossConfig(backup_every=horizon, gamma=0.985, lambda_=0.95,
                                                 weight_value_loss=1.0, weight_entropy_loss=0.001), env)
    ac = agent.actor_critic
    opt = torch.optim.AdamW(ac.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.0)
    # (LSTM + heads: their gradients are final before the encoder's last backward -- all-reduced from inside backward(), dist.py)
    reducer = GradAllReducer(list(ac.parameters()), early=[p for n, p in ac.named_parameters() if not n.startswith("encoder.")]) if use_dist else None

    def window():
        loss, metrics = ac()
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_mean()
        torch.nn.utils.clip_grad_norm_(ac.parameters(), 100.0)
        opt.step()
        opt.zero_grad(set_to_none=False)
        return loss

    window.reducer, window.env = reducer, env
    return agent, ac, window


def timed_windows(window, steps, warmup):
    for _ in range(warmup):
        window()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        window()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def measured_windows(window, steps=3, warmup=2):
    """(mean seconds per window, [device ms of every timed window]): `warmup` untimed windows first (the first window of a fresh
    configuration builds caches and is up to 50 % slow), then `steps` windows between two synchronisations, stamped by events."""
    for _ in range(warmup):
        window()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        window()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(steps)]


def stagger_episodes(env, horizon):
    """The steady state of the reference's training loop: every `end` desynchronises its env for good (its truncation then falls
    mid-window, world_model_env.py:71-72), so after a few thousand steps the episode lengths of a batch are spread over the
    horizon and B / horizon envs truncate at EVERY step.  Set that state directly: ep_len[r] = r mod horizon."""
    if hasattr(env, "set_episode_lengths"):
        env.set_episode_lengths(torch.arange(env.num_envs) % horizon)
    else:  # (an env without the host mirror: tools/ab_env_loop.sh runs the round-4 classes through this file)
        env.ep_len = (torch.arange(env.num_envs) % horizon).to(env.ep_len.device)


def _is_split(key):
    """kernels that run split-fp16 arithmetic (3 f16 MFMAs per algorithmic MAC) are priced against the f16 peak"""
    last_arg_true = key.rstrip(">").rstrip().endswith("true")  # (the SPLIT template argument is the last one: "..., true>" / "..., true>>")
    return (key.startswith("conv_f16ws") or key.startswith("attention_f16x2") or key.startswith("lowres_chain") or key.startswith("wgrad_ps_kernel")
            or ((key.startswith("conv1x1_stream") or key.startswith("conv_mfma") or key.startswith("wgrad_kernel")) and last_arg_true))


def kernel_table(summ, pmc_all, sq_all, top=8, cover=0.9):
    """`roofline.kernels`: the kernels of the instrumented window by measured time -- at least `top`, and as many as it takes to
    cover `cover` of the window's launch time -- each with its own roofline entry from the SAME LaunchProfiler pass: calls, average
    duration (HIP events), algorithmic GFLOP and bytes per launch (annotated by the host code that launches it), the fraction of
    the peak that bounds it, the PMC traffic ratio and the matrix-pipe busy share where the committed counter passes hold that
    kernel (profiles/pmc_traffic*.json, profiles/sq_counters.json)."""
    total_ms = sum(v["ms"] for v in summ.values())
    rows, acc = [], 0.0
    for key, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
        if len(rows) >= top and acc >= cover * total_ms:
            break
        acc += d["ms"]
        sec = d["ms"] * 1e-3
        tflops, gbs = d["flops"] / sec / 1e12, d["bytes"] / sec / 1e9
        split = _is_split(key)
        peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        frac_mfma, frac_hbm = tflops / peak, gbs / HBM_PEAK_GBS
        bound = "mfma" if (d["flops"] > 0 and frac_mfma * (3.0 if split else 1.0) >= frac_hbm) else ("hbm" if d["bytes"] > 0 else None)
        row = {"kernel": key, "calls": d["launches"], "avg_us": round(1e3 * d["ms"] / d["launches"], 2), "share_of_launch_time": round(d["ms"] / total_ms, 4),
               "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 4), "algorithmic_mb_per_launch": round(d["bytes"] / d["launches"] / 1e6, 3),
               "bound": bound, "achieved_tflops": round(tflops, 2), "achieved_gbs": round(gbs, 1),
               "frac": None if bound is None else round(frac_mfma if bound == "mfma" else frac_hbm, 4),
               "peak": None if bound is None else (peak if bound == "mfma" else HBM_PEAK_GBS), "unit": None if bound is None else ("TFLOP/s" if bound == "mfma" else "GB/s"),
               "executed_mfma_frac": round((3.0 if split else 1.0) * frac_mfma, 4) if d["flops"] > 0 else None}
        pmc = (pmc_all or {}).get(key)
        if pmc is not None and d["bytes"] > 0:
            row["traffic_mb_per_launch"] = round(pmc["hbm_bytes_per_launch"] / 1e6, 3)
            row["traffic_ratio"] = round(pmc["hbm_bytes_per_launch"] / (d["bytes"] / d["launches"]), 3)
        sq = (sq_all or {}).get(key)
        if sq is not None and sq.get("mfma_busy") is not None:
            row["mfma_busy"] = sq["mfma_busy"]
        rows.append(row)
    return rows, acc / total_ms


def dominant_kernel_roofline(window, nv, config_idx, world=1, custom=False):
    """One instrumented window (HIP events around every C-ABI launch) -> the `roofline` object: the dominant kernel's entry at the
    top level (the contract's shape) and `kernels`, the same for every kernel that matters (kernel_table)."""
    nv.PROFILER = nv.LaunchProfiler()
    try:
        window()
        summ = nv.PROFILER.summary()
    finally:
        nv.PROFILER = None
    key = max(summ, key=lambda k: summ[k]["ms"])
    d = summ[key]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    split = _is_split(key)
    peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
    pmc = pmc_all = sq_all = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json" if config_idx == 1 else f"pmc_traffic_cfg{config_idx}.json")
    if os.path.exists(pmc_path) and world == 1:
        pmc_all = json.load(open(pmc_path))
        pmc = pmc_all.get(key) if not custom else None  # keyed by the rocprofv3 kernel name (tools/pmc_to_profile.py)
    sq_path = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(sq_path) and config_idx == 1:
        sq = json.load(open(sq_path))
        sq_all, sq_src = sq.get("kernels", {}), f"profiles/sq_counters.json [{sq.get('profile_set')}]: rocprofv3 --pmc SQ counters over denoiser forwards at batch 256 (tools/pmc_sq.sh)"
    total_ms = sum(v["ms"] for v in summ.values())
    kernels, covered = kernel_table(summ, pmc_all, sq_all)
    return {
        "kernel": key, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB units;
        # tools/pmc_collect.sh -> tools/pmc_to_profile.py -> profiles/<set>_pmc_traffic.json), next to the algorithmic bytes
        "traffic": None if pmc is None else pmc["hbm_bytes_per_launch"],
        "traffic_source": None if pmc is None else f"profiles/{os.path.basename(pmc_path)} [{pmc.get('profile_set')}]: {pmc.get('workload')}",
        "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], "launches": d["launches"],
        "avg_launch_ms": d["ms"] / d["launches"], "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
        "algorithmic_hbm_gbs": d["bytes"] / (d["ms"] * 1e-3) / 1e9,
        "frac_hbm_peak": d["bytes"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "note": ("achieved = ALGORITHMIC fp32 FLOPs (2 per MAC) / measured kernel time; peak = dense f16 MFMA. The kernel "
                 "splits each fp32 operand into two fp16 pieces and issues 3 f16 MFMAs per algorithmic MAC (fp32-class "
                 "accuracy), so the matrix pipe executes 3x the algorithmic rate: executed_mfma_frac below."
                 if split else "exact-fp32 kernel: peak = fp32 MFMA/vector peak"),
        "executed_mfma_frac": (3.0 if split else 1.0) * achieved / peak,
        # the chip is power-managed: an MFMA-only stream on random operands is held to 1.58 GHz (tools/probe/clock_probe.hip)
        "power_limited_peak": F16_MFMA_SUSTAINED_TFLOPS if split else None,
        "executed_frac_of_power_limited_peak": (3.0 * achieved / F16_MFMA_SUSTAINED_TFLOPS) if split else None,
        "frac_of_fp32_direct_conv_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
        "kernel_share_of_launch_time": d["ms"] / total_ms,
        # the matrix pipe's busy share of the kernel's busy cycles, from the committed SQ counter passes (clock-free: both counters
        # tick in the same throttled domain) -- north_star's "rocprof MFMA utilisation"
        "mfma_busy": None if not sq_all or key not in sq_all else sq_all[key].get("mfma_busy"),
        "mfma_busy_source": None if not sq_all else sq_src,
        # every kernel that matters, each with its own roofline entry, from the same LaunchProfiler pass
        "kernels": kernels, "kernels_cover_launch_time": round(covered, 4),
        # every C-ABI entry point / kernel instantiation of the window, by measured time (the dominant one is chosen over ALL)
        "launch_time_ms": {k: round(v["ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])},
    }


def also_lines(device, args):
    """What the default line carries besides configs[1] (single GPU, after the timed region; none of it is part of `value`):
    configs[1] under the conditions the reference trains in (episodes that end), the other BASELINE configs that fit one GPU and
    the SURVEY §8(f) rows.  Every window-based line: 2 untimed windows, then 3 timed ones (`step_ms` = each of them)."""
    from types import SimpleNamespace

    from diamond_amd import native as nv

    out = {}
    t_all = time.perf_counter()

    def guarded(key, fn):  # an extra must never cost the line its headline value
        try:
            out[key] = fn()
        except Exception as e:  # noqa: BLE001
            out[key] = {"value": None, "error": repr(e)[:300]}
        torch.cuda.empty_cache()

    def env_stats(env, before):
        return {k: env.stats[k] - before.get(k, 0) for k in env.stats}

    def regimes():
        """configs[1] with episodes that END (reference world_model_env.py:77-82, env_loop.py:45-56: resets, V(final observation),
        burn-in inside the window): the synthetic reward/end model ends every env with probability p per step; `lockstep` starts
        from synchronised episodes (what 3 windows after a fresh start look like), `steady_state` from episode lengths spread over
        the horizon (what the reference's training loop converges to: B / horizon truncations at every step)."""
        agent, _, w = rollout_setup(device, 0, 64, 256, 15, 3, 1, (0, 0, 0, 0))
        env, res = w.env, {}
        base_dt, base_ms = measured_windows(w, 3, 2)
        res["no_ends"] = {"value": 256 * 15 / base_dt, "step_ms": base_ms}
        for name, p, stagger in (("p=0.003", 0.003, False), ("p=0.01", 0.01, False), ("p=0.5", 0.5, False),
                                 ("steady_state p=0", 1e-9, True), ("steady_state p=0.003", 0.003, True), ("steady_state p=0.01", 0.01, True)):
            set_end_rate(agent, p)
            env.reset_statistics()  # (a new regime: the env's running mean of ends per step starts over)
            if stagger:
                stagger_episodes(env, 15)
            else:
                env.set_episode_lengths(torch.zeros(256, dtype=torch.long))
            w()  # (the regime's own warm-up: the running averages of the speculation decision settle)
            before = dict(env.stats)
            dt, ms = measured_windows(w, 3, 1)
            st = env_stats(env, before)
            n = max(1, st["steps"])
            res[name] = {"value": 256 * 15 / dt, "step_ms": ms, "vs_no_ends": base_dt / dt, "steps_with_deaths": st["steps_with_deaths"] / n,
                         # (a step's deaths are resolved on the device into reset slots, sized before the deaths exist: env_loop.py)
                         "dead_rows_per_step": st["dead_rows"] / n, "reset_slots_per_step": st["slots"] / n,
                         "slot_overflows": st["slot_overflows"], "pool_rounds": st["pool_rounds"]}
        res["unit"] = "frames/s"
        res["workload"] = ("configs[1], end probability p per env-step through the synthetic reward/end head, 2 + 1 warm-up and 3 timed "
                           "windows per line; vs_no_ends = this line / the no-ends line measured by the same agent in the same process")
        return res

    def unbiased():
        # configs[1] with the UNBIASED synthetic reward/end model: ~half of the envs end at every step -> resets + reward/end
        # burn-in (reference world_model_env.py:77-89, env_loop.py:45-56) inside the timed window
        _, _, w = rollout_setup(device, 0, 64, 256, 15, 3, 1, (0, 0, 0, 0), bias_end_logits=False)
        dt, ms = measured_windows(w, 3, 2)
        return {"value": 256 * 15 / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "steps": 3, "warmup": 2, "step_ms": ms,
                "workload": "configs[1] without the end-logit bias: mid-window resets and burn-in passes in the timed window"}

    def config(idx, steps, warmup, roofline=False):
        c = CONFIGS[idx]
        attn = tuple(int(v) for v in c["attn_depths"].split(","))
        _, _, w = rollout_setup(device, 0, c["img_size"], c["batch"], c["horizon"], c["denoise_steps"], c["order"], attn)
        dt, ms = measured_windows(w, steps, warmup)
        flop_pf, _, _ = algorithmic_work(c["img_size"], c["denoise_steps"], c["order"], attn)
        fps = c["batch"] * c["horizon"] / dt
        res = {"value": fps, "unit": "frames/s", "ms_per_step": 1e3 * dt, "steps": steps, "warmup": warmup, "step_ms": ms,
               "global_batch": c["batch"], "algorithmic_tflops": fps * flop_pf / 1e12,
               "workload": f"{c['img_size']}x{c['img_size']}, batch {c['batch']}, {c['denoise_steps']} denoise steps order {c['order']}, "
                           f"attention {c['attn_depths']}"}
        if roofline:
            res["roofline"] = dominant_kernel_roofline(w, nv, idx)
        if idx == 4 and not args.no_cpu_baseline and getattr(args, "cpu_baseline_kind", "reference") == "reference":
            # (north_star: "256x256x3 CSGO batches ... alongside the reference CPU path timed on the same box's host cores")
            del w
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline_config4()
        return res

    def latency():
        lat = latency_line(SimpleNamespace(steps=200, warmup=12), eager=False)
        return dict({k: lat[k] for k in ("value", "unit", "steps", "frames_per_s", "sampler_only_graph_ms", "algorithmic_tflops", "frac_f16_mfma_peak")},
                    workload=lat["config"]["workload"])

    def train():
        tr = train_line(SimpleNamespace(batch=32, steps=20, warmup=3), eager=False)
        return dict({k: tr[k] for k in ("value", "unit", "steps", "frames_per_s", "loss", "algorithmic_tflops", "frac_f16_mfma_peak", "roofline")},
                    workload=tr["config"]["workload"] + ", batch 32")

    guarded("end_rate", regimes)
    guarded("unbiased_end_logits", unbiased)
    guarded("configs[3]", lambda: config(3, 3, 1))
    guarded("configs[4]", lambda: config(4, 3, 2, roofline=True))
    guarded("latency", latency)
    guarded("train", train)
    out["seconds"] = time.perf_counter() - t_all
    return out


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv, dry=False):
    """Re-execute this script under torch.distributed.run with N ranks on this node (127.0.0.1 rendezvous: the container hostname
    may not resolve).  Returns the launcher's exit code; the children inherit stdout, so rank 0's JSON line is this process's."""
    import subprocess

    argv = [a for a in argv if a != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    if dry:
        print(json.dumps({"self_launch": cmd, "env": {k: env[k] for k in ("MASTER_ADDR", "HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")},
                          "nproc_per_node": n}))
        return 0
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=str, default="1", choices=[str(k) for k in sorted(CONFIGS)] + ["latency", "train"],
                    help="BASELINE.json configs[] index (1, 3 or 4), or one of the SURVEY §8 'next' rows: latency = B=1 "
                         "interactive world-model env (f4, play.py), train = denoiser training step at the reference batch (f2)")
    ap.add_argument("--batch", type=int, default=None, help="imagination batch PER GPU")
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--order", type=int, default=None, choices=(1, 2))
    ap.add_argument("--img-size", type=int, default=None)
    ap.add_argument("--attn-depths", type=str, default=None,
                    help="denoiser attention per level; BASELINE configs[4] (256x256) uses 0,0,1,1")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed code path (RCCL process group, parameter broadcast, gradient all-reduce, replica "
                         "checksum) even at world size 1 (tests/test_gpu_dist.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-kind", choices=("reference", "port"), default="reference",
                    help="reference (default): time the reference's own modules where their bytecode travelled with the snapshot "
                         "(oracle/_ref, built by oracle/make_ref.py from /root/reference in the build container) beside the oracle port; "
                         "port: never execute oracle/_ref, time the oracle port only and quote the recorded reference number")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-exact-fp32", action="store_true")
    ap.add_argument("--no-end-logit-bias", action="store_true",
                    help="the synthetic reward/end model unbiased: ~half of the envs end at every step (mid-window resets + burn-in in the timed region)")
    ap.add_argument("--end-rate", type=float, default=None,
                    help="every env ends with this probability per step (through the synthetic reward/end head): mid-window resets at a stated rate")
    ap.add_argument("--stagger", action="store_true",
                    help="start the timed windows from episode lengths spread over the horizon (the steady state of the reference's "
                         "training loop: batch / horizon truncations at every step); the default of configs[1] together with "
                         f"--end-rate {STEADY_STATE_END_RATE}")
    ap.add_argument("--no-ends", action="store_true",
                    help="configs[1] as rounds 1-5 measured it: synchronised episodes, nobody ends mid-window (all envs truncate together "
                         "at the window boundary)")
    ap.add_argument("--foreach-adamw", action="store_true", help="--config train: the capturable FOREACH AdamW instead of the fused one (A/B)")
    ap.add_argument("--gc", choices=("default", "off", "window"), default="default",
                    help="A/B of python's cyclic garbage collector during the timed region: off = gc.disable(); window = disabled, with one "
                         "gc.collect() between windows (outside no kernel's critical path)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra measurements the default configs[1] line carries (`also`)")
    ap.add_argument("--pmc-calibrate", action="store_true",
                    help="two Heun updates over 256 MiB arrays before the window (tools/pmc_collect.sh: a known byte count for the FETCH_SIZE / "
                         "WRITE_SIZE unit corrections in the same rocprofv3 pass)")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no launcher environment: print the command / environment the self-launch would run "
                         "(one JSON line) instead of running it")
    ap.add_argument("--self-launch", action="store_true",
                    help="take the self-launch route at any N (tests/test_gpu_dist.py runs it at N = 1 on the one-GPU box; the ranks "
                         "then run the distributed branch as with --force-dist)")
    args = ap.parse_args()
    if (args.gpus > 1 or args.self_launch) and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, one process per GPU, like the
        # reference's main.py:22-27 (mp.spawn over torch.cuda.device_count()).  Rank 0 of the children prints the JSON line.
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.dry_launch))
    if args.config in ("latency", "train"):
        assert args.gpus == 1, "--config latency / train are single-GPU measurements"
        torch.cuda.set_device(0)
        line = latency_line(args) if args.config == "latency" else train_line(args)
        print(json.dumps(line))
        return
    args.config = int(args.config)
    preset = CONFIGS[args.config]
    # configs[1] with no regime flag = the steady state of the reference's loop (STEADY_STATE_NOTE)
    steady_default = (args.config == 1 and not (args.no_ends or args.no_end_logit_bias or args.stagger or args.end_rate is not None))
    if steady_default:
        args.stagger, args.end_rate = True, STEADY_STATE_END_RATE
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.steps is None:
        args.steps = 3 if args.config == 1 else (1 if args.config == 3 else 2)
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
    use_dist = world > 1 or args.force_dist or args.self_launch
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)  # "nccl" == RCCL on ROCm

    import diamond_amd as D
    from diamond_amd import ac_native
    from diamond_amd import engine as E
    from diamond_amd import native as nv
    from diamond_amd.dist import parameter_checksum

    if args.pmc_calibrate:
        # a kernel the window itself never launches (configs[1] is Euler-only), with a known byte count: 2 launches x
        # (4 x 256 MiB read, 256 MiB written)
        cal = [torch.randn(64 * 1024 * 1024, device=device) for _ in range(4)]
        for _ in range(2):
            D.DiffusionSampler._heun(cal[0], cal[1], cal[2], cal[3], 1.0, 0.5, -0.5)
        torch.cuda.synchronize()
        del cal
    torch.manual_seed(1234 + rank)
    attn = tuple(int(v) for v in args.attn_depths.split(","))
    agent, ac, window = rollout_setup(device, rank, args.img_size, args.batch, args.horizon, args.denoise_steps, args.order, attn, use_dist,
                                      bias_end_logits=not args.no_end_logit_bias, end_rate=args.end_rate)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def progress(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    progress("setup done")
    if args.gc != "default":
        import gc

        gc.collect()
        gc.disable()
        if args.gc == "window":
            inner = window

            def window():  # noqa: F811
                out = inner()
                gc.collect()
                return out

            window.reducer, window.env = inner.reducer, inner.env
    if args.stagger and args.warmup == 0:
        args.warmup = 1  # (the staggered state is set behind the env's first window: there is no env state before it)
    for i in range(args.warmup):
        window()
        if args.stagger and i == 0:
            stagger_episodes(window.env, args.horizon)
    fence()
    progress("warmup done")
    if use_dist:
        window.reducer.timing = True
    # per-step stamps WITHOUT synchronising: events on the current stream at the window boundaries, read after the fence
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        window()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if use_dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # replicas must still hold identical actor-critic parameters after the all-reduced steps
        cs = torch.tensor([parameter_checksum(ac)], device=device, dtype=torch.float64)
        gathered = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(gathered, cs)
        replicas_in_sync = all(float(g) == float(gathered[0]) for g in gathered)
        # self-description of the N > 1 run: what RCCL itself says the world is, every rank's own mean step time, and the device
        # time of the one gradient all-reduce per step (events around it on this rank)
        window.reducer.timing = False
        ar_ms = window.reducer.elapsed_ms()
        mine = torch.tensor([sum(step_ms) / len(step_ms), sum(ar_ms) / max(1, len(ar_ms))], device=device, dtype=torch.float64)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        dist_info = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "per_rank_step_ms": [round(float(t[0]), 2) for t in per_rank],
                     "allreduce_ms_per_step": [round(float(t[1]), 3) for t in per_rank],
                     "grad_bucket_mb": window.reducer.bucket.numel() * 4 / 1e6,
                     "early_slice_mb": window.reducer.early_numel * 4 / 1e6, "early_slice_allreduces_inside_backward": window.reducer.early_launches}
        if not replicas_in_sync and rank == 0:
            print(f"[bench] WARNING: replicas diverged: checksums {[float(g) for g in gathered]}", file=sys.stderr, flush=True)

    else:
        replicas_in_sync, dist_info = None, None
    progress(f"timed region done: {elapsed:.2f}s for {args.steps} steps")
    custom = any(getattr(args, k) != v for k, v in preset.items()) or args.no_end_logit_bias or \
        ((args.end_rate is not None or args.stagger or args.no_ends) and not steady_default)
    custom_flags = custom
    cfg_idx = args.config if world == 1 or args.config != 1 else 2
    cfg_name = f"configs[{cfg_idx}]" + (" (modified by flags)" if custom else "") + \
        (" (sharded over the GPUs)" if world > 1 else "")
    sampler = f"{args.denoise_steps} Euler denoise steps" if args.order == 1 else \
        f"{args.denoise_steps}-step 2nd-order Heun ({2 * args.denoise_steps - 1} denoiser calls per frame)"
    flop_pf, bytes_pf, den_calls = algorithmic_work(args.img_size, args.denoise_steps, args.order, attn)
    frames = args.batch * world * args.horizon * args.steps
    fps = frames / elapsed
    sz = f"{args.img_size}x{args.img_size}"
    metric = {1: "imagined frames/sec (64x64, 3 denoise steps, batch 256)",
              3: "imagined frames/sec (64x64, 50-step 2nd-order Heun, batch 256)",
              4: "imagined frames/sec (256x256, 3 denoise steps, 8 envs per GPU, attention [0,0,1,1])"}[args.config]
    line = {
        "metric": metric, "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "step_ms": [round(v, 2) for v in step_ms],  # (device time between window boundaries, this rank; diagnostic)
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": precision_label(E, ac_native), "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {sz}x3 frames, batch {args.batch}/GPU, horizon {args.horizon}, {sampler}"
                               f"{', denoiser attention at levels ' + args.attn_depths if any(attn) else ''}; step = "
                               f"ActorCritic.forward()+backward+all-reduce+clip+AdamW over one {args.horizon}-step imagined window; "
                               + ("UNBIASED synthetic end-logits: mid-window resets and burn-in inside the timed region" if args.no_end_logit_bias
                                  else STEADY_STATE_NOTE.format(p=args.end_rate) if steady_default
                                  else f"every env ends with probability {args.end_rate} per step (synthetic reward/end head)" if args.end_rate is not None
                                  else END_LOGIT_BIAS_NOTE)
                               + ("; episode lengths staggered over the horizon" if args.stagger and not steady_default else ""),
                   "regime": ("steady_state p=%g" % args.end_rate) if (args.stagger and args.end_rate is not None) else
                             ("unbiased" if args.no_end_logit_bias else ("p=%g" % args.end_rate if args.end_rate is not None else "no_ends")),
                   "env_loop": os.environ.get("DIAMOND_ENV_LOOP", "slots"),
                   "env_stats": dict(getattr(window.env, "stats", {})),
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (batch-sharded envs, flat-bucket "
                   "RCCL all-reduce of actor-critic grads)", "actor_critic_backend": ac.backend,
                   "world_model_precision": E.WORLD_MODEL_PRECISION, "actor_critic_precision": ac_native.AC_PRECISION,
                   "replicas_in_sync": replicas_in_sync, "distributed": dist_info},
        "whole_step_algorithmic": {"gflop_per_frame": flop_pf / 1e9, "mb_per_frame": bytes_pf / 1e6,
                                   "tflops": fps * flop_pf / 1e12 / world,
                                   "frac_fp32_peak": fps * flop_pf / 1e12 / world / FP32_MFMA_PEAK_TFLOPS,
                                   "hbm_gbs": fps * bytes_pf / 1e9 / world,
                                   "frac_hbm_peak": fps * bytes_pf / 1e9 / world / HBM_PEAK_GBS,
                                   "note": "north_star asks for >= 10k frames/s at >= 50 % of the HBM roofline; the path's arithmetic intensity is "
                                           "118 FLOP/B (SURVEY 8d), so 50 % of 8 TB/s would be 23.7k frames/s = 452 TFLOP/s of fp32 convolution, "
                                           "2.9x the exact-fp32 peak: the two targets are not self-consistent; the path is matrix-pipe / power "
                                           "bound (roofline), its HBM share is what these FLOPs need"},
    }

    if not args.no_roofline:
        # instrumented window (not part of `value`): HIP events around EVERY C-ABI launch (convolutions, attention, linears,
        # fused low-resolution levels, pointwise) on the stream they are launched on; the env replays no captured graph
        # while a profiler is installed (a replay would hide its kernels from the events)
        line["roofline"] = dominant_kernel_roofline(window, nv, args.config, world, custom_flags)
        progress("roofline window done")

    if args.config == 1 and not args.no_exact_fp32 and not custom:
        # the same window with every convolution on the exact-fp32 MFMA kernels (reported next to `value`, never as it)
        saved = (E.WORLD_MODEL_PRECISION, ac_native.AC_PRECISION)
        E.WORLD_MODEL_PRECISION, ac_native.AC_PRECISION = "f32", "f32"
        window()
        fence()
        t1 = time.perf_counter()
        window()
        fence()
        dt = time.perf_counter() - t1
        if use_dist:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        E.WORLD_MODEL_PRECISION, ac_native.AC_PRECISION = saved
        line["exact_fp32"] = {"value": args.batch * world * args.horizon / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "steps": 1,
                              "dtype": "f32 (every convolution on v_mfma_f32_16x16x4_f32: DIAMOND_CONV_PRECISION=f32 "
                                       "DIAMOND_AC_PRECISION=f32)"}
        progress("exact-fp32 window done")

    if args.config == 1 and world == 1 and not custom and not args.no_also:
        del agent, ac, window
        torch.cuda.empty_cache()
        line["also"] = also_lines(device, args)
        progress(f"also: {line['also']['seconds']:.1f}s")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_config4() if args.config == 4 else cpu_baseline(64, allow_reference=args.cpu_baseline_kind == "reference")
        progress("cpu baseline done")

    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
