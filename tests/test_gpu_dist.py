"""The distributed code path on the GPU box (one MI355X): RCCL ("nccl" backend) at world size 1 under
torch.distributed.run, exactly as the driver launches bench.py for N > 1.  No scaling number comes out of this -- it
proves that process-group creation, the flat parameter broadcast, the flat-bucket all-reduce (actor-critic and world-model
parameters), torch's DistributedDataParallel around ActorCritic, and bench.py's N > 1 branch (barrier, max-over-ranks
timing, replica checksum) all EXECUTE on RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _torchrun(script_args, port, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + "\n" + r.stderr[-2000:]
    return json.loads(lines[-1])


def test_rccl_world1_broadcast_allreduce_ddp():
    out = _torchrun([os.path.join(ROOT, "tests", "dist_gpu_worker.py")], 29611)
    print(out)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["broadcast_keeps_values"] and out["broadcast_bumps_versions"]
    assert out["wm_allreduce_ok"] and out["wm_bucket_mb"] > 10
    assert out["grads_finite"] and out["loss_equal"]
    assert out["ddp_grad_rel_diff"] < 1e-6, out  # same kernels, same seed: DDP's bucketed all-reduce changes nothing at world 1
    # the LSTM / head slice went out on a side stream from inside backward() (RCCL async work + stream waits), same gradients
    assert out["early_slice_launched_inside_backward"] and out["early_grad_rel_diff"] < 1e-6 and out["early_slice_mb"] > 10, out


def test_bench_distributed_branch_world1():
    """bench.py's N > 1 branch (init_process_group, broadcast_parameters, GradAllReducer per step, barrier + max-over-ranks
    timing, replica checksum all_gather) at world size 1, launched like the driver launches N > 1."""
    line = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--batch", "16",
                      "--horizon", "3", "--no-roofline", "--no-exact-fp32", "--no-cpu-baseline"], 29612)
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["replicas_in_sync"] is True


def test_bench_self_launch_world1():
    """`python bench.py --gpus N` with no launcher around it starts its own ranks (the reference launches itself: main.py:22-27);
    here the same route at N = 1 (--self-launch), i.e. bench.py -> torch.distributed.run -> one rank on RCCL -> the JSON line on
    the parent's stdout."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--steps", "1", "--warmup", "0",
                        "--batch", "16", "--horizon", "3", "--no-roofline", "--no-exact-fp32", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    d = line["config"]["distributed"]
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["replicas_in_sync"] is True
    assert d["rccl_world_size"] == 1 and d["backend"] == "nccl" and len(d["per_rank_step_ms"]) == 1
    assert d["allreduce_ms_per_step"][0] > 0 and d["grad_bucket_mb"] > 10
