"""GraphedTrainStep (diamond_amd/train_graph.py): the denoiser training step of the reference's trainer loop
(/root/reference/src/trainer.py:349-388: model(batch) -> backward -> clip -> AdamW) captured into one hipGraph must do
exactly what the eager step does -- same losses, same parameters after the same number of steps -- including weight
re-packing at every replay (a graph that kept reading the packed weights of capture time would stop learning)."""
from types import SimpleNamespace

import pytest
import torch

from tests.test_gpu_models import make_agent

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(seed, fused=False):
    import diamond_amd as D
    from diamond_amd.testing import synthetic_actions, synthetic_frames

    ag = make_agent()
    den = ag.denoiser
    den.train()
    den.setup_training(D.SigmaDistributionConfig(loc=-0.4, scale=1.2, sigma_min=2e-3, sigma_max=20))
    g = torch.Generator().manual_seed(seed)
    b, t = 4, 6
    batches = []
    for k in range(3):
        mask = torch.ones(b, t, dtype=torch.bool)
        mask[k, 5] = False
        batches.append(SimpleNamespace(obs=synthetic_frames(g, b, t, 3, 64, 64).to(DEV), act=synthetic_actions(g, 4, b, t).to(DEV),
                                       mask_padding=mask.to(DEV)))
    # deterministic "noise" that lives on the device (host RNG draws would be frozen into the graph as constants):
    # a fixed table indexed by shape, the same at every step for both runs
    table = {}

    def randn_fn(shape):
        if shape not in table:
            table[shape] = torch.randn(*shape, generator=torch.Generator().manual_seed(len(table) + 99)).to(DEV)
        return table[shape]

    den.randn_fn = randn_fn
    # fused=True: torch's single-kernel AdamW (the capturable foreach form divides every tensor by its 0-dim bias corrections with
    # one broadcast kernel each: 2 x 236 launches per step)
    opt = torch.optim.AdamW(den.parameters(), lr=3e-4, capturable=True, fused=fused)
    return den, opt, batches


def _eager_step(den, opt, batch):
    loss, _ = den(batch)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(den.parameters(), 1.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss.detach()


@pytest.mark.parametrize("fused", [False, True], ids=["foreach", "fused"])
def test_graphed_training_step_matches_the_eager_loop(fused):
    from diamond_amd.train_graph import GraphedTrainStep

    warm, steps = 2, 4
    den, opt, batches = _setup(5, fused)
    init = {k: v.detach().clone() for k, v in den.state_dict().items()}
    losses_e = []
    for i in range(warm):
        _eager_step(den, opt, batches[0])
    for i in range(steps):
        losses_e.append(float(_eager_step(den, opt, batches[i % 3])))
    params_e = {k: v.detach().clone() for k, v in den.named_parameters()}

    den2, opt2, batches2 = _setup(5, fused)
    den2.load_state_dict(init)
    gstep = GraphedTrainStep(den2, opt2, 1.0, batches2[0], warmup_steps=warm)
    losses_g = []
    for i in range(steps):
        loss, metrics = gstep(batches2[i % 3])
        losses_g.append(float(loss))
        assert float(metrics["loss_denoising"]) == losses_g[-1]
    torch.cuda.synchronize()
    print("eager", losses_e, "graph", losses_g)
    assert len(set(losses_g)) == steps, "the replayed step must see the new batch / the updated weights"
    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) <= 1e-6 * abs(a), (losses_e, losses_g)
    worst = 0.0
    for k, p in den2.named_parameters():
        d = float((p.detach() - params_e[k]).abs().max() / params_e[k].abs().max().clamp_min(1e-12))
        worst = max(worst, d)
    moved = max(float((p.detach() - init[k]).abs().max()) for k, p in den2.named_parameters())
    assert moved > 1e-4, "parameters did not train"
    assert worst < 1e-5, f"parameters after {warm}+{steps} steps differ from the eager loop by {worst:.3e}"


def test_graphed_step_rejects_a_non_capturable_optimizer_and_other_shapes():
    from diamond_amd.train_graph import GraphedTrainStep

    den, _, batches = _setup(6)
    with pytest.raises(AssertionError, match="capturable"):
        GraphedTrainStep(den, torch.optim.AdamW(den.parameters(), lr=1e-4), 1.0, batches[0])
