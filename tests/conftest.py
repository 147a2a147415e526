import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# DIAMOND_TESTS_ON_INTERPRETER=1 (development aid, off by default): run the `-m gpu` tests WITHOUT a GPU, on the SIMT-interpreter
# build of the kernels (tests/simt) -- a pre-flight for test logic and host code before GPU minutes are spent on it.  Tests
# that need the device itself (hipGraphs, events, RCCL, batch-256 shapes) fail or crawl there; it proves nothing about the GPU.
ON_INTERPRETER = os.environ.get("DIAMOND_TESTS_ON_INTERPRETER") == "1"


@pytest.fixture(scope="session", autouse=ON_INTERPRETER)
def _gpu_tests_on_the_interpreter():
    if not ON_INTERPRETER:
        yield
        return
    from tests.simt.host_harness import engine_on_interpreter

    torch.cuda.synchronize = lambda *a, **k: None
    with engine_on_interpreter():
        yield


def pytest_collection_modifyitems(config, items):
    if ON_INTERPRETER:
        for item in items:  # the device of the `-m gpu` test modules (module-scoped fixtures read it too)
            if hasattr(item.module, "DEV"):
                item.module.DEV = "cpu"
        return
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def reload_dmd_env():
    """The libraries cache their DIAMOND_* switches (DmdEnvInt, csrc/dmd_common.h): re-read them in whichever is loaded."""
    from diamond_amd import native as nv
    from tests.simt import loader as S

    for mod in (nv, S):
        L = getattr(mod, "_lib", None)
        if L is not None:
            L.dmd_reload_env()


@pytest.fixture
def dmd_env(monkeypatch):
    """dmd_env(DIAMOND_WGRAD_MAX_WG=7, ...): library switches for the duration of one test (None = unset)."""
    def set_(**kv):
        for k, v in kv.items():
            monkeypatch.delenv(k, raising=False) if v is None else monkeypatch.setenv(k, str(v))
        reload_dmd_env()

    yield set_
    monkeypatch.undo()
    reload_dmd_env()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


WEIGHT_SEED = 7


@pytest.fixture(scope="session")
def oracle_agent():
    """AgentSD (oracle state dicts) with the same name-keyed weights the fixtures were made with."""
    return make_oracle_agent()


def make_oracle_agent(dtype=torch.float32, attn_depths=(0, 0, 0, 0)):
    from diamond_amd.testing import fill_state_dict_
    from oracle import diamond_oracle as O

    shapes = load_golden("state_dict_keys.pt")
    sd = {k: torch.empty(s) for k, s in shapes.items()}
    if any(attn_depths):
        # attention parameters of the extra attention blocks (not in the default tree)
        lvls = [i for i, a in enumerate(attn_depths) if a]
        L = len(attn_depths)
        for lvl in lvls:
            for grp, n in ((f"d_blocks.{lvl}", 2), (f"u_blocks.{L - 1 - lvl}", 3)):
                for i in range(n):
                    p = f"denoiser.inner_model.unet.{grp}.resblocks.{i}.attn"
                    sd[p + ".norm.norm.weight"] = torch.empty(64)
                    sd[p + ".norm.norm.bias"] = torch.empty(64)
                    sd[p + ".qkv_proj.weight"] = torch.empty(192, 64, 1, 1)
                    sd[p + ".qkv_proj.bias"] = torch.empty(192)
                    sd[p + ".out_proj.weight"] = torch.empty(64, 64, 1, 1)
                    sd[p + ".out_proj.bias"] = torch.empty(64)
    fill_state_dict_(sd, WEIGHT_SEED)

    def sub(prefix):
        return {k[len(prefix) + 1:]: v.to(dtype) for k, v in sd.items() if k.startswith(prefix + ".")}

    return O.AgentSD(denoiser=sub("denoiser"), rew_end_model=sub("rew_end_model"), actor_critic=sub("actor_critic"),
                     dspec=O.DenoiserSpec(attn_depths=tuple(attn_depths)))
