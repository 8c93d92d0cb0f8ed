import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: whole-model CPU oracle run (tens of seconds)")


@pytest.fixture(scope="session")
def cfg():
    from pgtformer_amd.config import default_config

    return default_config()


@pytest.fixture(scope="session")
def manifest(cfg):
    from pgtformer_amd.manifest import pgtformer_manifest

    return pgtformer_manifest(cfg)


@pytest.fixture(scope="session")
def full_sd(manifest, cfg):
    """All 961 synthetic tensors (about 2 s, 520 MB)."""
    from pgtformer_amd.weightgen import generate_state_dict

    return generate_state_dict(manifest, cfg, seed=0)


@pytest.fixture(scope="session")
def golden_window():
    """The (3,3,512,512) fp32 input window the whole-model goldens were generated on."""
    import numpy as np
    import torch
    from pgtformer_amd.synth import make_clip, window_from_clip

    lq_u8, gt = make_clip(4, 512, seed=1234)
    win = window_from_clip(lq_u8, 1)
    x = torch.from_numpy(win.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    return x, win, gt
