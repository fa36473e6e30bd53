import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_ORACLE_WEIGHTS = {}      # (config, seed, dense) -> state dict of the drawn oracle UNet; at most two sets stay alive (6 GB each at SD1.5 widths)


def oracle_unet(cfg, num_views, num_frames, latent_hw, seed=0, dense=True):
    """The oracle UNet with ``init_synthetic_weights(seed, dense)`` weights (oracle.unet_ref.build_dense: same values as the
    two-step construction).  Weight sets are drawn once per session and shared, read-only, by every test that asks for the same
    (config, seed) — at the SD1.5 widths a draw is 1.5e9 values; the GPU tier used to spend minutes re-drawing them."""
    from oracle import unet_ref as O
    key = (repr(cfg), seed, dense)
    if key in _ORACLE_WEIGHTS:
        return O.build_dense(cfg, num_views, num_frames, latent_hw, state_dict=_ORACLE_WEIGHTS[key])
    model = O.build_dense(cfg, num_views, num_frames, latent_hw, seed=seed, dense=dense)
    while len(_ORACLE_WEIGHTS) >= 2:
        _ORACLE_WEIGHTS.pop(next(iter(_ORACLE_WEIGHTS)))
    _ORACLE_WEIGHTS[key] = {k: v.detach() for k, v in model.state_dict().items()}
    return model


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "processors.npz"))
