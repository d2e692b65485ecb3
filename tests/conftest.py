import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def smpl_model():
    from danet_densepose2smpl_amd import assets
    return assets.make_synthetic_smpl(0)


@pytest.fixture(scope='session')
def dp_tables(smpl_model):
    from danet_densepose2smpl_amd import assets
    dp = assets.make_synthetic_densepose(smpl_model, 0)
    return assets.densepose_render_tables(dp)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def rand_pose_shape(B, seed=1234, pose_sigma=0.2):
    """BASELINE.json configs[0] inputs: betas ~ N(0,1) clipped +-3, pose ~ N(0, 0.2^2)."""
    rng = np.random.default_rng(seed)
    betas = np.clip(rng.normal(0, 1, (B, 10)), -3, 3)
    pose = rng.normal(0, pose_sigma, (B, 72))
    return betas, pose


def record(name, values):
    """Append one measured-error record of a parity test to gpurun_out/parity_measured.jsonl (scratch; the closing run copies
    it to profiles/) -- tolerances in the tests are set from these measurements, not guessed."""
    import json
    d = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_measured.jsonl'), 'a') as f:
            f.write(json.dumps({'test': name, 'measured': values}) + '\n')
    except OSError:
        pass
