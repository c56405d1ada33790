import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as data:
        return {key: torch.from_numpy(np.asarray(data[key])) for key in data.files}


@pytest.fixture
def golden():
    return load_golden


RENDER_CASES = [
    "g4_render_n4_s32_step0",
    "g4_render_n4_s32_mid",
    "g4_render_n4_s32_late",
    "g4_render_n3_s20_mid",
    "g4_render_n16_s64_mid",
    "g4_render_n1_s32_late",
    "g17_render_n64_s128_mid",              # BASELINE config 5 shape: 4 wave rounds in pass 2, 64 instances
]
RESIDUAL_CASES = [
    "g10_render_residual_n3_s16",
    "g17_render_residual_n16_s64_mid",      # BASELINE config 3 shape: 2 wave rounds, 16 instances
    "g17_render_residual_n4_s100_late",     # the reference's own num_fine_samples: 4 wave rounds (199 points)
]
