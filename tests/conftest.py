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


@pytest.fixture(autouse=True)
def seeded_generators():
    """Every test starts from the same state of torch's global generators, run alone or in the suite: modules draw their initial parameters
    from them at construction (a detector's embeddings, a hypernetwork), and the renderer's conditioning -- one fine sample on the other side
    of a plateau -- turns a different draw into a different set of rays in the 1e-4 tail (round 5: a full-size parity test passed alone and
    failed in the suite on exactly that)."""
    torch.manual_seed(20240229)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(20240229)
    yield


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

# ---- observed error margins -----------------------------------------------------------------------------------------------
# Parity tests call margin(...) with what they observed next to what they tolerate; the table is printed at the end of the run
# (pytest -q shows it: it is part of the terminal summary, not captured output), so the record of a GPU test run carries the
# margins and not only "passed".
_MARGINS = []


def margin(test, what, observed, tolerance):
    """Record `observed` (a float) against `tolerance` under the label `test` / `what`; returns observed."""
    observed = float(observed)
    _MARGINS.append((str(test), str(what), observed, float(tolerance)))
    return observed


def pytest_terminal_summary(terminalreporter):
    if not _MARGINS:
        return
    worst = {}
    for test, what, observed, tolerance in _MARGINS:            # the worst case of every (test function, quantity) over its parametrisations
        key = (test.split("[")[0], what)
        ratio = observed / tolerance if tolerance > 0 else float("inf")
        if key not in worst or ratio > worst[key][0]:
            worst[key] = (ratio, test, observed, tolerance)
    terminalreporter.write_line(f"observed error margins ({len(_MARGINS)} records; worst case per test and quantity, then every record above half its tolerance)")
    for (name, what), (ratio, test, observed, tolerance) in sorted(worst.items()):
        terminalreporter.write_line(f"  {test[:70]:70s} {what[:26]:26s} {observed:9.2e} / {tolerance:8.1e} = {ratio:5.2f}")
    shown = {(entry[1], what) for (_, what), entry in worst.items()}
    for test, what, observed, tolerance in _MARGINS:
        ratio = observed / tolerance if tolerance > 0 else float("inf")
        if ratio >= 0.5 and (test, what) not in shown:
            terminalreporter.write_line(f"  {test[:70]:70s} {what[:26]:26s} {observed:9.2e} / {tolerance:8.1e} = {ratio:5.2f}")
