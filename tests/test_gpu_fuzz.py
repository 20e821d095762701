"""A slice of tools/fuzz_oracle.py as a test: random small scenes, frames, depths and kernel variants
against the CPU oracle, bit for bit."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    torch.cuda.set_device(0)
    return torch


def test_random_cases_match_the_oracle(pt, O, torch_cuda):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_oracle
    for seed in range(70000, 70400):
        ok, what = fuzz_oracle.one_case(seed)
        assert ok, (seed, what)
