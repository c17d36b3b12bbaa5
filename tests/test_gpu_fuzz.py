"""Randomized differential test, fixed seeds: HIP path == CPU oracle over random indexes / reads / parameters."""
import pytest

from tests.fuzz_common import run_trial

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(4))
def test_random_trials(block):
    tuples = 0
    for seed in range(1000 + 12 * block, 1000 + 12 * (block + 1)):
        ok, cfg, n = run_trial(seed)
        assert ok, cfg
        tuples += n
    assert tuples > 0
