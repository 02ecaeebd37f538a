"""A short, seeded slice of the randomised soak (tests/fuzz_gpu.py) inside the collected GPU suite: random sizes, curves,
input shapes (cancellation, doubling, identities), every tuning option incl. the workspace guards, the sharded entries
with simulated ranks, divisor-witness forests and scalar-witness batches -- every case against the oracle."""
import pytest

import fuzz_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [31337, 20261004])
def test_fuzz_slice(seed):
    assert fuzz_gpu.main(secs=25.0, seed=seed) >= 10
