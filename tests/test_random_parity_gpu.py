"""Random synthetic inputs through the product pipeline (code matrices: reads named by rows, distinct reads and known_paths'
per-read test on the device, read text on first use, native sparse flow and merge) and through the oracle pipeline (pure Python
restatement, strings): the same contigs, partitions and transcripts -- paired and single-end, K = 20 / 25 / 31, --partition 4 / 8 /
500.  tools/random_parity.py runs the same comparison over any number of seeds (200 + 40 larger ones were run for round 2)."""
import os, sys
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", list(range(3000, 3010)))
def test_random_input_equals_the_oracle(seed):
    import random_parity
    from shannon_amd import device
    ctx = device.Context(0)
    try:
        ok, text = random_parity.run_case(ctx, seed)
        assert ok, text
    finally:
        ctx.close()
