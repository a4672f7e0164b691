"""GPU test of the expert-parallel all-to-all (BASELINE config 5: csrc/all_reduce.hip: semipd_ep_dispatch /
semipd_ep_combine) against the oracle permutation (oracle/ops.py: ep_dispatch / ep_combine), bit for bit, with 2 / 4 / 8
ranks as processes on the one GPU (regions exchanged through hipIpc handles as between GPUs; the wire is HBM instead of
xGMI).  The reference has no all-to-all to compare with (ep_moe/layer.py:190 keeps every token on every rank and
all-reduces): parity unpinned, the oracle is the definition."""
import pytest

from test_gpu_all_reduce import run_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dispatch_and_combine_match_the_oracle_permutation_bit_for_bit(device, world):
    reports = run_world(world, worker="ep_worker.py", timeout=300)
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:
        assert r["cases"] == 5 * 8 + 4 + 12 + 3 and not r["bad"], r["bad"][:6]
