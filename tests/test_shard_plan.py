"""mic_hip_shard_plan (csrc/mic_host_io.hip): the contiguous, pixel-balanced cut of a batch's jobs over the devices of
mic_hip_set_devices -- the static assignment of the reference's fan-outs (parallelstrips.go:77-93).  Pure host logic: no GPU."""
import numpy as np


def _check(first, n, shards):
    assert len(first) == shards + 1 and first[0] == 0 and first[-1] == n
    assert all(a <= b for a, b in zip(first, first[1:]))


def test_equal_weights_cut_evenly(mic):
    for n, shards in ((8, 2), (64, 8), (10, 3), (7, 7), (5, 8), (0, 4), (1, 1)):
        first = mic.shard_plan([1000] * n, shards)
        _check(first, n, shards)
        sizes = [b - a for a, b in zip(first, first[1:])]
        assert max(sizes) - min(sizes) <= 1 or n < shards


def test_weighted_cut_is_balanced_to_within_one_item(mic):
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = int(rng.integers(1, 400)); shards = int(rng.integers(1, 9))
        w = rng.integers(1, 6_000_000, n)
        first = mic.shard_plan(w, shards)
        _check(first, n, shards)
        total = int(w.sum())
        for k in range(shards):
            got = int(w[first[k]: first[k + 1]].sum())
            assert got <= total / shards + int(w.max()) + 1      # never more than its share plus one item
        # the prefix up to boundary k reaches k / shards of the total, and would not without its last item
        for k in range(1, shards):
            pre = int(w[: first[k]].sum())
            assert pre >= (total * k) // shards or first[k] == n
            if first[k] > 0 and first[k] > first[k - 1]:
                assert pre - int(w[first[k] - 1]) < (total * k) // shards + 1


def test_zero_weights_and_bad_arguments(mic):
    first = mic.shard_plan([0, 0, 0, 0], 2)
    _check(first, 4, 2)
    assert first[1] == 2                                         # (an item of no weight counts as one pixel)
    import pytest
    with pytest.raises(mic.MicError):
        mic.shard_plan([1, 2], 0)
