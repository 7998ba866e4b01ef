"""CPU: dsmi_plan_shards (the native host's shard plan, comm.hip) against danspeech_amd.parallel.plan_shards, the plan the
world-2 gloo tests exercise end to end.  Host arithmetic only: no GPU, no RCCL."""
import numpy as np
import pytest


@pytest.mark.parametrize("n,world,seed", [(0, 2, 0), (1, 4, 1), (3, 8, 2), (32, 2, 3), (257, 8, 4), (64, 1, 5), (100, 3, 6)])
def test_native_plan_equals_python_plan(n, world, seed):
    from danspeech_amd import _native, parallel
    rng = np.random.default_rng(seed)
    lengths = rng.integers(1, 50, size=n).astype(np.int64) * 160            # few distinct values: plenty of ties (stable order matters)
    rank_of, slot_of = _native.plan_shards(lengths, world)
    shards = parallel.plan_shards(lengths, world)
    for r, idx in enumerate(shards):
        for slot, i in enumerate(idx):
            assert rank_of[i] == r and slot_of[i] == slot
        assert all(lengths[idx[j]] >= lengths[idx[j + 1]] for j in range(len(idx) - 1))     # every shard longest first
    assert sorted(np.concatenate(shards).tolist() if n else []) == list(range(n))


def test_config5_shaped_plan_is_balanced_over_eight_ranks():
    """BASELINE.json configs[4]: 1024 ragged clips of up to 30 s over 8 ranks.  Every rank gets 128 clips, longest first, and a
    share of the samples within 2 % of the mean (longest-first round-robin dealing: rank r gets the clips of rank r, r + 8, ...
    in length order), from both plans."""
    from danspeech_amd import _native, parallel
    rng = np.random.default_rng(55)
    lengths = (rng.uniform(4.0, 30.0, size=1024) * 16000).astype(np.int64)
    lengths[:8] = 480000
    rank_of, slot_of = _native.plan_shards(lengths, 8)
    shards = parallel.plan_shards(lengths, 8)
    totals = []
    for r, idx in enumerate(shards):
        assert len(idx) == 128
        assert all(rank_of[i] == r and slot_of[i] == k for k, i in enumerate(idx))
        assert all(lengths[idx[j]] >= lengths[idx[j + 1]] for j in range(len(idx) - 1))
        totals.append(int(lengths[idx].sum()))
    mean = float(np.mean(totals))
    assert max(abs(t - mean) for t in totals) < 0.02 * mean, totals


def test_native_plan_rejects_bad_arguments():
    from danspeech_amd import _native
    with pytest.raises(_native.DsmiError):
        _native.plan_shards([160, 320], 0)
