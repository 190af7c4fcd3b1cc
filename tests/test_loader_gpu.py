"""Feature loader upload path: pinned staging -> asynchronous copies on a copy stream, two slots, one batch ahead."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prefetched_batches_equal_host_packs(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd.loader import Clip, DeviceBatcher, FeaturePacker, FeaturePrefetcher
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    table = []
    for i in range(12):
        S, Sa = int(rng.integers(3, 40)), int(rng.integers(2, 60))
        np.save(tmp_path / f"v{i}_rgb.npy", rng.random((S, 1024), dtype=np.float32))
        np.save(tmp_path / f"v{i}_flow.npy", rng.random((S, 1024), dtype=np.float32))
        if i != 5:                                       # one clip without audio
            np.save(tmp_path / f"v{i}.npy", rng.random((Sa, 128), dtype=np.float32))
        dur = float(rng.random() * 50 + 5)
        a, b = sorted(rng.random(2) * dur)
        table.append(Clip(f"v{i}", f"caption {i}", float(a), float(b), dur))
    batches = [[0, 1, 2, 3], [4, 5, 6], [7, 8, 9, 10], [11, 0, 5], [3, 2]]
    ref_packer = FeaturePacker(str(tmp_path), str(tmp_path), pad_idx=1, pin=False)
    batcher = DeviceBatcher(FeaturePacker(str(tmp_path), str(tmp_path), pad_idx=1), dev)
    n = 0
    for ix, batch in zip(batches, FeaturePrefetcher(batcher, table, batches)):
        want = ref_packer.pack([table[i] for i in ix])
        fs = batch["feature_stacks"]
        # a consumer kernel on the current stream (the wait was enqueued by the prefetcher): the sums force real reads
        got = {k: fs[k].clone() for k in ("rgb", "flow", "audio")}
        torch.cuda.synchronize()
        for k in got:
            assert got[k].is_cuda and tuple(got[k].shape) == tuple(want[k].shape)
            assert torch.equal(got[k].cpu(), want[k]), (n, k)
        assert batch["video_ids"] == [table[i].video_id for i in ix] and batch["captions"] == [table[i].caption for i in ix]
        assert tuple(batch["starts"].shape) == (len(ix), 1) and batch["starts"].is_cuda
        n += 1
    assert n == len(batches)
