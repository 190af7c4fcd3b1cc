"""Tuning aid: feature loader throughput at the bench workload (B=16, Tv=256, Ta=800) -- host pack time, and batches/s of the
prefetching pipeline next to a consumer that holds each batch for one training step's time."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd.loader import Clip, DeviceBatcher, FeaturePacker, FeaturePrefetcher
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    table = []
    for i in range(48):
        S, Sa = 300 + int(rng.integers(0, 40)), 940 + int(rng.integers(0, 60))
        np.save(os.path.join(td, f"v{i}_rgb.npy"), rng.random((S, 1024), dtype=np.float32))
        np.save(os.path.join(td, f"v{i}_flow.npy"), rng.random((S, 1024), dtype=np.float32))
        np.save(os.path.join(td, f"v{i}.npy"), rng.random((Sa, 128), dtype=np.float32))
        table.append(Clip(f"v{i}", "c", 0.0, 85.0, 100.0))           # 85 % of the stack: ~256-290 video rows, ~800-850 audio rows
    packer = FeaturePacker(td, td, pad_idx=1)
    batches = [[(16 * j + k) % 48 for k in range(16)] for j in range(30)]
    packer.pack([table[i] for i in batches[0]])
    t0 = time.perf_counter()
    for ix in batches[:10]:
        h = packer.pack([table[i] for i in ix])
    t_pack = (time.perf_counter() - t0) / 10
    mb = sum(v.numel() for v in h.values()) * 4 / 1e6
    print(f"host pack: {t_pack * 1e3:.2f} ms per batch of {mb:.1f} MB ({mb / t_pack / 1e3:.2f} GB/s), threads={packer.threads}")
    batcher = DeviceBatcher(packer, dev)
    x = torch.zeros(1, device=dev)
    for hold_ms in (0.0, 8.4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for batch in FeaturePrefetcher(batcher, table, batches):
            x += batch["feature_stacks"]["rgb"][0, 0, 0]              # consumer touches the batch on the current stream
            if hold_ms:
                torch.cuda._sleep(int(hold_ms * 2.1e6))               # ~hold_ms of GPU time (cycles at ~2.1 GHz)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(batches)
        print(f"pipeline with a consumer holding each batch {hold_ms:.1f} ms: {dt * 1e3:.2f} ms per batch")
