"""Feature loader for the training / validation loops: the batch dict of
captioning_datasets/captioning_dataset.py:246-316 (`__getitem__` of the pre-batched dataset) without its per-sample
torch round trips on the training thread.

What the reference does per step: three `np.load` per clip, `torch.from_numpy(...).float()`, a crop
(captioning_datasets/load_features.py:14-35), `pad_sequence` of the three lists and three synchronous `.to(device)`.
Here:

  * the .npy files are memory-mapped and only the cropped rows are read, straight into PINNED staging buffers laid out as
    the batch tensors (B, T_max, 1024) / (B, T_max, 128) -- padding values written in the same pass (rgb, audio: pad_idx;
    flow: 0; a missing clip is one zero row, :268-278);
  * the three stacks go to the device as three asynchronous copies on a copy stream, double buffered, so that batch
    i + 1 is read and uploaded while batch i trains; the consumer stream waits on an event, not on the host;
  * `FeaturePrefetcher` runs that one batch ahead on a worker thread (np.load / memcpy release the GIL).

Same values as the reference's batch (tests/test_loader_cpu.py pins crop_bounds / pack against fixtures generated from the
reference's own functions), same dict keys, same dtypes.  The upload needs a GPU; packing is plain host code.
"""
from __future__ import annotations

import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

D_VIDEO, D_AUDIO = 1024, 128


def crop_bounds(S: int, start: float, end: float, duration: float) -> Optional[Tuple[int, int]]:
    """row range of captioning_datasets/load_features.py:14-35 (`crop_a_segment`) for a stack of S rows, or None when
    the crop is empty.  The quantiles are formed first (start / duration, then * S), as there: the rounding of the
    float product decides the index."""
    a = int(S * (start / duration))
    b = int(S * (end / duration))
    if a == b:
        if a == S:
            a -= 1
        else:
            b += 1
    lo, hi, _ = slice(a, b).indices(S)                 # what feature[a:b] keeps
    return (lo, hi) if hi > lo else None


# ---- the reference's per-clip functions under their own names and signatures (captioning_datasets/load_features.py), for
# callers that still assemble batches themselves (the reference's dataset class after `import bmhrl_amd.install`)
def fill_missing_features(method, feature_size):
    """load_features.py:8-12"""
    if method == 'random':
        return torch.rand(1, feature_size)
    if method == 'zero':
        return torch.zeros(1, feature_size).float()


def crop_a_segment(feature, start, end, duration):
    """load_features.py:14-35 on a (S, D) tensor or array: the kept rows (a view), or None"""
    r = crop_bounds(feature.shape[0], start, end, duration)
    return None if r is None else feature[r[0]:r[1], :]


def pad_segment(feature, max_feature_len, pad_idx):
    """load_features.py:38-44: rows appended up to max_feature_len with pad_idx"""
    S, D = feature.shape
    assert S <= max_feature_len
    out = feature.new_full((max_feature_len, D), pad_idx)
    out[:S] = feature
    return out


def load_features_from_npy(cfg, feature_names_list, video_id, start, end, duration, pad_idx, get_full_feat=False,
                           is_vatex=False):
    """load_features.py:47-99, same returns ({'audio','rgb','flow'[,'orig_feat_length']}, fp32 CPU tensors or None for a
    missing file); the crop reads only its rows from the memory-mapped file."""
    supported = {'i3d_features', 'vggish_features'}
    assert isinstance(feature_names_list, list) and len(feature_names_list) > 0 and set(feature_names_list).issubset(supported)
    stacks = {}
    if get_full_feat:
        stacks['orig_feat_length'] = {}

    def read(path, full_key, full_len, full_pad):
        m = _open(path)
        if m is None:
            return None
        if get_full_feat:
            stacks['orig_feat_length'][full_key] = m.shape[0]
            return pad_segment(torch.from_numpy(np.array(m, dtype=np.float32)), full_len, full_pad)
        r = crop_bounds(m.shape[0], start, end, duration)
        return None if r is None else torch.from_numpy(np.array(m[r[0]:r[1]], dtype=np.float32))

    if 'vggish_features' in feature_names_list:
        apath = './data/vggish_vatex/' if is_vatex else cfg.audio_features_path
        stacks['audio'] = read(os.path.join(apath, f'{video_id}.npy'), 'audio',
                               cfg.pad_feats_up_to['audio'] if get_full_feat else 0, pad_idx)
    if 'i3d_features' in feature_names_list:
        vpath = './data/i3d_vatex/' if is_vatex else cfg.video_features_path
        rgb_m, flow_m = _open(os.path.join(vpath, f'{video_id}_rgb.npy')), _open(os.path.join(vpath, f'{video_id}_flow.npy'))
        if rgb_m is None or flow_m is None:            # one try block in the reference: either file missing drops both
            stacks['rgb'] = stacks['flow'] = None
        else:
            assert rgb_m.shape == flow_m.shape
            stacks['rgb'] = read(os.path.join(vpath, f'{video_id}_rgb.npy'), 'rgb',
                                 cfg.pad_feats_up_to['video'] if get_full_feat else 0, pad_idx)
            stacks['flow'] = read(os.path.join(vpath, f'{video_id}_flow.npy'), 'flow',
                                  cfg.pad_feats_up_to['video'] if get_full_feat else 0, 0)
    return stacks


class Clip:
    """one row of the meta table: what `__getitem__` reads per index (captioning_dataset.py:251-253)"""
    __slots__ = ("video_id", "caption", "start", "end", "duration")

    def __init__(self, video_id: str, caption: str, start: float, end: float, duration: float):
        self.video_id, self.caption, self.start, self.end, self.duration = video_id, caption, start, end, duration


def _open(path: str):
    try:
        return np.load(path, mmap_mode="r")
    except FileNotFoundError:
        return None


class FeaturePacker:
    """Builds the feature stacks of one batch in host memory (pinned when a GPU is present)."""

    def __init__(self, video_features_path: str, audio_features_path: str, pad_idx: int, modality: str = "audio_video",
                 pin: Optional[bool] = None, threads: int = 8):
        self.vpath, self.apath, self.pad_idx = video_features_path, audio_features_path, float(pad_idx)
        self.threads = max(1, threads)                   # clips of a batch are copied in parallel (numpy releases the GIL)
        self._pool = None
        self.want_video = "video" in modality
        self.want_audio = "audio" in modality
        self.pin = torch.cuda.is_available() if pin is None else pin
        self._bufs: Dict[Tuple[str, int], torch.Tensor] = {}

    # -- one clip: views of the rows the reference would keep (no copy yet)
    def rows(self, c: Clip):
        rgb = flow = aud = None
        if self.want_audio:
            a = _open(os.path.join(self.apath, f"{c.video_id}.npy"))
            if a is not None:
                r = crop_bounds(a.shape[0], c.start, c.end, c.duration)
                aud = a[r[0]:r[1]] if r else None
        if self.want_video:
            r_ = _open(os.path.join(self.vpath, f"{c.video_id}_rgb.npy"))
            f_ = _open(os.path.join(self.vpath, f"{c.video_id}_flow.npy"))
            if r_ is not None and f_ is not None:        # the reference's try block drops both when either is missing
                if r_.shape != f_.shape:
                    raise AssertionError(f"rgb / flow shapes differ for {c.video_id}: {r_.shape} vs {f_.shape}")
                r = crop_bounds(r_.shape[0], c.start, c.end, c.duration)
                if r:
                    rgb, flow = r_[r[0]:r[1]], f_[r[0]:r[1]]
        return rgb, flow, aud

    def _staging(self, name: str, slot: int, shape) -> torch.Tensor:
        key = (name, slot)
        buf = self._bufs.get(key)
        n = int(np.prod(shape))
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=torch.float32, pin_memory=self.pin)
            self._bufs[key] = buf
        return buf[:n].view(*shape)

    def pack(self, clips: Sequence[Clip], slot: int = 0) -> Dict[str, torch.Tensor]:
        """host tensors {'rgb','flow','audio'} of one batch; `slot` selects the staging set (double buffering)."""
        per = [self.rows(c) for c in clips]
        B = len(clips)
        tv = max([1] + [r.shape[0] for r, _, _ in per if r is not None])
        ta = max([1] + [a.shape[0] for _, _, a in per if a is not None])
        out = {"rgb": self._staging("rgb", slot, (B, tv, D_VIDEO)), "flow": self._staging("flow", slot, (B, tv, D_VIDEO)),
               "audio": self._staging("audio", slot, (B, ta, D_AUDIO))}
        dsts = {k: v.numpy() for k, v in out.items()}

        def fill(i):
            for name, idx, pad in (("rgb", 0, self.pad_idx), ("flow", 1, 0.0), ("audio", 2, self.pad_idx)):
                dst, src = dsts[name], per[i][idx]
                n = 0 if src is None else src.shape[0]
                if n:
                    dst[i, :n] = src                      # the only read of the file: cropped rows -> staging (casts to fp32)
                    dst[i, n:] = pad
                else:
                    dst[i, :1] = 0.0                      # missing clip: one zero row, then padding
                    dst[i, 1:] = pad

        if self.threads > 1 and B > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.threads)
            list(self._pool.map(fill, range(B)))
        else:
            for i in range(B):
                fill(i)
        return out


class DeviceBatcher:
    """Uploads packed batches with asynchronous copies on a copy stream, two slots deep."""

    def __init__(self, packer: FeaturePacker, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("DeviceBatcher uploads to a GPU; on the CPU use FeaturePacker.pack directly")
        self.packer, self.device = packer, device
        self.copy_stream = torch.cuda.Stream(device=device)
        self._dev: Dict[Tuple[str, int], torch.Tensor] = {}
        self._ready = [torch.cuda.Event(), torch.cuda.Event()]
        self._consumed = [None, None]      # event recorded by the consumer once it is done with a slot
        self._slot = 0

    def _device_buf(self, name: str, slot: int, shape) -> torch.Tensor:
        key, n = (name, slot), int(np.prod(shape))
        buf = self._dev.get(key)
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
            self._dev[key] = buf
        return buf[:n].view(*shape)

    def upload(self, clips: Sequence[Clip]) -> Dict:
        """pack + enqueue the copies; returns the batch dict of captioning_dataset.py:296-307 whose tensors are valid for
        a consumer that called wait(batch) on its stream."""
        slot = self._slot
        self._slot ^= 1
        if self._consumed[slot] is not None:
            self._consumed[slot].synchronize()           # the step that used this slot two batches ago has finished
        host = self.packer.pack(clips, slot)
        feats = {}
        with torch.cuda.stream(self.copy_stream):
            for k, h in host.items():
                d = self._device_buf(k, slot, tuple(h.shape))
                d.copy_(h, non_blocking=True)
                feats[k] = d
            self._ready[slot].record(self.copy_stream)
        return {"video_ids": [c.video_id for c in clips], "captions": [c.caption for c in clips],
                "starts": torch.tensor([c.start for c in clips], device=self.device).unsqueeze(1),
                "ends": torch.tensor([c.end for c in clips], device=self.device).unsqueeze(1),
                "feature_stacks": feats, "_slot": slot}

    def wait(self, batch: Dict, stream: Optional[torch.cuda.Stream] = None):
        """make `stream` (default: the current one) wait for the batch's copies (device-side wait, no host sync)"""
        (stream or torch.cuda.current_stream(self.device)).wait_event(self._ready[batch["_slot"]])

    def release(self, batch: Dict, stream: Optional[torch.cuda.Stream] = None):
        """the consumer is done reading the batch on `stream`: its slot may be overwritten once that work completes"""
        ev = torch.cuda.Event()
        ev.record(stream or torch.cuda.current_stream(self.device))
        self._consumed[batch["_slot"]] = ev


class FeaturePrefetcher:
    """iterates over index batches; batch i + 1 is packed and uploaded on a worker thread while batch i is consumed"""

    def __init__(self, batcher: DeviceBatcher, table: Sequence[Clip], index_batches: Sequence[Sequence[int]]):
        self.batcher, self.table, self.index_batches = batcher, table, list(index_batches)

    def __iter__(self):
        nxt: List = [None]

        def work(ix):
            nxt[0] = self.batcher.upload([self.table[i] for i in ix])

        th = None
        for n, ix in enumerate(self.index_batches):
            if th is None:
                work(ix)
            else:
                th.join()
            batch = nxt[0]
            th = None
            if n + 1 < len(self.index_batches):
                th = threading.Thread(target=work, args=(self.index_batches[n + 1],), daemon=True)
                th.start()
            self.batcher.wait(batch)
            yield batch
            self.batcher.release(batch)
        if th is not None:
            th.join()
