"""Padding / causal masks -- same functions and semantics as the reference's model/masking.py:3-55.
Plain comparisons on the tensors' own device (they feed the attention kernels as byte masks)."""
import torch


def subsequent_mask(size, device=None):
    """(1, size, size) lower-triangular byte mask, built on `device` (no host->device copy: the training step is
    captured into a HIP graph).  reference model/masking.py:3-11"""
    return torch.tril(torch.ones(1, size, size, dtype=torch.uint8, device=device), 0)


_TRIL = {}


def _tril_bool(size, device):
    """the same triangle as a cached bool constant per (size, device): three launches less per step"""
    key = (size, str(device))
    t = _TRIL.get(key)
    if t is None:
        t = subsequent_mask(size, device).bool()
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):     # (a graph's private pool must not leak out)
            if len(_TRIL) > 64:
                _TRIL.clear()
            _TRIL[key] = t
    return t


def c_mask(trg, pad_idx):
    """key-padding & causal mask of a caption batch, (B, L, L).  reference :13-15"""
    pad = (trg != pad_idx).unsqueeze(-2)
    return pad & _tril_bool(trg.size(-1), trg.device)


def mask(src, trg, pad_idx, data_pad=0):
    """src (B, S) first feature column -> (B, 1, S); optional caption mask.  reference :18-25"""
    src_mask = (src != data_pad).unsqueeze(1)
    if trg is None:
        return src_mask
    return src_mask, c_mask(trg, pad_idx)


# which padding masks a modality string asks for: mask key -> (feature stack, takes the first feature column?)
_SOURCES = {
    'video': {'V_mask': ('rgb', True)},
    'audio': {'A_mask': ('audio', True)},
    'audio_video': {'V_mask': ('rgb', True), 'A_mask': ('audio', True)},
    'subs_audio_video': {'V_mask': ('rgb', True), 'A_mask': ('audio', True), 'S_mask': ('subs', False)},
}


def make_masks(feature_stacks, captions, modality, pad_idx):
    """reference :28-55 as a table: every stack of the modality gets its key-padding mask from its FIRST feature column
    (`rgb[:, :, 0]`, taken before the flow is added; `audio[:, :, 0]`; subtitles are token ids already), and the caption
    mask `C_mask` is added when captions are given."""
    try:
        wanted = _SOURCES[modality]
    except KeyError:
        raise ValueError(f'unknown modality {modality}') from None
    masks = {}
    for key, (stack, first_column) in wanted.items():
        feats = feature_stacks[stack]
        if first_column:
            if feats.dim() != 3:
                raise ValueError(f'{stack} features must be (B, T, D), got {tuple(feats.shape)}')
            feats = feats[:, :, 0]
        masks[key] = mask(feats, None, pad_idx)
    if captions is not None:
        masks['C_mask'] = c_mask(captions, pad_idx)
    return masks
