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


def make_masks(feature_stacks, captions, modality, pad_idx):
    """reference :28-55.  V_mask comes from rgb[:, :, 0] (before flow is added), A_mask from audio[:, :, 0]."""
    masks = {}
    if modality == 'video':
        if captions is None:
            masks['V_mask'] = mask(feature_stacks['rgb'][:, :, 0], None, pad_idx)
        else:
            masks['V_mask'], masks['C_mask'] = mask(feature_stacks['rgb'][:, :, 0], captions, pad_idx)
    elif modality == 'audio':
        assert len(feature_stacks['audio'].shape) == 3
        if captions is None:
            masks['A_mask'] = mask(feature_stacks['audio'][:, :, 0], None, pad_idx)
        else:
            masks['A_mask'], masks['C_mask'] = mask(feature_stacks['audio'][:, :, 0], captions, pad_idx)
    elif modality in ('audio_video', 'subs_audio_video'):
        assert len(feature_stacks['audio'].shape) == 3
        if captions is None:
            masks['V_mask'] = mask(feature_stacks['rgb'][:, :, 0], None, pad_idx)
        else:
            masks['V_mask'], masks['C_mask'] = mask(feature_stacks['rgb'][:, :, 0], captions, pad_idx)
        masks['A_mask'] = mask(feature_stacks['audio'][:, :, 0], None, pad_idx)
        if modality == 'subs_audio_video':
            masks['S_mask'] = mask(feature_stacks['subs'], None, pad_idx)
    else:
        raise ValueError(f'unknown modality {modality}')
    return masks
