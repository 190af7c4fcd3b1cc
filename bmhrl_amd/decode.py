"""Greedy decoding with the reference's decoder signature (epoch_loops/captioning_bmrl_loops.py:41-43,127-152):
arg-max autoregressive decode that stops when every sample has produced </s> or at max_len."""
import torch

from .model.masking import make_masks


def greedy_decode(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality, return_first=False, memoise=True):
    """memoise=True (default) runs the encoder and the fusion layers' memory projections once per clip batch instead of
    once per generated token; the tokens and log-probs are the same as with the reference's full re-run (memoise=False).
    Models without encode_memory() (anything but the HIP BMHrlAgent) always take the full re-run."""
    with torch.no_grad():
        B = feature_stacks['audio'].shape[0]
        device = feature_stacks['audio'].device
        done = torch.zeros(B, 1, dtype=torch.bool, device=device)
        trg = torch.full((B, 1), start_idx, dtype=torch.long, device=device)
        first = None
        x = ((feature_stacks['rgb'], feature_stacks['flow']), feature_stacks['audio'])
        memoise = memoise and hasattr(model, "encode_memory") and not model.training
        memory, kv_cache = None, {}
        while trg.size(-1) <= max_len and not bool(done.all()):
            masks = make_masks(feature_stacks, trg, modality, pad_idx)
            if memoise:
                if memory is None:
                    memory = model.encode_memory(x, masks)
                preds = model.inference_from_memory(memory, trg, masks, kv_cache)
            else:
                preds = model.inference(x, trg, masks)
            if first is None:
                first = preds[:, -1].clone()
            nxt = preds[:, -1].argmax(dim=-1, keepdim=True)
            trg = torch.cat([trg, nxt], dim=-1)
            done = done | (nxt == end_idx)
    return (trg, first) if return_first else trg


def bimodal_decoder(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality):
    return greedy_decode(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality)


bmhrl_greedy_decoder = bimodal_decoder
