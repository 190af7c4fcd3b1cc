"""Counterpart of the reference's epoch_loops/validation_loops.py for the HIP-backed agent: the one-by-one (greedy decode)
validation pass and the next-word loss pass, with the reference's signatures and result format.

validation_1by1_loop (:13-137) is where the decoder spends its time: every batch is decoded token by token.  Here the
decoder is bmhrl_amd.decode (per-clip memory K|V, incremental caption K|V cache, one HIP graph per token); the host side --
ids -> words, cut at </s>, capitalise, the ActivityNet-captions result dictionary and its JSON file -- keeps the reference's
format so that its evaluator reads the file unchanged.  The evaluator itself (evaluation/evaluate.py: Java METEOR, PTB
tokenizer) is outside the hot path: it is imported from the reference when that is on sys.path, otherwise the loop returns
after writing the predictions."""
import contextlib
import io
import json
import os
from time import time

import torch

from ..model.masking import make_masks


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


def _phase_references(cfg, phase):
    """reference-caption files and tIoU thresholds of a validation phase (:34-50)"""
    single = {"val_1": 0, "val_2": 1, "vatex_val": 2, "msrvtt_val": 3}
    if phase in single:
        return [cfg.reference_paths[single[phase]]], [0.5]
    if phase == "learned_props":
        assert len(cfg.tIoUs) == 4
        return cfg.reference_paths, cfg.tIoUs
    raise ValueError(f"unknown validation phase {phase!r}")


def tokens_to_sentences(ints_stack, itos, end_token="</s>"):
    """(B, T) ids -> sentences the way the reference filters them (:63-86): drop <s>, cut at the first </s>, join, capitalise"""
    out = []
    for ints in ints_stack:
        words = [itos[int(i)] for i in ints][1:]
        if end_token in words:
            words = words[:words.index(end_token)]
        out.append(" ".join(words).capitalize())
    return out


def predict_1by1(cfg, model, loader, decoder):
    """decode every batch of the loader -> the ActivityNet-captions submission dictionary (:16-24, :53-102)"""
    predictions = {"version": "VERSION 1.0", "external_data": {"used": True, "details": ""}, "results": {}}
    ds = loader.dataset
    agent = _unwrap(model)
    for batch in loader:
        ints = decoder(agent, batch["feature_stacks"], cfg.max_len, ds.start_idx, ds.end_idx, ds.pad_idx, cfg.modality)
        sentences = tokens_to_sentences(ints.cpu().numpy(), ds.train_vocab.itos)
        for video_id, start, end, sent in zip(batch["video_ids"], batch["starts"], batch["ends"], sentences):
            seg = {"sentence": sent, "timestamp": [start.item(), end.item()]}
            predictions["results"].setdefault(video_id, []).append(seg)
    return predictions


def calculate_metrics(reference_paths, submission_path, tIoUs, max_prop_per_vid, verbose=True, only_proposals=False):
    """per-tIoU and averaged scores of the reference's ANETcaptions evaluator (:160-183); needs the reference's
    evaluation package (and its Java METEOR) on sys.path -- not part of this package"""
    try:
        from evaluation.evaluate import ANETcaptions
    except Exception as e:  # noqa: BLE001
        raise NotImplementedError("the caption evaluator (evaluation/evaluate.py of the reference: PTB tokenizer, METEOR jar) is "
                                  "outside the hot path this package replaces; put the reference on sys.path to score") from e
    ev = ANETcaptions(reference_paths, submission_path, tIoUs, max_prop_per_vid, ["results", "version", "external_data"],
                      verbose, only_proposals)
    ev.evaluate()
    metrics = {tiou: {name: scores[i] for name, scores in ev.scores.items()} for i, tiou in enumerate(tIoUs)}
    metrics["Average across tIoUs"] = {name: sum(scores) / float(len(scores)) for name, scores in ev.scores.items()}
    return metrics


def validation_1by1_loop(cfg, model, loader, decoder, epoch, TBoard):
    """Greedy-decode the whole loader, write captioning_results_<phase>_e<epoch>.json under cfg.log_path and score it.
    Returns the metrics dictionary; None when cfg.log_path is None (as the reference); the predictions dictionary when the
    evaluator is not importable (see calculate_metrics)."""
    t_start = time()
    model.eval()
    loader.dataset.update_iterator()
    phase = loader.dataset.phase
    reference_paths, tIoUs = _phase_references(cfg, phase)
    predictions = predict_1by1(cfg, model, loader, decoder)
    if cfg.log_path is None:
        return None
    os.makedirs(cfg.log_path, exist_ok=True)
    path = os.path.join(cfg.log_path, f"captioning_results_{phase}_e{epoch}.json")
    if os.path.exists(path):          # another loader / pretrained model in the same run: keep both files
        path = path.replace(".json", f"_{time()}.json")
    with open(path, "w") as f:
        json.dump(predictions, f)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            val_metrics = calculate_metrics(reference_paths, path, tIoUs, cfg.max_prop_per_vid)
    except NotImplementedError:
        predictions["submission_path"] = path
        return predictions
    if TBoard is not None and phase != "learned_props":
        avg = val_metrics["Average across tIoUs"]
        for tag, key in (("meteor", "METEOR"), ("bleu4", "Bleu_4"), ("bleu3", "Bleu_3"), ("precision", "Precision"),
                         ("recall", "Recall")):
            TBoard.add_scalar(f"{phase}/{tag}", avg[key] * 100, epoch)
        TBoard.add_scalar(f"{phase}/duration_of_1by1", (time() - t_start) / 60, epoch)
    return val_metrics


def validation_next_word_loop(cfg, model, loader, decoder, criterion, epoch, TBoard, exp_name):
    """teacher-forced validation loss per batch, averaged over the loader (:139-158); `criterion` reduces to a scalar here
    (the captioning-module loops of the reference pass a summing criterion)"""
    model.eval()
    loader.dataset.update_iterator()
    pad_idx = loader.dataset.pad_idx
    total = 0.0
    for batch in loader:
        cap = batch["caption_data"].caption
        cap_in, cap_y = cap[:, :-1], cap[:, 1:]
        masks = make_masks(batch["feature_stacks"], cap_in, cfg.modality, pad_idx)
        with torch.no_grad():
            pred = model(batch["feature_stacks"], cap_in, masks)
            n_tokens = (cap_y != pad_idx).sum()
            total += (criterion(pred, cap_y) / n_tokens).item()
    return total / len(loader)
