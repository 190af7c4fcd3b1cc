"""Counterparts of the names scripts/train_rl_captioning_module.py:14-16 imports from the reference's
epoch_loops/captioning_bmrl_loops.py, for the BMHRL mode, over the HIP-backed agent.

Kept from the reference: loop signatures (cfg, models, scorer, loader, epoch, log_prefix, TBoard[, train_worker]), the
models dict {"captioning": (model, optimizer, criterion), "worker": (...), "manager": (...)}, the batch contract of the
loader, the step arithmetic (:837-877, :1148-1160) and its quirks (loss factor 4/20, clip after the
step, amplitude attached to the prediction).  Both call arities of the driver (:191 passes 8 arguments to the 7-parameter
warmstart loop) are accepted.  Host-side per-element Python loops of the reference (generate_synonyms, the manager's
segment loop) are vectorised tensor ops; string rewards come from the `scorer` object when it is given, otherwise a
`reward_fn(sampled_tokens, batch) -> (B, L) tensor` must be supplied in cfg.rl_reward_fn (BASELINE config 3: synthetic).
"""
import random

import torch

from ..decode import greedy_decode
from ..model.masking import make_masks


# ---------------------------------------------------------------------------------------------- decoders (:41-60, :127-152)
def bimodal_decoder(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality):
    return greedy_decode(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality)


bmhrl_greedy_decoder = bimodal_decoder


def audio_decoder(*a, **k):
    raise NotImplementedError("unimodal (AHRL) mode is outside the hot path (SURVEY.md section 2, row 3)")


video_decoder = detr_decoder = audio_decoder


# ---------------------------------------------------------------------------------------------- feature getters (:472-530)
def generate_synonyms(caption_idx, voc_size, p=0.3, end_idx=3, pad_idx=1, generator=None):
    """Random input-token corruption of the reference (:510-528), vectorised: every position before the first </s> is,
    with probability p, replaced by pad (80 %), a random token id in [2, V) (10 %) or kept (10 %); the first </s>
    becomes pad.  (The reference draws from Python's global `random`; the distribution is the same, the stream is not.)"""
    cap = caption_idx.clone()
    B, L = cap.shape
    dev = cap.device
    is_end = cap == end_idx
    first_end = torch.where(is_end.any(1), is_end.float().argmax(1), torch.full((B,), L, device=dev))
    pos = torch.arange(L, device=dev).unsqueeze(0)
    before = pos < first_end.unsqueeze(1)
    u = torch.rand(B, L, device=dev, generator=generator)
    v = torch.rand(B, L, device=dev, generator=generator)
    rnd_tok = torch.randint(2, voc_size, (B, L), device=dev, generator=generator)
    hit = before & (u < p)
    cap = torch.where(hit & (v < 0.8), torch.full_like(cap, pad_idx), cap)
    cap = torch.where(hit & (v >= 0.9), rnd_tok, cap)
    cap = torch.where(pos == first_end.unsqueeze(1), torch.full_like(cap, pad_idx), cap)
    return cap


def feature_getter(cfg, batch, loader, random_synonyms=0.3):
    """-> (caption_idx [corrupted input], caption_idx_y [target], ((rgb, flow), audio), masks); masks are built from the
    corrupted captions, V_mask from rgb (before flow is added), as in the reference (:487-509)."""
    src = batch['feature_stacks']
    caption = batch['caption_data'].caption
    caption_idx, caption_idx_y = caption[:, :-1], caption[:, 1:]
    voc = len(loader.dataset.train_vocab.itos)
    caption_idx = generate_synonyms(caption_idx, voc, random_synonyms)
    masks = make_masks(src, caption_idx, cfg.modality, loader.dataset.pad_idx)
    return caption_idx.contiguous(), caption_idx_y.contiguous(), ((src['rgb'], src['flow']), src['audio']), masks


def inference_feature_getter(trg, feature_stacks, modality, pad_idx):
    masks = make_masks(feature_stacks, trg, modality, pad_idx)
    return ((feature_stacks['rgb'], feature_stacks['flow']), feature_stacks['audio']), masks


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


# ---------------------------------------------------------------------------------------------- RL loss glue (:271-334)
def get_amplitude(score, sampled_probs, norm_reward_factor):
    return torch.clamp(score.float() * sampled_probs.float() * norm_reward_factor.float(), 0, 1)


def get_norm_reward_factor(train_worker, mask, segments):
    return (mask if train_worker else segments).sum(dim=-1).reshape(-1, 1)


def sample_actions(prediction, greedy, seed, seed_dev=None, row_offset=0):
    """a ~ Categorical(exp(prediction)) (worker) or arg-max (manager), and p(a) -- one kernel, no host sync (:283-286).
    seed_dev: optional device word added to the seed (a captured step draws fresh samples at every replay); row_offset: first
    row of this rank's share of the global batch (the uniform of a row is a function of (seed, global row))."""
    from .. import ops
    B, L, V = prediction.shape
    out = torch.empty(B, L, dtype=torch.int64, device=prediction.device)
    p = torch.empty(B, L, device=prediction.device)
    ops.sample_tokens(prediction.detach().contiguous(), V, out, p, B * L, V, greedy, seed, seed_dev, row_offset)
    return out, p


def biased_kl(train_worker, prediction, scorer, expected_scores, trg, trg_caption, mask, segments, device, biased_kldiv,
              stabilize, reward_fn=None, seed=None, seed_dev=None, row_offset=0):
    """The reference's biased_kl (:271-334), both branches.  Worker: a ~ Categorical(exp(prediction)), amplitude =
    clamp(score * p(a) * n_tokens_row, 0, 1).  Manager (:299-317): a = arg-max, score *= segments, amplitude =
    clamp(score * prod_{segment} p(a) * n_segments_row, 0, 1) and the expected scores are summed per segment, with the
    row-transition quirks of the reference's Python loop (rl_glue.manager_segments).  Returns (row sums of the divergence
    (B*L, 1), [score], [sampled], [amplitude]).  Rewards: `reward_fn(sampled, captions) -> (B, L)` when given (BASELINE
    config 3: synthetic), else the scorer's delta_*_worker / delta_*_manager."""
    from .. import rl_glue
    seed = random.getrandbits(62) if seed is None else seed
    sampled, _ = sample_actions(prediction, greedy=not train_worker, seed=seed, seed_dev=seed_dev, row_offset=row_offset)
    if reward_fn is not None:
        score = reward_fn(sampled, trg_caption)
    elif train_worker:
        score = scorer.delta_cider_worker(sampled, trg_caption)[0] if getattr(scorer, "type", "") == "CIDER" else \
            (scorer.delta_bleu_worker(sampled, trg_caption)[0] if getattr(scorer, "type", "") == "BLEU" else
             scorer.delta_meteor_worker(sampled, trg_caption, mask)[0])
    else:
        fn = {"CIDER": "delta_cider_manager", "BLEU": "delta_bleu_manager"}.get(getattr(scorer, "type", ""), "delta_meteor_manager")
        score = getattr(scorer, fn)(sampled, trg_caption, mask, segments)[0]
    score = score.to(device).float()
    n_row = get_norm_reward_factor(train_worker, mask, segments).float()
    if train_worker:
        if stabilize:
            score = (score - expected_scores) * mask.float()
        rows, amp = biased_kldiv.biased_kl_from_score(prediction, trg, sampled, score, n_row.expand_as(trg))
        return rows, [score], [sampled], [amp]
    score = score * segments.float()
    if stabilize:
        with torch.no_grad():
            p = torch.gather(prediction, 2, sampled.unsqueeze(-1)).squeeze(-1).exp()
            _, expected = rl_glue.manager_segments(p, expected_scores.float(), segments)
        score = (score - expected) * mask.float()
    rows, amp = biased_kldiv.biased_kl_from_segments(prediction, trg, sampled, score, n_row, segments)
    return rows, [score], [sampled], [amp]


# ---------------------------------------------------------------------------------------------- steps
def warmstart_bmhrl_bl(cfg, models, scorer, loader, epoch, log_prefix, TBoard, *extra):
    """Captioning half of the reference's warmstart step (:1132-1189): sum(LabelSmoothing) / n_tokens, Adam.  The value
    network regression that follows in the reference unpacks a 2-tuple into 3 names and is skipped."""
    cap_model, cap_optimizer, cap_criterion = models["captioning"]
    cap_model.train()
    loader.dataset.update_iterator()
    agent = _unwrap(cap_model)
    agent.teach_warmstart()          # (exploration stays as it is -- on after the constructor --, reference :572-575)
    total, n = 0.0, 0
    for batch in loader:
        cap_optimizer.zero_grad()
        caption_idx, caption_idx_y, x, masks = feature_getter(cfg, batch, loader)
        prediction = cap_model(x, caption_idx, masks)[0]
        n_tokens = (caption_idx_y != loader.dataset.pad_idx).sum()
        loss = torch.sum(cap_criterion(prediction, caption_idx_y)) / n_tokens
        loss.backward()
        cap_optimizer.step()
        total += float(loss)
        n += 1
    if TBoard is not None and n:
        TBoard.add_scalar('debug/train_loss_epoch', total / n, epoch)
    return total / max(n, 1)


def train_bmhrl_bl(cfg, models, scorer, loader, epoch, log_prefix, TBoard, train_worker=True, *extra):
    """Worker RL step of the reference (:797-890): forward, value head on detached features, biased KL with the sampled
    tokens' reward, loss / (n_tokens * 4/20), Adam (clip AFTER the step, as the reference), masked-MSE value update."""
    if not train_worker:
        raise NotImplementedError("the reference's manager training branch prints and raises (:852-854)")
    cap_model, cap_optimizer, cap_criterion = models["captioning"]
    wv_model, wv_optimizer, wv_criterion = models["worker"]
    cap_model.train()
    wv_model.train()
    loader.dataset.update_iterator()
    agent = _unwrap(cap_model)
    agent.teach_worker()
    pad_idx = loader.dataset.pad_idx
    loss_factor = 4.0 / 20.0
    reward_fn = getattr(cfg, "rl_reward_fn", None)
    total, n = 0.0, 0
    for batch in loader:
        cap_optimizer.zero_grad()
        wv_optimizer.zero_grad()
        caption_idx, caption_idx_y, x, masks = feature_getter(cfg, batch, loader)
        prediction, worker_feat, manager_feat, goal_feat, segment_labels = cap_model(x, caption_idx, masks)
        loss_mask = caption_idx_y != pad_idx
        n_tokens = loss_mask.sum()
        expected_value = wv_model((worker_feat.detach(), goal_feat.detach())).squeeze(-1)
        losses, scores, samples, amplitude = biased_kl(True, prediction, scorer, expected_value.detach(), caption_idx_y,
                                                       batch['captions'], loss_mask, segment_labels, prediction.device,
                                                       cap_criterion, getattr(cfg, "rl_stabilize", False), reward_fn)
        cap_loss = torch.sum(losses) / (n_tokens * loss_factor)
        cap_loss.backward()
        cap_optimizer.step()
        if getattr(cfg, "grad_clip", None) is not None:
            torch.nn.utils.clip_grad_norm_(cap_model.parameters(), cfg.grad_clip)
        value_loss = (wv_criterion(expected_value, scores[0].float()) * loss_mask.float()).mean()
        value_loss.backward()
        wv_optimizer.step()
        total += float(cap_loss)
        n += 1
    if TBoard is not None and n:
        TBoard.add_scalar('debug/train_loss_epoch', total / n, epoch)
    return total / max(n, 1)


def bmhrl_validation_next_word_loop(cfg, model, loader, decoder, criterion, epoch, TBoard, exp_name):
    """reference :189-216"""
    model.eval()
    loader.dataset.update_iterator()
    total, n = 0.0, 0
    for batch in loader:
        src = batch['feature_stacks']
        caption = batch['caption_data'].caption
        caption_idx, caption_idx_y = caption[:, :-1].contiguous(), caption[:, 1:].contiguous()
        masks = make_masks(src, caption_idx, cfg.modality, loader.dataset.pad_idx)
        with torch.no_grad():
            prediction = model(((src['rgb'], src['flow']), src['audio']), caption_idx, masks)[0]
            n_tokens = (caption_idx_y != loader.dataset.pad_idx).sum()
            total += float(torch.sum(criterion(prediction, caption_idx_y)) / n_tokens)
        n += 1
    return total / max(n, 1)


def bmhrl_inference(model, feature_stacks, max_len, start_idx, end_idx, pad_idx, modality, captions, batch):
    """reference :26-28 calls an undefined `inference` (NameError there); the name exists for the importers
    (sample/single_vid_bmhrl.py:11, scripts/test_model.py:6)"""
    raise NotImplementedError("bmhrl_inference is broken in the reference (undefined `inference`, :26-28)")


def bmhrl_test(cfg, models, loader):
    """reference :163-186: prints the ground truth and the greedy decode of every batch to stderr (its middle step calls an
    undefined `bimodal_inference` and is left out)."""
    import sys
    cap_model = models["captioning"]
    ds = loader.dataset
    sent = lambda idx: " ".join(ds.train_vocab.itos[int(i)] for i in idx)
    for batch in loader:
        src = batch['feature_stacks']
        caption_idx_y = batch['caption_data'].caption[:, 1:]
        print(f'Groundtruth: {sent(caption_idx_y[0])}', file=sys.stderr)
        synthesis = bmhrl_greedy_decoder(_unwrap(cap_model), src, cfg.max_len, ds.start_idx, ds.end_idx, ds.pad_idx, cfg.modality)
        print(f'Greedy Decoder: {sent(synthesis[0])}', file=sys.stderr)


def _not_hot_path(*a, **k):
    raise NotImplementedError("unimodal / DETR / analysis loops are outside the hot path (SURVEY.md section 2)")


train_audio_bl = train_video_bl = warmstart_audio_bl = warmstart_video_bl = _not_hot_path
analyze_bmhrl_div = train_detr_rl = reinforce_detr_rl = _not_hot_path
