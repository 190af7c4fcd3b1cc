"""One tiny forward + backward of the hot path on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke()).  The oracle is the checker only; this file lives under tests/ because nothing under bmhrl_amd/
may import the oracle."""
import os
import sys
from types import SimpleNamespace

import torch


def run_smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # the repository root
    if root not in sys.path:
        sys.path.insert(0, root)
    from bmhrl_amd import _lib, synthetic as syn
    from bmhrl_amd.loss.label_smoothing import LabelSmoothing
    from bmhrl_amd.model.bm_hrl_agent import BMHrlAgent
    from bmhrl_amd.model.masking import make_masks
    from oracle import bmhrl_oracle as O

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    _lib.load()
    dev = torch.device("cuda:0")
    cfg = syn.default_cfg(dout_p=0.0)                  # the reference widths (1024 / 128 / 300, d_model 1024, H 4, N 2)
    cfg.device = "cuda:0"
    V, B, Tv, Ta, L = 300, 2, 160, 200, 10             # Tv >= 128: the fused attention kernels are on the path
    ds = SimpleNamespace(trg_voc_size=V, train_vocab=SimpleNamespace(vectors=None))
    agent = BMHrlAgent(cfg, ds)
    shapes = {k: tuple(v.shape) for k, v in agent.state_dict().items()}
    sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=0)
    sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(cfg.d_model_caps, seed=1).items()})
    agent.load_state_dict(sd)
    agent.to(dev).eval()
    agent.set_inference_mode(True)
    b = syn.synthetic_batch(B, Tv, Ta, L, V, seed=3, d_vid=cfg.d_vid, d_aud=cfg.d_aud, min_len=3)
    cap = b["captions"]
    trg_in, trg_y = cap[:, :-1].contiguous(), cap[:, 1:].contiguous()
    fs = {k: b[k].to(dev) for k in ("rgb", "flow", "audio")}
    masks = make_masks(fs, trg_in.to(dev), "audio_video", 1)
    pred = agent(((fs["rgb"], fs["flow"]), fs["audio"]), trg_in.to(dev), masks)[0]
    loss = torch.sum(LabelSmoothing(0.7, 1)(pred, trg_y.to(dev))) / (trg_y != 1).sum().to(dev)
    loss.backward()
    torch.cuda.synchronize()
    ref = O.agent_forward(sd, cfg, (b["rgb"] + b["flow"], b["audio"]), trg_in, O.make_masks(b["rgb"], b["audio"], trg_in, 1))[0]
    ref_loss = O.warmstart_loss(ref, trg_y, 0.7, 1)
    diff = (pred.detach().cpu() - ref).abs()
    err = float(diff.max() / ref.abs().max())                               # max-norm metric
    err_elem = float((diff / ref.abs().clamp_min(1.0)).max())               # per element, relative, |log-prob| floor 1.0
    lerr = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    g = agent.bm_enc.encoder.layers[0].self_att_M1.linear_Q2d.weight.grad
    assert g is not None and bool(torch.isfinite(g).all())
    assert err < 1e-3 and lerr < 1e-3 and err_elem < 1e-3, (err, err_elem, lerr)
    print(f"smoke ok: log-prob err {err:.2e} (max-norm) / {err_elem:.2e} (per element, relative, floor 1.0), "
          f"loss rel err {lerr:.2e}, loss {float(loss):.5f}")


if __name__ == "__main__":
    run_smoke()
