"""Diagnostic: the DETR caption decoder layer by layer against the oracle on the oracle's own memory / object states."""
import sys, os, math, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from bmhrl_amd import synthetic as syn
from oracle import bmhrl_oracle as O
from bmhrl_amd.model.det_bmhrl_agent import DetrCaption
from bmhrl_amd.functional import LayerNormFn
dev = torch.device("cuda:0")
g = np.load("tests/golden/detr_agent.npz")
cfg = syn.tiny_cfg(d_model=64, d_model_video=64, d_vid=64, d_model_caps=20, rl_att_heads=4, rl_goal_d=8, dout_p=0.0)
cfg.pre_goal_attention = False; cfg.device = "cuda:0"
agent = DetrCaption(cfg, SimpleNamespace(trg_voc_size=41, train_vocab=SimpleNamespace(vectors=None)))
keys = [str(k) for k in g["keys"]]
shapes = {k: tuple(int(d) for d in str(s).split(",") if d != "") for k, s in zip(keys, g["shapes"])}
sd = syn.fill_state_dict({k: s for k, s in shapes.items() if not k.startswith("critic.")}, seed=13)
sd.update({"critic." + k: v for k, v in syn.synthetic_critic_state(20, seed=1).items()})
for leaf in ("weight", "bias"): sd[f"worker_decoder.norm.{leaf}"] = sd[f"manager_decoder.norm.{leaf}"]   # one shared module
agent.load_state_dict(sd); agent.to(dev).eval()
gen = torch.Generator().manual_seed(3)
x = torch.from_numpy(g["x_video"]); B, L = x.shape[0], 7
trg = torch.randint(4, 41, (B, L), generator=gen); trg[0, 5] = 3; trg[0, 6:] = 1
c_mask = ((trg != 1).unsqueeze(1) & torch.ones(L, L, dtype=torch.bool).tril().unsqueeze(0))
V_mask = torch.from_numpy(g["V_mask"])
H, V = 4, 41
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
with torch.no_grad():
    t2 = trg.clone(); t2[t2 == 3] = 1
    C = sd["emb_C.embedder.weight"][t2] * math.sqrt(20)
    vf = x
    for i in range(3): vf = O.conv1d_same_groupnorm(sd, f"input_proj.{i}", vf)
    cls, hs, _ = O.object_detect(sd, "object_detector", vf, V_mask, V)
    memory = O.detr_stack(sd, "encoder", 3, vf, lambda q, t: O.detr_encoder_layer(sd, q, t, V_mask, H), True, False)
    print("emb", rel(agent.emb_C(t2.to(dev)), C))
    xo, xm = C, C.to(dev)
    for i, layer in enumerate(agent.worker_decoder.layers):
        p = f"worker_decoder.layers.{i}"
        # sub-steps of the oracle layer
        qk = O.add_posenc(xo)
        b1 = O.mha(sd, p + ".self_attn", qk, qk, xo, O.causal_mask(c_mask, True), H)
        qk_m = agent.pos_enc_C(xm)
        b1m = layer.self_attn(qk_m, qk_m, xm, c_mask.to(dev), causal=True)
        print(i, "posenc", rel(qk_m, qk), "self_attn branch", rel(b1m, b1))
        t1 = O.layer_norm(sd, p + ".norm1", xo) + b1
        b2 = O.mha(sd, p + ".multihead_attn", qk, O.add_posenc(memory), memory, V_mask, H)
        b2m = layer.multihead_attn(qk.to(dev), agent.pos_enc(memory.to(dev)), memory.to(dev), V_mask.to(dev))
        print(i, "memory attn branch (oracle inputs)", rel(b2m, b2))
        b5 = O.mha(sd, p + ".detected_attention", qk, hs, hs, None, H)
        b5m = layer.detected_attention(qk.to(dev), hs.to(dev), hs.to(dev), None)
        print(i, "object attn branch (oracle inputs)", rel(b5m, b5))
        xo_next = O.detr_decoder_layer(sd, p, xo, memory, V_mask, None, c_mask, None, None, False, hs, H)
        xm_same = layer(xo.to(dev), memory.to(dev), V_mask.to(dev), agent.pos_enc, agent.pos_enc_C, c_mask.to(dev), None, None, None,
                        detected_objects=hs.to(dev), obj_mask=None)
        xm = layer(xm, memory.to(dev), V_mask.to(dev), agent.pos_enc, agent.pos_enc_C, c_mask.to(dev), None, None, None,
                   detected_objects=hs.to(dev), obj_mask=None)
        print(i, "layer on oracle input", rel(xm_same, xo_next), " chained", rel(xm, xo_next))
        xo = xo_next
    fo = O.layer_norm(sd, "worker_decoder.norm", xo)
    fm = LayerNormFn.apply(xm, agent.worker_decoder.norm.weight, agent.worker_decoder.norm.bias)
    print("final norm", rel(fm, fo))
    full = agent.worker_decoder(C.to(dev), memory.to(dev), V_mask.to(dev), agent.pos_enc, agent.pos_enc_C, c_mask.to(dev), None, None, None,
                                detected_objects=hs.to(dev), obj_mask=None)
    print("stack call", rel(full, fo))
