"""CPU tests of the host-side logic around the HIP path (no kernels are launched)."""
import numpy as np
import torch

from bmhrl_amd import synthetic as syn


def test_state_dict_layout_matches_reference(golden):
    from bmhrl_amd.model.bm_hrl_agent import agent_state_shapes
    g = golden("agent_tiny")
    ref = {k: eval(s) for k, s in zip(g["state_keys"].tolist(), g["state_shapes"].tolist())}
    mine = agent_state_shapes(syn.tiny_cfg(), 50)
    assert mine == ref and len(mine) == 307
    full = agent_state_shapes(syn.default_cfg(), 10172)
    assert full["bm_enc.encoder.layers.0.bi_modal_att_M1.linear_K2d.weight"] == (1024, 128)
    assert full["bm_enc.encoder.layers.1.bi_modal_att_M2.linear_d2Q.weight"] == (128, 1024)
    assert full["worker.core.projection.weight"] == (10172, 364)
    assert full["worker.goal_attention.linear_Q2d.weight"] == (1024, 64)
    # 70 322 365 parameters (+19 264 for the LinearCore that is registered twice)
    assert sum(int(np.prod(s)) for s in full.values()) == 70322365 + 19264


def test_masks_match_oracle():
    from bmhrl_amd.model.masking import make_masks
    from oracle import bmhrl_oracle as O
    b = syn.synthetic_batch(3, 9, 11, 6, 40, seed=1, d_vid=8, d_aud=4, min_len=2)
    trg = b["captions"][:, :-1]
    mine = make_masks({k: b[k] for k in ("rgb", "flow", "audio")}, trg, "audio_video", 1)
    ref = O.make_masks(b["rgb"], b["audio"], trg, 1)
    for k in ("V_mask", "A_mask", "C_mask"):
        assert torch.equal(mine[k].bool(), ref[k].bool()), k
    assert not mine["V_mask"][1, 0, -1]          # padded tail rows are masked out


def test_generate_synonyms_semantics():
    from bmhrl_amd.epoch_loops.captioning_bmrl_loops import generate_synonyms
    g = torch.Generator().manual_seed(0)
    cap = torch.tensor([[2, 10, 11, 12, 3, 1, 1], [2, 20, 21, 22, 23, 24, 3], [2, 5, 6, 7, 8, 9, 10]])
    out = generate_synonyms(cap.repeat(400, 1), 50, p=0.3, generator=g).view(400, 3, 7)
    assert (out[:, 0, 4] == 1).all() and (out[:, 1, 6] == 1).all()          # first </s> becomes pad
    assert (out[:, 0, 5:] == 1).all()                                        # nothing after it is touched
    changed = (out[:, 2, :] != cap[2]).float().mean()
    assert 0.2 < float(changed) < 0.34                                       # 30 % hit, of which 90 % change the token
    to_pad = ((out[:, 2, :] == 1) & (cap[2] != 1)).float().mean()
    assert 0.18 < float(to_pad) < 0.30                                       # 80 % of the hits -> pad id


def test_flat_adam_cpu_matches_torch_and_bucket_contents():
    from bmhrl_amd.train import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=0.1)
    opt = FlatAdam(ps, lr=1e-2, weight_decay=0.1)
    for step in range(3):
        for p, r in zip(ps, ref):
            g = torch.randn_like(p)
            p.grad = g.clone() if not (step == 1 and p.dim() == 1) else None     # one missing gradient
            r.grad = g.clone() if p.grad is not None else torch.zeros_like(r)
        opt.gather_grads(); opt.step()
        opt_ref.step()
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r, atol=1e-6), (p - r).abs().max()
        assert p.data_ptr() >= opt.flat.data_ptr()                              # parameters live in the flat bucket


def test_trainable_bucket_excludes_unreachable_parameters():
    from types import SimpleNamespace
    from bmhrl_amd.model.bm_hrl_agent import BMHrlAgent
    from bmhrl_amd.train import trainable_bucket
    cfg = syn.tiny_cfg(); cfg.device = "cpu"
    agent = BMHrlAgent(cfg, SimpleNamespace(trg_voc_size=30, train_vocab=SimpleNamespace(vectors=None)))
    names = {id(p): n for n, p in agent.named_parameters()}
    got = {names[id(p)] for p in trainable_bucket(agent)}
    assert not any(n.startswith(("critic.", "manager_core.")) or ".feed_forward.fc" in n and "_fus." in n for n in got)
    assert "bm_enc.encoder.layers.0.feed_forward_M1.fc1.weight" in got and "emb_C.embedder.weight" in got
    # every parameter the reference's warmstart loss reaches is in the bucket (fixture: gradients produced by the reference)
    agent.teach_worker()
    got_w = {names[id(p)] for p in trainable_bucket(agent)}
    assert "manager.linear.weight" not in got_w and "worker.core.projection.weight" in got_w


def test_install_aliases_reference_paths():
    import importlib
    import sys
    import bmhrl_amd.install as inst
    inst.install()
    assert sys.modules["model.bm_hrl_agent"].BMHrlAgent.__module__ == "bmhrl_amd.model.bm_hrl_agent"
    loops = importlib.import_module("epoch_loops.captioning_bmrl_loops")
    for name in ("bimodal_decoder", "audio_decoder", "video_decoder", "bmhrl_validation_next_word_loop", "train_bmhrl_bl",
                 "warmstart_bmhrl_bl", "train_audio_bl", "train_video_bl", "warmstart_audio_bl", "warmstart_video_bl",
                 "analyze_bmhrl_div", "train_detr_rl", "reinforce_detr_rl", "detr_decoder"):
        assert hasattr(loops, name), name


def test_agent_refuses_cpu_inputs():
    """The product has no CPU path: a forward on CPU tensors must fail loudly, not fall back."""
    import pytest
    from types import SimpleNamespace
    from bmhrl_amd.model.bm_hrl_agent import BMHrlAgent
    from bmhrl_amd.model.masking import make_masks
    cfg = syn.tiny_cfg(); cfg.device = "cpu"
    agent = BMHrlAgent(cfg, SimpleNamespace(trg_voc_size=30, train_vocab=SimpleNamespace(vectors=None))).eval()
    agent.set_inference_mode(True)
    b = syn.synthetic_batch(2, 5, 6, 4, 30, seed=0, d_vid=cfg.d_vid, d_aud=cfg.d_aud, min_len=2)
    trg = b["captions"][:, :-1]
    masks = make_masks({k: b[k] for k in ("rgb", "flow", "audio")}, trg, "audio_video", 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        agent((b["rgb"] + b["flow"], b["audio"]), trg, masks)


def test_validation_1by1_host_side(tmp_path):
    """validation_1by1_loop's host half (reference epoch_loops/validation_loops.py:53-118) with a stub decoder: ids -> words,
    cut at </s>, capitalise, results grouped per video, the JSON file name and the duplicate-file rule"""
    import json
    from types import SimpleNamespace
    import torch
    from bmhrl_amd.epoch_loops.validation_loops import tokens_to_sentences, validation_1by1_loop
    itos = ["<unk>", "<blank>", "<s>", "</s>", "a", "man", "runs", "fast"]
    assert tokens_to_sentences([[2, 4, 5, 6, 3, 7, 7], [2, 5, 6, 7, 7, 7, 7], [2, 3, 4, 4, 4, 4, 4]], itos) == \
        ["A man runs", "Man runs fast fast fast fast", ""]
    ds = SimpleNamespace(start_idx=2, end_idx=3, pad_idx=1, phase="val_1", train_vocab=SimpleNamespace(itos=itos),
                         update_iterator=lambda: None)
    batches = [dict(feature_stacks={"k": i}, video_ids=["v1", "v2"] if i == 0 else ["v1", "v3"],
                    starts=torch.tensor([[0.0], [1.5]]), ends=torch.tensor([[2.0], [3.5]])) for i in range(2)]
    loader = SimpleNamespace(dataset=ds, __iter__=None)

    class Loader(list):
        dataset = ds
    calls = []

    def decoder(model, fs, max_len, start, end, pad, modality):
        calls.append((model, fs["k"], max_len, start, end, pad, modality))
        return torch.tensor([[2, 4, 5, 3, 1], [2, 6, 7, 7, 7]])

    class Model:
        module = "the-agent"
        def eval(self):
            calls.append("eval")
    cfg = SimpleNamespace(max_len=4, modality="audio_video", log_path=None, reference_paths=["r1", "r2", "r3", "r4"],
                          max_prop_per_vid=100, tIoUs=[0.3, 0.5, 0.7, 0.9])
    assert validation_1by1_loop(cfg, Model(), Loader(batches), decoder, 3, None) is None       # log_path None -> None
    assert calls[0] == "eval" and calls[1] == ("the-agent", 0, 4, 2, 3, 1, "audio_video")
    cfg.log_path = str(tmp_path / "logs")
    out = validation_1by1_loop(cfg, Model(), Loader(batches), decoder, 3, None)
    path = tmp_path / "logs" / "captioning_results_val_1_e3.json"
    saved = json.load(open(path))
    assert saved["version"] == "VERSION 1.0" and saved["external_data"] == {"used": True, "details": ""}
    assert saved["results"]["v1"] == [{"sentence": "A man", "timestamp": [0.0, 2.0]}] * 2
    assert saved["results"]["v3"] == [{"sentence": "Runs fast fast fast", "timestamp": [1.5, 3.5]}]
    assert out["submission_path"] == str(path)           # no evaluator in this image: the predictions come back
    out2 = validation_1by1_loop(cfg, Model(), Loader(batches), decoder, 3, None)
    assert out2["submission_path"] != str(path) and out2["submission_path"].startswith(str(path)[:-5] + "_")


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r02_bench.json is a verbatim bench.py line: the fields of the driver's contract are all there"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r02_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "caption-train steps/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16" and "workload" in d["config"]
    assert abs(d["value"] - d["steps"] * d["n_gpus"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["launch_us"] * 1e-6) / 1e12) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_bench_refuses_to_report_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own ranks -- or, when the GPUs are not there, exits non-zero
    instead of measuring one rank and calling it two (no GPU in this container: the refusal is what can be checked here)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0          # (no GPU here; on a GPU box: "--gpus 2 but WORLD_SIZE=1")


def test_detr_agent_checkpoint_layout_matches_the_reference():
    """DetrCaption (model/det_bmhrl_agent.py) builds without a GPU and has the reference module's state-dict keys and shapes
    (recorded from the reference's own module in tests/golden/detr_agent.npz); the driver's import resolves"""
    import os
    from types import SimpleNamespace
    import numpy as np
    import bmhrl_amd.install  # noqa: F401
    from model.det_bmhrl_agent import DetrCaption
    from bmhrl_amd import synthetic as syn
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detr_agent.npz"), allow_pickle=False))
    cfg = syn.tiny_cfg(d_model=64, d_model_video=64, d_vid=64, d_model_caps=20, rl_att_heads=4, rl_goal_d=8, dout_p=0.0)
    cfg.pre_goal_attention = False
    agent = DetrCaption(cfg, SimpleNamespace(trg_voc_size=41, train_vocab=SimpleNamespace(vectors=None)))
    want = {str(k): tuple(int(d) for d in str(s).split(",") if d != "") for k, s in zip(g["keys"], g["shapes"])}
    assert {k: tuple(v.shape) for k, v in agent.state_dict().items()} == want
    assert agent.name == "detr_agent" and agent.manager.exploration
    agent.teach_worker()
    assert not agent.manager.exploration and all(p.requires_grad for p in agent.linear.parameters())
    assert not any(p.requires_grad for p in agent.manager_decoder.parameters())


def test_flat_adam_adopts_gradient_homes_and_keeps_values():
    """FlatAdam.adopt_homes (data-parallel steps produce the leaf gradients in the flat bucket): parameters whose gradients
    tile one recorded allocation are moved next to each other inside their bucket, values and Adam state move with them, the
    allocation maps to one contiguous run of the bucket, and whatever does not tile an allocation stays on the copy path."""
    import torch
    from bmhrl_amd.functional import ScratchState
    from bmhrl_amd.train import FlatAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(8, 4), (4,), (8, 4), (4,), (12, 4), (3,), (16,)]          # w_a, b_a, w_b, b_b, w_c, odd, lone
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    opt = FlatAdam(params, lr=1e-3)
    opt.set_buckets([4, 3])
    opt.exp_avg.copy_(torch.randn(opt.n, generator=g))
    before = opt.in_param_order(opt.flat).clone(), opt.in_param_order(opt.exp_avg).clone()
    st = ScratchState()
    stacked_w = torch.randn(16, 4, generator=g)          # [w_a; w_b]: one allocation, two parameters of bucket 0
    stacked_b = torch.randn(8, generator=g)              # [b_a; b_b]
    own = torch.randn(12, 4, generator=g)                # w_c alone
    mixed = torch.randn(3 + 16, generator=g)             # odd + lone: a member of 3 elements -> cannot be glued
    other = torch.randn(5, generator=g)                  # an allocation no parameter uses
    st.log = [(0, 0, 64, stacked_w.data_ptr(), stacked_w), (0, 64, 8, stacked_b.data_ptr(), stacked_b),
              (1, 0, 48, own.data_ptr(), own), (0, 72, 19, mixed.data_ptr(), mixed), (0, 92, 5, other.data_ptr(), other)]
    grads = [stacked_w[:8], stacked_b[:4], stacked_w[8:], stacked_b[4:], own, mixed[:3], mixed[3:]]
    for p, gr in zip(params, grads):
        p.grad = gr
    placed = opt.adopt_homes(st)
    assert placed == 64 + 8 + 48
    assert torch.equal(opt.in_param_order(opt.flat), before[0]) and torch.equal(opt.in_param_order(opt.exp_avg), before[1])
    for p, v in zip(params, before[0].split([p.numel() for p in params])):
        assert torch.equal(p.data.reshape(-1), v) and p.data.data_ptr() >= opt.flat.data_ptr()
    assert set(st.homes) == {(0, 0), (0, 64), (1, 0)} and st.home_buckets[0] is opt.grad
    # the run of [w_a; w_b] is one contiguous piece of the bucket, w_a first
    ia, ib = [k for k, p in enumerate(opt.params) if p is params[0]][0], [k for k, p in enumerate(opt.params) if p is params[2]][0]
    assert ib == ia + 1 and opt.offsets[ib] == opt.offsets[ia] + 32
    assert st.homes[(0, 0)].data_ptr() == opt.grad_views[ia].data_ptr() and st.homes[(0, 0)].numel() == 64
    # buckets keep their members
    assert {id(p) for p in opt.params[:4]} == {id(p) for p in params[:4]}
    # a step that produces the gradients at home copies only the rest
    for k, p in enumerate(opt.params):
        p.grad = opt.grad_views[k] if any(p is q for q in params[:5]) else torch.ones_like(p)
    opt.grad.zero_()
    opt.grad_views[ia].fill_(2.0)
    opt.gather_grads()
    assert float(opt.grad_views[ia].min()) == 2.0                     # untouched (its own memory)
    odd = [k for k, p in enumerate(opt.params) if p is params[5]][0]
    assert float(opt.grad_views[odd].min()) == 1.0                    # copied
