"""Expose the HIP-backed modules under the reference's import paths, so that the reference's driver
(scripts/train_rl_captioning_module.py) picks them up unchanged:

    import bmhrl_amd.install  # before `from model.bm_hrl_agent import BMHrlAgent`

Only the hot-path modules are aliased (model.bm_hrl_agent, model.blocks, model.multihead_attention, model.masking,
model.encoder, model.decoder, model.utils, model.det_bmhrl_agent, model.object_detector,
loss.label_smoothing, loss.biased_kl, epoch_loops.captioning_bmrl_loops, epoch_loops.validation_loops,
captioning_datasets.load_features); everything
else keeps resolving to the reference's own files."""
import importlib
import sys
import types

ALIASES = {
    "model.bm_hrl_agent": "bmhrl_amd.model.bm_hrl_agent",
    "model.blocks": "bmhrl_amd.model.blocks",
    "model.multihead_attention": "bmhrl_amd.model.multihead_attention",
    "model.masking": "bmhrl_amd.model.masking",
    "model.encoder": "bmhrl_amd.model.encoder",
    "model.decoder": "bmhrl_amd.model.decoder",
    "model.utils": "bmhrl_amd.model.utils",
    "model.det_bmhrl_agent": "bmhrl_amd.model.det_bmhrl_agent",
    "model.object_detector": "bmhrl_amd.model.object_detector",
    "loss.label_smoothing": "bmhrl_amd.loss.label_smoothing",
    "loss.biased_kl": "bmhrl_amd.loss.biased_kl",
    "epoch_loops.captioning_bmrl_loops": "bmhrl_amd.epoch_loops.captioning_bmrl_loops",
    "epoch_loops.validation_loops": "bmhrl_amd.epoch_loops.validation_loops",
    "captioning_datasets.load_features": "bmhrl_amd.loader",
}


def install():
    for pkg in ("model", "loss", "epoch_loops", "captioning_datasets"):
        if pkg not in sys.modules:
            try:
                importlib.import_module(pkg)          # the reference's package, when it is on sys.path
            except Exception:
                sys.modules[pkg] = types.ModuleType(pkg)
                sys.modules[pkg].__path__ = []
    for ref_name, ours in ALIASES.items():
        mod = importlib.import_module(ours)
        sys.modules[ref_name] = mod
        setattr(sys.modules[ref_name.split(".")[0]], ref_name.split(".")[1], mod)
    return sorted(ALIASES)


install()
