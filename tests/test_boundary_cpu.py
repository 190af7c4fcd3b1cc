"""Drop-in boundary: every name the reference's own sources import from a module that
`bmhrl_amd.install` aliases must resolve after `import bmhrl_amd.install` -- in particular the import block of the driver
(scripts/train_rl_captioning_module.py:14-26) and of model/det_bmhrl_agent.py:1-9.

The list of names is a committed fixture (tests/golden/boundary_names.json) produced by parsing (ast) the reference in the
build container (tests/golden/make_boundary_names.py); when the reference is present the fixture is re-derived and compared."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "boundary_names.json")
REF = "/root/reference"


def _load():
    return json.load(open(FIX))


def test_every_imported_name_resolves():
    import bmhrl_amd.install as inst      # noqa: F401  (aliases on import)
    names = _load()
    assert set(names) <= set(inst.ALIASES)
    missing = []
    for mod, wanted in names.items():
        m = sys.modules[mod]
        for n in wanted:
            if n != "*" and not hasattr(m, n):
                missing.append(f"{mod}.{n} (imported at {wanted[n][0]})")
    assert not missing, missing


def test_driver_import_block_names_are_in_the_fixture():
    """the names of the two import blocks VERDICT r01 names, spelled out (so the fixture cannot silently shrink)"""
    names = _load()
    for n in ("BMHrlAgent", "BMManagerValueFunction", "BMWorkerValueFunction", "AudioAgent", "VideoAgent",
              "SegmentCritic", "UnimodalFusion", "Worker", "Manager", "WorkerCore", "LinearCore"):
        assert n in names["model.bm_hrl_agent"], n
    for n in ("bimodal_decoder", "audio_decoder", "video_decoder", "bmhrl_validation_next_word_loop", "train_bmhrl_bl",
              "warmstart_bmhrl_bl", "train_audio_bl", "train_video_bl", "warmstart_audio_bl", "warmstart_video_bl",
              "analyze_bmhrl_div", "train_detr_rl", "reinforce_detr_rl", "detr_decoder"):
        assert n in names["epoch_loops.captioning_bmrl_loops"], n
    assert set(names["loss.biased_kl"]) == {"BiasedKL", "Reinforce"}


def test_out_of_hot_path_names_raise_on_construction():
    import bmhrl_amd.install  # noqa: F401
    from model.bm_hrl_agent import AudioAgent, UnimodalFusion, VideoAgent
    for cls in (AudioAgent, VideoAgent, UnimodalFusion):
        with pytest.raises(NotImplementedError):
            cls(None, None)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_fixture_matches_the_reference_sources(tmp_path):
    before = _load()
    gen = os.path.join(HERE, "golden", "make_boundary_names.py")
    keep = open(FIX).read()
    try:
        subprocess.check_call([sys.executable, gen, REF], stdout=subprocess.DEVNULL)
        assert _load() == before
    finally:
        open(FIX, "w").write(keep)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_driver_import_statements_execute(tmp_path):
    """execute the driver's `from <aliased module> import ...` statements (those of :14-26 that touch aliased modules)
    in a fresh interpreter after `import bmhrl_amd.install`"""
    import ast
    src = open(os.path.join(REF, "scripts", "train_rl_captioning_module.py")).read()
    import bmhrl_amd.install as inst
    stmts = [ast.get_source_segment(src, n) for n in ast.parse(src).body
             if isinstance(n, ast.ImportFrom) and n.module in inst.ALIASES]
    assert len(stmts) >= 3
    code = "import sys\nsys.path.insert(0, %r)\nimport bmhrl_amd.install\n" % os.path.dirname(HERE) + "\n".join(stmts) + "\nprint('ok')\n"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
