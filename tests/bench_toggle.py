"""Tuning aid: bench.py with class-level switches flipped, e.g.
   python tests/bench_toggle.py BMHrlAgent.critic_side_stream=0 BMEncoderLayer.modality_side_stream=0 -- --steps 20 --warmup 5"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bmhrl_amd.model.bm_hrl_agent as m
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for a in args[:cut]:
    k, v = a.split("=")
    cls, attr = k.split(".")
    setattr(getattr(m, cls), attr, bool(int(v)))
sys.argv = [os.path.join(root, "bench.py")] + args[cut + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
