"""Tuning aid: bench.py with the segment critic replaced by constant labels -- the step time a free critic would give
(the upper bound of anything done to csrc/critic.hip).  usage: python tests/probes/no_critic.py [bench.py arguments]"""
import os
import runpy
import sys
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from bmhrl_amd.model import bm_hrl_agent as m  # noqa: E402


def no_critic(self, emb, threshold):
    B, L, _ = emb.shape
    return torch.zeros(B, L, 1, device=emb.device), torch.zeros(B, L, dtype=torch.int32, device=emb.device)


if os.environ.get("PROBE_KEEP_CRITIC", "0") != "1":
    m.SegmentCritic.score_and_labels = no_critic
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
