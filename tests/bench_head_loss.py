"""Tuning aid: the warmstart loss tail at the reference shape (480 rows x 10 172 logits): ops.head_loss against
log_softmax_ + smooth_kl_fwd + token_loss_reduce + smooth_kl_bwd, event-timed back to back."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rows, V = 480, 10172
ldg = (V + 7) // 8 * 8
logits = torch.randn(rows, V, device=dev) * 3
trg = torch.randint(2, V, (rows,), device=dev)
trg[-40:] = 1
x = logits.clone()
row_loss = torch.empty(rows, device=dev)
out = torch.empty(2, device=dev)
gb = torch.zeros(rows, ldg, dtype=torch.bfloat16, device=dev)
counter = torch.zeros(4, dtype=torch.int32, device=dev)
one = torch.ones(1, device=dev)


def four():
    ops.log_softmax_(x, V, rows, V)
    ops.smooth_kl_fwd(x, V, trg, None, None, None, 0.7, 1, -1, row_loss, None, rows, V)
    ops.token_loss_reduce(row_loss, trg, rows, 1, None, 1.0, out[0:1], out[1:2])
    ops.smooth_kl_bwd(x, V, trg, None, None, None, 0.7, 1, -1, out[1:2], gb, ldg, None, rows, V, wrt_logits=True, loss_scale2=one)


def fused():
    ops.head_loss(x, V, trg, 0.7, 1, None, 1.0, one, row_loss, out, gb, ldg, counter, rows, V)


for name, fn in (("four kernels", four), ("head_loss", fused), ("four kernels", four), ("head_loss", fused)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
