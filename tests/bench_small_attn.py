"""Tuning aid: the one-launch short-sequence attention core alone (forward, backward) at the step's two shapes."""
import math
import torch
from bmhrl_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, S, dk in ((32, 4, 30, 256), (16, 2, 30, 512)):
    D = H * dk
    qkv = (torch.randn(B * S, 3 * D, device=dev) * 0.3).bfloat16()
    dO = (torch.randn(B * S, D, device=dev) * 0.3).bfloat16()
    O = torch.empty(B * S, D, dtype=torch.bfloat16, device=dev)
    P = torch.empty(B, H, S, 32, dtype=torch.bfloat16, device=dev)
    dqkv = torch.empty_like(qkv)
    mask = torch.tril(torch.ones(S, S, dtype=torch.bool, device=dev)).unsqueeze(0).repeat(B, 1, 1).contiguous()
    sc = 1.0 / math.sqrt(dk)
    tf = timeit(lambda: ops.small_attention_fwd(qkv, qkv, qkv, O, P, 32, mask, S * S, S, B, H, S, S, dk, sc, 3 * D, 3 * D, 3 * D, D,
                                                q_off=0, k_off=D, v_off=2 * D, dropout_p=0.1, seed=5))
    tb = timeit(lambda: ops.small_attention_bwd(dO, D, P, 32, qkv, qkv, qkv, dqkv, dqkv, dqkv, mask, S * S, S, B, H, S, S, dk, sc,
                                                3 * D, 3 * D, 3 * D, 3 * D, 3 * D, 3 * D, q_off=0, k_off=D, v_off=2 * D, dq_off=0,
                                                dk_off=D, dv_off=2 * D))
    print(f"B {B} H {H} S {S} dk {dk}: forward {tf:.1f} us, backward {tb:.1f} us")
