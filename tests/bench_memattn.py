"""Tuning aid: the one-launch few-query memory attention core (ops.memory_attention) alone, forward and backward, at the two
shapes of the step (video 256 x 1024, audio 800 x 128), against the three launches it replaces."""
import math
import torch
from bmhrl_amd import ops

dev = torch.device("cuda:0")
B, H, L = 16, 4, 30
B2 = 2 * B


def timeit(fn, n=50):
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


for Sk, dm in ((256, 1024), (800, 128)):
    Skp, ldt = (Sk + 7) & ~7, (Sk + 15) & ~15
    mem = torch.randn(B, Sk, dm, device=dev)
    y = torch.empty(B * Sk, dm, dtype=torch.bfloat16, device=dev)
    yt = torch.empty(B, dm, ldt, dtype=torch.bfloat16, device=dev)
    ops.cast_memory(mem, y, yt, B, Sk, dm, ldt)
    y2 = y.repeat(2, 1).contiguous()
    ldq, ldp = 2 * H * dm, 2 * H * Skp
    QD = (torch.randn(B2 * L, ldq, device=dev) * 0.3).bfloat16()
    PD = torch.zeros(B2 * L, ldp, dtype=torch.bfloat16, device=dev)
    Cx = torch.zeros(B2 * L, H * dm, dtype=torch.bfloat16, device=dev)
    mask = torch.ones(B2, 1, Sk, dtype=torch.bool, device=dev)
    scale = 1.0 / 16
    t_f = timeit(lambda: ops.memory_attention(False, QD, H * dm, ldq, y, yt, ldt, PD, ldp, H * Skp, Cx, H * dm, mask, Sk, B, B2, H, L, Sk, dm, scale))
    t_b = timeit(lambda: ops.memory_attention(True, QD, 0, ldq, y, yt, ldt, PD, ldp, H * Skp, Cx, H * dm, mask, Sk, B, B2, H, L, Sk, dm, scale))
    S = torch.empty(B2, L, H, Skp, device=dev)

    def gemm_path():
        ops.gemm(QD, y2, L, Sk, dm, lda=ldq, ldb=dm, a_off=H * dm, batch=(B2, H), a_strides=(L * ldq, dm), b_strides=(Sk * dm, 0),
                 C_f32=S, ldc=H * Skp, c_strides=(L * H * Skp, Skp), alpha=scale, mask=mask, mask_sb1=Sk, mask_sm=0)
        ops.softmax_rows(S, Skp, PD, Skp, B2 * L * H, Sk, rows_per_group=H, group_stride=ldp)
        ops.gemm(PD, y2, L, dm, Sk, lda=ldp, ldb=dm, b_trans=True, batch=(B2, H), a_strides=(L * ldp, Skp), b_strides=(Sk * dm, 0),
                 C_bf16=Cx, ldcb=H * dm, cb_strides=(L * H * dm, dm))
    t_g = timeit(gemm_path)
    t_c = timeit(lambda: ops.cast_memory(mem, y, yt, B, Sk, dm, ldt))
    print(f"Sk {Sk} dm {dm}: fused fwd {t_f:.1f} us, fused bwd {t_b:.1f} us, GEMM + softmax + GEMM {t_g:.1f} us, cast_memory {t_c:.1f} us")
