"""Tuning aid: the two big batched GEMMs of the audio self-attention backward in the absorbed form (P recompute with the
PROB epilogue, dS with the DSCORE epilogue): (B, L, H, .) operands, K = 128.  BMHRL_GEMM_DBG=1/2/3 cut the kernel short
(launch only / + prologue / + main loop) for a breakdown."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bmhrl_amd import ops
dev = torch.device("cuda:0")
B, L, H, Sk, dm = 16, 800, 4, 800, 128
Qp = torch.randn(B, L, H, dm, device=dev).to(torch.bfloat16)
X = torch.randn(B, Sk, dm, device=dev).to(torch.bfloat16)
P = torch.empty(B, L, H, Sk, dtype=torch.bfloat16, device=dev)
dS = torch.empty(B, L, H, Sk, dtype=torch.bfloat16, device=dev)
rmax = torch.zeros(B, H, L, device=dev); rsum = torch.full((B, H, L), 800.0, device=dev); delta = torch.zeros(B, H, L, device=dev)
m8 = torch.ones(B, Sk, dtype=torch.uint8, device=dev)
pstr = (L * H * Sk, Sk)
kinds = {
    "plain bf16": lambda: ops.gemm(Qp, X, L, Sk, dm, lda=H * dm, ldb=dm, batch=(B, H), a_strides=(L * H * dm, dm), b_strides=(Sk * dm, 0),
                                   C_bf16=P, ldcb=H * Sk, cb_strides=pstr),
    "PROB": lambda: ops.gemm(Qp, X, L, Sk, dm, lda=H * dm, ldb=dm, batch=(B, H), a_strides=(L * H * dm, dm), b_strides=(Sk * dm, 0),
                             C_bf16=P, ldcb=H * Sk, cb_strides=pstr, epilogue=ops.EPI_PROB, alpha=1 / 16, mask=m8, mask_sb1=Sk, mask_sm=0,
                             rowvec=rmax, rowvec2=rsum, rv_strides=(H * L, L)),
    "DSCORE": lambda: ops.gemm(Qp, X, L, Sk, dm, lda=H * dm, ldb=dm, batch=(B, H), a_strides=(L * H * dm, dm), b_strides=(Sk * dm, 0),
                               C_bf16=dS, ldcb=H * Sk, cb_strides=pstr, epilogue=ops.EPI_DSCORE, alpha=1 / 16, rowvec=delta,
                               rv_strides=(H * L, L), aux=P, ldaux=H * Sk, aux_strides=pstr),
}
for name, run in kinds.items():
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(10): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) * 100:7.1f} us   (dbg={os.environ.get('BMHRL_GEMM_DBG', '0')})")
