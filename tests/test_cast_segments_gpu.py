"""bmhrl_cast_segments (one-launch refresh of all bf16 weight shadows / concatenated biases) and ShadowCache.refresh."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cast_segments_matches_per_tensor_casts():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    shapes = [(1024, 300), (64, 364), (3, 5), (4100, 128), (1, 4096), (17, 1026)]     # aligned, padded, odd, multi-block
    srcs = [torch.randn(r, c, generator=g).to(dev) for r, c in shapes]
    dsts = [torch.zeros(r, ops.pad8(c), dtype=torch.bfloat16, device=dev) for r, c in shapes]
    bias_src = [torch.randn(n, generator=g).to(dev) for n in (1024, 300, 5000)]
    bias_dst = torch.zeros(sum(b.numel() for b in bias_src), device=dev)
    rows, blk, off = [], 0, 0
    for s, d in zip(srcs, dsts):
        rows.append([s.data_ptr(), d.data_ptr(), s.shape[0], s.shape[1], d.shape[1], blk])
        blk += (s.numel() + ops.SEG_ELEMS_PER_BLOCK - 1) // ops.SEG_ELEMS_PER_BLOCK
    for b in bias_src:
        rows.append([b.data_ptr(), bias_dst.data_ptr() + 4 * off, 1, b.numel(), 0, blk])
        off += b.numel()
        blk += (b.numel() + ops.SEG_ELEMS_PER_BLOCK - 1) // ops.SEG_ELEMS_PER_BLOCK
    table = torch.tensor(rows, dtype=torch.int64).to(dev)
    ops.cast_segments(table, len(rows), blk)
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        c = s.shape[1]
        assert torch.equal(d[:, :c], s.to(torch.bfloat16))
        assert float(d[:, c:].float().abs().sum()) == 0.0          # padding untouched
    assert torch.equal(bias_dst, torch.cat(bias_src))


def test_shadow_refresh_tracks_weight_updates():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd.functional import ShadowCache
    dev = torch.device("cuda:0")
    ws = [torch.nn.Parameter(torch.randn(64, 300, device=dev)) for _ in range(3)]
    bs = [torch.nn.Parameter(torch.randn(64, device=dev)) for _ in range(3)]
    cache = ShadowCache()
    w0 = cache.weight(*ws)
    b0 = cache.bias(*bs)
    with torch.no_grad():
        for p in ws + bs:
            p.data.mul_(2.0)            # like the fused Adam: no version bump
    cache.invalidate()
    cache.refresh()
    w1, b1 = cache.weight(*ws), cache.bias(*bs)
    assert w1.data_ptr() == w0.data_ptr() and b1.data_ptr() == b0.data_ptr()        # same buffers, no re-allocation
    assert torch.equal(w1[:, :300], torch.cat([p.detach() for p in ws]).to(torch.bfloat16))
    assert torch.equal(b1, torch.cat([p.detach() for p in bs]))


def test_cast_colsum_and_gemm_epilogue_colsum():
    """Bias gradients taken in the producing kernels: cast + column sums, and column sums of a GEMM's output tile."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bmhrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    for rows, cols in ((480, 300), (4096, 1024), (37, 5), (12800, 128), (1000, 64), (333, 20), (64, 600)):
        x = torch.randn(rows, cols, generator=g).to(dev)
        y = torch.zeros(rows, ops.pad8(cols), dtype=torch.bfloat16, device=dev)
        cs = torch.zeros(cols, device=dev)
        ops.cast_colsum_bf16(x, cols, y, y.shape[1], rows, cols, cs, scale=0.5)
        ref = (x * 0.5).to(torch.bfloat16)
        assert torch.equal(y[:, :cols], ref)
        assert float((cs - ref.float().sum(0)).abs().max()) < 1e-3 * max(1.0, float(ref.float().sum(0).abs().max()))
    # GEMM with a bf16 output and column sums, batched over heads (column index = head * dk + n)
    B, H, M, N, K = 2, 3, 200, 64, 96
    A = torch.randn(B, H, M, K, generator=g).to(dev).to(torch.bfloat16)
    Bm = torch.randn(B, H, N, K, generator=g).to(dev).to(torch.bfloat16)
    out = torch.zeros(B * M, H * N, dtype=torch.bfloat16, device=dev)
    cs = torch.zeros(H * N, device=dev)
    ops.gemm(A, Bm, M, N, K, lda=K, ldb=K, batch=(B, H), a_strides=(H * M * K, M * K), b_strides=(H * N * K, N * K),
             C_bf16=out, ldcb=H * N, cb_strides=(M * H * N, N), colsum=cs, colsum_sb2=N)
    ref = torch.einsum("bhmk,bhnk->bmhn", A.float(), Bm.float()).reshape(B * M, H * N)
    assert float((out.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    assert float((cs - ref.sum(0)).abs().max()) < 5e-3 * float(ref.sum(0).abs().max())
