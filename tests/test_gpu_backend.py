"""GPU: the host-side mirrors of the reference interfaces (sgl_kernel op surface, attention backend, KV pools, AWQ
linear method) drive the HIP kernels and agree with the C oracle."""
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Mode:
    def __init__(self, decode):
        self._d = decode

    def is_decode_or_idle(self):
        return self._d

    def is_extend(self):
        return not self._d


def _runner(Hq, Hkv, D, max_bs=8, max_ctx=512, layers=2, pool=2048, dtype=torch.bfloat16):
    from sglang_security_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator

    r2t = ReqToTokenPool(max_bs, max_ctx, DEV)
    kv = MHATokenToKVPool(pool, 1, dtype, Hkv, D, layers, DEV)
    alloc = TokenToKVPoolAllocator(pool, dtype, DEV, kv)
    cfg = types.SimpleNamespace(num_attention_heads=Hq, head_dim=D, get_num_kv_heads=lambda tp: Hkv)
    mr = types.SimpleNamespace(device=DEV, req_to_token_pool=r2t, token_to_kv_pool=kv, model_config=cfg, tp_size=1,
                               server_args=types.SimpleNamespace(triton_attention_num_kv_splits=8))
    return mr, r2t, kv, alloc


def _layer(i, Hq, Hkv, D):
    return types.SimpleNamespace(layer_id=i, tp_q_head_num=Hq, tp_k_head_num=Hkv, tp_v_head_num=Hkv, qk_head_dim=D,
                                 v_head_dim=D, scaling=1.0 / math.sqrt(D), logit_cap=0.0)


def test_pools_and_set_kv_buffer():
    mr, r2t, kv, alloc = _runner(8, 2, 128)
    assert alloc.available_size() == 2048
    a = alloc.alloc(10)
    b = alloc.alloc(5)
    assert a.tolist() == list(range(1, 11)) and b.tolist() == list(range(11, 16))  # slot 0 is never handed out
    assert alloc.alloc(5000) is None
    alloc.free(a)
    assert alloc.available_size() == 2048 - 5
    alloc.free_group_begin()
    alloc.free(b[:2])
    alloc.free(b[2:])
    assert alloc.available_size() == 2048 - 5
    alloc.free_group_end()
    assert alloc.available_size() == 2048
    rows = r2t.alloc(3)
    assert rows == [0, 1, 2] and r2t.available_size() == 5
    r2t.free(rows)
    k = torch.randn(7, 2, 128, device=DEV).to(torch.bfloat16)
    v = torch.randn(7, 2, 128, device=DEV).to(torch.bfloat16)
    loc = torch.tensor([5, 9, 1, 77, 300, 2047, 2048], device=DEV)
    kv.set_kv_buffer(_layer(1, 8, 2, 128), loc, k, v)
    assert torch.equal(kv.get_key_buffer(1)[loc], k) and torch.equal(kv.get_value_buffer(1)[loc], v)
    assert kv.get_key_buffer(0).abs().sum() == 0 and kv.get_key_buffer(1)[0].abs().sum() == 0


@pytest.mark.parametrize("graph_state", [False, True])
@pytest.mark.parametrize("share", [True, False])
@pytest.mark.parametrize("Hq,Hkv,D,pool_dtype", [(16, 4, 128, torch.bfloat16), (14, 2, 64, torch.bfloat16),
                                                (16, 4, 128, torch.float8_e4m3fn), (16, 4, 128, torch.float8_e5m2fnuz)])
def test_backend_decode_matches_oracle(graph_state, share, Hq, Hkv, D, pool_dtype):
    """Radix-style batch (shared 100-token prefix + private tails) through init_forward_metadata / forward_decode with
    nothing but the reference's own arguments: the backend finds the group on the device (share=True: the shared-prefix
    kernel runs) or is told to keep to the per-request kernel; eager and fixed-address (graph) metadata.  Llama heads,
    Qwen2-0.5B's (head size 64: pairs of kv heads), and fp8 pools."""
    import oracle
    from sglang_security_amd import ops
    from sglang_security_amd.attention import SglkAttnBackend

    B = 5
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D, dtype=pool_dtype)
    g = torch.Generator().manual_seed(0)
    prefix = alloc.alloc(100)
    lens, slots_new = [], []
    for b in range(B):
        tail = alloc.alloc(3 + 7 * b)
        L = 100 + tail.numel()
        r2t.req_to_token[b, :100] = prefix.int()
        r2t.req_to_token[b, 100:L] = tail.int()
        lens.append(L)
        slots_new.append(int(tail[-1]))
    for layer_buf in (kv.get_key_buffer(0), kv.get_value_buffer(0)):
        layer_buf.copy_(torch.randn(layer_buf.shape, generator=g).to(torch.bfloat16))
    seq_lens = torch.tensor(lens, device=DEV)
    q = torch.randn(B, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    k_new = torch.randn(B, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    v_new = torch.randn(B, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    fb = types.SimpleNamespace(forward_mode=_Mode(True), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=seq_lens, seq_lens_sum=sum(lens), out_cache_loc=torch.tensor(slots_new, device=DEV),
                               token_to_kv_pool=kv)
    be = SglkAttnBackend(mr, share_prefix=share)
    layer = _layer(0, Hq, Hkv, D)
    if graph_state:
        be.init_cuda_graph_state(8)
        be.init_forward_metadata_capture_cuda_graph(B, B, fb.req_pool_indices, seq_lens)
        be.init_forward_metadata_replay_cuda_graph(B, fb.req_pool_indices, seq_lens, sum(lens))
    else:
        be.init_forward_metadata(fb)
    md = be.forward_metadata
    assert (md.shared_plan is not None) == share
    if share:
        plan = ops.decode_shared_plan_read(md.shared_plan, B)
        assert plan["num_groups"] == 1 and plan["order"] == list(range(B)) and plan["group_indptr"] == [0, B]
        assert plan["prefix_lens"] == [100] and plan["n_items"] > 0
    o = be.forward(q, k_new, v_new, layer, fb)
    # oracle over the pool AFTER the write of the new token
    kb, vb = kv.get_key_buffer(0).cpu(), kv.get_value_buffer(0).cpu()
    assert torch.equal(kb[slots_new].view(torch.uint8), k_new.cpu().view(B, Hkv, D).to(pool_dtype).view(torch.uint8))
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    ind = torch.cat([r2t.req_to_token[b, : lens[b]].cpu() for b in range(B)])
    if pool_dtype != torch.bfloat16:      # the oracle sees an fp8 pool through the values its codes stand for
        kb, vb = kb.float().to(torch.bfloat16), vb.float().to(torch.bfloat16)
    want = oracle.decode_attention(q.cpu().view(B, Hq, D), kb, vb, ptr, ind, torch.ones(B, dtype=torch.int32), layer.scaling)
    assert (o.float().cpu().view(B, Hq, D) - want).abs().max().item() < 2e-2
    if share:
        assert ops.decode_shared_workspace_status(md.shared_workspace, B, Hq) == 0


def test_backend_decode_nobody_shares():
    """A batch whose requests share nothing: the plan holds no prefix item, and the same launch (which a captured graph
    could not change) walks every request's whole context as its remainder -- same numbers as the oracle."""
    import oracle
    from sglang_security_amd import ops
    from sglang_security_amd.attention import SglkAttnBackend

    Hq, Hkv, D, B = 8, 2, 128, 3
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D, max_ctx=1536)
    g = torch.Generator().manual_seed(5)
    lens = [70, 33, 1290]        # the last one long enough to span two tail blocks
    for b in range(B):
        r2t.req_to_token[b, : lens[b]] = alloc.alloc(lens[b]).int()
    for buf in (kv.k_buffer[0], kv.v_buffer[0]):
        buf.copy_(torch.randn(buf.shape, generator=g).to(torch.bfloat16))
    seq_lens = torch.tensor(lens, device=DEV)
    q = torch.randn(B, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    fb = types.SimpleNamespace(forward_mode=_Mode(True), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=seq_lens, seq_lens_sum=sum(lens), out_cache_loc=None, token_to_kv_pool=kv)
    be = SglkAttnBackend(mr)
    be.init_forward_metadata(fb)
    plan = ops.decode_shared_plan_read(be.forward_metadata.shared_plan, B)
    assert plan["n_items"] == 0 and plan["num_groups"] == B and plan["prefix_lens"] == [0, 0, 0]
    layer = _layer(0, Hq, Hkv, D)
    o = be.forward(q, None, None, layer, fb, save_kv_cache=False)
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    ind = torch.cat([r2t.req_to_token[b, : lens[b]].cpu() for b in range(B)])
    want = oracle.decode_attention(q.cpu().view(B, Hq, D), kv.k_buffer[0].cpu(), kv.v_buffer[0].cpu(), ptr, ind,
                                   torch.ones(B, dtype=torch.int32), layer.scaling)
    assert (o.float().cpu().view(B, Hq, D) - want).abs().max().item() < 2e-2


def test_backend_graph_replay_in_batch_prefix_shape():
    """The shape of benchmark/bench_in_batch_prefix/bench_in_batch_prefix.py:112-116 -- 10 prefixes x 32 requests, the
    requests of the ten groups interleaved in the batch as a scheduler would leave them -- through a REAL HIP graph:
    captured once on one batch, replayed on another whose groups differ (other prefixes, other lengths, other members),
    with only init_forward_metadata_replay_cuda_graph in between.  The shared-prefix path must equal the per-request
    kernel on the same pool (both are within the reference's tolerance of the oracle; between themselves they differ
    by summation order only)."""
    from sglang_security_amd import ops
    from sglang_security_amd.attention import SglkAttnBackend

    Hq, Hkv, D = 32, 8, 128
    n_prefix, per, B = 10, 32, 320
    P, S = 96, 24
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D, max_bs=B + 1, max_ctx=256, pool=32768)
    g = torch.Generator().manual_seed(7)
    for buf in (kv.k_buffer[0], kv.v_buffer[0]):
        buf.copy_(torch.randn(buf.shape, generator=g).to(torch.bfloat16))
    layer = _layer(0, Hq, Hkv, D)

    def batch(seed, plen_of, slen_of):
        """Rows of req_to_token for 10 groups x 32 requests, members interleaved: request b belongs to group b % 10."""
        rows, lens = [], []
        prefixes = [alloc.alloc(plen_of(gi)) for gi in range(n_prefix)]
        for b in range(B):
            gi = b % n_prefix
            tail = alloc.alloc(slen_of(b))
            rows.append(torch.cat([prefixes[gi], tail]))
            lens.append(rows[-1].numel())
        return rows, lens

    be_shared, be_plain = SglkAttnBackend(mr, share_prefix=True), SglkAttnBackend(mr, share_prefix=False)
    outs = {}
    static_rpi = torch.arange(B, device=DEV, dtype=torch.int32)
    static_lens = torch.ones(B, device=DEV, dtype=torch.int32)
    q = torch.randn(B, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    graphs = {}
    for name, be in (("shared", be_shared), ("plain", be_plain)):
        be.init_cuda_graph_state(B)
        rows, lens = batch(1, lambda gi: P, lambda b: S)
        for b in range(B):
            r2t.req_to_token[b, : lens[b]] = rows[b].int()
        static_lens.copy_(torch.tensor(lens, dtype=torch.int32))
        be.init_forward_metadata_capture_cuda_graph(B, B, static_rpi, static_lens)
        fb = types.SimpleNamespace(forward_mode=_Mode(True), batch_size=B, req_pool_indices=static_rpi, seq_lens=static_lens,
                                   seq_lens_sum=sum(lens), out_cache_loc=None, token_to_kv_pool=kv)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            be.forward(q, None, None, layer, fb, save_kv_cache=False)      # warm-up outside the capture
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=stream):
                outs[name] = be.forward(q, None, None, layer, fb, save_kv_cache=False)
        graphs[name] = gr
    # a different batch: ragged prefixes (one of them shorter than a tile: that group does not share), ragged tails
    rows, lens = batch(2, lambda gi: 17 if gi == 3 else 64 + 9 * gi, lambda b: 1 + (b * 7) % 37)
    for b in range(B):
        r2t.req_to_token[b, : lens[b]] = rows[b].int()
    static_lens.copy_(torch.tensor(lens, dtype=torch.int32))
    for name, be in (("shared", be_shared), ("plain", be_plain)):
        be.init_forward_metadata_replay_cuda_graph(B, static_rpi, static_lens, sum(lens))
        graphs[name].replay()
    torch.cuda.synchronize()
    plan = ops.decode_shared_plan_read(be_shared.cuda_graph_shared_plan, B)
    assert plan["num_groups"] == n_prefix
    assert sorted(plan["order"]) == list(range(B))
    for gi in range(n_prefix):                      # members of a group are the requests b with b % 10 == gi, in order
        lo, hi = plan["group_indptr"][gi], plan["group_indptr"][gi + 1]
        assert plan["order"][lo:hi] == list(range(gi, B, n_prefix))
        assert plan["prefix_lens"][gi] == (0 if gi == 3 else 64 + 9 * gi)
    assert ops.decode_shared_workspace_status(be_shared.cuda_graph_shared_workspace, B, Hq) == 0
    a, b_ = outs["shared"].float().cpu(), outs["plain"].float().cpu()
    assert torch.isfinite(a).all()
    torch.testing.assert_close(a, b_, atol=2e-2, rtol=1e-2)
    # and against the oracle on a sample of requests from different groups
    import oracle

    pick = [0, 3, 13, 157, 319]
    ptr = torch.zeros(len(pick) + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.tensor([lens[i] for i in pick]), 0)
    ind = torch.cat([rows[i].int().cpu() for i in pick])
    want = oracle.decode_attention(q.cpu().view(B, Hq, D)[pick], kv.k_buffer[0].cpu(), kv.v_buffer[0].cpu(), ptr, ind,
                                   torch.ones(len(pick), dtype=torch.int32), layer.scaling)
    assert (a.view(B, Hq, D)[pick] - want).abs().max().item() < 2e-2


def test_backend_extend_matches_oracle():
    import oracle
    from sglang_security_amd.attention import SglkAttnBackend

    Hq, Hkv, D, B = 8, 2, 128, 3
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D)
    g = torch.Generator().manual_seed(1)
    pre = [40, 0, 17]
    ext = [33, 70, 5]
    for buf in (kv.k_buffer[1], kv.v_buffer[1]):
        buf.copy_(torch.randn(buf.shape, generator=g).to(torch.bfloat16))
    locs = []
    for b in range(B):
        p = alloc.alloc(pre[b])
        e = alloc.alloc(ext[b])
        r2t.req_to_token[b, : pre[b]] = p.int()
        r2t.req_to_token[b, pre[b]: pre[b] + ext[b]] = e.int()
        locs.append(e)
    T = sum(ext)
    q = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    fb = types.SimpleNamespace(forward_mode=_Mode(False), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=torch.tensor([p + e for p, e in zip(pre, ext)], device=DEV),
                               extend_prefix_lens=torch.tensor(pre, device=DEV), extend_seq_lens=torch.tensor(ext, device=DEV),
                               out_cache_loc=torch.cat(locs), token_to_kv_pool=kv)
    be = SglkAttnBackend(mr)
    layer = _layer(1, Hq, Hkv, D)
    kb0, vb0 = kv.k_buffer[1].cpu().clone(), kv.v_buffer[1].cpu().clone()
    be.init_forward_metadata(fb)
    o = be.forward(q, k, v, layer, fb)
    assert torch.equal(kv.k_buffer[1][fb.out_cache_loc].cpu(), k.cpu().view(T, Hkv, D))  # new tokens were cached
    qo = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32)
    kp = torch.tensor([0] + list(torch.tensor(pre).cumsum(0)), dtype=torch.int32)
    ki = torch.cat([r2t.req_to_token[b, : pre[b]].cpu() for b in range(B)])
    want = oracle.extend_attention(q.cpu().view(T, Hq, D), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D), kb0, vb0, qo,
                                   kp, ki, layer.scaling, 0.0, True)
    assert (o.float().cpu().view(T, Hq, D) - want).abs().max().item() < 2e-2


@pytest.mark.parametrize("graph_state", [False, True])
def test_backend_sliding_window(graph_state):
    """A model with sliding-window layers (model_runner.sliding_window_size = W, layer.sliding_window_size = W): decode
    attends to the last W + 1 tokens of each request, extend to the window of the cached prefix under the reference's
    mask plus the causal new tokens (triton_backend.py:75-131, 759-816); a full-attention layer of the same model keeps
    the whole sequence.  Eager and fixed-address (graph) metadata."""
    import oracle
    from sglang_security_amd.attention import SglkAttnBackend

    Hq, Hkv, D, B, W = 8, 2, 128, 4, 16
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D)
    mr.sliding_window_size = W
    g = torch.Generator().manual_seed(5)
    lens = [5, 17, 18, 90]                       # shorter than, equal to and longer than the window
    for b, L in enumerate(lens):
        r2t.req_to_token[b, :L] = alloc.alloc(L).int()
    for buf in (kv.k_buffer[0], kv.v_buffer[0]):
        buf.copy_(torch.randn(buf.shape, generator=g).to(torch.bfloat16))
    seq_lens = torch.tensor(lens, device=DEV)
    q = torch.randn(B, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    fb = types.SimpleNamespace(forward_mode=_Mode(True), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=seq_lens, seq_lens_sum=sum(lens), out_cache_loc=None, token_to_kv_pool=kv)
    be = SglkAttnBackend(mr)
    if graph_state:
        be.init_cuda_graph_state(8)
        be.init_forward_metadata_capture_cuda_graph(B, B, fb.req_pool_indices, seq_lens)
        be.init_forward_metadata_replay_cuda_graph(B, fb.req_pool_indices, seq_lens, sum(lens))
    else:
        be.init_forward_metadata(fb)
    kb, vb = kv.k_buffer[0].cpu(), kv.v_buffer[0].cpu()
    ones = torch.ones(B, dtype=torch.int32)
    for window_layer in (True, False):
        layer = _layer(0, Hq, Hkv, D)
        layer.sliding_window_size = W if window_layer else -1
        o = be.forward(q, None, None, layer, fb, save_kv_cache=False)
        keep = [min(L, W + 1) if window_layer else L for L in lens]
        ptr = torch.tensor([0] + list(torch.tensor(keep).cumsum(0)), dtype=torch.int32)
        ind = torch.cat([r2t.req_to_token[b, lens[b] - keep[b]: lens[b]].cpu() for b in range(B)])
        want = oracle.decode_attention(q.cpu().view(B, Hq, D), kb, vb, ptr, ind, ones, layer.scaling)
        assert (o.float().cpu().view(B, Hq, D) - want).abs().max().item() < 2e-2
    if graph_state:
        return
    # extend over a cached prefix
    pre, ext = [40, 3, 0, 17], [12, 30, 9, 5]
    for b in range(B):
        r2t.req_to_token[b, : pre[b] + ext[b]] = alloc.alloc(pre[b] + ext[b]).int()
    T = sum(ext)
    qe = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    ke = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    ve = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    fb2 = types.SimpleNamespace(forward_mode=_Mode(False), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                                seq_lens=torch.tensor([p + e for p, e in zip(pre, ext)], device=DEV),
                                extend_prefix_lens=torch.tensor(pre, device=DEV), extend_seq_lens=torch.tensor(ext, device=DEV),
                                out_cache_loc=None, token_to_kv_pool=kv)
    be.init_forward_metadata(fb2)
    layer = _layer(0, Hq, Hkv, D)
    layer.sliding_window_size = W
    o = be.forward(qe, ke, ve, layer, fb2, save_kv_cache=False)
    wpre = [min(p, W + 1) for p in pre]
    qo = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32)
    kp = torch.tensor([0] + list(torch.tensor(wpre).cumsum(0)), dtype=torch.int32)
    ki = torch.cat([r2t.req_to_token[b, pre[b] - wpre[b]: pre[b]].cpu() for b in range(B)])
    want = oracle.extend_attention_window(qe.cpu().view(T, Hq, D), ke.cpu().view(T, Hkv, D), ve.cpu().view(T, Hkv, D), kb, vb,
                                          qo, kp, ki, W, layer.scaling)
    assert (o.float().cpu().view(T, Hq, D) - want).abs().max().item() < 2e-2


def test_awq_linear_method():
    import oracle
    from sglang_security_amd.quantization import AWQConfig, AWQLinearMethod

    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    K, N = 1024, 768
    g = torch.Generator().manual_seed(2)
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, generator=g)
    sc = (torch.rand(K // 128, N, generator=g) * 0.01 + 0.001).to(torch.float16)
    bias = torch.randn(N, generator=g).to(torch.float16)
    for keep in (True, False):
        method = AWQLinearMethod(cfg, keep_reference_layout=keep)
        layer = torch.nn.Module()
        method.create_weights(layer, K, [N // 2, N // 2], K, N, torch.float16, device=DEV)
        assert layer.qweight.shape == (K, N // 8) and layer.qzeros.shape == (K // 128, N // 8) and layer.scales.shape == (K // 128, N)
        layer.qweight.data.copy_(qw)
        layer.qzeros.data.copy_(qz)
        layer.scales.data.copy_(sc)
        method.process_weights_after_loading(layer)
        assert (layer.qweight is None) == (not keep)
        for M in (1, 48, 300):  # decode sizes take the fused kernel, 300 rows the dequantise + matmul path
            x = torch.randn(2, M, K, generator=g).to(torch.float16)[0:1]
            out = method.apply(layer, x.to(DEV), bias.to(DEV))
            assert out.shape == (1, M, N)
            want = oracle.awq_gemm(x[0], qw, sc, qz, bias).float()
            tol = 1e-2 * max(1.0, want.abs().max().item())
            assert (out[0].float().cpu() - want).abs().max().item() <= tol
    # the re-laid-out copy dequantises to the same bits as the reference layout
    from sglang_security_amd import ops

    w_ref = ops.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    w_pk = ops.awq_dequantize_packed(ops.awq_repack(qw.to(DEV)), sc.to(DEV), qz.to(DEV))
    assert torch.equal(w_ref.view(torch.int16), w_pk.view(torch.int16))
    with pytest.raises(ValueError):
        AWQConfig(weight_bits=8)


def test_sgl_kernel_surface():
    """A call of each mirrored op with the reference's argument order, against the oracle."""
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    g = torch.Generator().manual_seed(3)
    x = torch.randn(9, 1024, generator=g).to(torch.float16)
    w = torch.randn(1024, generator=g).to(torch.float16)
    out = sk.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    assert torch.equal(out.cpu().view(torch.int16), oracle.rmsnorm(x, w, 1e-6).view(torch.int16))
    res = torch.randn(9, 1024, generator=g).to(torch.float16)
    xd, rd = x.to(DEV).clone(), res.to(DEV).clone()
    sk.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-6)
    wx, wr = oracle.fused_add_rmsnorm(x, res, w, 1e-6)
    assert torch.equal(xd.cpu().view(torch.int16), wx.view(torch.int16)) and torch.equal(rd.cpu().view(torch.int16), wr.view(torch.int16))
    a = torch.randn(2, 5, 512, generator=g).to(torch.bfloat16)
    y = sk.silu_and_mul(a.to(DEV))
    assert y.shape == (2, 5, 256)
    assert torch.equal(y.cpu().view(torch.int16).reshape(10, 256), oracle.silu_and_mul(a.reshape(10, 512)).view(torch.int16))
    with pytest.raises(ValueError):
        sk.silu_and_mul(torch.zeros(2, 6, dtype=torch.float16, device=DEV))
    with pytest.raises(ValueError):
        sk.apply_rope_with_cos_sin_cache_inplace(torch.zeros(1, device=DEV), xd, xd, 128, torch.zeros(4, 128, dtype=torch.float16, device=DEV))
    va, vb = torch.randn(4, 3, 64, generator=g), torch.randn(4, 3, 64, generator=g)
    sa, sb = torch.randn(4, 3, generator=g), torch.randn(4, 3, generator=g)
    vm, sm = sk.merge_state_v2(va.to(DEV), sa.to(DEV), vb.to(DEV), sb.to(DEV))
    wv, ws = oracle.merge_state(va, sa, vb, sb)
    torch.testing.assert_close(vm.cpu(), wv, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(sm.cpu(), ws, atol=1e-5, rtol=1e-5)
    sk.install("sgl_kernel_under_test")
    import importlib

    assert importlib.import_module("sgl_kernel_under_test").rmsnorm is sk.rmsnorm


def test_torch_ops_registration():
    """torch.ops.sgl_kernel.* with the reference's schemas (common_extension.cc:48-121) dispatch to the HIP kernels."""
    import oracle
    from sglang_security_amd.sgl_kernel.torch_ops import register_torch_ops

    register_torch_ops()
    register_torch_ops()  # idempotent
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 512, generator=g).to(torch.bfloat16)
    w = torch.randn(512, generator=g).to(torch.bfloat16)
    out = torch.empty_like(x, device=DEV)
    torch.ops.sgl_kernel.rmsnorm.default(out, x.to(DEV), w.to(DEV), 1e-5, False)
    assert torch.equal(out.cpu().view(torch.int16), oracle.rmsnorm(x, w, 1e-5).view(torch.int16))
    a = torch.randn(5, 256, generator=g).to(torch.bfloat16)
    o2 = torch.empty(5, 128, dtype=torch.bfloat16, device=DEV)
    torch.ops.sgl_kernel.silu_and_mul.default(o2, a.to(DEV), 0)
    assert torch.equal(o2.cpu().view(torch.int16), oracle.silu_and_mul(a).view(torch.int16))
    qw = torch.randint(-(2 ** 31), 2 ** 31 - 1, (256, 16), dtype=torch.int32, generator=g)
    qz = torch.randint(-(2 ** 31), 2 ** 31 - 1, (2, 16), dtype=torch.int32, generator=g)
    sc = torch.rand(2, 128, generator=g).to(torch.float16)
    wd = torch.ops.sgl_kernel.awq_dequantize.default(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    assert torch.equal(wd.cpu().view(torch.int16), oracle.awq_dequantize(qw, sc, qz).view(torch.int16))
    a8 = (torch.randn(8, 64, generator=g) * 0.5).to(torch.float8_e4m3fn)
    b8 = (torch.randn(32, 64, generator=g) * 0.5).to(torch.float8_e4m3fn)
    sa, sb = torch.rand(8, 1, generator=g), torch.rand(32, 1, generator=g)
    mm = torch.ops.sgl_kernel.fp8_scaled_mm.default(a8.to(DEV), b8.to(DEV).t(), sa.to(DEV), sb.to(DEV), torch.bfloat16, None)
    want = oracle.fp8_scaled_mm(a8, b8.t(), sa, sb, torch.bfloat16)
    torch.testing.assert_close(mm.cpu().float(), want.float(), rtol=2e-2, atol=2e-2)


def test_mla_pool_and_decode():
    """MLA latent pool (576 = 512 latent + 64 rope per token, one kv head): pool write, then absorbed-MLA decode (keys =
    whole row, values = its first 512 elements, a strided view of the same rows), against the oracle."""
    import oracle
    from sglang_security_amd import ops
    from sglang_security_amd.mem_cache.memory_pool import MLATokenToKVPool

    pool = MLATokenToKVPool(256, 1, torch.bfloat16, 512, 64, 2, DEV)
    g = torch.Generator().manual_seed(6)
    n = 90
    nope = torch.randn(n, 1, 512, generator=g).to(torch.bfloat16)
    rope = torch.randn(n, 1, 64, generator=g).to(torch.bfloat16)
    loc = (torch.randperm(255, generator=g)[:n] + 1).to(DEV)
    layer = types.SimpleNamespace(layer_id=1)
    pool.set_mla_kv_buffer(layer, loc, nope.to(DEV), rope.to(DEV))
    kb, vb = pool.get_kv_buffer(1)
    assert torch.equal(kb[loc, 0, :512].cpu(), nope[:, 0]) and torch.equal(kb[loc, 0, 512:].cpu(), rope[:, 0])
    assert vb.shape[-1] == 512 and not vb.is_contiguous()
    B, Hq = 2, 16
    lens = [50, 40]
    q = torch.randn(B, Hq, 576, generator=g).to(torch.bfloat16)
    ind = loc[:90].int()
    ptr = torch.tensor([0, 50, 90], dtype=torch.int32)
    sm = 1.0 / math.sqrt(192)
    want = oracle.decode_attention(q, kb.cpu(), vb.cpu().contiguous(), ptr, ind.cpu(), torch.tensor([2, 1], dtype=torch.int32), sm)
    o = torch.empty(B, Hq, 512, dtype=torch.bfloat16, device=DEV)
    al = torch.empty(B, Hq, 4, 512, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, 4, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb, vb, o, ptr.to(DEV), ind, al, lse, torch.tensor([2, 1], dtype=torch.int32, device=DEV),
                             4, sm, 0.0)
    assert (o.float().cpu() - want).abs().max().item() < 2e-2


@pytest.mark.parametrize("share", [None, True])
def test_backend_mla_decode_shared_prefix(share):
    """The backend over an MLA latent pool: left to itself it keeps absorbed MLA on the per-request kernels (the
    plan-driven form measured slower), share_prefix=True routes the batch through the plan; both against the oracle."""
    import oracle
    from sglang_security_amd import ops
    from sglang_security_amd.attention import SglkAttnBackend
    from sglang_security_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool

    Hq, B, P = 16, 5, 70
    r2t = ReqToTokenPool(8, 512, DEV)
    pool = MLATokenToKVPool(1024, 1, torch.bfloat16, 512, 64, 1, DEV)
    g = torch.Generator().manual_seed(8)
    kb = pool.get_key_buffer(0)
    kb.copy_((torch.randn(kb.shape, generator=g) * 0.5).to(torch.bfloat16))
    lens, nxt = [], 1 + P
    for b in range(B):
        t = 3 + 11 * b
        r2t.req_to_token[b, :P] = torch.arange(1, 1 + P, dtype=torch.int32)
        r2t.req_to_token[b, P:P + t] = torch.arange(nxt, nxt + t, dtype=torch.int32)
        nxt += t
        lens.append(P + t)
    cfg = types.SimpleNamespace(num_attention_heads=Hq, head_dim=576, v_head_dim=512, get_num_kv_heads=lambda tp: 1)
    mr = types.SimpleNamespace(device=DEV, req_to_token_pool=r2t, token_to_kv_pool=pool, model_config=cfg, tp_size=1,
                               server_args=types.SimpleNamespace(triton_attention_num_kv_splits=8))
    seq_lens = torch.tensor(lens, device=DEV)
    fb = types.SimpleNamespace(forward_mode=_Mode(True), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=seq_lens, seq_lens_sum=sum(lens), out_cache_loc=None, token_to_kv_pool=pool)
    be = SglkAttnBackend(mr, share_prefix=share)
    be.init_forward_metadata(fb)
    md = be.forward_metadata
    assert (md.shared_plan is not None) == (share is True)
    if share:
        view = ops.decode_shared_plan_read(md.shared_plan, B)
        assert view["prefix_lens"] == [P] and view["n_items"] > 0
    layer = types.SimpleNamespace(layer_id=0, tp_q_head_num=Hq, tp_k_head_num=1, tp_v_head_num=1, qk_head_dim=576,
                                  v_head_dim=512, scaling=1.0 / math.sqrt(192), logit_cap=0.0)
    q = (torch.randn(B, Hq * 576, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    o = be.forward(q, None, None, layer, fb, save_kv_cache=False)
    torch.cuda.synchronize()
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    ind = torch.cat([r2t.req_to_token[b, : lens[b]].cpu() for b in range(B)])
    kc = kb.cpu()
    want = oracle.decode_attention(q.cpu().view(B, Hq, 576), kc, kc[..., :512].contiguous(), ptr, ind,
                                   torch.ones(B, dtype=torch.int32), layer.scaling)
    assert (o.float().cpu().view(B, Hq, 512) - want.float()).abs().max().item() < 2e-2
    if share:
        assert ops.decode_shared_workspace_status(md.shared_workspace, B, Hq) == 0


class _SpecMode:
    def __init__(self, name):
        self._n = name

    def is_decode_or_idle(self):
        return self._n == "decode"

    def is_target_verify(self):
        return self._n == "target_verify"

    def is_draft_extend(self):
        return self._n == "draft_extend"

    def is_extend(self):
        return self._n != "decode"


def test_backend_target_verify_uses_the_tree_mask():
    """A target-verify batch through init_forward_metadata / forward_extend (triton_backend.py:338-374): the backend
    builds qo_indptr, kv_indptr / kv_indices over the cached sequences and mask_indptr from seq_lens, takes custom_mask
    from spec_info, and the draft tokens attend the prefix plus their ancestors only."""
    import oracle
    from sglang_security_amd.attention import SglkAttnBackend

    Hq, Hkv, D, B, nd = 8, 2, 128, 3, 4
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D)
    mr.server_args.speculative_num_draft_tokens = nd
    g = torch.Generator().manual_seed(11)
    lens = [20, 7, 33]
    for b, L in enumerate(lens):
        r2t.req_to_token[b, :L] = alloc.alloc(L).int()
    new_loc = alloc.alloc(B * nd)
    for buf in (kv.k_buffer[0], kv.v_buffer[0]):
        buf.copy_(torch.randn(buf.shape, generator=g).to(torch.bfloat16))
    # per request a chain 0 <- 1 <- 2 and a sibling 3 of node 1 (parent 0): ancestor masks over the draft tokens
    anc = torch.tensor([[1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [1, 0, 0, 1]], dtype=torch.bool)
    rows = []
    for L in lens:
        rows.append(torch.cat([torch.ones(nd, L, dtype=torch.bool), anc], dim=1).reshape(-1))
    custom_mask = torch.cat(rows).to(DEV)
    seq_lens = torch.tensor(lens, device=DEV)
    spec = types.SimpleNamespace(custom_mask=custom_mask)
    fb = types.SimpleNamespace(forward_mode=_SpecMode("target_verify"), batch_size=B * nd,
                               req_pool_indices=torch.arange(B, device=DEV), seq_lens=seq_lens, seq_lens_sum=sum(lens),
                               out_cache_loc=new_loc, token_to_kv_pool=kv, spec_info=spec)
    be = SglkAttnBackend(mr)
    be.init_forward_metadata(fb)
    md = be.forward_metadata
    assert md.qo_indptr.tolist() == [0, nd, 2 * nd, 3 * nd] and md.max_extend_len == nd
    assert md.kv_indptr.tolist() == [0, 20, 27, 60]
    assert md.mask_indptr.tolist() == [0, nd * 24, nd * 24 + nd * 11, nd * 24 + nd * 11 + nd * 37]
    assert md.custom_mask is custom_mask
    q = torch.randn(B * nd, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(B * nd, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(B * nd, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    layer = _layer(0, Hq, Hkv, D)
    o = be.forward(q, k, v, layer, fb)
    want = oracle.extend_attention_masked(
        q.cpu().view(-1, Hq, D), k.cpu().view(-1, Hkv, D), v.cpu().view(-1, Hkv, D), kv.k_buffer[0].cpu(),
        kv.v_buffer[0].cpu(), md.qo_indptr.cpu(), md.kv_indptr.cpu(), md.kv_indices.cpu(),
        custom_mask.cpu().to(torch.uint8), md.mask_indptr.cpu(), True, layer.scaling)
    torch.testing.assert_close(o.float().cpu().view(-1, Hq, D), want, rtol=2e-2, atol=2e-2)
    # node 3 must not see node 1: changing node 1's key / value leaves node 3's output alone and moves node 2's
    k2, v2 = k.clone(), v.clone()
    k2.view(B, nd, -1)[:, 1] += 1.0
    v2.view(B, nd, -1)[:, 1] += 1.0
    o2 = be.forward(q, k2, v2, layer, fb, save_kv_cache=False)
    a, b2 = o.view(B, nd, -1), o2.view(B, nd, -1)
    assert torch.equal(a[:, 3], b2[:, 3]) and torch.equal(a[:, 0], b2[:, 0])
    assert not torch.equal(a[:, 2], b2[:, 2])


def test_backend_draft_decode_and_draft_extend_take_spec_info_indices():
    """Draft decode reads kv_indptr / kv_indices from spec_info (triton_backend.py:317-319: topk rows per request);
    draft extend asks spec_info.generate_attn_arg_prefill for its arguments (:375-391)."""
    import oracle
    from sglang_security_amd.attention import SglkAttnBackend

    Hq, Hkv, D, B, topk = 8, 2, 128, 2, 2
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D)
    g = torch.Generator().manual_seed(12)
    for buf in (kv.k_buffer[0], kv.v_buffer[0]):
        buf.copy_(torch.randn(buf.shape, generator=g).to(torch.bfloat16))
    lens = [9, 14]
    # topk branches per request: the shared sequence plus one private slot each
    per_row, ptr = [], [0]
    for b, L in enumerate(lens):
        shared = alloc.alloc(L).int()
        for t in range(topk):
            per_row.append(torch.cat([shared, alloc.alloc(1).int()]))
            ptr.append(ptr[-1] + L + 1)
    spec = types.SimpleNamespace(kv_indptr=torch.tensor(ptr, dtype=torch.int32, device=DEV),
                                 kv_indices=torch.cat(per_row).to(DEV))
    fb = types.SimpleNamespace(forward_mode=_SpecMode("decode"), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=torch.tensor(lens, device=DEV), seq_lens_sum=sum(lens),
                               out_cache_loc=None, token_to_kv_pool=kv, spec_info=spec)
    be = SglkAttnBackend(mr)
    be.init_forward_metadata(fb)
    md = be.forward_metadata
    assert md.kv_indptr is spec.kv_indptr and md.num_kv_splits.shape[0] == B * topk
    q = torch.randn(B * topk, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    layer = _layer(0, Hq, Hkv, D)
    o = be.forward(q, None, None, layer, fb, save_kv_cache=False)
    want = oracle.decode_attention(q.cpu().view(-1, Hq, D), kv.k_buffer[0].cpu(), kv.v_buffer[0].cpu(), spec.kv_indptr.cpu(),
                                   spec.kv_indices.cpu(), torch.ones(B * topk, dtype=torch.int32), layer.scaling)
    assert (o.float().cpu().view(-1, Hq, D) - want).abs().max().item() < 2e-2

    # draft extend: 2 and 3 accepted tokens on top of the cached sequences, plain causal (no mask)
    acc = torch.tensor([2, 3], device=DEV)
    pre_idx = torch.cat([per_row[0][: lens[0]], per_row[2][: lens[1]]]).to(DEV)

    def gen_args(req_pool_indices, seq_lens, _unused, req_to_token):
        qo = torch.tensor([0, 2, 5], dtype=torch.int32, device=DEV)
        kp = torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int32, device=DEV)
        return pre_idx, kp, qo, None

    spec2 = types.SimpleNamespace(generate_attn_arg_prefill=gen_args, accept_length=acc)
    fb2 = types.SimpleNamespace(forward_mode=_SpecMode("draft_extend"), batch_size=B,
                                req_pool_indices=torch.arange(B, device=DEV), seq_lens=torch.tensor(lens, device=DEV),
                                seq_lens_sum=sum(lens), out_cache_loc=None, token_to_kv_pool=kv, spec_info=spec2)
    be.init_forward_metadata(fb2)
    md = be.forward_metadata
    assert md.max_extend_len == 3 and md.custom_mask is None and md.mask_indptr is None
    T = 5
    q = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    o = be.forward(q, k, v, layer, fb2, save_kv_cache=False)
    want = oracle.extend_attention(q.cpu().view(-1, Hq, D), k.cpu().view(-1, Hkv, D), v.cpu().view(-1, Hkv, D),
                                   kv.k_buffer[0].cpu(), kv.v_buffer[0].cpu(), md.qo_indptr.cpu(), md.kv_indptr.cpu(),
                                   md.kv_indices.cpu(), layer.scaling)
    torch.testing.assert_close(o.float().cpu().view(-1, Hq, D), want, rtol=2e-2, atol=2e-2)
