"""Per-kernel parity of libmtam_hip.so against CPU references (GPU box only).

Each test feeds the same seeded inputs to one C-ABI entry point and to a float64
restatement of the reference ops (the oracle where one exists, plain numpy
otherwise).  Tolerances are stated per test; integer outputs are bit-exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D = 128


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.fixture(scope="module")
def ops(hip_lib):
    from mtamrecommender_amd import hip_ops
    return hip_ops


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 70, 45), (6400, 128, 256), (130, 384, 129)])
@pytest.mark.parametrize("split", [False, True])          # fp32 MFMA chain / six bf16 products of the split operands
def test_gemm_store(ops, ta, tb, M, N, K, split):
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (Bm.T if tb else Bm).astype(np.float64)
    C = torch.full((M, N), 7.0, device="cuda")
    ops.gemm(dev(A), dev(Bm), C, trans_a=bool(ta), trans_b=bool(tb), split=split)
    assert rel_err(C.cpu().numpy(), ref) < 2e-6     # fp32 fma chain over K <= 256; the split products are no worse


def test_gemm_is_ordered_fma_chain(ops):
    """The fp32 MFMA result is bit-for-bit a k-ordered fmaf chain (what the
    scoring contract for top-K relies on)."""
    rng = np.random.default_rng(5)
    M, N, K = 37, 50, 128
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bt = rng.standard_normal((N, K)).astype(np.float32)
    C = torch.zeros((M, N), device="cuda")
    ops.gemm(dev(A), dev(Bt), C, trans_b=True, split=False)
    import oracle.c_oracle as co
    ref = co.score_fma(A, Bt)
    assert np.array_equal(C.cpu().numpy(), ref)


@pytest.mark.parametrize("split,tb", [(False, False), (True, False), (True, True)])
def test_gemm_epilogues(ops, split, tb):
    """tb: B stored [N, K] (C = A B^T) -- with split operands and few tiles that is the 32 x 64 tiling."""
    import functools
    rng = np.random.default_rng(11)
    M, N, K = 150, 200, 96
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    Bdev = dev(np.ascontiguousarray(Bm.T)) if tb else dev(Bm)

    def ops_gemm(a, b_unused, c, **kw):
        return ops.gemm(a, Bdev, c, trans_b=tb, split=split, **kw)
    bias = rng.standard_normal(N).astype(np.float32)
    aux = rng.standard_normal((M, N)).astype(np.float32)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    acc = A.astype(np.float64) @ Bm.astype(np.float64)
    tol = 2e-6

    C = torch.zeros((M, N), device="cuda")
    ops_gemm(dev(A), dev(Bm), C, epilogue=ops.EPI_BIAS, bias=dev(bias))
    assert rel_err(C.cpu().numpy(), acc + bias) < tol
    ops_gemm(dev(A), dev(Bm), C, epilogue=ops.EPI_BIAS_RELU, bias=dev(bias))
    assert rel_err(C.cpu().numpy(), np.maximum(acc + bias, 0)) < tol
    aux_out = torch.zeros((M, N), device="cuda")
    ops_gemm(dev(A), dev(Bm), C, epilogue=ops.EPI_RELU_ADD, aux_in=dev(aux), aux_out=aux_out)
    assert rel_err(aux_out.cpu().numpy(), np.maximum(acc, 0)) < tol
    assert rel_err(C.cpu().numpy(), np.maximum(acc, 0) + aux) < tol
    C = dev(c0).clone()
    ops_gemm(dev(A), dev(Bm), C, epilogue=ops.EPI_ACCUM)
    assert rel_err(C.cpu().numpy(), c0 + acc) < tol
    C = dev(c0).clone()
    ops_gemm(dev(A), dev(Bm), C, epilogue=ops.EPI_ACCUM_MASK, aux_in=dev(aux), aux_out=aux_out)
    assert rel_err(C.cpu().numpy(), c0 + acc) < tol
    assert rel_err(aux_out.cpu().numpy(), np.where(aux > 0, c0 + acc, 0)) < tol
    add2 = rng.standard_normal((M, N)).astype(np.float32)
    C = dev(c0).clone()
    ops_gemm(dev(A), dev(Bm), C, epilogue=ops.EPI_ACCUM2_MASK, bias=dev(add2), aux_in=dev(aux), aux_out=aux_out)
    assert rel_err(C.cpu().numpy(), c0 + acc + add2) < tol
    assert rel_err(aux_out.cpu().numpy(), np.where(aux > 0, c0 + acc + add2, 0)) < tol


@pytest.mark.parametrize("x3", [False, True])
@pytest.mark.parametrize("split", [1, 4, 25])
def test_gemm_split_k_atomic(ops, split, x3):
    rng = np.random.default_rng(split)
    K, M, N = 1850, 128, 384          # K = B*L of a short final batch (37 x 50)
    A = rng.standard_normal((K, M)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    C = dev(c0).clone()
    ops.gemm(dev(A), dev(Bm), C, trans_a=True, epilogue=ops.EPI_ATOMIC, split_k=split, split=x3)
    ref = c0 + A.T.astype(np.float64) @ Bm.astype(np.float64)
    assert rel_err(C.cpu().numpy(), ref) < 5e-6


def test_gemm_submatrix_views(ops):
    """Leading dimensions: B operand is a column block of a wider matrix."""
    rng = np.random.default_rng(3)
    R, wide = 200, 384
    A = rng.standard_normal((R, D)).astype(np.float32)
    G = rng.standard_normal((R, wide)).astype(np.float32)
    Gd = dev(G)
    C = torch.zeros((D, 256), device="cuda")
    ops.gemm(dev(A), Gd.view(-1)[128:], C, trans_a=True, M=D, N=256, K=R, lda=D, ldb=wide, ldc=256)
    ref = A.T.astype(np.float64) @ G[:, 128:384].astype(np.float64)
    assert rel_err(C.cpu().numpy(), ref) < 2e-6


@pytest.mark.parametrize("split", [False, True])
def test_gemm_batched(ops, split):
    """(sample, head) batches over sub-matrices of packed buffers, as the self-attention encoder uses them."""
    import functools
    gemm_batched = functools.partial(ops.gemm_batched, split=split)
    rng = np.random.default_rng(21)
    Bn, H, L, d = 3, 4, 37, 32
    Dm = H * d
    qkv = rng.standard_normal((Bn * L, 3 * Dm)).astype(np.float32)
    out = torch.full((Bn, H, L, L), 9.0, device="cuda")
    qd = dev(qkv)
    gemm_batched(qd, qd.view(-1)[Dm:], out, L, L, d, 3 * Dm, (L * 3 * Dm, d), 3 * Dm, (L * 3 * Dm, d), L,
                     (H * L * L, L * L), (Bn, H), trans_b=True)
    q3 = qkv.reshape(Bn, L, 3 * Dm).astype(np.float64)
    ref = np.einsum("bihc,bjhc->bhij", q3[:, :, :Dm].reshape(Bn, L, H, d), q3[:, :, Dm:2 * Dm].reshape(Bn, L, H, d))
    assert rel_err(out.cpu().numpy(), ref) < 2e-6
    # W V with accumulate into a strided output, and the transposed-A form
    w = rng.standard_normal((Bn, H, L, L)).astype(np.float32)
    o0 = rng.standard_normal((Bn * L, Dm)).astype(np.float32)
    o = dev(o0).clone()
    gemm_batched(dev(w), qd.view(-1)[2 * Dm:], o, L, d, L, L, (H * L * L, L * L), 3 * Dm, (L * 3 * Dm, d), Dm,
                     (L * Dm, d), (Bn, H), epilogue=ops.EPI_ACCUM)
    v = q3[:, :, 2 * Dm:].reshape(Bn, L, H, d)
    ref = o0.reshape(Bn, L, H, d) + np.einsum("bhij,bjhc->bihc", w.astype(np.float64), v)
    assert rel_err(o.cpu().numpy().reshape(Bn, L, H, d), ref) < 2e-6
    o2 = torch.zeros((Bn * L, Dm), device="cuda")
    gemm_batched(dev(w), qd.view(-1)[2 * Dm:], o2, L, d, L, L, (H * L * L, L * L), 3 * Dm, (L * 3 * Dm, d), Dm,
                     (L * Dm, d), (Bn, H), trans_a=True)
    ref = np.einsum("bhji,bjhc->bihc", w.astype(np.float64), v)
    assert rel_err(o2.cpu().numpy().reshape(Bn, L, H, d), ref) < 2e-6


def test_gemm_rejects_bad_arguments(ops):
    from mtamrecommender_amd._lib import MtamHipError
    a = torch.zeros((8, 8), device="cuda")
    with pytest.raises(MtamHipError):
        ops.gemm(a, a, a, epilogue=ops.EPI_BIAS)            # bias missing
    with pytest.raises(MtamHipError):
        ops.gemm(a, a, a, split_k=2)                        # split-K without atomic epilogue
    with pytest.raises(MtamHipError):
        ops.gemm(a.cpu(), a, a)                             # host tensor


def test_colsum(ops):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((777, 300)).astype(np.float32)
    out = torch.ones(300, device="cuda")
    ops.colsum_atomic(dev(x), out)
    assert rel_err(out.cpu().numpy(), 1.0 + x.astype(np.float64).sum(0)) < 1e-5


# ------------------------------------------------------------- embedding
def _emb_case(B, L, seed, rows=(97, 13, None, 31)):
    rng = np.random.default_rng(seed)
    vi, vc, vp, vu = rows[0], rows[1], (L + 3 if rows[2] is None else rows[2]), rows[3]
    tabs = [rng.standard_normal((v, D)).astype(np.float32) for v in (vi, vc, vp, vu)]
    sl = rng.integers(2, L + 1, size=B).astype(np.int32)
    if B > 1:
        sl[0] = L                                   # one sample without padding
    ids = [np.zeros((B, L), np.int32) for _ in range(3)]
    for b in range(B):
        for t, v in zip(ids, (vi, vc, vp)):
            t[b, :sl[b]] = rng.integers(0, v, size=sl[b])
    uid = rng.integers(0, vu, size=B).astype(np.int32)
    return tabs, ids, uid, sl


@pytest.mark.parametrize("B,L", [(1, 2), (5, 7), (128, 50), (37, 50)])
@pytest.mark.parametrize("with_user", [1, 0])
def test_emb_gather(ops, B, L, with_user):
    tabs, ids, uid, sl = _emb_case(B, L, B * 100 + L)
    R = B * L
    ic = torch.zeros((R, 2 * D), device="cuda")
    pos = torch.zeros((R, D), device="cuda")
    usr = torch.zeros((B, D), device="cuda")
    npart = ops.emb_gather_partials(B, L)
    part = torch.full((npart,), 123.0, device="cuda")
    ops.emb_gather_fwd(dev(tabs[0]), dev(tabs[1]), dev(tabs[2]), dev(tabs[3]), dev(ids[0]), dev(ids[1]),
                       dev(ids[2]), dev(uid), B, L, with_user, ic, pos, usr, part)
    I, C, P, U = tabs[0][ids[0].ravel()], tabs[1][ids[1].ravel()], tabs[2][ids[2].ravel()], tabs[3][uid]
    assert np.array_equal(ic.cpu().numpy(), np.concatenate([I, C], axis=1))     # row copy: bit exact
    assert np.array_equal(pos.cpu().numpy(), P)
    assert np.array_equal(usr.cpu().numpy(), U)
    sq = sum(float((a.astype(np.float64) ** 2).sum()) for a in (I, C, P)) + \
        (float((U.astype(np.float64) ** 2).sum()) if with_user else 0.0)
    assert abs(float(part.double().sum()) - sq) / sq < 1e-6


@pytest.mark.parametrize("B,L", [(1, 2), (5, 7), (128, 50), (130, 9), (67, 50)])
@pytest.mark.parametrize("with_user", [1, 0])
def test_emb_scatter_add(ops, B, L, with_user):
    tabs, ids, uid, sl = _emb_case(B, L, B * 10 + L + 1)
    rng = np.random.default_rng(B + L)
    R = B * L
    live = (np.arange(L)[None, :] < sl[:, None]).reshape(R, 1)
    d_ic = (rng.standard_normal((R, 2 * D)) * live).astype(np.float32)          # zero at padded slots
    d_pos = (rng.standard_normal((R, D)) * live).astype(np.float32)
    I, C, P, U = tabs[0][ids[0].ravel()], tabs[1][ids[1].ravel()], tabs[2][ids[2].ravel()], tabs[3][uid]
    ic = np.concatenate([I, C], axis=1)
    reg = 0.37
    g0 = [rng.standard_normal(t.shape).astype(np.float32) for t in tabs]        # pre-existing content
    g = [dev(x).clone() for x in g0]
    part = torch.full((ops.emb_scatter_partials(B, L),), 9.0, device="cuda")
    ops.emb_scatter_add_bwd(dev(d_ic), dev(d_pos), dev(ic), dev(P), dev(U), dev(ids[0]), dev(ids[1]),
                            dev(ids[2]), dev(uid), dev(sl), B, L, reg, with_user, g[0], g[1], g[2], g[3], part)
    ref = [x.astype(np.float64) for x in g0]
    ci = d_ic[:, :D].astype(np.float64) + reg * I
    cc = d_ic[:, D:].astype(np.float64) + reg * C
    cp = d_pos.astype(np.float64) + reg * P
    np.add.at(ref[0], ids[0].ravel(), ci)
    np.add.at(ref[1], ids[1].ravel(), cc)
    np.add.at(ref[2], ids[2].ravel(), cp)
    sq = float((ci ** 2).sum() + (cc ** 2).sum() + (cp ** 2).sum())
    if with_user:
        cu = reg * U.astype(np.float64)
        np.add.at(ref[3], uid, cu)
        sq += float((cu ** 2).sum())
    for got, want in zip(g, ref):
        assert rel_err(got.cpu().numpy(), want) < 1e-5     # float atomics: order-dependent rounding
    assert abs(float(part.double().sum()) - sq) / sq < 1e-5


@pytest.mark.parametrize("B,L,with_user", [(128, 50, 1), (1, 2, 1), (5, 7, 0), (33, 50, 1)])
def test_emb_scatter_add_fused_sources(ops, B, L, with_user):
    """mtam_emb_scatter_add_bwd_fused: the [item | category] gradient rows formed inside the kernel from
    d_z . W4^T (no [R, 2D] buffer) and the position rows read from the table through the ids, against the plain
    entry point fed with the same gradient computed by the GEMM kernel."""
    tabs, ids, uid, sl = _emb_case(B, L, B * 10 + L + 2)
    rng = np.random.default_rng(B + 3 * L)
    R = B * L
    live = (np.arange(L)[None, :] < sl[:, None]).reshape(R, 1)
    d_z = dev((rng.standard_normal((R, D)) * live).astype(np.float32))          # zero rows at padded slots
    W4 = dev((rng.standard_normal((2 * D, D)) * 0.1).astype(np.float32))
    d_pos = dev((rng.standard_normal((R, D)) * live).astype(np.float32))
    I, C, P, U = tabs[0][ids[0].ravel()], tabs[1][ids[1].ravel()], tabs[2][ids[2].ravel()], tabs[3][uid]
    ic = dev(np.concatenate([I, C], axis=1))
    reg = 0.37
    g0 = [rng.standard_normal(t.shape).astype(np.float32) for t in tabs]
    args = (dev(ids[0]), dev(ids[1]), dev(ids[2]), dev(uid), dev(sl), B, L, reg, with_user)
    # reference: the GEMM kernel's d_ic through the plain entry point
    d_ic = torch.zeros((R, 2 * D), device="cuda")
    ops.gemm(d_z, W4, d_ic, trans_b=True)
    g_ref = [dev(x).clone() for x in g0]
    part_ref = torch.zeros(ops.emb_scatter_partials(B, L), device="cuda")
    ops.emb_scatter_add_bwd(d_ic, d_pos, ic, dev(P), dev(U), *args, g_ref[0], g_ref[1], g_ref[2], g_ref[3], part_ref)
    g = [dev(x).clone() for x in g0]
    part = torch.full((ops.emb_scatter_partials(B, L),), 9.0, device="cuda")
    ops.emb_scatter_add_bwd(None, d_pos, ic, None, dev(U), *args, g[0], g[1], g[2], g[3], part,
                            pos_table=dev(tabs[2]), d_z=d_z, W4=W4)
    for got, want in zip(g, g_ref):
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 1e-5      # float atomics + two fp32 summation orders
    assert abs(float(part.double().sum()) - float(part_ref.double().sum())) < 1e-5 * float(part_ref.double().sum())


@pytest.mark.parametrize("B,L,n_g", [(128, 50, 61 * 4096), (5, 7, 4096 * 3 + 1234), (33, 50, 100)])
def test_emb_scatter_add_with_the_norm_rider(ops, B, L, n_g):
    """mtam_emb_scatter_add_bwd_norm: the scatter-add is unchanged and the riding workgroups do what
    mtam_sqnorm_state_loss does -- the same partials bit for bit (the same per-block arithmetic), the same Adam state,
    the loss to float64-summation rounding (1024 threads sum it in another order)."""
    tabs, ids, uid, sl = _emb_case(B, L, B * 10 + L + 5)
    rng = np.random.default_rng(B + 5 * L)
    R = B * L
    live = (np.arange(L)[None, :] < sl[:, None]).reshape(R, 1)
    d_ic = dev((rng.standard_normal((R, 2 * D)) * live).astype(np.float32))
    d_pos = dev((rng.standard_normal((R, D)) * live).astype(np.float32))
    I, C, P, U = tabs[0][ids[0].ravel()], tabs[1][ids[1].ravel()], tabs[2][ids[2].ravel()], tabs[3][uid]
    ic = dev(np.concatenate([I, C], axis=1))
    g0 = [rng.standard_normal(t.shape).astype(np.float32) for t in tabs]
    args = (dev(ids[0]), dev(ids[1]), dev(ids[2]), dev(uid), dev(sl), B, L, 0.37, 1)
    gd = dev(rng.standard_normal(n_g).astype(np.float32))
    l2, ce = dev(rng.uniform(0, 2, 4832).astype(np.float32)), dev(rng.uniform(0, 9, B).astype(np.float32))
    lr = dev(np.array([1e-3], np.float32))
    nb = ops.sqnorm_blocks(n_g)
    out = []
    for ride in (False, True):
        g = [dev(x).clone() for x in g0]
        part = torch.zeros(ops.emb_scatter_partials(B, L), device="cuda")
        npart = torch.full((nb + 7,), 3.0, device="cuda")
        state = dev(np.array([0.0, 0.9, 0.999, 1e-8, 0.81, 0.998, 0.0, 0.0], np.float32))
        loss = torch.zeros(4, device="cuda")
        norm = dict(g=gd, n=n_g, partials=npart, offset=2, lr=lr, adam_state=state, l2_partial=l2, ce=ce, B=B,
                    reg=1e-4, ce_scale=1.0 / B, loss=loss)
        ops.emb_scatter_add_bwd(d_ic, d_pos, ic, dev(P), dev(U), *args, g[0], g[1], g[2], g[3], part,
                                norm=norm if ride else None)
        if not ride:
            ops.sqnorm_state_loss(gd, n_g, npart, 2, lr, state, l2, l2.numel(), ce, B, 1e-4, 1.0 / B, loss)
        out.append((g, part, npart, state, loss))
    (g_a, part_a, np_a, st_a, loss_a), (g_b, part_b, np_b, st_b, loss_b) = out
    for x, y in zip(g_a, g_b):
        assert rel_err(y.cpu().numpy(), x.cpu().numpy()) < 1e-5          # float atomics: order-dependent rounding
    assert abs(float(part_a.double().sum()) - float(part_b.double().sum())) < 1e-5 * float(part_a.double().sum())
    assert torch.equal(np_a, np_b) and float(np_b[0]) == 3.0 and float(np_b[2 + nb]) == 3.0
    assert abs(float(np_b[2:2 + nb].double().sum()) - float((gd.double() ** 2).sum())) < 1e-5 * float((gd.double() ** 2).sum())
    assert torch.equal(st_a, st_b) and float(st_b[4]) != 0.81
    assert float((loss_a[:3] - loss_b[:3]).abs().max()) <= 1e-6 * float(loss_a[:3].abs().max())


def test_rows_gather_range(ops):
    """mtam_rows_gather_range: the rows of a ROW RANGE by catalog row number, zeros for rows outside it -- summed over
    the ranges of a partition every id gets exactly its row (what the "sharded-table" reduce-scatter relies on)."""
    rng = np.random.default_rng(9)
    V, n = 1003, 777
    E = dev(rng.standard_normal((V, D)).astype(np.float32))
    ids = dev(rng.integers(0, V, n).astype(np.int32))
    ids[0], ids[1] = 0, V - 1
    total = torch.zeros((n, D), device="cuda")
    for lo, hi in ((0, 126), (126, 630), (630, V)):
        out = torch.full((n, D), 5.0, device="cuda")
        ops.rows_gather_range(E[lo:hi], lo, ids, out)
        owned = (ids >= lo) & (ids < hi)
        assert torch.equal(out[owned], E[ids[owned].long()]) and not bool(out[~owned].any())
        total += out
    assert torch.equal(total, E[ids.long()])


def test_emb_scatter_item_row_ranges(ops):
    """mtam_emb_scatter_add_bwd_range (data-parallel row-sharded scoring): with the item rows cut into ranges, the
    ranges' scatter-adds -- the owner's full call restricted to its rows, and the item-only form that applies ANOTHER
    rank's slots -- add up to the one-piece scatter-add, item table and the other three alike."""
    rng = np.random.default_rng(3)
    B, L, V, C, U = 37, 11, 301, 13, 29
    R = B * L
    f = lambda *s: dev(rng.standard_normal(s).astype(np.float32))
    sl = rng.integers(1, L + 1, B)
    live = np.arange(L)[None, :] < sl[:, None]
    ids = {k: dev((rng.integers(0, n, (B, L)) * live).astype(np.int32)) for k, n in (("item", V), ("cat", C), ("pos", L + 3))}
    ids["item"][0, 0] = V - 1
    user = dev(rng.integers(0, U, B).astype(np.int32))
    sl_d = dev(sl.astype(np.int32))
    d_ic, d_x, ic, pos, usr = f(R, 2 * D), f(R, D), f(R, 2 * D), f(R, D), f(B, D)
    d_ic[~dev(live.reshape(-1))] = 0.0                       # padded slots carry a zero upstream gradient
    g = lambda: [torch.zeros((n, D), device="cuda") for n in (V, C, L + 3, U)]
    part = lambda: torch.zeros(ops.emb_scatter_partials(B, L), device="cuda")
    whole = g()
    ops.emb_scatter_add_bwd(d_ic, d_x, ic, pos, usr, ids["item"], ids["cat"], ids["pos"], user, sl_d, B, L, 5e-3, 1,
                            *whole, part())
    cuts = (0, 104, 208, V)
    own = g()                                               # "rank 0": every table, its own item rows
    ops.emb_scatter_add_bwd(d_ic, d_x, ic, pos, usr, ids["item"], ids["cat"], ids["pos"], user, sl_d, B, L, 5e-3, 1,
                            *own, part(), item_range=(cuts[0], cuts[1]))
    assert not bool(own[0][cuts[1]:].any())
    for t in (1, 2, 3):
        assert float((own[t] - whole[t]).abs().max()) <= 1e-5 * float(whole[t].abs().max())
    for lo, hi in zip(cuts[1:-1], cuts[2:]):                # "ranks 1, 2": the same slots, item rows only
        before = own[0].clone()
        ops.emb_scatter_add_items_range(d_ic, ic, ids["item"].reshape(-1), sl_d, B, L, 5e-3, own[0], part(), (lo, hi))
        changed = (own[0] != before).any(1)
        assert not bool(changed[:lo].any()) and not bool(changed[hi:].any())
    assert float((own[0] - whole[0]).abs().max()) <= 1e-5 * float(whole[0].abs().max())


# ------------------------------------------------------------------- GRU
# Variable names come from the oracle's own tables (oracle/family.py, oracle/specs.py), not from the product's;
# the ROW ORDER of the packed kernel operands is the C ABI's (include/mtam_hip.h: tvec [8, D], tparams [5, L]).
from oracle.family import CELL_SCOPE, TIME_GATE_VARS as TIME_GATE  # noqa: E402
from oracle import specs as ORACLE_SPECS  # noqa: E402
GRU_SCOPE = CELL_SCOPE["decay_new"]
GRU_USED = ("_time_kernel_w1", "_time_kernel_b1", "_time_history_w1", "_time_w1", "_time_b1", "_time_kernel_w2",
            "_time_w12", "_time_b12")            # mtam_tagru_fwd's tvec rows, include/mtam_hip.h
assert sorted(GRU_USED) == sorted(n for n, _ in ORACLE_SPECS.DECAY_NEW_LIVE)


def _gru_weights(rng):
    w = {GRU_SCOPE + "gates/kernel": rng.uniform(-0.1, 0.1, (2 * D, 2 * D)),
         GRU_SCOPE + "gates/bias": rng.uniform(0.5, 1.5, 2 * D),
         GRU_SCOPE + "candidate/kernel": rng.uniform(-0.1, 0.1, (2 * D, D)),
         GRU_SCOPE + "candidate/bias": rng.uniform(-0.2, 0.2, D)}
    for n in GRU_USED:
        w[GRU_SCOPE + n] = rng.uniform(-0.15, 0.15, D)
    w[GRU_SCOPE + "_time_w1"] = rng.uniform(-0.01, 0.01, D)     # times are raw hours
    return {k: v.astype(np.float32) for k, v in w.items()}


def _gru_device_args(w):
    Wg, Wc = w[GRU_SCOPE + "gates/kernel"], w[GRU_SCOPE + "candidate/kernel"]
    Wx = np.concatenate([Wg[:D], Wc[:D]], axis=1)                 # [D, 3D]
    bx = np.concatenate([w[GRU_SCOPE + "gates/bias"], w[GRU_SCOPE + "candidate/bias"]])
    tvec = np.stack([w[GRU_SCOPE + n] for n in GRU_USED])
    return Wx, bx, np.ascontiguousarray(Wg[D:]), np.ascontiguousarray(Wc[D:]), tvec


@pytest.mark.parametrize("B,L,n_kv", [(3, 5, 256), (128, 50, 256), (16, 50, 512), (7, 9, 32)])
def test_gru_launches_carry_the_kv_work(ops, B, L, n_kv):
    """mtam_tagru_fwd_kv / mtam_tagru_bwd_dkv: the decoder's K/V projection relu(x Wkv + bkv) and its gradient
    towards x (d_x += d_kv Wkv^T) ride in the GRU's own launches as extra workgroups.  The recurrence's outputs are
    BIT-IDENTICAL to the launches without the riders (the same code runs on the same inputs), the riders' results
    match float64 to the tolerance of the GEMMs they replace (2e-5) and the split-bf16 GEMM to fp32 rounding."""
    rng = np.random.default_rng(B * 5 + L + n_kv)
    w = _gru_weights(rng)
    R = B * L
    x = rng.uniform(-0.5, 0.5, (R, D)).astype(np.float32)
    tl = np.floor(rng.exponential(24.0, R)).astype(np.float32)
    sl = rng.integers(2, L + 1, size=B).astype(np.int32)
    Wx, bx, whg, whc, tvec = _gru_device_args(w)
    xproj = dev((x.astype(np.float64) @ Wx + bx).astype(np.float32))
    xd, tld, sld, whg, whc, tvec = dev(x), dev(tl), dev(sl), dev(whg), dev(whc), dev(tvec)
    Wkv = dev((rng.standard_normal((D, n_kv)) * 0.1).astype(np.float32))
    bkv = dev(rng.standard_normal(n_kv).astype(np.float32))
    img, img_t = (torch.zeros(3 * D * n_kv, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    ops.split_weight_images(Wkv, img)
    ops.split_weight_rows(Wkv, img_t)
    z = lambda *shape: torch.full(shape, 5.0, device="cuda")
    plain = [z(R, D), z(B, D), z(R, 5 * D)]
    ops.tagru_fwd(xproj, xd, tld, sld, whg, whc, tvec, B, L, *plain)
    ridden, kv = [z(R, D), z(B, D), z(R, 5 * D)], z(R, n_kv)
    ops.tagru_fwd(xproj, xd, tld, sld, whg, whc, tvec, B, L, *ridden, kv=(img, bkv, kv))
    for a, b in zip(plain, ridden):
        assert torch.equal(a, b)
    # ... and with the recurrent weights taken from their register-order image instead of through LDS: the same bits
    w_image = torch.zeros(ops.gru_weight_image_floats(), device="cuda")
    ops.gru_weight_image(whg, whc, w_image)
    direct = [z(R, D), z(B, D), z(R, 5 * D)]
    ops.tagru_fwd(xproj, xd, tld, sld, whg, whc, tvec, B, L, *direct, w_image=w_image)
    for a, b in zip(plain, direct):
        assert torch.equal(a, b)
    ref = np.maximum(x.astype(np.float64) @ Wkv.double().cpu().numpy() + bkv.double().cpu().numpy(), 0.0)
    assert rel_err(kv.cpu().numpy(), ref) < 2e-5
    want = z(R, n_kv)
    ops.gemm(xd, Wkv, want, epilogue=ops.EPI_BIAS_RELU, bias=bkv)
    assert rel_err(kv.cpu().numpy(), want.cpu().numpy()) < 2e-6
    pre = x.astype(np.float64) @ Wkv.double().cpu().numpy() + bkv.double().cpu().numpy()
    clear = np.abs(pre) > 1e-4          # (the relu mask is exact where the pre-activation is not within rounding of zero)
    assert np.array_equal((kv.cpu().numpy() > 0)[clear], (pre > 0)[clear])
    if n_kv != 256:
        with pytest.raises(RuntimeError):           # the backward rider is built for one decoder block
            ops.tagru_bwd(dev(x[:B]), xd, tld, sld, whg, whc, tvec, plain[2], B, L, z(R, 3 * D), z(R, D), z(R, D),
                          torch.zeros((B, 8, D), device="cuda"), dkv=(z(R, n_kv), img_t, z(R, D)))
        return
    d_short = dev(rng.standard_normal((B, D)).astype(np.float32))
    d_kv = dev(rng.standard_normal((R, n_kv)).astype(np.float32))
    d_x0 = rng.standard_normal((R, D)).astype(np.float32)
    outs_a = [z(R, 3 * D), z(R, D), z(R, D), torch.zeros((B, 8, D), device="cuda")]
    ops.tagru_bwd(d_short, xd, tld, sld, whg, whc, tvec, plain[2], B, L, *outs_a)
    outs_b, d_x = [z(R, 3 * D), z(R, D), z(R, D), torch.zeros((B, 8, D), device="cuda")], dev(d_x0)
    ops.tagru_bwd(d_short, xd, tld, sld, whg, whc, tvec, plain[2], B, L, *outs_b, dkv=(d_kv, img_t, d_x))
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    ref = d_x0.astype(np.float64) + d_kv.double().cpu().numpy() @ Wkv.double().cpu().numpy().T
    assert rel_err(d_x.cpu().numpy(), ref) < 2e-5
    want = dev(d_x0)
    ops.gemm(d_kv, Wkv, want, trans_b=True, epilogue=ops.EPI_ACCUM)
    assert rel_err(d_x.cpu().numpy(), want.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("B,L", [(3, 5), (16, 50), (128, 50)])
def test_tagru_fwd_bwd(ops, B, L):
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(B + 17 * L)
    w = _gru_weights(rng)
    x = rng.uniform(-0.5, 0.5, (B, L, D)).astype(np.float32)
    tl = np.floor(rng.exponential(24.0, (B, L))).astype(np.float32)
    sl = rng.integers(2, L + 1, size=B).astype(np.int32)
    sl[0] = 2
    sl[-1] = L
    Wx, bx, whg, whc, tvec = _gru_device_args(w)
    xproj = (x.reshape(-1, D).astype(np.float64) @ Wx + bx).astype(np.float32)
    R = B * L
    hs = torch.full((R, D), 5.0, device="cuda")
    short = torch.zeros((B, D), device="cuda")
    save = torch.zeros((R, 5 * D), device="cuda")
    xd, tld, sld = dev(x.reshape(R, D)), dev(tl.reshape(R)), dev(sl)
    ops.tagru_fwd(dev(xproj), xd, tld, sld, dev(whg), dev(whc), dev(tvec), B, L, hs, short, save)

    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref_hs = O.time_aware_gru(wt, xt, torch.tensor(tl, dtype=torch.float64), torch.tensor(sl).long() - 1)
    ref_short = O.gather_indexes(ref_hs, torch.tensor(sl).long() - 2)
    assert rel_err(hs.cpu().numpy().reshape(B, L, D), ref_hs.detach().numpy()) < 2e-5
    assert rel_err(short.cpu().numpy(), ref_short.detach().numpy()) < 2e-5

    d_short = rng.standard_normal((B, D)).astype(np.float32)
    (ref_short * torch.tensor(d_short, dtype=torch.float64)).sum().backward()
    d_xproj = torch.full((R, 3 * D), 3.0, device="cuda")
    rh = torch.full((R, D), 3.0, device="cuda")
    d_xt = torch.full((R, D), 3.0, device="cuda")
    d_tv = torch.zeros((B, 8, D), device="cuda")
    ops.tagru_bwd(dev(d_short), xd, tld, sld, dev(whg), dev(whc), dev(tvec), save, B, L, d_xproj, rh, d_xt, d_tv)
    # gradients of the parameters the kernel owns directly
    got_tv = d_tv.cpu().numpy().astype(np.float64).sum(0)
    for i, n in enumerate(GRU_USED):
        assert rel_err(got_tv[i], wt[GRU_SCOPE + n].grad.numpy()) < 1e-4, n
    # the rest goes through the pre-activation gradients: check them via the weight gradients
    dxp = d_xproj.cpu().numpy().astype(np.float64)
    sv = save.cpu().numpy().astype(np.float64)
    hprev = sv[:, 4 * D:5 * D]
    live = (np.arange(L)[None, :] < (sl[:, None] - 1)).reshape(R)
    assert np.all(dxp[~live] == 0) and np.all(rh.cpu().numpy()[~live] == 0) and np.all(d_xt.cpu().numpy()[~live] == 0)
    x2 = x.reshape(R, D).astype(np.float64)
    gWg = np.concatenate([x2[live].T @ dxp[live, :2 * D], hprev[live].T @ dxp[live, :2 * D]])
    gWc = np.concatenate([x2[live].T @ dxp[live, 2 * D:], rh.cpu().numpy().astype(np.float64)[live].T @ dxp[live, 2 * D:]])
    assert rel_err(gWg, wt[GRU_SCOPE + "gates/kernel"].grad.numpy()) < 1e-4
    assert rel_err(gWc, wt[GRU_SCOPE + "candidate/kernel"].grad.numpy()) < 1e-4
    assert rel_err(dxp.sum(0)[:2 * D], wt[GRU_SCOPE + "gates/bias"].grad.numpy()) < 1e-4
    # d_x: time-gate path (d_xt) + x-projection path
    got_dx = d_xt.cpu().numpy().astype(np.float64) + dxp @ Wx.T.astype(np.float64)
    assert rel_err(got_dx, xt.grad.numpy().reshape(R, D)) < 1e-4


# -------------------------------------------------------------- attention
def _attn_case(rng, B, L, H):
    scope, inner = "blk/", "vanilla_attention"
    w = {}
    for layer in ("dense", "dense_1", "dense_2"):
        w[scope + layer + "/kernel"] = rng.uniform(-0.15, 0.15, (D, D))
        w[scope + layer + "/bias"] = rng.uniform(-0.1, 0.1, D)
    s = scope + inner + "/"
    w[s + "_time_input_w"] = rng.uniform(-0.15, 0.15, (D, D))
    for n in TIME_GATE:
        w[s + n] = rng.uniform(-0.3, 0.3, (1, L))
    w[s + "ln/Variable"] = rng.uniform(-0.2, 0.2, D)
    w[s + "ln/Variable_1"] = rng.uniform(0.8, 1.2, D)
    w = {k: v.astype(np.float32) for k, v in w.items()}
    q = rng.uniform(-1, 1, (B, 1, D)).astype(np.float32)
    x = rng.uniform(-0.5, 0.5, (B, L, D)).astype(np.float32)
    sl = rng.integers(2, L + 1, size=B).astype(np.int32)
    sl[0] = 2
    sl[-1] = L
    tk = np.cumsum(np.floor(rng.exponential(24.0, (B, L))), axis=1).astype(np.float32) + 250000
    tq = (tk[np.arange(B), sl - 1] + 0).astype(np.float32)       # mask slot carries the target time
    return scope, inner, w, q, x, sl, tq, tk


@pytest.mark.parametrize("B,L,H", [(4, 6, 1), (9, 50, 2), (128, 50, 1), (8, 200, 8)])
def test_ta_attn_decode_fwd_bwd(ops, B, L, H):
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(B * 3 + L + H)
    scope, inner, w, q, x, sl, tq, tk = _attn_case(rng, B, L, H)
    s = scope + inner + "/"
    R = B * L
    wkv = np.concatenate([w[scope + "dense_1/kernel"], w[scope + "dense_2/kernel"]], axis=1)
    bkv = np.concatenate([w[scope + "dense_1/bias"], w[scope + "dense_2/bias"]])
    kv = np.maximum(x.reshape(R, D).astype(np.float64) @ wkv + bkv, 0).astype(np.float32)
    wqt = np.concatenate([w[scope + "dense/kernel"], w[s + "_time_input_w"]], axis=1)
    tparams = np.concatenate([w[s + n] for n in TIME_GATE], axis=0)           # [5, L]
    nsave = ops.ta_attn_decode_save_floats(L, H)
    out = torch.zeros((B, D), device="cuda")
    save = torch.zeros((B, nsave), device="cuda")
    args = dict(x=dev(x.reshape(R, D)), kv=dev(kv), tq=dev(tq), tk=dev(tk.reshape(R)), sl=dev(sl),
                wqt=dev(wqt), tp=dev(tparams), g=dev(w[s + "ln/Variable_1"]))
    ops.ta_attn_decode_fwd(dev(q.reshape(B, D)), args["x"], args["kv"], 2 * D, 0, D, args["tq"], args["tk"],
                           args["sl"], args["wqt"], dev(w[scope + "dense/bias"]), args["tp"],
                           dev(w[s + "ln/Variable"]), args["g"], B, L, H, out, save)

    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref, _ = O.time_aware_multihead_attention(
        wt, scope, inner, qt, xt, torch.tensor(sl).long(), torch.ones(B).long(),
        torch.tensor(tq, dtype=torch.float64).unsqueeze(1), torch.tensor(tk, dtype=torch.float64), H)
    assert rel_err(out.cpu().numpy(), ref.detach().numpy().reshape(B, D)) < 2e-5

    d_out = rng.standard_normal((B, D)).astype(np.float32)
    (ref.reshape(B, D) * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
    d_dec = torch.zeros((B, D), device="cuda")
    d_kv = torch.full((R, 2 * D), 4.0, device="cuda")
    d_x = torch.full((R, D), 4.0, device="cuda")
    d_qt = torch.zeros((B, 2 * D), device="cuda")
    d_tp = torch.zeros((B, 5, L), device="cuda")
    d_ln = torch.zeros((B, 2, D), device="cuda")
    ops.ta_attn_decode_bwd(dev(d_out), dev(q.reshape(B, D)), args["x"], args["kv"], 2 * D, 0, D, args["tq"],
                           args["tk"], args["sl"], args["wqt"], args["tp"], args["g"], save, B, L, H, 0,
                           d_dec, d_kv, d_x, d_qt, d_tp, d_ln)
    tol = 2e-4
    x2 = x.reshape(R, D).astype(np.float64)
    dkv = d_kv.cpu().numpy().astype(np.float64)
    dqt = d_qt.cpu().numpy().astype(np.float64)
    q2 = q.reshape(B, D).astype(np.float64)
    assert rel_err(x2.T @ dkv[:, :D], wt[scope + "dense_1/kernel"].grad.numpy()) < tol
    assert rel_err(x2.T @ dkv[:, D:], wt[scope + "dense_2/kernel"].grad.numpy()) < tol
    assert rel_err(dkv.sum(0)[:D], wt[scope + "dense_1/bias"].grad.numpy()) < tol
    assert rel_err(q2.T @ dqt[:, :D], wt[scope + "dense/kernel"].grad.numpy()) < tol
    assert rel_err(q2.T @ dqt[:, D:], wt[s + "_time_input_w"].grad.numpy()) < tol
    assert rel_err(dqt.sum(0)[:D], wt[scope + "dense/bias"].grad.numpy()) < tol
    got_tp = d_tp.cpu().numpy().astype(np.float64).sum(0)
    for i, n in enumerate(TIME_GATE):
        assert rel_err(got_tp[i], wt[s + n].grad.numpy()[0]) < tol, n
    got_ln = d_ln.cpu().numpy().astype(np.float64).sum(0)
    assert rel_err(got_ln[0], wt[s + "ln/Variable"].grad.numpy()) < tol
    assert rel_err(got_ln[1], wt[s + "ln/Variable_1"].grad.numpy()) < tol
    assert rel_err(d_dec.cpu().numpy(), qt.grad.numpy().reshape(B, D)) < tol
    got_dx = d_x.cpu().numpy().astype(np.float64) + dkv @ wkv.T.astype(np.float64)
    assert rel_err(got_dx, xt.grad.numpy().reshape(R, D)) < tol
    # padded keys carry exactly zero gradient (the scatter kernel relies on it)
    pad = (np.arange(L)[None, :] >= sl[:, None]).reshape(R)
    assert np.all(dkv[pad] == 0) and np.all(d_x.cpu().numpy()[pad] == 0)
    # accumulate_dx adds on top
    d_x2 = torch.full((R, D), 1.5, device="cuda")
    ops.ta_attn_decode_bwd(dev(d_out), dev(q.reshape(B, D)), args["x"], args["kv"], 2 * D, 0, D, args["tq"],
                           args["tk"], args["sl"], args["wqt"], args["tp"], args["g"], save, B, L, H, 1,
                           d_dec, d_kv, d_x2, d_qt, d_tp, d_ln)
    assert np.allclose(d_x2.cpu().numpy(), d_x.cpu().numpy() + 1.5, atol=1e-6)


# ------------------------------------------------------------ layer norm
def test_normalize_with_residual(ops):
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(10)
    rows = 41
    x = rng.standard_normal((rows, D)).astype(np.float32)
    res = rng.standard_normal((rows, D)).astype(np.float32)
    beta = rng.standard_normal(D).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, D).astype(np.float32)
    y = torch.zeros((rows, D), device="cuda")
    save = torch.zeros((rows, D + 1), device="cuda")
    ops.layer_norm_fwd(dev(x), dev(beta), dev(gamma), 1e-8, rows, y, save, resid=dev(res), form=1)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = O.normalize(xt + torch.tensor(res, dtype=torch.float64), torch.tensor(beta, dtype=torch.float64),
                      torch.tensor(gamma, dtype=torch.float64))
    assert rel_err(y.cpu().numpy(), ref.detach().numpy()) < 2e-6
    dy = rng.standard_normal((rows, D)).astype(np.float32)
    (ref * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    dx = torch.zeros((rows, D), device="cuda")
    dbg = torch.zeros((2, D), device="cuda")
    ops.layer_norm_bwd(dev(dy), dev(gamma), save, rows, dx, dbg)
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < 1e-5


def test_seq_row_gather_scatter_and_relu_bwd(ops):
    rng = np.random.default_rng(12)
    B, L = 7, 9
    src = rng.standard_normal((B * L, D)).astype(np.float32)
    sl = rng.integers(2, L + 1, size=B).astype(np.int32)
    out = torch.zeros((B, D), device="cuda")
    ops.seq_row_gather(dev(src), dev(sl), -1, B, L, out)
    assert np.array_equal(out.cpu().numpy(), src.reshape(B, L, D)[np.arange(B), sl - 1])
    d_src = torch.full((B * L, D), 5.0, device="cuda")
    ops.seq_row_scatter(out, dev(sl), -1, B, L, d_src)
    want = np.zeros((B, L, D), np.float32)
    want[np.arange(B), sl - 1] = out.cpu().numpy()
    assert np.array_equal(d_src.cpu().numpy().reshape(B, L, D), want)
    dd = rng.standard_normal((B * L, D)).astype(np.float32)
    yy = rng.standard_normal((B * L, D)).astype(np.float32)
    dt = dev(dd).clone()
    ops.relu_bwd_inplace(dt, dev(yy), dt.numel())
    assert np.array_equal(dt.cpu().numpy(), np.where(yy > 0, dd, 0))


def test_layer_norm(ops):
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(9)
    rows = 37
    x = rng.standard_normal((rows, D)).astype(np.float32)
    beta = rng.standard_normal(D).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, D).astype(np.float32)
    y = torch.zeros((rows, D), device="cuda")
    save = torch.zeros((rows, D + 1), device="cuda")
    ops.layer_norm_fwd(dev(x), dev(beta), dev(gamma), 1e-12, rows, y, save)   # contrib form, no residual
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    ref = O.layer_norm(xt, bt, gt)
    assert rel_err(y.cpu().numpy(), ref.detach().numpy()) < 2e-6
    dy = rng.standard_normal((rows, D)).astype(np.float32)
    (ref * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    dx = torch.zeros((rows, D), device="cuda")
    dbg = torch.zeros((2, D), device="cuda")
    ops.layer_norm_bwd(dev(dy), dev(gamma), save, rows, dx, dbg)
    assert rel_err(dx.cpu().numpy(), xt.grad.numpy()) < 1e-5
    assert rel_err(dbg.cpu().numpy()[0], bt.grad.numpy()) < 1e-5
    assert rel_err(dbg.cpu().numpy()[1], gt.grad.numpy()) < 1e-5


# ------------------------------------------------------------ softmax CE
@pytest.mark.parametrize("B,V", [(3, 10), (128, 3709), (5, 20000)])
def test_softmax_ce(ops, B, V):
    rng = np.random.default_rng(V)
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    target = rng.integers(0, V, size=B).astype(np.int32)
    lse = torch.zeros(B, device="cuda")
    ce = torch.zeros(B, device="cuda")
    lg = dev(logits)
    part = torch.zeros(ops.softmax_ce_partials(B, V), device="cuda")
    scale = 1.0 / B
    ops.softmax_ce(lg, V, dev(target), B, V, scale, lse, ce, lg, part)         # gradient in place
    l64 = logits.astype(np.float64)
    m = l64.max(1, keepdims=True)
    ref_lse = (m[:, 0] + np.log(np.exp(l64 - m).sum(1)))
    assert rel_err(lse.cpu().numpy(), ref_lse) < 1e-6
    assert rel_err(ce.cpu().numpy(), ref_lse - l64[np.arange(B), target]) < 1e-6
    g = np.exp(l64 - ref_lse[:, None])
    g[np.arange(B), target] -= 1
    assert np.abs(lg.cpu().numpy() - g * scale).max() < 1e-6 * scale * 10
    l2p = dev(rng.uniform(0, 1, 77).astype(np.float32))
    loss = torch.zeros(3, device="cuda")
    ops.loss_reduce(l2p, 77, ce, B, 5e-5, 1.0 / B, loss)
    l2 = 0.5 * float(l2p.double().sum())
    ces = float(ce.double().sum())
    assert np.allclose(loss.cpu().numpy(), [5e-5 * l2 + ces / B, l2, ces / B], rtol=1e-5)


@pytest.mark.parametrize("B,V", [(3, 10), (128, 3709), (7, 16384), (5, 20000)])
def test_softmax_ce_loss_fused(ops, B, V):
    """One-launch form (V <= 16384) and the multi-launch fallback give the same numbers, call after call
    (the arrival ticket resets itself)."""
    rng = np.random.default_rng(V + B)
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    target = rng.integers(0, V, size=B).astype(np.int32)
    l2p = rng.uniform(0, 1, 77).astype(np.float32)
    l64 = logits.astype(np.float64)
    m = l64.max(1, keepdims=True)
    ref_lse = m[:, 0] + np.log(np.exp(l64 - m).sum(1))
    ref_ce = ref_lse - l64[np.arange(B), target]
    g = np.exp(l64 - ref_lse[:, None])
    g[np.arange(B), target] -= 1
    part = torch.zeros(ops.softmax_ce_partials(B, V) + 4, device="cuda")
    lse, ce, loss = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(3, device="cuda")
    for _ in range(3):
        lg = dev(logits)
        ops.softmax_ce_loss(lg, V, dev(target), B, V, 1.0 / B, lse, ce, lg, part, dev(l2p), 77, 5e-5, 1.0 / B, loss)
        assert rel_err(lse.cpu().numpy(), ref_lse) < 1e-6
        assert rel_err(ce.cpu().numpy(), ref_ce) < 1e-6
        assert np.abs(lg.cpu().numpy() - g / B).max() < 1e-5 / B
        l2 = 0.5 * float(l2p.astype(np.float64).sum())
        assert np.allclose(loss.cpu().numpy(), [5e-5 * l2 + ref_ce.mean(), l2, ref_ce.mean()], rtol=1e-5)


# ------------------------------------------------------------------ top-K
@pytest.mark.parametrize("rows,V,k", [(4, 10, 50), (7, 3709, 50), (3, 70000, 50), (5, 300, 1), (2, 64, 64),
                                      (4, 1000003, 50)])
def test_topk_bit_exact(ops, rows, V, k):
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(rows + V + k)
    scores = rng.standard_normal((rows, V)).astype(np.float32)
    scores[0, : min(V, 40)] = 1.25           # a run of exact ties inside the top-k
    if rows > 1:
        scores[1] = np.round(scores[1] * 2) / 2          # heavy ties everywhere
    if rows > 2:
        scores[2, ::3] = -0.0
        scores[2, 1::3] = 0.0
    if rows > 3:
        scores[3] = 0.75                                  # a constant row: more ties than the tie list holds
    idx = torch.zeros((rows, k), dtype=torch.int32, device="cuda")
    val = torch.zeros((rows, k), device="cuda")
    ops.topk(dev(scores), V, rows, V, k, idx, val)
    kk = min(k, V)
    ref = O.top_k(scores, kk)
    got = idx.cpu().numpy()
    assert np.array_equal(got[:, :kk], ref)
    assert np.all(got[:, kk:] == -1)
    assert np.array_equal(val.cpu().numpy()[:, :kk] + 0.0, np.take_along_axis(scores, ref, 1) + 0.0)


@pytest.mark.parametrize("rows,V,k", [(5, 1000003, 50), (4, 262144, 20), (3, 2031617, 64)])
def test_topk_two_level_matches_single_pass(ops, rows, V, k):
    """Long rows with a workspace: segments -> candidates -> final k.  Same lists as the oracle's stable
    sort, including ties that straddle segment boundaries, a constant row (all ties: indices 0..k-1) and a
    last segment shorter than k (V = 31 segments x 65,536 + 1)."""
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(V + k)
    scores = rng.standard_normal((rows, V)).astype(np.float32)
    scores[0, ::1000] = 9.5                                # equal maxima spread over every segment
    scores[1] = np.round(scores[1] * 2) / 2                # heavy ties everywhere
    scores[2] = 0.75                                       # a constant row
    if rows > 3:
        scores[3, -3:] = [40.0, 41.0, 40.0]                # winners at the very end of the row
    nbytes = ops.topk_workspace_bytes(rows, V, k)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, device="cuda")
    idx = torch.zeros((rows, k), dtype=torch.int32, device="cuda")
    val = torch.zeros((rows, k), device="cuda")
    ld = (V + 3) // 4 * 4
    padded = np.full((rows, ld), 99.0, np.float32)         # the pad columns must not be read
    padded[:, :V] = scores
    ops.topk(dev(padded), ld, rows, V, k, idx, val, workspace=ws)
    ref = O.top_k(scores, k)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(val.cpu().numpy() + 0.0, np.take_along_axis(scores, ref, 1) + 0.0)
    assert ops.topk_workspace_bytes(rows, 3709, k) == 0


# ------------------------------------------------------------ clip + Adam
def test_clip_and_adam(ops):
    rng = np.random.default_rng(1)
    n = 4096 * 3 + 5
    g = rng.standard_normal(n).astype(np.float32)
    p0 = rng.standard_normal(n).astype(np.float32)
    m0 = (rng.standard_normal(n) * 0.1).astype(np.float32)
    v0 = (rng.uniform(0, 0.1, n)).astype(np.float32)
    nb = ops.sqnorm_blocks(n)
    assert nb == 4
    part = torch.zeros(nb + 2, device="cuda")
    part[nb] = 2.5
    part[nb + 1] = 0.5
    ops.sqnorm_partial(dev(g), n, part)
    total = float((g.astype(np.float64) ** 2).sum()) + 3.0
    scale = torch.zeros(2, device="cuda")
    ops.clip_scale(part, nb + 2, 1.0, scale)
    norm = np.sqrt(total)
    assert abs(float(scale[1]) - norm) / norm < 1e-6
    assert abs(float(scale[0]) - 1.0 * min(1 / norm, 1.0)) < 1e-7
    ops.clip_scale(part, nb + 2, 1e6, scale)                 # norm below the clip: scale == 1
    assert float(scale[0]) == 1.0
    # Adam bookkeeping on the device: lr_t from the current beta powers, then the powers advance
    state = dev(np.array([0, 0.9, 0.999, 1e-8, 0.9, 0.999, 0, 0], np.float32))
    lr = dev(np.array([1e-3, 0, 0, 0], np.float32))
    b1p, b2p = np.float32(0.9), np.float32(0.999)
    for _ in range(3):
        ops.clip_scale(part, nb + 2, 1.0, scale, lr, state)
        st = state.cpu().numpy()
        want = np.float32(1e-3) * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p)
        assert abs(st[0] - want) <= 1e-6 * want
        b1p, b2p = np.float32(b1p * np.float32(0.9)), np.float32(b2p * np.float32(0.999))
        assert st[4] == b1p and st[5] == b2p
    # fused sqnorm + clip + Adam bookkeeping: same result as the two separate launches, repeatably
    part2 = torch.zeros(nb + 2, device="cuda")
    part2[nb] = 2.5
    part2[nb + 1] = 0.5
    scale2 = torch.zeros(2, device="cuda")
    ticket = torch.zeros(4, dtype=torch.int32, device="cuda")
    state2 = dev(np.array([0, 0.9, 0.999, 1e-8, 0.9, 0.999, 0, 0], np.float32))
    for rep in range(3):
        ops.sqnorm_clip_scale(dev(g), n, part2, 0, nb + 2, 1.0, scale2, lr, state2, ticket)
        assert abs(float(scale2[1]) - norm) / norm < 1e-6 and int(ticket[0]) == 0
    assert np.allclose(state2.cpu().numpy(), state.cpu().numpy(), rtol=1e-6)
    sc = np.float32(scale[0].item())
    lr_t, b1, b2, eps = np.float32(1e-3), np.float32(0.9), np.float32(0.999), np.float32(1e-8)
    hyper = dev(np.array([lr_t, b1, b2, eps], np.float32))
    blk = ops.adam_block()
    for sparse_begin in (n, 0, blk):
        p, m, v = dev(p0).clone(), dev(m0).clone(), dev(v0).clone()
        ops.adam(p, m, v, dev(g), n, scale, hyper, sparse_begin)
        gs = g * sc
        sparse = np.arange(n) >= sparse_begin
        mr = np.where(sparse, m0 * b1 + gs * (np.float32(1) - b1), m0 + (gs - m0) * (np.float32(1) - b1))
        vr = np.where(sparse, v0 * b2 + (gs * gs) * (np.float32(1) - b2), v0 + (gs * gs - v0) * (np.float32(1) - b2))
        pr = p0 - (lr_t * mr) / (np.sqrt(vr) + eps)
        assert np.allclose(m.cpu().numpy(), mr, rtol=1e-6, atol=1e-9)
        assert np.allclose(v.cpu().numpy(), vr, rtol=1e-6, atol=1e-9)
        assert np.allclose(p.cpu().numpy(), pr, rtol=1e-6, atol=1e-7)


def test_grouped_weight_gradient_gemm_and_colsum(ops):
    rng = np.random.default_rng(8)
    shapes = [(1850, 128, 384, 7), (1850, 256, 128, 3), (37, 128, 256, 1), (700, 128, 128, 16)]   # K, M, N, split
    probs, refs, outs = [], [], []
    for K, M, N, split in shapes:
        A = rng.standard_normal((K, M + 5)).astype(np.float32)         # lda > M
        Bm = rng.standard_normal((K, N)).astype(np.float32)
        c0 = rng.standard_normal((M, N)).astype(np.float32)
        C = dev(c0).clone()
        probs.append(dict(A=dev(A), lda=M + 5, B=dev(Bm), ldb=N, C=C, ldc=N, M=M, N=N, K=K, split_k=split))
        refs.append(c0 + A[:, :M].T.astype(np.float64) @ Bm.astype(np.float64))
        outs.append(C)
    ops.gemm_tn_atomic_grouped(probs)
    for C, ref in zip(outs, refs):
        assert rel_err(C.cpu().numpy(), ref) < 5e-6
    jobs, wants = [], []
    for rows, cols, ld in [(777, 300, 300), (128, 128, 256), (5, 70, 70)]:
        x = rng.standard_normal((rows, ld)).astype(np.float32)
        out = torch.ones(cols, device="cuda")
        jobs.append((dev(x), rows, cols, ld, out))
        wants.append(1.0 + x[:, :cols].astype(np.float64).sum(0))
    ops.colsum_atomic_multi(jobs)
    for (_, _, _, _, out), want in zip(jobs, wants):
        assert rel_err(out.cpu().numpy(), want) < 1e-5


# ------------------------------------------------- sgd / adadelta / rmsprop
@pytest.mark.parametrize("kind", ["sgd", "adadelta", "rmsprop"])
def test_opt_update_matches_the_tf_formulas(ops, kind):
    """mtam_opt_update vs oracle.apply_slot_optimizer on one flat buffer laid out like the model's:
    [dense 4096 | row-sparse table 40 rows | dense-gradient table 24 rows]."""
    import oracle.mtam_oracle as O
    rng = np.random.default_rng(11)
    nd, r1, r2 = 4096, 40, 24
    arrays = {"w": rng.standard_normal(nd).astype(np.float32),
              "embedding_layer/category": rng.standard_normal((r1, D)).astype(np.float32),
              "embedding_layer/item": rng.standard_normal((r2, D)).astype(np.float32)}
    touched = np.array([0, 3, 4, 17, 39])
    state = O.SlotState(kind, arrays)
    flat = lambda d: np.concatenate([d["w"], d["embedding_layer/category"].ravel(), d["embedding_layer/item"].ravel()])
    p = dev(flat(arrays))
    s1, s2 = dev(flat(state.s1)), dev(flat(state.s2))
    scale = dev(np.array([0.37, 0.0], np.float32))
    lr = dev(np.array([0.05, 0, 0, 0], np.float32))
    for step in range(3):
        grads = {"w": rng.standard_normal(nd).astype(np.float32),
                 "embedding_layer/category": np.zeros((r1, D), np.float32),
                 "embedding_layer/item": rng.standard_normal((r2, D)).astype(np.float32)}
        grads["embedding_layer/category"][touched] = rng.standard_normal((len(touched), D)).astype(np.float32)
        O.apply_slot_optimizer(arrays, state, grads, np.float32(0.37), 0.05, {"category_list": touched})
        ops.opt_update(kind, p, s1, s2, dev(flat(grads)), p.numel(), scale, lr, nd, nd + r1 * D)
        touched = np.array([1, 3, 20])
    assert rel_err(p.cpu().numpy(), flat(arrays)) < 2e-6
    if kind != "sgd":
        assert rel_err(s1.cpu().numpy(), flat(state.s1)) < 2e-6
        assert rel_err(s2.cpu().numpy(), flat(state.s2)) < 2e-5


def test_gemm_store_sq_epilogue(ops):
    """STORE_SQ: C as STORE, plus per-wave partial sums of C^2 (ragged tile edges excluded)."""
    rng = np.random.default_rng(21)
    M, N, K = 3709, 128, 128                      # the dense item gradient: logits^T [V, B] x pred [B, D]
    a = rng.standard_normal((K, M)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    c = torch.zeros((M, N), device="cuda")
    n = ops.gemm_sq_partials(M, N)
    assert n == 4 * 58 * 2
    part = torch.full((n,), 7.0, device="cuda")
    ops.gemm(dev(a), dev(b), c, trans_a=True, epilogue=ops.EPI_STORE_SQ, aux_out=part)
    ref = a.astype(np.float64).T @ b.astype(np.float64)
    assert rel_err(c.cpu().numpy(), ref) < 1e-5
    assert abs(float(part.double().sum()) - float((ref ** 2).sum())) / float((ref ** 2).sum()) < 1e-5
    c2 = torch.zeros((M, N), device="cuda")
    ops.gemm(dev(a), dev(b), c2, trans_a=True, split=False)      # (STORE_SQ always takes the fp32 chain)
    assert torch.equal(c, c2)


def test_gemm_dual_source(ops):
    """C = A B^T + A2 B2^T with the ACCUM2_MASK epilogue (the d_x product of the MTAM backward)."""
    rng = np.random.default_rng(31)
    M, N, K, K2 = 333, 128, 384, 256
    a, b = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    a2, b2 = rng.standard_normal((M, K2)).astype(np.float32), rng.standard_normal((N, K2)).astype(np.float32)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    add2 = rng.standard_normal((M, N)).astype(np.float32)
    gate = rng.standard_normal((M, N)).astype(np.float32)
    c, out = dev(c0).clone(), torch.zeros((M, N), device="cuda")
    ops.gemm_dual(dev(a), dev(b), dev(a2), dev(b2), c, trans_b=True, epilogue=ops.EPI_ACCUM2_MASK,
                  bias=dev(add2), aux_in=dev(gate), aux_out=out)
    ref = c0.astype(np.float64) + add2 + a.astype(np.float64) @ b.T.astype(np.float64) \
        + a2.astype(np.float64) @ b2.T.astype(np.float64)
    assert rel_err(c.cpu().numpy(), ref) < 1e-5
    assert rel_err(out.cpu().numpy(), np.where(gate > 0, ref, 0.0)) < 1e-5


# ------------------------------------------------------------------ bf16 scoring without logits
def test_f32_to_bf16_is_round_to_nearest_even(ops):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(4096).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 4096).astype(np.float32)
    x[:4] = [0.0, -0.0, 1.00390625, 1.01171875]          # exact halves between two bf16 values: ties to even
    src = dev(x)
    dst = torch.full((4096 + 512,), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.f32_to_bf16(src, dst)
    assert torch.equal(dst[:4096].view(torch.int16), src.bfloat16().view(torch.int16))
    assert bool((dst[4096:] == 0).all())


def _score16_case(B, V, seed):
    rng = np.random.default_rng(seed)
    E = (rng.standard_normal((V, D)) * 0.2).astype(np.float32)
    P = rng.standard_normal((B, D)).astype(np.float32)
    target = rng.integers(0, V, B).astype(np.int32)
    target[0] = V - 1
    E16 = torch.empty((V, D), dtype=torch.bfloat16, device="cuda")
    return E, P, target, E16


@pytest.mark.parametrize("B,V", [(128, 3709), (100, 1000), (37, 63), (256, 70001), (130, 300007)])
def test_score16_lse_logits_and_backward(ops, B, V):
    """csrc/score16.hip against float64 products of the SAME bf16-rounded operands: the forward differs from
    them by fp32 summation order only (1e-5); the backward also rounds G to bf16 (8 significant bits per
    element, unbiased) before its two products, hence 4e-3 of the largest gradient entry."""
    E, P, target, E16 = _score16_case(B, V, B + V)
    Bp = ops.score16_batch_pad(B)
    assert Bp % 128 == 0 and Bp >= B
    ops.f32_to_bf16(dev(E).view(-1), E16.view(-1))
    P16 = torch.empty((Bp, D), dtype=torch.bfloat16, device="cuda")
    ops.f32_to_bf16(dev(P).view(-1), P16.view(-1))
    tgt = dev(target)
    lse = torch.zeros(B, device="cuda")
    ce = torch.zeros(B, device="cuda")
    partial = torch.zeros(ops.score16_partials(B, V), device="cuda")
    ops.score16_lse(E16, P16, tgt, B, V, partial, lse, ce)

    Er = E16.double()
    Pr = P16[:B].double()
    ref_logits = Pr @ Er.T
    ref_lse = torch.logsumexp(ref_logits, 1)
    ref_ce = ref_lse - ref_logits.gather(1, tgt.long()[:, None])[:, 0]
    assert float((lse.double() - ref_lse).abs().max()) < 1e-5 * float(ref_lse.abs().max())
    assert float((ce.double() - ref_ce).abs().max()) < 2e-5 * float(ref_lse.abs().max())

    ld = (V + 3) // 4 * 4
    logits = torch.full((B, ld), -77.0, device="cuda")
    ops.score16_logits(E16, P16, B, V, logits, ld)
    assert float((logits[:, :V].double() - ref_logits).abs().max()) < 1e-5 * float(ref_logits.abs().max())
    assert bool((logits[:, V:] == -77.0).all())

    scale = 1.0 / B
    G = (torch.exp(ref_logits - ref_lse[:, None]))
    G[torch.arange(B), tgt.long()] -= 1.0
    G *= scale
    ref_dpred = G @ Er
    ref_dE = G.T @ Pr
    d_pred = torch.zeros((B, D), device="cuda")
    dE = torch.full((V, D), 5.0, device="cuda")
    sq = torch.zeros(ops.score16_sq_partials(V), device="cuda")
    ops.score16_bwd(E16, P16, lse, tgt, B, V, scale, d_pred, dE, sq)
    assert float((d_pred.double() - ref_dpred).abs().max()) < 4e-3 * float(ref_dpred.abs().max())
    assert float((dE.double() - ref_dE).abs().max()) < 4e-3 * float(ref_dE.abs().max())
    # gradient rows of the softmax sum to ~0 over the catalog: dE^T 1 = pred^T (G 1) -- the bf16 rounding of G
    # leaves a residue well below one rounding step of the largest entry
    assert abs(float(sq.double().sum()) - float((dE.double() ** 2).sum())) < 1e-5 * float((dE.double() ** 2).sum())


def test_gather_reads_item_rows_from_the_bf16_copy(ops):
    """mtam_emb_gather_fwd_item16: item rows = the bf16 image widened to fp32 (bit-exact), every other output
    identical to the fp32 call, the L2 partials are those of the values actually gathered."""
    rng = np.random.default_rng(11)
    B, L, V, NC, NP, NU = 37, 20, 500, 31, 23, 50
    tabs = [rng.standard_normal((n, D)).astype(np.float32) for n in (V, NC, NP, NU)]
    ids = [rng.integers(0, n, B * L).astype(np.int32) for n in (V, NC, NP)] + [rng.integers(0, NU, B).astype(np.int32)]
    t = [dev(x) for x in tabs]
    i = [dev(x) for x in ids]
    item16 = torch.empty((V, D), dtype=torch.bfloat16, device="cuda")
    ops.f32_to_bf16(t[0].view(-1), item16.view(-1))
    outs = []
    for use16 in (None, item16):
        ic = torch.zeros((B * L, 2 * D), device="cuda")
        pos = torch.zeros((B * L, D), device="cuda")
        user = torch.zeros((B, D), device="cuda")
        l2 = torch.zeros(ops.emb_gather_partials(B, L), device="cuda")
        ops.emb_gather_fwd(t[0], t[1], t[2], t[3], i[0], i[1], i[2], i[3], B, L, 1, ic, pos, user, l2, item16=use16)
        outs.append((ic, pos, user, l2))
    (ic0, pos0, user0, l20), (ic1, pos1, user1, l21) = outs
    assert torch.equal(ic1[:, :D], item16[i[0].long()].float())
    assert torch.equal(ic0[:, :D], t[0][i[0].long()])
    assert torch.equal(ic1[:, D:], ic0[:, D:]) and torch.equal(pos1, pos0) and torch.equal(user1, user0)
    want = float((ic1.double() ** 2).sum() + (pos1.double() ** 2).sum() + (user1.double() ** 2).sum())
    assert abs(float(l21.double().sum()) - want) < 1e-5 * want


@pytest.mark.parametrize("B,V", [(128, 3709), (100, 1000), (37, 63), (256, 7001), (130, 70007), (128, 200003),
                                 (61, 65536), (1, 17), (3, 32), (129, 33)])
@pytest.mark.parametrize("form", ["split", "native"])
def test_score32_lse_and_backward(ops, B, V, form):
    """csrc/score32.hip (fp32 logits-free scoring) against float64 products of the same fp32 operands:
    fp32 products and sums, so 2e-5 of the largest entry (G itself is formed in fp32 here).  Both forms: native
    fp32 MFMA, and (the default) every fp32 product as six bf16 MFMA terms of three-way operand splits: the same
    tolerances hold, and the tighter ones below show that nothing of fp32's accuracy is given up."""
    if form == "native" and V > 100000:
        pytest.skip("the native pair is not used at this size")
    ops.score32_set_split_min_rows(1 if form == "split" else 0)
    try:
        _score32_case(ops, B, V, form)
    finally:
        ops.score32_set_split_min_rows(1)


@pytest.mark.parametrize("B,V,cuts", [(128, 3709, (0, 1240, 2480, 3709)), (300, 70007, (0, 8751, 70007)),
                                      (37, 1003, (0, 126, 252, 378, 504, 630, 756, 882, 1003))])
def test_score32_row_ranges_combine_to_the_whole_catalog(ops, B, V, cuts):
    """mtam_score32_lse_range / mtam_score32_bwd_range (data-parallel row-sharded scoring): the catalog cut into row
    ranges, each scored on its own -- per-range (lse, target logit) combine to the whole catalog's lse / cross entropy,
    the ranges' dE blocks ARE the rows of the whole dE (complete, nothing to sum) and the ranges' d_pred shares add
    up to the whole d_pred.  Against the one-piece kernels (fp32 rounding) and float64."""
    rng = np.random.default_rng(B + V)
    E = dev((rng.standard_normal((V, D)) * 0.2).astype(np.float32))
    P = dev(rng.standard_normal((B, D)).astype(np.float32))
    target = rng.integers(0, V, B).astype(np.int32)
    target[0], target[1 % B] = V - 1, cuts[1]              # the last row; the first row of the second range
    tgt = dev(target)
    z = lambda *s: torch.zeros(s, device="cuda")
    lse_w, ce_w = z(B), z(B)
    ops.score32_lse(E, P, tgt, B, V, z(ops.score32_partials(B, V)), lse_w, ce_w)
    parts, logit = [], z(B)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        lp, tl = z(B), z(B)
        ops.score32_lse(E[lo:hi], P, tgt, B, hi - lo, z(ops.score32_partials(B, hi - lo)), lp, tl, row0=lo)
        owned = (tgt >= lo) & (tgt < hi)
        assert not bool(tl[~owned].any())                   # a target outside the range contributes no logit
        parts.append(lp)
        logit += tl
    lse = torch.logsumexp(torch.stack(parts), 0)
    ref = torch.logsumexp(P.double() @ E.double().T, 1)
    assert float((lse.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    assert float((lse - lse_w).abs().max()) < 2e-6 * float(lse_w.abs().max())
    assert float(((lse - logit) - ce_w).abs().max()) < 2e-5 * float(lse_w.abs().max())
    # backward: the whole catalog in one piece, then range by range with the combined lse
    scale = 1.0 / B
    d_pred_w, dE_w = z(B, D), z(V, D)
    ops.score32_bwd(E, P, lse, tgt, B, V, scale, d_pred_w, dE_w)
    d_pred, dE = z(B, D), torch.full((V, D), 9.0, device="cuda")
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        share = z(B, D)
        ops.score32_bwd(E[lo:hi], P, lse, tgt, B, hi - lo, scale, share, dE[lo:hi], row0=lo)
        d_pred += share
    assert float((dE - dE_w).abs().max()) <= 2e-6 * float(dE_w.abs().max())
    assert float((d_pred - d_pred_w).abs().max()) <= 1e-5 * float(d_pred_w.abs().max())
    G = torch.exp(P.double() @ E.double().T - ref[:, None])
    G[torch.arange(B), tgt.long()] -= 1.0
    assert float((dE.double() - (G * scale).T @ P.double()).abs().max()) < 2e-5 * float(dE_w.abs().max())


@pytest.mark.parametrize("B,V,fused", [(128, 3709, True), (128, 3712, True), (100, 1000, True), (37, 257, True),
                                       (1, 5376, True), (128, 5377, False), (128, 7169, False), (129, 3709, False), (37, 63, False),
                                       (128, 256, True), (128, 225, True), (128, 224, False)])
def test_score32_train_one_call(ops, B, V, fused):
    """mtam_score32_train: the loss terms and both scoring gradients in one call -- ONE launch (x3::train_small_kernel:
    grid barrier, d_pred summed in slab order) when every slab gets a resident workgroup, lse + bwd otherwise.  Against
    float64 products of the same fp32 operands and against the two-pass kernels; three calls in a row on the same work
    buffer (the barrier words return to rest) give bit-identical results (no float atomics in the fused form)."""
    assert ops.score32_train_is_fused(B, V) == fused
    rng = np.random.default_rng(B * 7 + V)
    E = dev((rng.standard_normal((V, D)) * 0.2).astype(np.float32))
    P = dev(rng.standard_normal((B, D)).astype(np.float32))
    target = rng.integers(0, V, B).astype(np.int32)
    target[0], target[B - 1] = V - 1, 0
    if B > 2:
        target[1] = (V // 32) * 32 - 1 if V >= 64 else 1        # last row of the last full slab
    tgt = dev(target)
    z = lambda *s: torch.zeros(s, device="cuda")
    scale = 1.0 / B
    OFF = 1.0 / 1024            # d_pred is ACCUMULATED: it starts from this (small: the sums round at its ulp, 1e-10)
    work = ops.score32_train_work(B, V)
    outs = []
    for _ in range(3):
        lse, ce, d_pred = z(B), z(B), torch.full((B, D), OFF, device="cuda")
        dE, sq = torch.full((V, D), 5.0, device="cuda"), z(ops.score32_sq_partials(V))
        ops.score32_train(E, P, tgt, B, V, scale, work, lse, ce, d_pred, dE, sq)
        outs.append((lse, ce, d_pred, dE, sq))
    torch.cuda.synchronize()
    lse, ce, d_pred, dE, sq = outs[0]
    if fused:
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    ref_logits = P.double() @ E.double().T
    ref_lse = torch.logsumexp(ref_logits, 1)
    ref_ce = ref_lse - ref_logits.gather(1, tgt.long()[:, None])[:, 0]
    assert float((lse.double() - ref_lse).abs().max()) < 1e-5 * float(ref_lse.abs().max())
    assert float((ce.double() - ref_ce).abs().max()) < 2e-5 * float(ref_lse.abs().max())
    G = torch.exp(ref_logits - ref_lse[:, None])
    G[torch.arange(B), tgt.long()] -= 1.0
    G *= scale
    ref_dpred, ref_dE = G @ E.double(), G.T @ P.double()
    assert float((d_pred.double() - OFF - ref_dpred).abs().max()) < 2e-5 * float(ref_dpred.abs().max()) + 1e-8
    assert float((dE.double() - ref_dE).abs().max()) < 2e-5 * float(ref_dE.abs().max())
    assert abs(float(sq.double().sum()) - float((dE.double() ** 2).sum())) < 1e-5 * float((dE.double() ** 2).sum())
    # the two-pass kernels on the same operands: equal to fp32 rounding
    lse2, ce2, d_pred2, dE2 = z(B), z(B), z(B, D), z(V, D)
    ops.score32_lse(E, P, tgt, B, V, z(ops.score32_partials(B, V)), lse2, ce2)
    ops.score32_bwd(E, P, lse2, tgt, B, V, scale, d_pred2, dE2)
    assert float((lse - lse2).abs().max()) <= 2e-6 * float(lse2.abs().max())
    assert float((ce - ce2).abs().max()) <= 2e-5 * float(lse2.abs().max())
    assert float((dE - dE2).abs().max()) <= 4e-6 * float(dE2.abs().max())
    assert float((d_pred - OFF - d_pred2).abs().max()) <= 1e-5 * float(d_pred2.abs().max()) + 1e-8


def test_score32_rejects_buffers_sized_under_the_other_form(ops):
    """The partial-buffer counts depend on the form (mtam_score32_set_split_min_rows is process-global state): a
    buffer sized under one form is REJECTED when the other is in force, not overrun -- and the backward's
    squared-norm partials must be exactly what its consumer will sum."""
    B, V = 128, 200003
    ops.score32_set_split_min_rows(1)
    try:
        n_split, n_sq_split = ops.score32_partials(B, V), ops.score32_sq_partials(V)
        ops.score32_set_split_min_rows(0)
        n_native, n_sq_native = ops.score32_partials(B, V), ops.score32_sq_partials(V)
        assert n_split != n_native
        small, big = sorted([(n_split, 1), (n_native, 0)])
        E = torch.zeros((V, D), device="cuda")
        P = torch.zeros((B, D), device="cuda")
        tgt = torch.zeros(B, dtype=torch.int32, device="cuda")
        lse, ce = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
        ops.score32_set_split_min_rows(big[1])
        with pytest.raises(RuntimeError, match="partial buffer holds"):
            ops.score32_lse(E, P, tgt, B, V, torch.zeros(small[0], device="cuda"), lse, ce)
        if n_sq_split != n_sq_native:
            ops.score32_set_split_min_rows(1)
            with pytest.raises(RuntimeError, match="sq_partial holds"):
                ops.score32_bwd(E, P, lse, tgt, B, V, 1.0 / B, torch.zeros((B, D), device="cuda"),
                                torch.zeros((V, D), device="cuda"), torch.zeros(n_sq_native, device="cuda"))
    finally:
        ops.score32_set_split_min_rows(1)


def _score32_case(ops, B, V, form):
    rng = np.random.default_rng(B * 3 + V)
    E = dev((rng.standard_normal((V, D)) * 0.2).astype(np.float32))
    P = dev(rng.standard_normal((B, D)).astype(np.float32))
    target = rng.integers(0, V, B).astype(np.int32)
    target[0] = V - 1
    tgt = dev(target)
    lse = torch.zeros(B, device="cuda")
    ce = torch.zeros(B, device="cuda")
    partial = torch.zeros(ops.score32_partials(B, V), device="cuda")
    ops.score32_lse(E, P, tgt, B, V, partial, lse, ce)
    ref_logits = P.double() @ E.double().T
    ref_lse = torch.logsumexp(ref_logits, 1)
    ref_ce = ref_lse - ref_logits.gather(1, tgt.long()[:, None])[:, 0]
    assert float((lse.double() - ref_lse).abs().max()) < 1e-5 * float(ref_lse.abs().max())
    assert float((ce.double() - ref_ce).abs().max()) < 2e-5 * float(ref_lse.abs().max())
    scale = 1.0 / B
    G = torch.exp(ref_logits - ref_lse[:, None])
    G[torch.arange(B), tgt.long()] -= 1.0
    G *= scale
    ref_dpred, ref_dE = G @ E.double(), G.T @ P.double()
    d_pred = torch.zeros((B, D), device="cuda")
    dE = torch.full((V, D), 5.0, device="cuda")
    sq = torch.zeros(ops.score32_sq_partials(V), device="cuda")
    ops.score32_bwd(E, P, lse, tgt, B, V, scale, d_pred, dE, sq)
    assert float((d_pred.double() - ref_dpred).abs().max()) < 2e-5 * float(ref_dpred.abs().max())
    assert float((dE.double() - ref_dE).abs().max()) < 2e-5 * float(ref_dE.abs().max())
    assert abs(float(sq.double().sum()) - float((dE.double() ** 2).sum())) < 1e-5 * float((dE.double() ** 2).sum())
    if form == "split" and V >= 1000:
        # fp32-level accuracy of the split products: as close to float64 as a native fp32 matmul of the operands is
        nat_lse = torch.logsumexp(P @ E.T, 1).double()
        assert float((lse.double() - ref_lse).abs().max()) <= 2.0 * float((nat_lse - ref_lse).abs().max()) + 1e-6
        nat_dE = (G.float().T @ P).double()
        assert float((dE.double() - ref_dE).abs().max()) <= 2.0 * float((nat_dE - ref_dE).abs().max()) + \
            1e-7 * float(ref_dE.abs().max())


def _weight_images(ops, W4, Wkv, Wx):
    """The bf16 operand images of the three weight matrices, as mtam_split_weight_images writes them."""
    n_kv, n_x = (Wkv.shape[1] if Wkv is not None else 0), Wx.shape[1]
    buf = torch.zeros(ops.seq_chain_images_elems(n_kv, n_x), dtype=torch.bfloat16, device="cuda")
    for which, W in ((0, W4), (1, Wkv), (2, Wx)):
        if W is not None:
            ops.split_weight_images(W, buf[ops.seq_chain_image_offset(which, n_x):])
    return buf


def test_weight_images_layout_and_exactness(ops):
    """mtam_split_weight_images: W = W1 + W2 + W3 to the last bit of fp32 (three bf16 terms carry 24 significant
    bits), stored [K / 8][N][8] per term -- checked element by element against the layout the header states."""
    rng = np.random.default_rng(5)
    K, N = 24, 20
    W = (rng.standard_normal((K, N)) * np.exp(rng.uniform(-8, 8, (K, N)))).astype(np.float32)
    img = torch.zeros(3 * K * N, dtype=torch.bfloat16, device="cuda")
    ops.split_weight_images(dev(W), img)
    terms = img.float().cpu().numpy().reshape(3, K // 8, N, 8).transpose(0, 1, 3, 2).reshape(3, K, N)
    assert np.array_equal(terms[0], dev(W).bfloat16().float().cpu().numpy())      # term 1 = round-to-nearest bf16
    total = terms[0].astype(np.float64) + terms[1] + terms[2]
    assert np.array_equal(total.astype(np.float32), W) and np.abs(total - W).max() <= 2.0 ** -24 * np.abs(W).max()


def test_gru_weight_image_layout(ops, hip_lib):
    """mtam_gru_weight_image: a permutation of wh_g [128, 256] and wh_c [128, 128] -- element (k, n) at
    mtam_gru_weight_image_pos(which, k, n), every position hit exactly once -- and mtam_adam_images (gru_which 1 / 2)
    keeps it current from the values it updates."""
    rng = np.random.default_rng(2)
    whg, whc = rng.standard_normal((D, 2 * D)).astype(np.float32), rng.standard_normal((D, D)).astype(np.float32)
    n = ops.gru_weight_image_floats()
    assert n == D * 3 * D
    img = torch.full((n,), 7.0, device="cuda")
    ops.gru_weight_image(dev(whg), dev(whc), img)
    got = img.cpu().numpy()
    pos_g = np.array([[hip_lib.mtam_gru_weight_image_pos(0, k, c) for c in range(2 * D)] for k in range(D)])
    pos_c = np.array([[hip_lib.mtam_gru_weight_image_pos(1, k, c) for c in range(D)] for k in range(D)])
    assert sorted(np.concatenate([pos_g.ravel(), pos_c.ravel()]).tolist()) == list(range(n))
    assert np.array_equal(got[pos_g], whg) and np.array_equal(got[pos_c], whc)
    assert hip_lib.mtam_gru_weight_image_pos(0, D, 0) == -1 and hip_lib.mtam_gru_weight_image_pos(1, 0, D) == -1
    # kept current by the optimizer launch
    blk = ops.adam_block()
    total = 16 * blk
    p0 = rng.standard_normal(total).astype(np.float32)
    p0[64:64 + whg.size] = whg.ravel()
    p0[40000:40000 + whc.size] = whc.ravel()
    g, m0, v0 = rng.standard_normal(total).astype(np.float32), np.zeros(total, np.float32), np.full(total, 0.01, np.float32)
    scale, hyper = dev(np.array([1.0, 1.0], np.float32)), dev(np.array([1e-2, 0.9, 0.999, 1e-8], np.float32))
    p, m, v = dev(p0), dev(m0), dev(v0)
    img2 = torch.zeros(n, device="cuda")
    descs = ops.weight_image_descs([(64, D, 2 * D, img2, None, "gru_g"), (40000, D, D, img2, None, "gru_c")])
    ops.adam_images(p, m, v, dev(g), total, scale, hyper, total, descs)
    want = torch.zeros(n, device="cuda")
    ops.gru_weight_image(p[64:64 + whg.size].view(D, 2 * D), p[40000:40000 + whc.size].view(D, D), want)
    assert torch.equal(img2, want) and not torch.equal(p[64:64 + whg.size].cpu(), torch.from_numpy(whg.ravel()))
    p_ref, m_ref, v_ref = dev(p0), dev(m0), dev(v0)
    ops.adam(p_ref, m_ref, v_ref, dev(g), total, scale, hyper, total)
    assert torch.equal(p, p_ref) and torch.equal(m, m_ref) and torch.equal(v, v_ref)


def test_adam_rewrites_the_weight_images(ops):
    """mtam_adam_images = mtam_adam (same parameters, same slots, bit for bit) + the bf16 operand images of the
    listed matrices re-written from the UPDATED values -- what mtam_split_weight_images gives afterwards -- with and
    without the bf16 copy of the tail, matrices at odd places of the flat space, elements outside them untouched."""
    rng = np.random.default_rng(11)
    blk = ops.adam_block()
    n = 3 * blk + 128 * 40 + 8
    # (begin, K, N): inside the dense part [0, 2 blk), not block-aligned, the last one ending exactly at its end
    mats = [(256, 16, 128), (256 + 16 * 128 + 4, 128, 36), (2 * blk - 8 * 128, 8, 128)]
    g, p0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    m0, v0 = (rng.standard_normal(n) * 0.1).astype(np.float32), rng.uniform(0, 0.1, n).astype(np.float32)
    scale = dev(np.array([0.7, 1.0], np.float32))
    hyper = dev(np.array([1e-3, 0.9, 0.999, 1e-8], np.float32))
    for copy in (False, True):
        p, m, v = dev(p0), dev(m0), dev(v0)
        ops.adam(p, m, v, dev(g), n, scale, hyper, 2 * blk)
        p2, m2, v2 = dev(p0), dev(m0), dev(v0)
        imgs = [torch.full((3 * K * N,), 7.0, dtype=torch.bfloat16, device="cuda") for _, K, N in mats]
        imgs_r = [torch.full((3 * K * N,), 7.0, dtype=torch.bfloat16, device="cuda") if N % 8 == 0 else None
                  for _, K, N in mats]
        descs = ops.weight_image_descs([(b, K, N, im, ir) for (b, K, N), im, ir in zip(mats, imgs, imgs_r)])
        copy16 = torch.zeros(n - 2 * blk, dtype=torch.bfloat16, device="cuda") if copy else None
        ops.adam_images(p2, m2, v2, dev(g), n, scale, hyper, 2 * blk, descs, copy16=copy16, copy_begin=2 * blk)
        assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)
        for (b, K, N), im in zip(mats, imgs):
            want = torch.zeros_like(im)
            ops.split_weight_images(p2[b:b + K * N].view(K, N), want)
            assert torch.equal(im, want)
        for (b, K, N), ir in zip(mats, imgs_r):
            if ir is not None:                      # the images of the transpose (the backward's operands)
                want = torch.zeros_like(ir)
                ops.split_weight_rows(p2[b:b + K * N].view(K, N), want)
                assert torch.equal(ir, want)
                # the layout the header states: the wimg layout of W^T -- [N / 8][K][8] per term
                terms = ir.float().view(3, N // 8, K, 8).permute(0, 2, 1, 3).reshape(3, K, N)
                assert torch.equal((terms[0].double() + terms[1].double() + terms[2].double()).float(),
                                   p2[b:b + K * N].view(K, N))
        if copy:
            assert torch.equal(copy16, p2[2 * blk:].bfloat16())
    with pytest.raises(RuntimeError):           # K must be a multiple of 8
        ops.adam_images(p2, m2, v2, dev(g), n, scale, hyper, 2 * blk, ops.weight_image_descs([(0, 12, 128, imgs[0])]))
    with pytest.raises(RuntimeError):           # a matrix among the table rows (IndexedSlices update form) is refused
        ops.adam_images(p2, m2, v2, dev(g), n, scale, hyper, blk, ops.weight_image_descs([mats[2] + (imgs[2],)]))


@pytest.mark.parametrize("n_extra,with_images", [(0, True), (5357, True), (300, False), (16000, False), (17000, False)])
def test_clip_inside_the_optimizer_launch_equals_the_ticket_pair(ops, n_extra, with_images):
    """mtam_sqnorm_state_loss + mtam_adam_images_clip (no arrival ticket: partials only, then EVERY workgroup of the
    optimizer launch sums them and derives the clip scale) against mtam_sqnorm_clip_scale + mtam_adam_images: the
    same norm, scale, Adam state, loss, parameters, slots and images, bit for bit; three steps in a row."""
    rng = np.random.default_rng(n_extra + 3)
    blk = ops.adam_block()
    n_dense, n = 5 * blk, 7 * blk + 128 * 11
    nb = ops.sqnorm_blocks(n_dense)
    n_part = nb + n_extra
    mats = [(256, 16, 128), (2 * blk + 4, 128, 36)] if with_images else []
    p0 = rng.standard_normal(n).astype(np.float32)
    m0, v0 = (rng.standard_normal(n) * 0.1).astype(np.float32), rng.uniform(0, 0.1, n).astype(np.float32)
    l2, ce = dev(rng.uniform(0, 2, 977).astype(np.float32)), dev(rng.uniform(0, 9, 37).astype(np.float32))
    lr = dev(np.array([1e-3], np.float32))

    def run(new):
        p, m, v = dev(p0), dev(m0), dev(v0)
        state = dev(np.array([0.0, 0.9, 0.999, 1e-8, 0.9, 0.999, 0.0, 0.0], np.float32))
        scale, loss = torch.zeros(2, device="cuda"), torch.zeros(4, device="cuda")
        ticket = torch.zeros(4, dtype=torch.int32, device="cuda")
        imgs = [torch.full((3 * K * N,), 7.0, dtype=torch.bfloat16, device="cuda") for _, K, N in mats]
        descs = ops.weight_image_descs([(b, K, N, im) for (b, K, N), im in zip(mats, imgs)]) if mats else None
        out = []
        for step in range(3):
            g = dev((np.random.default_rng(step).standard_normal(n) * (3.0 if step == 1 else 0.01)).astype(np.float32))
            part = torch.zeros(n_part + 4, device="cuda")
            part[nb:n_part] = dev(np.random.default_rng(step + 9).uniform(0, 1e-3, n_extra).astype(np.float32))
            if new:
                ops.sqnorm_state_loss(g, n_dense, part, 0, lr, state, l2, l2.numel(), ce, ce.numel(), 1e-4, 1 / 37.0, loss)
                ops.adam_images_clip(p, m, v, g, n, part, n_part, 5.0, scale, state, n_dense, descs)
            else:
                ops.sqnorm_clip_scale(g, n_dense, part, 0, n_part, 5.0, scale, lr, state, ticket, l2, l2.numel(), ce,
                                      ce.numel(), 1e-4, 1 / 37.0, loss)
                if descs is not None:
                    ops.adam_images(p, m, v, g, n, scale, state, n_dense, descs)
                else:
                    ops.adam(p, m, v, g, n, scale, state, n_dense)
            out.append([t.clone() for t in (p, m, v, scale, state, loss[:3])] + [im.clone() for im in imgs])
        return out

    if n_extra + nb > ops.adam_clip_max_partials():
        with pytest.raises(RuntimeError):
            run(True)
        return
    a, b = run(False), run(True)
    for step, (x, y) in enumerate(zip(a, b)):
        for i, (s, t) in enumerate(zip(x, y)):
            assert torch.equal(s, t), (step, i)
    assert float(a[1][3][0]) < 1.0 and float(a[0][3][0]) == 1.0      # step 1 is clipped, step 0 is not


@pytest.mark.parametrize("words,slots,with_images", [(32768, 5, True), (4096 + 12, 3, False), (8, 1, False), (65536 + 4, 2, True)])
def test_optimizer_launch_hands_over_the_next_feed(ops, words, slots, with_images):
    """mtam_adam_images_clip_feed: the update itself is mtam_adam_images_clip's bit for bit, and one more workgroup
    copies ring slot (cursor % slots) into the arena and advances the cursor -- seven launches in a row walk the ring
    (wrap-around included), nothing outside the arena is written, the ring is left as it was.  An arena inside the
    ring, a slot pitch that is not a multiple of 4 words or a missing cursor are refused."""
    rng = np.random.default_rng(words + slots)
    blk = ops.adam_block()
    n_dense, n = 3 * blk, 4 * blk + 128 * 7
    nb = ops.sqnorm_blocks(n_dense)
    mats = [(256, 16, 128)] if with_images else []
    p0 = rng.standard_normal(n).astype(np.float32)
    m0, v0 = (rng.standard_normal(n) * 0.1).astype(np.float32), rng.uniform(0, 0.1, n).astype(np.float32)
    ring_np = rng.integers(-2 ** 31, 2 ** 31 - 1, (slots, words), dtype=np.int64).astype(np.int32)

    def run(feed):
        p, m, v = dev(p0), dev(m0), dev(v0)
        state = dev(np.array([1e-3, 0.9, 0.999, 1e-8, 0.9, 0.999, 0.0, 0.0], np.float32))
        scale = torch.zeros(2, device="cuda")
        imgs = [torch.full((3 * K * N,), 7.0, dtype=torch.bfloat16, device="cuda") for _, K, N in mats]
        descs = ops.weight_image_descs([(b, K, N, im) for (b, K, N), im in zip(mats, imgs)]) if mats else None
        ring = torch.from_numpy(ring_np).cuda()
        guard = torch.full((words + 64,), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
        arena = guard[32:32 + words]
        cursor = torch.tensor([3], dtype=torch.int32, device="cuda")
        out = []
        for step in range(7):
            g = dev((np.random.default_rng(step).standard_normal(n) * 0.01).astype(np.float32))
            part = dev(np.random.default_rng(step + 9).uniform(0, 1e-3, nb + 11).astype(np.float32))
            ops.adam_images_clip(p, m, v, g, n, part, nb + 11, 5.0, scale, state, n_dense, descs,
                                 feed=(ring, arena, cursor) if feed else None)
            if feed:
                assert int(cursor.item()) == 3 + step + 1
                assert torch.equal(arena, ring[(3 + step) % slots]), step
                assert bool((guard[:32] == 0x5a5a5a5a).all()) and bool((guard[32 + words:] == 0x5a5a5a5a).all())
            out.append([t.clone() for t in (p, m, v, scale)] + [im.clone() for im in imgs])
        if feed:
            assert np.array_equal(ring.cpu().numpy(), ring_np)
        return out

    for x, y in zip(run(False), run(True)):
        for s, t in zip(x, y):
            assert torch.equal(s, t)
    p, m, v, g = dev(p0), dev(m0), dev(v0), dev(p0)
    state, scale = dev(np.array([1e-3, 0.9, 0.999, 1e-8, 0.9, 0.999, 0.0, 0.0], np.float32)), torch.zeros(2, device="cuda")
    part = torch.ones(nb, device="cuda")
    ring = torch.zeros((4, 64), dtype=torch.int32, device="cuda")
    cursor = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):            # the arena is one of the ring's own slots
        ops.adam_images_clip(p, m, v, g, n, part, nb, 5.0, scale, state, n_dense, None, feed=(ring, ring[1], cursor))
    ring6 = torch.zeros((4, 66), dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):            # slot pitch not a multiple of 4 words
        ops.adam_images_clip(p, m, v, g, n, part, nb, 5.0, scale, state, n_dense, None,
                             feed=(ring6, torch.zeros(66, dtype=torch.int32, device="cuda"), cursor))


@pytest.mark.parametrize("x3", [False, True])
@pytest.mark.parametrize("R,n_kv,n_x", [(6400, 256, 384), (100, 512, 384), (33, 0, 384), (777, 256, 640)])
def test_seq_chain_fwd_matches_the_three_products(ops, R, n_kv, n_x, x3):
    """mtam_seq_chain_fwd = dense4emb (relu, + position), K/V projection (bias, relu) and the GRU input
    projection (bias) of the forward step; float64 reference, fp32 MFMA sums: 2e-5 of the largest entry.
    x3: the same products as six bf16-MFMA terms of split operands (weights pre-split into images) -- the SAME
    tolerance, and no further from float64 than 2 x a native fp32 matmul of the operands is."""
    rng = np.random.default_rng(R + n_kv + n_x)
    ic = rng.standard_normal((R, 2 * D)).astype(np.float32)
    pos = rng.standard_normal((R, D)).astype(np.float32)
    W4 = (rng.standard_normal((2 * D, D)) * 0.1).astype(np.float32)
    Wkv = (rng.standard_normal((D, max(n_kv, 32))) * 0.1).astype(np.float32)[:, :n_kv]
    bkv = rng.standard_normal(n_kv).astype(np.float32)
    Wx = (rng.standard_normal((D, n_x)) * 0.1).astype(np.float32)
    bx = rng.standard_normal(n_x).astype(np.float32)
    zr = torch.full((R, D), 9.0, device="cuda")
    x = torch.full((R, D), 9.0, device="cuda")
    kv = torch.full((R, max(n_kv, 1)), 9.0, device="cuda")
    xproj = torch.full((R, n_x), 9.0, device="cuda")
    Wkv_d = dev(np.ascontiguousarray(Wkv)) if n_kv else None
    images = _weight_images(ops, dev(W4), Wkv_d, dev(Wx)) if x3 else None
    ops.seq_chain_fwd(dev(ic), dev(W4), dev(pos), R, Wkv_d, dev(bkv) if n_kv else None, dev(Wx), dev(bx), zr, x,
                      kv if n_kv else None, xproj, w_images=images)
    z64 = ic.astype(np.float64) @ W4.astype(np.float64)
    if x3:
        nat = (dev(ic) @ dev(W4)).double().cpu().numpy()
        got_z = np.where(zr.cpu().numpy() > 0, zr.cpu().numpy().astype(np.float64), z64)     # (relu'd entries: skip)
        assert np.abs(got_z - z64).max() <= 2.0 * np.abs(nat - z64).max() + 1e-7 * np.abs(z64).max()
    zr64 = np.maximum(z64, 0.0)
    x64 = zr64 + pos
    assert rel_err(zr.cpu().numpy(), zr64) < 2e-5 and rel_err(x.cpu().numpy(), x64) < 2e-5
    assert rel_err(xproj.cpu().numpy(), x64 @ Wx.astype(np.float64) + bx) < 2e-5
    if n_kv:
        assert rel_err(kv.cpu().numpy(), np.maximum(x64 @ Wkv.astype(np.float64) + bkv, 0.0)) < 2e-5
    # the relu mask the backward uses is exact where the pre-activation is not within rounding of zero
    clear = np.abs(z64) > 1e-4
    assert np.array_equal((zr.cpu().numpy() > 0)[clear], (z64 > 0)[clear])


@pytest.mark.parametrize("R,n_kv", [(6400, 256), (100, 128), (33, 0), (777, 256), (1, 256)])
def test_seq_chain_bwd_matches_the_two_gemms(ops, R, n_kv):
    """mtam_seq_chain_bwd (the backward's sequence-side chain as one stripe kernel): d_x += d_xproj Wx^T + d_kv Wkv^T
    + d_xt, d_z = d_x where zr > 0, d_ic = d_z W4^T -- against float64 (2e-5 of the largest entry, the tolerance of
    the GEMMs it replaces; no further from float64 than 2 x a native fp32 matmul) and against the two-launch form
    (mtam_gemm_f32_dual ACCUM2_MASK + mtam_gemm_f32): the same mask, values to fp32 rounding."""
    rng = np.random.default_rng(R + n_kv)
    n_x = 384
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    d_xproj, d_kv, d_xt, d_x0 = f(R, n_x), f(R, max(n_kv, 1))[:, :n_kv], f(R, D), f(R, D)
    zr = np.maximum(f(R, D), 0.0)
    Wx, Wkv, W4 = f(D, n_x) * 0.1, np.ascontiguousarray((f(D, max(n_kv, 8)) * 0.1)[:, :n_kv]), f(2 * D, D) * 0.1
    n_img = ops.seq_chain_images_elems(n_kv, n_x)
    img_r = torch.zeros(n_img, dtype=torch.bfloat16, device="cuda")
    for which, W in ((0, W4), (1, Wkv), (2, Wx)):
        if W.size:
            ops.split_weight_rows(dev(W), img_r[ops.seq_chain_image_offset(which, n_x):])
    d_x, d_z, d_ic = dev(d_x0), torch.full((R, D), 9.0, device="cuda"), torch.full((R, 2 * D), 9.0, device="cuda")
    d_kv_d = dev(np.ascontiguousarray(d_kv)) if n_kv else None
    ops.seq_chain_bwd(dev(d_xproj), d_kv_d, dev(d_xt), dev(zr), R, d_x, d_z, d_ic, img_r)
    f64 = lambda a: a.astype(np.float64)
    dx64 = f64(d_x0) + f64(d_xproj) @ f64(Wx).T + (f64(d_kv) @ f64(Wkv).T if n_kv else 0.0) + f64(d_xt)
    dz64 = np.where(zr > 0, dx64, 0.0)
    dic64 = dz64 @ f64(W4).T
    assert rel_err(d_x.cpu().numpy(), dx64) < 2e-5 and rel_err(d_ic.cpu().numpy(), dic64) < 2e-5
    got_z = d_z.cpu().numpy()
    assert np.array_equal(got_z != 0, (zr > 0) & (d_x.cpu().numpy() != 0)) and rel_err(got_z, dz64) < 2e-5
    assert np.array_equal(got_z[zr > 0], d_x.cpu().numpy()[zr > 0])          # d_z is d_x itself under the mask
    nat = (dev(d_x0) + dev(d_xproj) @ dev(Wx).T + (d_kv_d @ dev(Wkv).T if n_kv else 0.0) + dev(d_xt)).double().cpu().numpy()
    assert np.abs(d_x.cpu().numpy() - dx64).max() <= 2.0 * np.abs(nat - dx64).max() + 1e-7 * np.abs(dx64).max()
    # the two-launch form it replaces
    d_x2, d_z2, d_ic2 = dev(d_x0), torch.zeros((R, D), device="cuda"), torch.zeros((R, 2 * D), device="cuda")
    if n_kv:
        ops.gemm_dual(dev(d_xproj), dev(Wx), d_kv_d, dev(Wkv), d_x2, trans_b=True, epilogue=ops.EPI_ACCUM2_MASK,
                      bias=dev(d_xt), aux_in=dev(zr), aux_out=d_z2)
    else:
        ops.gemm(dev(d_xproj), dev(Wx), d_x2, trans_b=True, epilogue=ops.EPI_ACCUM2_MASK, bias=dev(d_xt),
                 aux_in=dev(zr), aux_out=d_z2)
    ops.gemm(d_z2, dev(W4), d_ic2, trans_b=True)
    assert rel_err(d_x.cpu().numpy(), d_x2.cpu().numpy()) < 1e-5 and rel_err(d_ic.cpu().numpy(), d_ic2.cpu().numpy()) < 1e-5
    assert torch.equal(d_z != 0, d_z2 != 0)
    with pytest.raises(RuntimeError):           # widths are multiples of 128 ...
        ops.seq_chain_bwd(dev(d_xproj)[:, :320].contiguous(), d_kv_d, dev(d_xt), dev(zr), R, d_x, d_z, d_ic, img_r)
    with pytest.raises(RuntimeError):           # ... and the staged stripe holds n_x + n_kv <= 640 (one decoder block)
        ops.seq_chain_bwd(dev(d_xproj), torch.zeros((R, 512), device="cuda"), dev(d_xt), dev(zr), R, d_x, d_z, d_ic, img_r)
    assert ops.seq_chain_bwd_max_k() == 640


@pytest.mark.parametrize("rows,V,slab", [(5, 300000, 131072), (3, 65536 * 3, 65536), (2, 70000, 65536 * 2)])
def test_topk_stream_equals_topk_on_the_stored_matrix(ops, rows, V, slab):
    """mtam_topk_stream_slab / _finish (evaluation without stored logits, Model/base_model.py:194-202) against
    mtam_topk on the whole matrix: identical lists, with equal values planted on both sides of slab and
    segment boundaries (lower index first), a row whose best entries are all in the last partial segment, and
    -inf / -0.0 entries."""
    rng = np.random.default_rng(rows * 7 + V)
    k = 50
    x = rng.standard_normal((rows, V)).astype(np.float32)
    seg = ops.TOPK_STREAM_SEG
    # ties across a segment boundary and across a slab boundary, above everything else in the row
    x[0, [min(i, V - 1) for i in (seg - 1, seg, seg + 1, slab - 1, slab, 5)]] = 9.0
    x[1 % rows, V - 30:] = 7.0                         # the best 30 tie in the last (partial) segment
    x[1 % rows, :40] = 7.0                             # ... with 40 more at the very start: lower indices win
    x[rows - 1, 100:200] = -np.inf
    x[rows - 1, 300] = -0.0
    x[rows - 1, 301] = 0.0
    full = dev(x)
    want = torch.zeros((rows, k), dtype=torch.int32, device="cuda")
    want_v = torch.zeros((rows, k), device="cuda")
    ops.topk(full, V, rows, V, k, want, want_v)
    ws = torch.full(((ops.topk_stream_workspace_bytes(rows, V, k) + 3) // 4,), float("nan"), device="cuda")
    scratch = torch.empty((rows, slab), device="cuda")
    for col0 in range(0, V, slab):
        width = min(slab, V - col0)
        scratch[:, :width] = full[:, col0:col0 + width]
        ops.topk_stream_slab(scratch, slab, rows, col0, width, V, k, ws)
    got = torch.zeros((rows, k), dtype=torch.int32, device="cuda")
    got_v = torch.zeros((rows, k), device="cuda")
    ops.topk_stream_finish(ws, rows, V, k, got, got_v)
    assert torch.equal(got, want) and torch.equal(got_v, want_v)
    ref = np.argsort(-x.astype(np.float64), axis=1, kind="stable")[:, :k]
    assert np.array_equal(got.cpu().numpy(), ref)
    with pytest.raises(RuntimeError):                  # a slab must start on a segment boundary
        ops.topk_stream_slab(scratch, slab, rows, 1000, 10, V, k, ws)


@pytest.mark.parametrize("x3", [False, True])
@pytest.mark.parametrize("B,L,with_user,train", [(128, 50, 1, True), (7, 9, 0, True), (33, 50, 1, False), (1, 2, 1, True)])
def test_seq_chain_gather_fwd_equals_gather_then_chain(ops, B, L, with_user, train, x3):
    """mtam_seq_chain_gather_fwd (the four lookups folded into the forward's first GEMM kernel) against
    mtam_emb_gather_fwd + mtam_seq_chain_fwd: bit-identical zr / x / kv / xproj / user rows and (training) the
    [item | category] copy; the l2 sum to rounding; side ranges cleared; out-of-range ids clamped the same way."""
    rng = np.random.default_rng(B * 13 + L)
    R, n_kv, n_x = B * L, 256, 384
    V, C, U = 777, 31, 55
    T = {k: dev(rng.standard_normal((n, D)).astype(np.float32) * 0.3)
         for k, n in (("item", V), ("cat", C), ("pos", L + 3), ("user", U))}
    ids = dict(item=rng.integers(0, V, (B, L)), cat=rng.integers(0, C, (B, L)), pos=rng.integers(0, L + 3, (B, L)),
               user=rng.integers(0, U, B))
    ids["item"][0, 0] = V + 5                      # clamped like the gather kernel clamps
    ids = {k: dev(v.astype(np.int32)) for k, v in ids.items()}
    W4 = dev(rng.standard_normal((2 * D, D)).astype(np.float32) * 0.05)
    Wkv, bkv = dev(rng.standard_normal((D, n_kv)).astype(np.float32) * 0.05), dev(rng.standard_normal(n_kv).astype(np.float32))
    Wx, bx = dev(rng.standard_normal((D, n_x)).astype(np.float32) * 0.05), dev(rng.standard_normal(n_x).astype(np.float32))
    z = lambda *s: torch.full(s, 7.0, device="cuda")
    # reference: two kernels
    ic, pos, user = z(R, 2 * D), z(R, D), z(B, D)
    l2a = torch.zeros(ops.emb_gather_partials(B, L), device="cuda")
    ops.emb_gather_fwd(T["item"], T["cat"], T["pos"], T["user"], ids["item"], ids["cat"], ids["pos"], ids["user"], B, L,
                       with_user, ic, pos, user, l2a)
    ref = [z(R, D), z(R, D), z(R, n_kv), z(R, n_x)]
    images = _weight_images(ops, W4, Wkv, Wx) if x3 else None         # (both forms: split-bf16 products too)
    ops.seq_chain_fwd(ic, W4, pos, R, Wkv, bkv, Wx, bx, *ref, w_images=images)
    # fused
    got = [z(R, D), z(R, D), z(R, n_kv), z(R, n_x)]
    ic2, user2 = z(R, 2 * D), z(B, D)
    l2b = torch.full((ops.seq_chain_gather_partials(B, L) + 37,), 3.0, device="cuda")
    ca, cb = torch.ones(4 * 1000, device="cuda"), torch.ones(4 * 77, device="cuda")
    ops.seq_chain_gather_fwd(T["item"], T["cat"], T["pos"], T["user"], ids["item"], ids["cat"], ids["pos"], ids["user"],
                             B, L, with_user, W4, Wkv, bkv, Wx, bx, ic2 if train else None, user2, l2b, *got,
                             clear=(ca, cb), w_images=images)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert torch.equal(user2, user)
    assert torch.equal(ic2, ic) if train else bool((ic2 == 7.0).all())
    assert abs(float(l2b.double().sum()) - float(l2a.double().sum())) < 1e-6 * float(l2a.double().sum())
    assert not bool(l2b[ops.seq_chain_gather_partials(B, L):].any())
    assert not bool(ca.any()) and not bool(cb.any())
