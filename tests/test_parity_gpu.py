"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer CN counts and patterns bit-exact; fp32 pooled vectors and
scores within 1e-5 (atol) + 1e-5 (rtol).
"""
import json
import os
from functools import partial
from types import SimpleNamespace

import pytest
import torch

from oracle import ocn_oracle as O
from tests.helpers import batch, close, make_graph, product_adj2, spm_equal, to_product

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(__file__), "golden")

CASES = [  # n, avg_deg, max_deg, B, seed, isolated
    (64, 5, 20, 37, 0, 3),
    (500, 8, 100, 300, 1, 10),
    (3000, 12, 400, 2048, 2, 0),
    (20000, 10, 600, 8192, 3, 100),
]


@pytest.fixture(scope="module", params=CASES, ids=lambda c: f"n{c[0]}_B{c[3]}")
def case(request, hiplib):
    n, avg, mx, B, seed, iso = request.param
    oadj = make_graph(n, avg, mx, seed, isolated=iso)
    oadj2 = O.adj2_sparse(oadj)
    e = batch(oadj, B, seed + 50)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    return SimpleNamespace(n=n, B=B, seed=seed, oadj=oadj, oadj2=oadj2, e=e, adj=adj, adj2=adj2,
                           ocn1=O.adjoverlap(oadj, oadj, e), ocn2=O.adjoverlap(oadj, oadj2, e))


def test_adj2_pattern_bit_exact(case):
    assert case.adj2.nnz() == case.oadj2.nnz
    assert spm_equal(case.adj2, case.oadj2)


def test_adjoverlap_counts_and_pattern_bit_exact(case):
    from ocn_amd.utils import adjoverlap
    e = case.e.to(DEV)
    h1, h2 = adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e)
    assert h1.counts().cpu().tolist() == torch.bincount(case.ocn1.row, minlength=case.B).tolist()
    assert h2.counts().cpu().tolist() == torch.bincount(case.ocn2.row, minlength=case.B).tolist()
    assert spm_equal(h1.materialize(), case.ocn1)
    assert spm_equal(h2.materialize(), case.ocn2)
    assert h1.sizes() == [case.B, case.n]


def test_fused_flags_histograms(case):
    from ocn_amd.utils import CNState
    st = CNState(case.adj, case.adj, case.adj2, case.e.to(DEV))
    st.check_status()
    assert st.cnt1.cpu().tolist() == torch.bincount(case.ocn1.row, minlength=case.B).tolist()
    assert st.cnt2.cpu().tolist() == torch.bincount(case.ocn2.row, minlength=case.B).tolist()
    hist = st.hist_counts().cpu()
    assert hist[:, 0].tolist() == torch.bincount(case.ocn1.col, minlength=case.n).tolist()
    assert hist[:, 1].tolist() == torch.bincount(case.ocn2.col, minlength=case.n).tolist()
    union = torch.unique(torch.cat([O.spm2elem(case.ocn1), O.spm2elem(case.ocn2)]))
    assert hist[:, 2].tolist() == torch.bincount(union & 0xFFFFFFFF, minlength=case.n).tolist()
    assert spm_equal(st.materialize(1), case.ocn1) and spm_equal(st.materialize(2), case.ocn2)
    assert int(st.off[-1]) == int(case.oadj.rowcount()[case.e[0]].sum())


@pytest.mark.parametrize("H", [32, 64, 256, 48])
@pytest.mark.parametrize("ip", [0.0, 0.37])
def test_cn5_pool(case, H, ip):
    from ocn_amd.utils import CNState
    g = torch.Generator().manual_seed(case.seed)
    x = torch.randn(case.n, H, generator=g)
    xcn1, xcn2, aux = O.cn5_pool(x, case.ocn1, case.ocn2, torch.tensor([ip]))
    st = CNState(case.adj, case.adj, case.adj2, case.e.to(DEV))
    w = st.weights_cn5(torch.tensor([ip], device=DEV))
    g1, g2, gij = st.gather(w, x.to(DEV))
    # column weights: w1 = 1/S1 and — innerprod != 0 included, where S2 is summed entry by entry in the reference's
    # order (ocn_cn_colsum_exact) — inv2 = 1/S2 are the oracle's numbers bit for bit
    cols1 = torch.unique(case.ocn1.col)
    assert torch.equal(w[:, 0].cpu()[cols1], aux["inv1"][cols1])
    colsu = torch.unique(torch.cat([case.ocn1.col, case.ocn2.col]))
    assert torch.equal(w[:, 2].cpu()[colsu], (1 / aux["S2"])[colsu]), "S2 in the reference's summation order"
    assert torch.equal(gij.cpu(), x[case.e[0]] * x[case.e[1]])
    assert torch.equal(g1.cpu(), xcn1), "ncn1 pooling is order- and rounding-exact"
    if H % 4 == 0 and H in (16, 32, 64, 128, 256, 512):
        assert torch.equal(g2.cpu(), xcn2), "ncn2 pooling is order- and rounding-exact"
    else:
        assert close(g2, xcn2)


@pytest.mark.parametrize("sum_fill", [0.0, 1.0, 2.74])
def test_cn7_pool(case, sum_fill):
    from ocn_amd.utils import CNState
    H = 64
    x = torch.randn(case.n, H, generator=torch.Generator().manual_seed(7))
    xcn1, xcn2, _ = O.cn7_pool(x, case.ocn1, case.ocn2, sum_fill)
    st = CNState(case.adj, case.adj, case.adj2, case.e.to(DEV))
    g1, g2, _ = st.gather(st.weights_cn7(sum_fill), x.to(DEV))
    assert torch.equal(g1.cpu(), xcn1)
    assert torch.equal(g2.cpu(), xcn2)


@pytest.mark.parametrize("name,ln,tailact,two", [("cn5", True, True, False), ("cn5", False, False, True),
                                                 ("cn7", True, True, False)])
def test_predictor_scores(case, name, ln, tailact, two):
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    H = 64
    torch.manual_seed(case.seed)
    x = torch.randn(case.n, H)
    pred = predictor_dict[name](H, H, 1, 3, 0.05, 0.4, ln, use_xlin=True, tailact=tailact,
                                twolayerlin=two, beta=1.0).eval()
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    args = SimpleNamespace(sum=2.74)
    if name == "cn5":
        ref = O.cn5_forward(sd, x, case.ocn1, case.ocn2, case.e, ln, tailact, two)
    else:
        ref = O.cn7_forward(sd, x, case.ocn1, case.ocn2, case.e, args.sum, ln, tailact, two)
    e = case.e.to(DEV)
    with torch.no_grad():
        out = pred.to(DEV)(x.to(DEV), case.adj, adjoverlap(case.adj, case.adj, e),
                           adjoverlap(case.adj, case.adj2, e), e, args)
    assert out.shape == (case.B, 1) and out.dtype == torch.float32
    assert close(out, ref), (out.cpu() - ref).abs().max()


@pytest.mark.parametrize("name", ["cn5", "cn7"])
def test_backward_matches_oracle_autograd(case, name):
    """Training drop-in: gradients w.r.t. the embeddings and every used parameter against torch
    autograd through the oracle's op sequence (fp32 atomics: tolerance, not bits)."""
    if case.B > 4096:
        pytest.skip("oracle autograd at this size takes minutes")
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    H = 32
    torch.manual_seed(case.seed + 17)
    x = torch.randn(case.n, H)
    pred = predictor_dict[name](H, H, 1, 3, 0.0, 0.0, True).eval()       # eval + enable_grad: no dropout noise
    args = SimpleNamespace(sum=1.0)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in pred.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    ref = (O.cn5_forward(sd, xr, case.ocn1, case.ocn2, case.e, True) if name == "cn5"
           else O.cn7_forward(sd, xr, case.ocn1, case.ocn2, case.e, args.sum, True))
    wgt = torch.randn(case.B, 1, generator=torch.Generator().manual_seed(1))
    (ref * wgt).sum().backward()
    pred = pred.to(DEV)
    e = case.e.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = pred(xd, case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e, args)
    assert out.requires_grad and close(out, ref)
    (out * wgt.to(DEV)).sum().backward()
    scale = xr.grad.abs().max().item()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= 2e-5 * max(1.0, scale)
    for k, p in pred.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k      # xcnlin / xcn4lin are never used
            continue
        g = sd[k].grad
        assert (p.grad.cpu() - g).abs().max().item() <= 2e-5 * max(1.0, g.abs().max().item()), k


def test_heads_under_autograd_run_on_the_library_linear(case, monkeypatch):
    """With autograd on, the heads' Linear layers (forward and input gradient) go through ocn_linear_bf16x6
    (model._LinearFn); switching that off (torch modules) gives the same scores and gradients to rounding."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    H = 64
    torch.manual_seed(case.seed + 5)
    x = torch.randn(case.n, H, device=DEV)
    pred = predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    e = case.e.to(DEV)
    calls = {"fwd": 0, "bwd": 0}
    lin, lin_t = ops.linear, ops.linear_t
    monkeypatch.setattr(ops, "linear", lambda *a, **k: (calls.__setitem__("fwd", calls["fwd"] + 1), lin(*a, **k))[1])
    monkeypatch.setattr(ops, "linear_t", lambda *a, **k: (calls.__setitem__("bwd", calls["bwd"] + 1), lin_t(*a, **k))[1])
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "train_linear", on)
        xd = x.clone().requires_grad_(True)
        pred.zero_grad(set_to_none=True)
        out = pred(xd, case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e, None)
        out.square().sum().backward()
        res[on] = (out.detach(), xd.grad, {k: p.grad.clone() for k, p in pred.named_parameters() if p.grad is not None})
        if on:
            assert calls["fwd"] >= 9 and calls["bwd"] >= 9                   # 9 Linear(H, H) of the head, each way
    assert close(res[True][0], res[False][0])
    for a, b in [(res[True][1], res[False][1])] + [(res[True][2][k], res[False][2][k]) for k in res[False][2]]:
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())


def test_training_mode_updates_running_innerprod(case):
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    H = 16
    pred = predictor_dict["cn5"](H, H, 1, 3, 0.0).to(DEV).train()
    e = case.e.to(DEV)
    x = torch.randn(case.n, H, device=DEV, requires_grad=True)
    ip1 = O.cn5_batch_innerprod(case.ocn1, case.ocn2).item()
    out = pred.multidomainforward(x, case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e)
    out.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    assert pred.innerprod.item() == pytest.approx(ip1, rel=1e-5, abs=1e-6) and pred.n == 1
    pred.multidomainforward(x, case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e)
    assert pred.innerprod.item() == pytest.approx(ip1, rel=1e-5, abs=1e-6) and pred.n == 2   # mean of two equal batches


@pytest.mark.parametrize("conv", ["puregcn", "gin", "gcn", "puremean"])
def test_encoder_backward_matches_oracle_autograd(hiplib, conv):
    import ocn_amd.model as M
    n, H = 400, 32
    oadj = make_graph(n, 8, 60, 33, isolated=4)
    adj = to_product(oadj, DEV)
    torch.manual_seed(8)
    enc = M.GCN(H, H, H, 2, 0.0, True, True, -1, conv, True).eval()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in enc.state_dict().items()}
    x = torch.randn(n, H)
    xr = x.clone().requires_grad_(True)
    ref = O.gcn_forward(sd, xr, oadj, num_layers=2, conv_fn=conv, ln=True, res=True, jk=True)
    wgt = torch.randn(n, H, generator=torch.Generator().manual_seed(2))
    (ref * wgt).sum().backward()
    enc = enc.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = enc(xd, adj)
    assert close(out, ref, atol=2e-5, rtol=2e-5)
    (out * wgt.to(DEV)).sum().backward()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= 3e-5 * max(1.0, xr.grad.abs().max().item())
    for k, p in enc.named_parameters():
        g = sd[k].grad
        assert g is not None and (p.grad.cpu() - g).abs().max().item() <= 3e-5 * max(1.0, g.abs().max().item()), k


# ---- golden: hand-derived Appendix C --------------------------------------------------------
def test_appendix_c_on_gpu(hiplib):
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    g = json.load(open(os.path.join(GOLD, "appendix_c.json")))["path_graph"]
    und = torch.tensor(g["undirected_edges"]).t()
    adj = SparseTensor.from_edge_index(und.to(DEV), sparse_sizes=(4, 4)).to_symmetric()
    adj2 = product_adj2(adj)
    r, c, _ = adj2.coo()
    rows = {str(i): c[r == i].cpu().tolist() for i in range(4)}
    assert rows == g["a2_rows"]
    e = torch.tensor(g["batch"]).t().contiguous().to(DEV)
    st = CNState(adj, adj, adj2, e)
    assert st.cnt1.cpu().tolist() == g["cn1_counts"] and st.cnt2.cpu().tolist() == g["cn2_counts"]
    hc = st.hist_counts().cpu()
    assert hc[:, 0].tolist() == g["S1"] and hc[:, 1].tolist() == g["cn2_colsum"]
    eye = torch.eye(4, device=DEV).repeat(1, 4).contiguous()           # H = 16, h[k] = one-hot(k) x4
    for key, ip in (("cn5_innerprod_0", 0.0), ("cn5_innerprod_0.37", 0.37)):
        st = CNState(adj, adj, adj2, e)
        w = st.weights_cn5(torch.tensor([ip], device=DEV))
        _, g2, _ = st.gather(w, eye)
        dense = g2[:, :4].cpu()                                        # row e = ncn2[e, :]
        got = [dense[r_, c_].item() for r_, c_ in g["union_pattern"]]
        assert got == pytest.approx(g[key]["ncn2"], rel=2e-6, abs=1e-7)
    # Q2: a column hit by exactly one edge of the batch is zeroed in cn5
    st = CNState(adj, adj, adj2, torch.tensor([[0], [1]], device=DEV))
    w = st.weights_cn5(torch.tensor([0.0], device=DEV))
    assert w[2, 0].item() == 0.0
    g1, _, _ = st.gather(w, eye)
    assert g1.abs().max().item() == 0.0
    # pygho route, cn7 on it, cn6 (hand-derived, tests/golden/make_golden.py)
    from ocn_amd.utils import CNState3
    wk = g["walk_route"]
    st = CNState(adj, None, None, e, walk=True)
    assert st.cnt1.cpu().tolist() == g["cn1_counts"] and st.cnt2.cpu().tolist() == [len(r_) for r_ in wk["cn2_rows"]]
    assert st.hist_counts().cpu()[:, 3].tolist() == wk["walk_colsum"]
    x1, x2, _ = st.gather(st.weights_cn7(1.0), eye)
    assert torch.allclose(x1[:, :4].cpu(), torch.tensor(g["cn7_walk"]["xcn1"]), rtol=1e-6, atol=0)
    assert torch.equal(x2[:, :4].cpu(), torch.tensor(g["cn7_walk"]["xcn2"]))
    k6 = g["cn6_innerprod_0"]
    adj3 = SparseTensor.from_torch_sparse_coo_tensor(adj2.to_torch_sparse_coo_tensor() @ adj.to_torch_sparse_coo_tensor(), False)
    r3, c3, _ = adj3.coo()
    for r_, cols in k6["a3_rows"].items():
        assert c3[r3 == int(r_)].cpu().tolist() == cols
    st3 = CNState3(adj, adj2, adj3, e)
    assert st3.cnt3.cpu().tolist() == [len(r_) for r_ in k6["cn3_rows"]]
    x3 = st3.gather(*st3.weights(torch.zeros(1, device=DEV)), eye)[2]
    assert torch.allclose(x3[:, :4].cpu(), torch.tensor(k6["xcn3"]), rtol=1e-6, atol=0)


def test_oracle_vectors_regression(hiplib):
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    for rec in json.load(open(os.path.join(GOLD, "oracle_vectors.json"))):
        n, H = rec["n"], rec["H"]
        ei = torch.tensor(rec["edge_index"])
        adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n)).to_symmetric()
        adj2 = product_adj2(adj)
        assert adj2.nnz() == rec["a2_nnz"]
        e = torch.tensor(rec["batch"]).to(DEV)
        x = torch.randn(n, H, generator=torch.Generator().manual_seed(rec["x_seed"])).to(DEV)
        st = CNState(adj, adj, adj2, e)
        assert st.cnt1.cpu().tolist() == rec["cn1_counts"] and st.cnt2.cpu().tolist() == rec["cn2_counts"]
        g1, g2, _ = st.gather(st.weights_cn5(torch.tensor([0.0], device=DEV)), x)
        assert g1[0].cpu().tolist() == pytest.approx(rec["cn5_ip0.0"]["xcn1_row0"], rel=1e-6, abs=1e-6)
        assert g2[0].cpu().tolist() == pytest.approx(rec["cn5_ip0.0"]["xcn2_row0"], rel=1e-5, abs=1e-6)
        st = CNState(adj, adj, adj2, e)
        g1, g2, _ = st.gather(st.weights_cn7(2.74), x)
        assert g1[0].cpu().tolist() == pytest.approx(rec["cn7_sum2.74"]["xcn1_row0"], rel=1e-6, abs=1e-6)
        assert g2[0].cpu().tolist() == pytest.approx(rec["cn7_sum2.74"]["xcn2_row0"], rel=1e-6, abs=1e-6)
        if "a3_nnz" in rec and H in (16, 32, 64, 128, 256, 512):      # ocn_cn_gather3's widths
            from ocn_amd.utils import CNState3
            adj3 = SparseTensor.from_torch_sparse_coo_tensor(
                adj2.to_torch_sparse_coo_tensor() @ adj.to_torch_sparse_coo_tensor(), False)
            assert adj3.nnz() == rec["a3_nnz"]
            for ip in (0.0, 0.37):
                st3 = CNState3(adj, adj2, adj3, e)
                assert st3.cnt3.cpu().tolist() == rec["cn3_counts"]
                g = st3.gather(*st3.weights(torch.tensor([ip], device=DEV)), x)[2]
                want = rec[f"cn6_ip{ip}"]
                scale = max(1.0, max(abs(v) for v in want["xcn3_row0"]))
                assert g[0].cpu().tolist() == pytest.approx(want["xcn3_row0"], rel=1e-5, abs=(1e-6 if ip == 0.0 else 1e-5 * scale))


@pytest.mark.parametrize("n,avg,bs", [(300, 40, 64), (1000, 120, 256), (2100, 30, 1024)])
def test_block_route_adj2_on_the_integer_matrix_cores(hiplib, n, avg, bs):
    """utils.block_matrix_multiply (utils.py:287-323) as dense int8 MFMA block products: the offset-correct pattern equals
    the sparse A² (and the oracle's block loop), the bit rows match the CSR, and fold_quirk=True reproduces the
    reference's block-local accumulation as the oracle's switch does."""
    from ocn_amd.utils import block_matrix_multiply
    oadj = make_graph(n, avg, min(n - 1, 6 * avg), n + bs)
    adj = to_product(oadj, DEV)
    want = O.adj2_by_block(oadj, bs)
    got = block_matrix_multiply(adj, bs)
    assert got.nnz() == want.nnz and spm_equal(got, want)
    assert spm_equal(got, O.adj2_sparse(oadj)) and spm_equal(product_adj2(adj), want)
    bits = got._bitmap
    r, c, _ = got.coo()
    probe = (bits[r, c >> 5] >> (c & 31).to(torch.int32)) & 1
    assert bool(probe.all()) and int(torch.ops.aten.bitwise_and(bits, -1).ne(0).sum()) > 0
    pop = sum(((bits >> k) & 1).sum() for k in range(32))
    assert int(pop) == got.nnz()
    fold = block_matrix_multiply(adj, bs, fold_quirk=True)
    wantf = O.adj2_by_block(oadj, bs, fold_quirk=True)
    assert fold.nnz() == wantf.nnz and spm_equal(fold, wantf)


# ---- edge cases ---------------------------------------------------------------------------
def test_edge_cases(hiplib):
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import CNState, adjoverlap
    oadj = make_graph(200, 6, 50, 11, isolated=20)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    oadj2 = O.adj2_sparse(oadj)
    # empty batch
    e0 = torch.zeros(2, 0, dtype=torch.long, device=DEV)
    st = CNState(adj, adj, adj2, e0)
    g1, g2, gij = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), torch.randn(200, 32, device=DEV))
    assert g1.shape == (0, 32) and int(st.off[0]) == 0
    pred = predictor_dict["cn5"](32, 32, 1, 3, 0.0).to(DEV).eval()
    with torch.no_grad():
        out = pred(torch.randn(200, 32, device=DEV), adj, adjoverlap(adj, adj, e0), adjoverlap(adj, adj2, e0), e0)
    assert out.shape == (0, 1)
    # isolated endpoints, self pairs, duplicated edges, every edge identical
    e = torch.tensor([[199, 0, 5, 5, 5, 7, 190], [0, 199, 5, 9, 9, 7, 195]])
    st = CNState(adj, adj, adj2, e.to(DEV))
    c1, c2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e)
    assert st.cnt1.cpu().tolist() == torch.bincount(c1.row, minlength=7).tolist()
    assert st.cnt2.cpu().tolist() == torch.bincount(c2.row, minlength=7).tolist()
    x = torch.randn(200, 32)
    r1, r2, _ = O.cn5_pool(x, c1, c2, torch.tensor([0.0]))
    g1, g2, _ = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), x.to(DEV))
    assert close(g1, r1) and close(g2, r2)
    # out-of-range endpoint -> IndexError like the reference's index_select, not a GPU fault
    with pytest.raises(IndexError):
        CNState(adj, adj, adj2, torch.tensor([[0], [200]], device=DEV))
    # graph with no edges at all
    from ocn_amd.sparse import SparseTensor
    empty = SparseTensor.from_edge_index(torch.zeros(2, 0, dtype=torch.long, device=DEV), sparse_sizes=(50, 50))
    e2 = product_adj2(empty)
    assert e2.nnz() == 0
    st = CNState(empty, empty, e2, torch.tensor([[1, 2], [3, 4]], device=DEV))
    assert st.cnt1.cpu().tolist() == [0, 0]


def test_hub_rows_longer_than_a_wave(hiplib):
    """Star + clique: source rows much longer than 64 and target rows of thousands of entries."""
    from ocn_amd.utils import CNState
    n = 3000
    hub = torch.stack([torch.zeros(n - 1, dtype=torch.long), torch.arange(1, n)])
    ring = torch.stack([torch.arange(1, n - 1), torch.arange(2, n)])
    oadj = O.to_symmetric(O.from_edge_index(torch.cat([hub, ring], 1), n))
    oadj2 = O.adj2_sparse(oadj)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    assert spm_equal(adj2, oadj2)
    e = torch.tensor([[0, 0, 5, 17, 2999], [0, 9, 0, 18, 1]])
    st = CNState(adj, adj, adj2, e.to(DEV))
    c1, c2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e)
    assert st.cnt1.cpu().tolist() == torch.bincount(c1.row, minlength=5).tolist()
    assert st.cnt2.cpu().tolist() == torch.bincount(c2.row, minlength=5).tolist()
    x = torch.randn(n, 256)
    r1, r2, _ = O.cn5_pool(x, c1, c2, torch.tensor([0.0]))
    g1, g2, _ = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), x.to(DEV))
    assert torch.equal(g1.cpu(), r1)
    # rows 0 and 1 have the 2999-entry hub as source: a whole workgroup fetches their embedding rows, one lane group adds
    # them in ascending column order — the reference's sequential sum, bit for bit, like every short row
    assert int(adj.storage.rowcount()[0]) > 1024
    assert torch.equal(g2.cpu(), r2)
    for H in (32, 64, 128, 512):              # other lane-group widths of the hub-row kernel
        xh = torch.randn(n, H, generator=torch.Generator().manual_seed(H))
        for ip in (0.0, 0.6):
            q1, q2, _ = O.cn5_pool(xh, c1, c2, torch.tensor([ip]))
            st2 = CNState(adj, adj, adj2, e.to(DEV))
            p1, p2, _ = st2.gather(st2.weights_cn5(torch.tensor([ip], device=DEV)), xh.to(DEV))
            assert torch.equal(p1.cpu(), q1) and torch.equal(p2.cpu(), q2)


@pytest.mark.parametrize("H", [32, 64, 128, 256, 512])
def test_hub_rows_in_a_large_batch(hiplib, H):
    """Batches above 4096 rows take other hub-row kernels than the small ones: one wave per hub row for narrow
    embeddings (the ddi shape), a 256-thread workgroup for wide ones.  Two hubs (1500 and 2500 neighbours) among
    5000 candidates: the pooled vectors of every row equal the oracle's, bit for bit."""
    from ocn_amd.utils import CNState
    n, B = 4000, 5000
    g = torch.Generator().manual_seed(H)
    star_a = torch.stack([torch.zeros(2500, dtype=torch.long), torch.arange(2, 2502)])
    star_b = torch.stack([torch.ones(1500, dtype=torch.long), torch.arange(2400, 3900)])
    rnd = torch.randint(2, n, (2, 6000), generator=g)
    oadj = O.to_symmetric(O.from_edge_index(torch.cat([star_a, star_b, rnd], 1), n))
    oadj2 = O.adj2_sparse(oadj)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    e = torch.randint(0, n, (2, B), generator=g)
    e[0, ::7] = 0                                           # hub sources, spread over the batch
    e[0, 3::11] = 1
    st = CNState(adj, adj, adj2, e.to(DEV))
    assert int(adj.storage.rowcount()[:2].min()) > 1024 and st.B > 4096
    c1, c2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e)
    assert st.cnt1.cpu().tolist() == torch.bincount(c1.row, minlength=B).tolist()
    x = torch.randn(n, H, generator=g)
    for ip in (0.0, 0.45):
        r1, r2, _ = O.cn5_pool(x, c1, c2, torch.tensor([ip]))
        st = CNState(adj, adj, adj2, e.to(DEV))
        g1, g2, gx = st.gather(st.weights_cn5(torch.tensor([ip], device=DEV)), x.to(DEV))
        assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2)
        assert torch.equal(gx.cpu(), x[e[0]] * x[e[1]])


# ---- encoders -----------------------------------------------------------------------------
ENC = [  # cls, conv_fn, layers, in, hid, ln, res, jk, max_x
    ("GCN", "puregcn", 1, 40, 64, False, False, True, -1),
    ("GCN", "puregcn", 3, 40, 64, False, True, False, 500),
    ("GCN", "gin", 1, 128, 256, True, False, True, -1),
    ("GCN", "gcn", 2, 32, 32, True, True, True, -1),
    ("GCN", "puremean", 2, 16, 32, False, True, True, -1),
    ("GCN", "puremax", 1, 16, 32, False, False, False, -1),
    ("GCN2", "gcn", 1, 64, 64, True, False, True, 500),
    ("GCN3", "gcn", 5, 128, 32, True, True, True, -1),
]


@pytest.mark.parametrize("cls,conv,L,fin,hid,ln,res,jk,max_x", ENC, ids=lambda v: str(v))
def test_encoders(hiplib, cls, conv, L, fin, hid, ln, res, jk, max_x):
    import ocn_amd.model as M
    n = 500
    oadj = make_graph(n, 8, 80, 21, isolated=5)
    adj = to_product(oadj, DEV)
    torch.manual_seed(5)
    enc = getattr(M, cls)(fin, hid, hid, L, 0.1, ln, res, max_x, conv, jk, 0.0, xdropout=0.3,
                          taildropout=0.2).eval()
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    x = torch.randint(0, max_x + 1, (n,)) if max_x >= 0 else torch.randn(n, fin)
    ref = O.gcn_forward(sd, x, oadj, num_layers=L, conv_fn=conv, ln=ln, res=res, jk=jk, max_x=max_x,
                        variant={"GCN": 1, "GCN2": 2, "GCN3": 3}[cls])
    with torch.no_grad():
        out = enc.to(DEV)(x.to(DEV), adj)
    assert close(out, ref, atol=2e-5, rtol=2e-5), (out.cpu() - ref).abs().max()


# ---- size-independent properties at the benchmark shape -------------------------------------
def test_properties_collab_shape(hiplib):
    """ogbl-collab-shaped graph, B = 65536: sizes the oracle cannot finish in seconds, so the
    checks are invariants of the domain."""
    from ocn_amd.synth import dataset_like, sample_edges
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    ei, n, _ = dataset_like("collab", seed=0)
    adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n)).to_symmetric()
    adj2 = product_adj2(adj)
    r, c, _ = adj.coo()
    e = sample_edges(r.cpu(), c.cpu(), n, 65536, seed=1).to(DEV)
    st = CNState(adj, adj, adj2, e)
    st.check_status()
    hist = st.hist_counts()
    # histogram mass == per-edge counts; union between max and sum
    assert int(hist[:, 0].sum()) == int(st.cnt1.sum()) and int(hist[:, 1].sum()) == int(st.cnt2.sum())
    assert bool((hist[:, 2] <= hist[:, 0] + hist[:, 1]).all()) and bool((hist[:, 2] >= torch.maximum(hist[:, 0], hist[:, 1])).all())
    # cn1 is symmetric in (i, j); counts are invariant under a permutation of the batch
    st_sw = CNState(adj, adj, adj2, e.flip(0).contiguous())
    assert torch.equal(st_sw.cnt1, st.cnt1)
    perm = torch.randperm(e.shape[1], device=DEV)
    st_p = CNState(adj, adj, adj2, e[:, perm].contiguous())
    assert torch.equal(st_p.cnt1, st.cnt1[perm]) and torch.equal(st_p.cnt2, st.cnt2[perm])
    assert torch.equal(st_p.hist_counts(), hist)
    # A² contains A's 2-walk closure: diagonal present for every non-isolated node, symmetric nnz
    deg = adj.storage.rowcount()
    r2, c2, _ = adj2.coo()
    assert int((r2 == c2).sum()) == int((deg > 0).sum())
    # pooling is linear in h and its column weights sum to S1-normalised mass
    H = 256
    h1, h2 = torch.randn(n, H, device=DEV), torch.randn(n, H, device=DEV)
    w = st.weights_cn5(torch.zeros(1, device=DEV))
    a1, a2, _ = st.gather(w, h1)
    b1, b2, _ = st.gather(w, h2)
    s1, s2, _ = st.gather(w, h1 + h2)
    assert torch.allclose(s1, a1 + b1, atol=1e-4, rtol=1e-4) and torch.allclose(s2, a2 + b2, atol=1e-4, rtol=1e-4)
    ones = torch.ones(n, H, device=DEV)
    o1, o2, _ = st.gather(w, ones)
    # with h = 1: Σ_e xcn1[e] = #{c : S1[c] >= 2} and Σ_e xcn2[e] = #{c : colsum(cn2)[c] >= 1}
    k1, k2 = int((hist[:, 0] >= 2).sum()), int((hist[:, 1] >= 1).sum())
    assert abs(o1[:, 0].double().sum().item() - k1) <= 1e-4 * k1 + 1e-3
    assert abs(o2[:, 0].double().sum().item() - k2) <= 1e-4 * k2 + 1e-3


# ---- MLP-head glue kernels (floating point: plain torch fp32 reference) ----------------------
@pytest.mark.parametrize("H", [16, 64, 256, 512])
def test_rows_ln_relu_and_combine3(hiplib, H):
    from ocn_amd import ops
    torch.manual_seed(H)
    x = (torch.randn(1000, H) * 3 + 0.5).to(DEV)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    for relu in (False, True):
        ref = torch.nn.functional.layer_norm(x, (H,), g, b, 1e-5)
        ref = torch.relu(ref) if relu else ref
        got = ops.rows_ln_relu(x, g, b, 1e-5, relu)
        assert torch.allclose(got, ref, atol=2e-6, rtol=2e-6), (got - ref).abs().max()
    y = x.clone()
    assert ops.rows_ln_relu(y, g, b, 1e-5, True, inplace=True).data_ptr() == y.data_ptr()
    c = torch.tensor([0.73, 0.41, -1.3], device=DEV)
    a1, a2, a3 = torch.randn(3, 1000, H, device=DEV)
    assert torch.equal(ops.combine3(c, a1, a2, a3), c[0] * a1 + c[1] * a2 + c[2] * a3)


# ---- the pygho route: get_cn1_cn2 with walk-count values (ppa / citation2 drivers) -----------
def test_walk_route_counts_values_and_pools(case):
    from ocn_amd.utils import CNState, get_cn1_cn2
    oc1, oc2 = O.get_cn1_cn2(case.oadj, case.e)
    h1, h2 = get_cn1_cn2(case.adj, case.e.to(DEV))
    assert h1.counts().cpu().tolist() == torch.bincount(oc1.row, minlength=case.B).tolist()
    assert h2.counts().cpu().tolist() == torch.bincount(oc2.row, minlength=case.B).tolist()
    assert spm_equal(h1.materialize(), oc1)
    m2 = h2.materialize()
    assert spm_equal(m2, oc2) and m2.storage.value().cpu().tolist() == oc2.val.tolist()
    st = CNState(case.adj, None, None, case.e.to(DEV), walk=True)
    hc = st.hist_counts().cpu()
    assert hc[:, 0].tolist() == torch.bincount(oc1.col, minlength=case.n).tolist()
    assert hc[:, 1].tolist() == torch.bincount(oc2.col, minlength=case.n).tolist()
    assert hc[:, 3].tolist() == torch.zeros(case.n, dtype=torch.long).index_add_(0, oc2.col, oc2.val.long()).tolist()
    x = torch.randn(case.n, 64, generator=torch.Generator().manual_seed(3))
    for sum_fill in (0.0, 1.0):
        r1, r2, _ = O.cn7_pool(x, oc1, oc2, sum_fill)
        st = CNState(case.adj, None, None, case.e.to(DEV), walk=True)
        g1, g2, _ = st.gather(st.weights_cn7(sum_fill), x.to(DEV))
        assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2)
    for ip in (0.0, 0.37):
        r1, r2, _ = O.cn5_pool(x, oc1, oc2, torch.tensor([ip]))
        st = CNState(case.adj, None, None, case.e.to(DEV), walk=True)
        g1, g2, _ = st.gather(st.weights_cn5(torch.tensor([ip], device=DEV)), x.to(DEV))
        assert torch.equal(g1.cpu(), r1)
        assert torch.equal(g2.cpu(), r2), (g2.cpu() - r2).abs().max()     # valued cn2, innerprod != 0 included


def test_pygho_shim_scores_equal_the_direct_handles(case):
    """The drivers' own get_cn1_cn2 (restated in tests/test_shims.py), executed against shims/pygho on the GPU: same
    counts, same explicit matrices and bit-identical scores as ocn_amd.utils.get_cn1_cn2 — for both predictors of the
    pygho drivers — and a deferred vector touched as a tensor materialises the explicit matrix."""
    from tests.test_shims import pygho_namespace
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import get_cn1_cn2
    ns = pygho_namespace()
    adj = ns["wrap"](case.adj)
    e = case.e.to(DEV)
    s1, s2 = ns["get_cn1_cn2"](adj, e)
    d1, d2 = get_cn1_cn2(case.adj, e)
    assert torch.equal(s1.counts(), d1.counts()) and torch.equal(s2.counts(), d2.counts())
    oc1, oc2 = O.get_cn1_cn2(case.oadj, case.e)
    assert spm_equal(s1.materialize(), oc1) and spm_equal(s2.materialize(), oc2)
    torch.manual_seed(4)
    x = torch.randn(case.n, 64, device=DEV)
    for name in ("cn5", "cn7"):
        pred = predictor_dict[name](64, 64, 1, 3, 0.0, 0.0, True).to(DEV).eval()
        args = SimpleNamespace(sum=0.0)
        with torch.no_grad():
            via_shim = pred(x, adj, ns["get_cn1_cn2"](adj, e)[0], ns["get_cn1_cn2"](adj, e)[1], e, args)   # …_ppa.py:201: two calls
            direct = pred(x, case.adj, d1, d2, e, args)
        assert torch.equal(via_shim, direct)
    # the deferred vectors as tensors
    from pygho.backend.Spspmm import spsphadamard, spspmm
    Ei, Ej = adj.index_select([0], e[0].unsqueeze(0)), adj.index_select([0], e[1].unsqueeze(0))
    coo = spsphadamard(Ei, spspmm(Ej, 1, adj, 0)).to_torch_sparse_coo()
    row, col = coo.indices()
    assert row.cpu().tolist() == oc2.row.tolist() and col.cpu().tolist() == oc2.col.tolist()
    assert coo.values().cpu().tolist() == oc2.val.tolist()


def test_fold_quirk_switch_on_the_ddi_call(hiplib, monkeypatch):
    """ops.adj2_fold_quirk routes utils.sparse_tensor_multiply — the unchanged ddi command's call — to either reading of
    utils.py:318-321; both equal the oracle's."""
    from ocn_amd import ops
    from ocn_amd.utils import sparse_tensor_multiply
    oadj = make_graph(700, 60, 300, 77)
    adj = to_product(oadj, DEV)
    assert spm_equal(sparse_tensor_multiply(adj, 256), O.adj2_by_block(oadj, 256))
    monkeypatch.setattr(ops, "adj2_fold_quirk", True)
    assert spm_equal(sparse_tensor_multiply(adj, 256), O.adj2_by_block(oadj, 256, fold_quirk=True))


def _walk_raw(adj, e, nds):
    from ocn_amd import ops
    order, off, flags, wc, hist, c1, c2, status, _ = ops.cn_flags(
        adj._rowptr, adj._col, None, None, e[0].contiguous(), e[1].contiguous(), adj.size(1), adj.max_rowcount(),
        walk=True, nds=nds)
    n = int(off[-1])
    assert int(status[0]) == 0
    return flags[:n].cpu(), wc[:n].cpu(), ops.hist_counts(hist).cpu(), c1.cpu(), c2.cpu()


@pytest.mark.parametrize("hubs", [0, 3])
def test_walk_route_swept_from_either_endpoint(hiplib, hubs):
    """cn2[e,k] = #2-walks j -> m -> k can be enumerated from i's rows or from j's rows
    (ocn_hip.h: ocn_walk_rev_offsets): forward-only, reverse-everywhere (a doctored cost vector) and
    the cost-based mix give identical flags, walk counts, histograms and per-edge counts — and the mix
    equals the oracle."""
    n, B = 6000, 3000
    oadj = make_graph(n, 9, 300, 11, isolated=20)
    if hubs:    # hub nodes with thousands of neighbours (beyond the forward sweep's LDS set, within and beyond the reverse one's)
        g = torch.Generator().manual_seed(5)
        extra = [torch.stack([torch.full((d,), v), torch.randperm(n, generator=g)[:d]])
                 for v, d in zip(range(hubs), (5500, 2500, 900))]
        ei = torch.cat([torch.stack([oadj.row, oadj.col])] + extra, 1)
        ei = ei[:, ei[0] != ei[1]]
        oadj = O.to_symmetric(O.from_edge_index(ei, n))
    e = batch(oadj, B, 77)
    if hubs:    # hub -> leaf, leaf -> hub and hub -> hub candidates
        e[:, :40] = torch.stack([torch.arange(40) % hubs, torch.arange(100, 140)])
        e[:, 40:80] = torch.stack([torch.arange(200, 240), torch.arange(40) % hubs])
        e[:, 80:83] = torch.tensor([[0, 1, 2], [1, 2, 0]])
    adj = to_product(oadj, DEV)
    ed = e.to(DEV)
    nds = adj.neighbor_degree_sum()
    deg = oadj.rowcount()
    assert nds.cpu().tolist() == torch.zeros(n, dtype=torch.long).index_add_(0, oadj.row, deg[oadj.col]).tolist()
    fwd = _walk_raw(adj, ed, None)
    mix = _walk_raw(adj, ed, nds)
    forced = torch.zeros_like(nds); forced[ed[0]] = 1 << 50; forced[ed[1]] = 0     # src rows look expensive, dst rows free
    forced[ed[0][ed[0] == ed[1]]] = 0
    rev = _walk_raw(adj, ed, forced)
    for a, b, c in zip(fwd, mix, rev):
        assert torch.equal(a, b) and torch.equal(a, c)
    oc1, oc2 = O.get_cn1_cn2(oadj, e)
    assert mix[3].tolist() == torch.bincount(oc1.row, minlength=B).tolist()
    assert mix[4].tolist() == torch.bincount(oc2.row, minlength=B).tolist()
    assert mix[2][:, 3].tolist() == torch.zeros(n, dtype=torch.long).index_add_(0, oc2.col, oc2.val.long()).tolist()
    if hubs:    # the cost model does pick the reverse sweep for hub -> leaf candidates
        i, j = e[0], e[1]
        picks = (2 * nds.cpu()[j] + deg[i] * ((deg[j] + 15) // 16) + 2 * deg[i] < nds.cpu()[i])
        assert picks[:40][i[:40] < 2].all() and not picks[40:80].any()


@pytest.mark.parametrize("n_src,per", [(7, 150), (1, 1000), (40, 3)])
def test_walk_route_shared_source_sweep(hiplib, n_src, per):
    """Candidates sharing a source (the MRR layout: per source many negatives) are swept together
    (ocn_cn_walk_group): flags, walk counts, histograms and per-edge counts equal the per-candidate sweeps bit for
    bit, and the oracle's counts."""
    from ocn_amd import ops
    n = 4000
    oadj = make_graph(n, 9, 300, 17)
    adj = to_product(oadj, DEV)
    g = torch.Generator().manual_seed(n_src * 1000 + per)
    deg = oadj.rowcount()
    hubs = torch.argsort(deg, descending=True)[: max(n_src // 2, 1)]
    srcs = torch.cat([hubs, torch.randint(0, n, (n_src - hubs.numel(),), generator=g)])[:n_src]
    src = srcs.repeat_interleave(per)
    dst = torch.randint(0, n, (src.numel(),), generator=g)
    dst[::7] = oadj.col[torch.randint(0, oadj.nnz, (dst[::7].numel(),), generator=g)]       # some high-overlap targets
    perm = torch.randperm(src.numel(), generator=g)
    e = torch.stack([src[perm], dst[perm]])
    B = e.shape[1]
    assert B <= 4096
    ed = e.to(DEV)
    nds = adj.neighbor_degree_sum()
    keep = ops.walk_share_min
    try:
        ops.walk_share_min = 0
        ref = _walk_raw(adj, ed, nds)
        ops.walk_share_min = 2
        got = _walk_raw(adj, ed, nds)
    finally:
        ops.walk_share_min = keep
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    oc1, oc2 = O.get_cn1_cn2(oadj, e)
    assert got[3].tolist() == torch.bincount(oc1.row, minlength=B).tolist()
    assert got[4].tolist() == torch.bincount(oc2.row, minlength=B).tolist()
    assert got[2][:, 3].tolist() == torch.zeros(n, dtype=torch.long).index_add_(0, oc2.col, oc2.val.long()).tolist()


@pytest.mark.parametrize("name", ["cn5", "cn7"])
def test_walk_route_predictor_scores(case, name):
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import get_cn1_cn2
    H = 32
    torch.manual_seed(case.seed + 9)
    x = torch.randn(case.n, H)
    pred = predictor_dict[name](H, H, 1, 3, 0.0, 0.0, True, use_xlin=True, tailact=True).eval()
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    oc1, oc2 = O.get_cn1_cn2(case.oadj, case.e)
    args = SimpleNamespace(sum=1.0)
    ref = (O.cn5_forward(sd, x, oc1, oc2, case.e, True, True) if name == "cn5"
           else O.cn7_forward(sd, x, oc1, oc2, case.e, args.sum, True, True))
    e = case.e.to(DEV)
    c1, c2 = get_cn1_cn2(case.adj, e)
    with torch.no_grad():
        out = pred.to(DEV)(x.to(DEV), case.adj, c1, c2, e, args)
    assert close(out, ref), (out.cpu() - ref).abs().max()


# ---- bf16x6 MFMA Linear (floating point: plain torch fp32 reference, fp64 ground truth) --------
@pytest.mark.parametrize("M,K,N", [(1, 16, 32), (300, 64, 64), (1000, 256, 256), (4099, 128, 256), (257, 256, 128)])
def test_linear_bf16x6(hiplib, M, K, N):
    from ocn_amd import ops
    torch.manual_seed(M + N)
    x = (torch.randn(M, K) * 2).to(DEV)
    lin = torch.nn.Linear(K, N).to(DEV)
    ln = torch.nn.LayerNorm(N).to(DEV)
    with torch.no_grad():
        ln.weight.normal_(); ln.bias.normal_()
        out1 = torch.nn.Linear(N, 1).to(DEV)
        exact = (x.double() @ lin.weight.double().t() + lin.bias.double())
        ref32 = torch.nn.functional.linear(x, lin.weight, lin.bias)
        got = ops.linear(x, lin.weight, lin.bias)
        scale = exact.abs().max().item()
        err_mine, err_f32 = (got.double() - exact).abs().max().item(), (ref32.double() - exact).abs().max().item()
        assert err_mine <= max(2.0 * err_f32, 2e-7 * scale), (err_mine, err_f32)      # as accurate as an fp32 GEMM
        assert torch.allclose(got, ref32, atol=2e-6 * scale, rtol=1e-5)
        assert torch.equal(ops.linear(x, lin.weight, None, relu=True), torch.relu(ops.linear(x, lin.weight)))
        y = torch.relu(ln(ref32))
        assert torch.allclose(ops.linear(x, lin.weight, lin.bias, (ln.weight, ln.bias, ln.eps), True), y, atol=1e-5, rtol=1e-5)
        d = ops.linear(x, lin.weight, lin.bias, (ln.weight, ln.bias, ln.eps), True, (out1.weight, out1.bias))
        assert d.shape == (M, 1) and torch.allclose(d, out1(y), atol=2e-5, rtol=1e-5)
        lin.weight.mul_(2.0)                                                          # in-place update -> panel rebuilt
        assert torch.allclose(ops.linear(x, lin.weight, lin.bias), torch.nn.functional.linear(x, lin.weight, lin.bias),
                              atol=4e-6 * scale, rtol=1e-5)


def test_processing_order_is_a_permutation_grouped_by_source(hiplib):
    from ocn_amd import ops
    n, B = 100000, 50000
    src = torch.randint(0, n, (B,), device=DEV)
    order = torch.empty(B, dtype=torch.int64, device=DEV)
    # the workspace is handed over zeroed once and comes back zeroed (ocn_hip.h): two calls on the same buffer
    ws = torch.zeros(int(hiplib.ocn_order_workspace_bytes(n)) // 8 + 1, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.check(hiplib.ocn_order_by_node(ops.ptr(src), B, n, ops.ptr(order), ops.ptr(ws), ops.stream_ptr()), "order")
        assert torch.equal(torch.sort(order).values, torch.arange(B, device=DEV))
        s = src[order]
        assert bool((s[1:] >= s[:-1]).all())
        counters = ws[: (n * 4 + 15) // 16 * 2]                    # the per-node counters (int32[n], 16-byte padded)
        assert not bool(counters.any()) and not bool(ws[-(int(hiplib.ocn_scan_workspace_bytes(n)) // 8):].any())
        src = torch.randint(0, n, (B,), device=DEV)


def test_chained_scan_matches_cumsum(hiplib):
    """Inputs beyond one tile are scanned by one launch whose tiles chain through the workspace; it is left zero."""
    from ocn_amd import ops
    for n in (1, 2047, 16385, 100003, 3000017):
        cnt = torch.randint(0, 1000, (n,), dtype=torch.int32, device=DEV)
        ws = torch.zeros(int(hiplib.ocn_scan_workspace_bytes(n)) // 8 + 1, dtype=torch.int64, device=DEV)
        out = torch.empty(n + 1, dtype=torch.int64, device=DEV)
        for _ in range(2):
            ops.check(hiplib.ocn_scan_i32(ops.ptr(cnt), n, ops.ptr(out), ops.ptr(ws), ops.stream_ptr()), "scan")
            want = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(cnt.long(), 0)])
            assert torch.equal(out, want) and not bool(ws.any())


@pytest.mark.parametrize("H", [64, 256])
@pytest.mark.parametrize("ip", [0.37, 37.5, 2500.0, -3.0])
def test_cn5_scores_with_trained_innerprod(case, ip, H):
    """A trained checkpoint carries innerprod != 0 (running mean of Σ cn2 ⊙ ncn1 over training
    batches, typically 1e1..1e4): the orthogonalisation branch must hold the 1e-5 bar too."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    torch.manual_seed(case.seed + 3)
    x = torch.randn(case.n, H)
    pred = predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, True).eval()           # as the drivers build it
    with torch.no_grad():
        pred.innerprod.fill_(ip)
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    ref = O.cn5_forward(sd, x, case.ocn1, case.ocn2, case.e, True)
    e = case.e.to(DEV)
    with torch.no_grad():
        out = pred.to(DEV)(x.to(DEV), case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 1e-5 + 1e-5 * ref.abs().max().item(), err


@pytest.mark.parametrize("ip", [0.37, 250.0])
def test_cn5_trained_innerprod_collab_shape(hiplib, ip):
    """The same on a collab-shaped graph (scale 0.05: 11 793 nodes, H = 256, B = 8192): S2 bit-equal to the
    oracle's sequential index_add_, pooled vectors bit-equal, scores within 1e-5; columns with more than 64
    union entries (the workgroup sort path of ocn_cn_colsum_exact) are present."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like, sample_edges
    from ocn_amd.utils import CNState, adjoverlap
    ei, n, shape = dataset_like("collab", seed=0, scale=0.05)
    oadj = O.to_symmetric(O.from_edge_index(ei, n))
    oadj2 = O.adj2_sparse(oadj)
    adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n)).to_symmetric()
    adj2 = product_adj2(adj)
    H, B = 256, 8192
    e = sample_edges(oadj.row, oadj.col, n, B, seed=5)
    torch.manual_seed(11)
    x = torch.randn(n, H)
    oc1, oc2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e)
    xcn1, xcn2, aux = O.cn5_pool(x, oc1, oc2, torch.tensor([ip]))
    st = CNState(adj, adj, adj2, e.to(DEV))
    assert int(st.hist_counts()[:, 2].max()) > 64
    w = st.weights_cn5(torch.tensor([ip], device=DEV))
    colsu = torch.unique(torch.cat([oc1.col, oc2.col]))
    assert torch.equal(w[:, 2].cpu()[colsu], (1 / aux["S2"])[colsu])
    g1, g2, _ = st.gather(w, x.to(DEV))
    assert torch.equal(g1.cpu(), xcn1) and torch.equal(g2.cpu(), xcn2)
    pred = predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, True).eval()
    with torch.no_grad():
        pred.innerprod.fill_(ip)
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    ref = O.cn5_forward(sd, x, oc1, oc2, e, True)
    ed = e.to(DEV)
    with torch.no_grad():
        out = pred.to(DEV)(x.to(DEV), adj, adjoverlap(adj, adj, ed), adjoverlap(adj, adj2, ed), ed)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 1e-5 + 1e-5 * ref.abs().max().item(), err


def test_colsum_exact_on_a_column_longer_than_the_lds_sort(hiplib):
    """A star centre that is a common neighbour of every candidate: one column with B = 10 000 > 8192 union
    entries (in-memory sort path of ocn_cn_colsum_exact), next to ordinary short columns."""
    from ocn_amd.utils import CNState
    m, extra = 3000, 4000
    g = torch.Generator().manual_seed(77)
    star = torch.stack([torch.zeros(m, dtype=torch.long), torch.arange(1, m + 1)])
    rnd = torch.randint(1, m + 1, (2, extra), generator=g)
    n = m + 1
    oadj = O.to_symmetric(O.from_edge_index(torch.cat([star, rnd[:, rnd[0] != rnd[1]]], dim=1), n))
    oadj2 = O.adj2_sparse(oadj)
    B = 10000
    e = torch.randint(1, m + 1, (2, B), generator=g)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    oc1, oc2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e)
    x = torch.randn(n, 32, generator=g)
    for ip in (0.41, 9000.0):
        xcn1, xcn2, aux = O.cn5_pool(x, oc1, oc2, torch.tensor([ip]))
        st = CNState(adj, adj, adj2, e.to(DEV))
        assert int(st.hist_counts()[0, 2]) == B
        w = st.weights_cn5(torch.tensor([ip], device=DEV))
        colsu = torch.unique(torch.cat([oc1.col, oc2.col]))
        assert torch.equal(w[:, 2].cpu()[colsu], (1 / aux["S2"])[colsu])
        g1, g2, _ = st.gather(w, x.to(DEV))
        assert torch.equal(g1.cpu(), xcn1) and torch.equal(g2.cpu(), xcn2)


@pytest.mark.parametrize("name", ["cn5", "cn7"])
def test_predictors_accept_materialised_cn_matrices(case, name):
    """The reference's own adjoverlap / get_cn1_cn2 hand the predictor explicit [B, N] SparseTensors: those are
    converted to the flag form (CNState.from_materialized) and score exactly as the lazy handles do."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap, get_cn1_cn2
    H = 64
    torch.manual_seed(case.seed + 21)
    x = torch.randn(case.n, H, device=DEV)
    pred = predictor_dict[name](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    with torch.no_grad():
        pred.innerprod.fill_(0.8)
    e = case.e.to(DEV)
    args = SimpleNamespace(sum=2.74)
    with torch.no_grad():
        for h1, h2 in ((adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e)), get_cn1_cn2(case.adj, e)):
            want = pred(x, case.adj, h1, h2, e, args)
            got = pred(x, case.adj, h1.materialize(), h2.materialize(), e, args)
            assert torch.equal(got, want)
            mixed = pred(x, case.adj, h1, h2.materialize(), e, args)
            assert torch.equal(mixed, want)


# ---- valued, non-symmetric adjacencies (what DropAdj hands the encoder in training) -----------
@pytest.mark.parametrize("kind", ["puregcn", "gcnconv", "pureconv2", "sum"])
def test_valued_spmm_forward_and_transpose_backward(hiplib, kind):
    from ocn_amd import ops
    from ocn_amd.model import _spmm
    from ocn_amd.sparse import SparseTensor
    torch.manual_seed(4)
    n, F = 300, 32
    dense = (torch.rand(n, n) < 0.04).float() * (1.0 + torch.rand(n, n))      # valued, not symmetric
    dense.fill_diagonal_(0)
    r, c = dense.nonzero(as_tuple=True)
    adj = SparseTensor(row=r.to(DEV), col=c.to(DEV), value=dense[r, c].to(DEV), sparse_sizes=(n, n))
    x = torch.randn(n, F)
    xr = x.clone().requires_grad_(True)
    deg = dense.sum(1)
    nrm = torch.rsqrt(1 + deg)
    if kind == "puregcn":                                   # model.py:50-55 with a valued adj_t
        xs = nrm[:, None] * xr
        ref = nrm[:, None] * (dense @ xs + xs)
        kw = dict(pre="n", post="n", mode="sum", edge_scale=False, self_mode=1)
    elif kind == "gcnconv":                                 # D^-1/2 (A_w + I) D^-1/2
        a = dense + torch.eye(n)
        ref = (nrm[:, None] * a * nrm[None, :]) @ xr
        kw = dict(pre="n", mode="sum", edge_scale=True, self_mode=2)
    elif kind == "pureconv2":                               # (A_w ⊙ n nᵀ) x, no self term
        ref = (nrm[:, None] * dense * nrm[None, :]) @ xr
        kw = dict(pre="n", mode="sum", edge_scale=True)
    else:
        ref = dense @ xr
        kw = dict(mode="sum")
    nd = ops.deg_rsqrt(adj._rowptr, 1.0, val=adj._value)
    assert torch.allclose(nd.cpu(), nrm, atol=1e-6, rtol=1e-6)
    kw = {k: (nd if v == "n" else v) for k, v in kw.items()}
    w = torch.randn(n, F, generator=torch.Generator().manual_seed(1))
    (ref * w).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    out = _spmm(adj, xd, **kw)
    assert torch.allclose(out.cpu(), ref.detach(), atol=2e-5, rtol=2e-5)
    (out * w.to(DEV)).sum().backward()
    assert torch.allclose(xd.grad.cpu(), xr.grad, atol=2e-5, rtol=2e-5), (xd.grad.cpu() - xr.grad).abs().max()


def test_encoder_trains_through_dropadj(hiplib):
    import ocn_amd.model as M
    oadj = make_graph(400, 8, 60, 41)
    adj = to_product(oadj, DEV)
    torch.manual_seed(0)
    enc = M.GCN(32, 64, 64, 3, 0.1, False, True, 400, "puregcn", False, 0.5, taildropout=0.2).to(DEV).train()   # ddi-like
    out = enc(torch.arange(400, device=DEV), adj)
    out.square().mean().backward()
    g = enc.xemb[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max().item() > 0


def test_walk_route_training_innerprod(case):
    from ocn_amd.utils import CNState
    oc1, oc2 = O.get_cn1_cn2(case.oadj, case.e)
    ref = O.cn5_batch_innerprod(oc1, oc2).item()
    st = CNState(case.adj, None, None, case.e.to(DEV), walk=True)
    assert st.cn5_batch_innerprod().item() == pytest.approx(ref, rel=1e-5, abs=1e-5)
    st2 = CNState(case.adj, case.adj, case.adj2, case.e.to(DEV))
    assert st2.cn5_batch_innerprod().item() == pytest.approx(O.cn5_batch_innerprod(case.ocn1, case.ocn2).item(), rel=1e-5, abs=1e-5)


def test_bitmap_and_csr_search_of_adj2_agree(case):
    """A² is probed through dense bit rows when they fit the budget, otherwise searched as a CSR row
    (LDS sample + binary search): both must give the oracle's flags."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    assert case.adj2._bitmap is not None and case.adj2._bitmap.shape == (case.n, (case.n + 31) // 32)
    plain = SparseTensor(rowptr=case.adj2._rowptr, col=case.adj2._col, sparse_sizes=case.adj2.sparse_sizes())
    assert plain._bitmap is None
    a = CNState(case.adj, case.adj, case.adj2, case.e.to(DEV))
    b = CNState(case.adj, case.adj, plain, case.e.to(DEV))
    assert torch.equal(a.cnt2, b.cnt2) and torch.equal(a.cnt1, b.cnt1)
    assert torch.equal(a.hist, b.hist) and torch.equal(a.flags[: int(a.off[-1])], b.flags[: int(b.off[-1])])
    assert b.cnt2.cpu().tolist() == torch.bincount(case.ocn2.row, minlength=case.B).tolist()
    # the bit rows themselves: popcount per row == CSR row length
    bits = case.adj2._bitmap.view(torch.uint8)
    pop = torch.zeros(case.n, dtype=torch.int64, device=DEV)
    for s in range(8):
        pop += ((bits >> s) & 1).sum(dim=1)
    assert torch.equal(pop, case.adj2.storage.rowcount())


def test_bitmap_and_csr_search_of_adj_agree(case, monkeypatch):
    """cn1 membership: A itself is probed through dense bit rows when they fit ``ops.a1_bitmap_max_bytes`` (every small
    graph, ogbl-ddi), otherwise the target row is searched (LDS copy + binary search: collab and larger).  Both give
    the oracle's flags; the bit rows are the CSR pattern."""
    from ocn_amd import ops
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    e = case.e.to(DEV)
    with_bits = SparseTensor(rowptr=case.adj._rowptr, col=case.adj._col, sparse_sizes=case.adj.sparse_sizes())
    bits = with_bits.bit_rows()
    assert bits is not None and bits.shape == (case.n, (case.n + 31) // 32)
    dense = torch.zeros(case.n, bits.shape[1] * 32, dtype=torch.bool, device=DEV)
    r, c, _ = case.adj.coo()
    dense[r, c] = True
    got = ((bits.view(torch.uint8).unsqueeze(-1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).bool().reshape(case.n, -1)
    assert torch.equal(got, dense)
    a = CNState(case.adj, with_bits, case.adj2, e)
    monkeypatch.setattr(ops, "a1_bitmap_max_bytes", 0)
    searched = SparseTensor(rowptr=case.adj._rowptr, col=case.adj._col, sparse_sizes=case.adj.sparse_sizes())
    assert searched.bit_rows() is None
    b = CNState(case.adj, searched, case.adj2, e)
    assert torch.equal(a.cnt1, b.cnt1) and torch.equal(a.cnt2, b.cnt2) and torch.equal(a.hist, b.hist)
    assert torch.equal(a.flags[: int(a.off[-1])], b.flags[: int(b.off[-1])])
    assert b.cnt1.cpu().tolist() == torch.bincount(case.ocn1.row, minlength=case.B).tolist()


@pytest.mark.parametrize("ip", [0.0, 0.37])
def test_integration_md_ctypes_stub_runs_as_written(case, ip):
    """The ctypes stub of INTEGRATION.md §3 (what a maintainer would write against include/ocn_hip.h), executed
    verbatim: same counts and bit-identical pooled vectors as the library's own Python host."""
    import re
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import CNState
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = re.search(r"One candidate batch, cn5:\n\n```python\n(.*?)```", open(os.path.join(root, "INTEGRATION.md")).read(), re.S).group(1)
    H = 64
    torch.manual_seed(9)
    h = torch.randn(case.n, H, device=DEV)
    predictor = predictor_dict["cn5"](H, H, 1, 3, 0.0).to(DEV).eval()
    with torch.no_grad():
        predictor.innerprod.fill_(ip)
    e = case.e.to(DEV)
    env = dict(dev=DEV, B=case.B, N=case.n, H=H, max_deg=max(case.adj.max_rowcount(), 1), h=h, predictor=predictor,
               rowptrA=case.adj._rowptr, colA=case.adj._col, rowptrA2=case.adj2._rowptr, colA2=case.adj2._col,
               bitrowsA2=case.adj2._bitmap, src=e[0].contiguous(), dst=e[1].contiguous())
    cwd = os.getcwd()
    os.chdir(root)                                        # the stub opens "ocn_amd/libocn_hip.so"
    try:
        exec(compile(code, "INTEGRATION.md", "exec"), env)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    st = CNState(case.adj, case.adj, case.adj2, e)
    assert torch.equal(env["cnt1"], st.cnt1) and torch.equal(env["cnt2"], st.cnt2)
    x1, x2, xij = st.gather(st.weights_cn5(predictor.innerprod), h)
    assert torch.equal(env["out"][0], x1) and torch.equal(env["out"][1], x2) and torch.equal(env["out"][2], xij)


@pytest.mark.parametrize("name", ["cn5", "cn7"])
def test_two_phase_scoring_keeps_two_batches_in_flight(case, name):
    """predictor.begin / .finish (the pipelined loop of the sharded bench): the intersection pass of batch t + 1 is
    enqueued before batch t is finished, on alternating scratch sets — every batch scores exactly as forward() does."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    H = 64
    torch.manual_seed(case.seed + 3)
    x = torch.randn(case.n, H, device=DEV)
    pred = predictor_dict[name](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    args = SimpleNamespace(sum=0.7)
    g = torch.Generator().manual_seed(4)
    batches = [case.e.to(DEV)[:, torch.randperm(case.B, generator=g).to(DEV)][:, : max(case.B - 3 * q, 1)].contiguous() for q in range(4)]

    def handles(e):
        return adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e)

    with torch.no_grad():
        ref = [pred(x, case.adj, *handles(e), e, args).clone() for e in batches]
        outs, tok = [], pred.begin(x, case.adj, *handles(batches[0]), batches[0], slot=0)
        for t in range(len(batches)):
            nxt = pred.begin(x, case.adj, *handles(batches[t + 1]), batches[t + 1], slot=t + 1) if t + 1 < len(batches) else None
            outs.append(pred.finish(x, tok, args).clone())
            tok = nxt
    for a, b in zip(outs, ref):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        pred.begin(x, case.adj, *handles(batches[0]), batches[0])         # autograd on: not the scoring path


@pytest.mark.parametrize("name", ["cn5", "cn7"])
def test_two_stream_scoring_loop_equals_one_stream(case, name, monkeypatch):
    """pipeline.overlapped_steps / score_edges: phase A of batch t + 1 on a second HIP stream beside phase B of batch t,
    ordered by events (B of t after A of t; A of t + depth after B of t: they share a scratch set) — twelve ragged batches
    score exactly as on one stream and as forward() does, with two and with three batches in flight (the side streams are reused)."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.pipeline import overlapped_steps, score_edges
    from ocn_amd.utils import adjoverlap
    H = 64
    torch.manual_seed(case.seed + 5)
    x = torch.randn(case.n, H, device=DEV)
    pred = predictor_dict[name](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    if name == "cn5":
        pred.innerprod.fill_(0.37)             # a trained model: the order-exact column sums run in phase A, on the side stream
    args = SimpleNamespace(sum=0.7)
    g = torch.Generator().manual_seed(9)
    batches = [case.e.to(DEV)[:, torch.randperm(case.B, generator=g).to(DEV)][:, : max(case.B - 5 * q, 1)].contiguous() for q in range(12)]

    def handles(e):
        return adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e)

    def begin(it):
        return pred.begin(x, case.adj, *handles(batches[it]), batches[it], slot=it, args=args if it % 2 else None)

    with torch.no_grad():
        ref = [pred(x, case.adj, *handles(e), e, args).clone() for e in batches]
        for overlap, batch in ((True, None), (False, None), (True, 1), (True, None)):      # batch=1: the deep loop (three in flight)
            outs = [o.clone() for o in overlapped_steps(begin, lambda tok: pred.finish(x, tok, args), len(batches), overlap=overlap,
                                                        batch=batch)]
            torch.cuda.synchronize()
            for a, b in zip(outs, ref):
                assert torch.equal(a, b)
        edges = torch.cat(batches, 1).t().contiguous()              # [n, 2], the split_edge layout
        bs = max(case.B // 3, 1)
        monkeypatch.setattr(ops, "overlap_min_batch", 0)
        two = score_edges(pred, x, case.adj, case.adj2, edges, bs, args)
        monkeypatch.setattr(ops, "overlap_streams", False)
        one = score_edges(pred, x, case.adj, case.adj2, edges, bs, args)
    assert torch.equal(two, one)


def test_phase_a_carries_weights_class_order_and_schedule(hiplib):
    """The scoring loops' phase A at H = 256 with batches the pooling schedule applies to (multiples of 32, large enough to
    be processed in source order): begin() leaves column weights, class-major order and the longest-first schedule, finish()
    only pools and runs the heads — scores equal forward()'s bit for bit, fresh and trained model, with the extras in phase
    A and (OCN_PHASE_A_EXTRAS=0's path) in phase B."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.pipeline import overlapped_steps
    from ocn_amd.utils import adjoverlap
    n, H, B = 6000, 256, 4096
    oadj = make_graph(n, 14, 500, seed=21)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    torch.manual_seed(2)
    x = torch.randn(n, H, device=DEV)
    pred = predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    batches = [batch(oadj, B, 40 + q).to(DEV) for q in range(5)]

    def handles(e):
        return adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e)

    with torch.no_grad():
        for ip in (0.0, 0.37):
            pred.innerprod.fill_(ip)
            ref = [pred(x, adj, *handles(e), e).clone() for e in batches]
            for extras in (True, False):
                ops.phase_a_extras = extras
                try:
                    seen = []

                    def begin(it):
                        tok = pred.begin(x, adj, *handles(batches[it]), batches[it], slot=it)
                        seen.append((tok[2] is not None, getattr(tok[0], "_cls_decided", False), getattr(tok[0], "_sched_ready", False)))
                        return tok
                    outs = [o.clone() for o in overlapped_steps(begin, lambda tok: pred.finish(x, tok), len(batches), batch=B)]
                finally:
                    ops.phase_a_extras = True
                torch.cuda.synchronize()
                assert all(w for w, _, _ in seen) and all(c == extras and s == extras for _, c, s in seen), seen
                for a, b in zip(outs, ref):
                    assert torch.equal(a, b)


def test_order_sensitive_column_sum_known_answer_on_the_gpu(hiplib):
    """tests/golden/order_sensitive_colsum.json (hand-derived): five entries in one column whose fp32 sum is 2^25 only in
    ascending batch-row order (2^25 + 4 with the small entries first, or with one rounding of the exact sum)."""
    from ocn_amd.utils import CNState
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "order_sensitive_colsum.json")))
    oadj = O.to_symmetric(O.from_edge_index(torch.tensor(g["undirected_edges"]).t(), g["n"]))
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    e = torch.tensor(g["batch"]).t().contiguous()
    k = g["column"]
    ip = torch.tensor([g["innerprod"]], device=DEV)
    st = CNState(adj, adj, adj2, e.to(DEV))
    assert st.cnt1.cpu().tolist() == [1, 0, 0, 0, 1] and st.cnt2.cpu().tolist() == [0, 1, 1, 1, 0]
    w = st.weights_cn5(ip)
    assert w[k, 2].item() == 1.0 / g["S2_reference_order"] != 1.0 / g["S2_ones_first_or_rounded_once"]
    x = torch.randn(g["n"], 32, generator=torch.Generator().manual_seed(2))
    r1, r2, _ = O.cn5_pool(x, O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, O.adj2_sparse(oadj), e), ip.cpu())
    g1, g2, _ = st.gather(w, x.to(DEV))
    assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2)


@pytest.mark.parametrize("H", [16, 64, 256, 48])
def test_order_sensitive_pooling_known_answer_on_the_gpu(hiplib, H):
    """tests/golden/order_sensitive_pooling.json (hand-derived): 2^23 only in ascending column order — every pooling kernel
    layout (lane groups of 4 / 16 / 64 lanes, the generic width)."""
    from ocn_amd.utils import CNState
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "order_sensitive_pooling.json")))
    oadj = O.to_symmetric(O.from_edge_index(torch.tensor(g["undirected_edges"]).t(), g["n"]))
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    e = torch.tensor(g["batch"]).t().contiguous().to(DEV)
    x = torch.zeros(g["n"], H, device=DEV)
    for k, v in g["x_rows"].items():
        x[int(k)] = v
    for st in (CNState(adj, adj, adj2, e), CNState(adj, None, None, e, walk=True)):
        x1, _, _ = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), x)
        assert x1.unique().tolist() == [g["xcn1_ascending_column_order"]]


@pytest.mark.parametrize("H,B", [(32, 2), (64, 2), (256, 2), (64, 5000), (256, 5000)])
def test_order_sensitive_pooling_of_a_hub_row(hiplib, H, B):
    """The hub-row kernels against a hand-derived value: source 0 and target 1 share 1500 neighbours (2..1501), the pair
    is scored twice (weight 1/2 per column), x[2] = 2^24 and x[k] = 1 elsewhere.  Added in ascending column order the
    pooled value is 2^23 — each later 0.5 is absorbed (ties to even); ANY order that adds some of the small terms
    together first (segments, partial sums per lane group) ends above it.  B = 2: workgroup kernel of small batches;
    B = 5000: one wave per hub row (H <= 64) / the 256-thread workgroup kernel."""
    from ocn_amd.utils import CNState
    n, m = 1502, 1500
    nb = torch.arange(2, 2 + m)
    ei = torch.cat([torch.stack([torch.zeros(m, dtype=torch.long), nb]), torch.stack([torch.ones(m, dtype=torch.long), nb])], 1)
    oadj = O.to_symmetric(O.from_edge_index(ei, n))
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    e = torch.tensor([[0, 0], [1, 1]])
    if B > 2:                                               # filler candidates (2, 3): their entries are in columns 0 and 1 only
        e = torch.cat([e, torch.tensor([[2], [3]]).expand(2, B - 2)], 1)
    e = e.contiguous().to(DEV)
    x = torch.ones(n, H, device=DEV)
    x[2] = 2.0 ** 24
    st = CNState(adj, adj, adj2, e)
    assert int(adj.storage.rowcount()[0]) == m > 1024 and st.cnt1[:2].cpu().tolist() == [m, m]
    x1, _, _ = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), x)
    assert x1[:2].unique().tolist() == [2.0 ** 23]


@pytest.mark.parametrize("F", [16, 64, 128, 256])
def test_order_sensitive_spmm_known_answer(hiplib, F):
    """The encoders' SpMM (spmm_add of model.py:42-55) against hand-derived values: row 0 has the neighbours 1 < 2 < 3 < 4 with
    x[1] = 2^24 and x[2..4] = 1.  Ascending column order: 2^24 + 1 = 2^24 three times -> 2^24 (descending: 3 + 2^24 ->
    2^24 + 4).  With the self term of PureConv added AFTER the neighbours (y = A x + x, x[0] = 1): still 2^24."""
    from ocn_amd import ops
    rowptr = torch.tensor([0, 4, 4, 4, 4, 4], dtype=torch.int64, device=DEV)
    col = torch.tensor([1, 2, 3, 4], dtype=torch.int32, device=DEV)
    x = torch.ones(5, F, device=DEV)
    x[1] = 2.0 ** 24
    y = ops.spmm_csr(rowptr, col, x)
    assert y[0].unique().tolist() == [2.0 ** 24] and y[1:].abs().max().item() == 0.0
    y = ops.spmm_csr(rowptr, col, x, self_mode=1)                       # neighbours first, then the row's own x
    assert y[0].unique().tolist() == [2.0 ** 24]
    acc = torch.tensor(0.0)
    for v in (1.0, 1.0, 1.0, 2.0 ** 24):                                # what a descending sum would give
        acc = acc + v
    assert acc.item() == 2.0 ** 24 + 4


def test_block_route_known_answers_on_a_path(hiplib):
    """The ddi block route (int8 MFMA blocks -> bit rows -> CSR) against patterns derived by hand for a path graph: A^2 has
    (i, i) and (i, i +- 2); with the reference's block-local indices (fold_quirk) every entry lands on
    (i % block, j % block).  70 nodes, 32-wide blocks: a ragged last block and entries that cross block borders."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import block_matrix_multiply
    n, bs = 70, 32
    ei = torch.stack([torch.arange(n - 1), torch.arange(1, n)]).to(DEV)
    adj = SparseTensor.from_edge_index(ei, sparse_sizes=(n, n)).to_symmetric()
    full = {(i, i) for i in range(n)} | {(i, i + 2) for i in range(n - 2)} | {(i + 2, i) for i in range(n - 2)}
    fold = {(i % bs, j % bs) for i, j in full}
    for quirk, want in ((False, full), (True, fold)):
        a2 = block_matrix_multiply(adj, bs, fold_quirk=quirk)
        r, c, _ = a2.coo()
        assert set(zip(r.cpu().tolist(), c.cpu().tolist())) == want


def test_complete_bipartite_closed_forms(hiplib):
    """K_{a,b} with a = 1100, b = 1500 (every row longer than 1024: the hub paths of the intersection kernels): every
    count has a closed form, no oracle involved.  Same-side pair (i, j in A): cn1 = B (b entries), cn2 = empty.  Cross pair
    (i in A, j in B): cn1 = empty; N2(j) = B, so cn2 = N(i) = B (b entries), and on the walk route each of them is reached by
    |N(k) n N(j)| = a walks.  Column histograms follow by counting the pairs."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    a, b = 1100, 1500
    n = a + b
    A, Bs = torch.arange(a), torch.arange(a, n)
    ei = torch.stack([A.repeat_interleave(b), Bs.repeat(a)]).to(DEV)
    adj = SparseTensor.from_edge_index(ei, sparse_sizes=(n, n)).to_symmetric()
    adj2 = product_adj2(adj)
    g = torch.Generator().manual_seed(5)
    n_same, n_cross = 37, 53
    same = torch.stack([torch.randint(0, a, (n_same,), generator=g), torch.randint(0, a, (n_same,), generator=g)])
    cross = torch.stack([torch.randint(0, a, (n_cross,), generator=g), torch.randint(a, n, (n_cross,), generator=g)])
    e = torch.cat([same, cross], 1)[:, torch.randperm(n_same + n_cross, generator=g)].contiguous()
    is_cross = e[1] >= a
    want1 = torch.where(is_cross, 0, b).tolist()
    want2 = torch.where(is_cross, b, 0).tolist()
    for st in (CNState(adj, adj, adj2, e.to(DEV)), CNState(adj, None, None, e.to(DEV), walk=True)):
        assert st.cnt1.cpu().tolist() == want1 and st.cnt2.cpu().tolist() == want2
        hc = st.hist_counts().cpu()                       # [N, 4] = n1, n2, n_union, walks
        assert hc[:a].abs().max().item() == 0             # every source is in A: all entries are columns of B
        assert hc[a:, 0].unique().tolist() == [n_same] and hc[a:, 1].unique().tolist() == [n_cross]
        assert hc[a:, 2].unique().tolist() == [n_same + n_cross]
        if st.walk:
            assert hc[a:, 3].unique().tolist() == [n_cross * a]
            wc = st.wc[: int(st.off[-1])].view(-1, b).cpu()   # every source has exactly the b neighbours of side B
            assert torch.equal(wc, torch.where(is_cross, a, 0).view(-1, 1).expand(-1, b).to(wc.dtype))


@pytest.mark.parametrize("H", [64, 256])
def test_complete_bipartite_pooled_vectors_in_closed_form(hiplib, H):
    """The CN stage end to end without the oracle: K_{1100,1500}, 32 same-side and 64 cross pairs, integer-valued embeddings.
    Every weight is a power of two (1/32 = 1/S1, 1/64 = 1/S2 of a fresh cn5) and every partial sum an exact fp32 number, so
    xcn1 = (sum of h over side B) / 32 on same-side rows, xcn2 = (the same sum) / 64 on cross rows (pattern route, cn5) or
    1100 x that sum (walk route, cn7: raw walk counts), zero elsewhere; x_i * x_j exactly.  All rows are hub rows."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    a, b, n_same, n_cross = 1100, 1500, 32, 64
    n = a + b
    ei = torch.stack([torch.arange(a).repeat_interleave(b), torch.arange(a, n).repeat(a)]).to(DEV)
    adj = SparseTensor.from_edge_index(ei, sparse_sizes=(n, n)).to_symmetric()
    adj2 = product_adj2(adj)
    g = torch.Generator().manual_seed(H)
    same = torch.stack([torch.randint(0, a, (n_same,), generator=g), torch.randint(0, a, (n_same,), generator=g)])
    cross = torch.stack([torch.randint(0, a, (n_cross,), generator=g), torch.randint(a, n, (n_cross,), generator=g)])
    e = torch.cat([same, cross], 1)[:, torch.randperm(n_same + n_cross, generator=g)].contiguous().to(DEV)
    is_cross = (e[1] >= a).view(-1, 1)
    h = torch.randint(-3, 4, (n, H), generator=g).float().to(DEV)
    sum_b = h[a:].double().sum(0).float().view(1, H)                   # |sum| <= 4500: exact in fp32 in any order
    zero = torch.zeros(1, H, device=DEV)
    st = CNState(adj, adj, adj2, e)
    x1, x2, xij = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), h)
    assert torch.equal(x1, torch.where(is_cross, zero, sum_b / 32)) and torch.equal(x2, torch.where(is_cross, sum_b / 64, zero))
    assert torch.equal(xij, h[e[0]] * h[e[1]])
    st = CNState(adj, None, None, e, walk=True)
    x1, x2, _ = st.gather(st.weights_cn7(0.5), h)
    assert torch.equal(x1, torch.where(is_cross, zero, sum_b / 32)) and torch.equal(x2, torch.where(is_cross, sum_b * a, zero))


@pytest.mark.parametrize("H", [64, 256])
def test_large_batch_pooled_vectors_in_closed_form(hiplib, H):
    """The large-batch path (one-launch prep, processing order, slot records, packed pooling kernel) in closed form:
    K_{300,500} (rows below the hub threshold), 16384 same-side and 16384 cross pairs -> weights 2^-14, integer embeddings,
    every partial sum exact."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState
    a, b, half = 300, 500, 16384
    n = a + b
    ei = torch.stack([torch.arange(a).repeat_interleave(b), torch.arange(a, n).repeat(a)]).to(DEV)
    adj = SparseTensor.from_edge_index(ei, sparse_sizes=(n, n)).to_symmetric()
    adj2 = product_adj2(adj)
    g = torch.Generator().manual_seed(H + 1)
    same = torch.stack([torch.randint(0, a, (half,), generator=g), torch.randint(0, a, (half,), generator=g)])
    cross = torch.stack([torch.randint(0, a, (half,), generator=g), torch.randint(a, n, (half,), generator=g)])
    e = torch.cat([same, cross], 1)[:, torch.randperm(2 * half, generator=g)].contiguous().to(DEV)
    is_cross = (e[1] >= a).view(-1, 1)
    h = torch.randint(-3, 4, (n, H), generator=g).float().to(DEV)
    sum_b = h[a:].double().sum(0).float().view(1, H)
    zero = torch.zeros(1, H, device=DEV)
    st = CNState(adj, adj, adj2, e)
    assert st.order is not None and st.rec is not None
    assert st.cnt1.tolist() == torch.where(is_cross.view(-1), 0, b).tolist()
    x1, x2, xij = st.gather(st.weights_cn5(torch.zeros(1, device=DEV)), h)
    assert torch.equal(x1, torch.where(is_cross, zero, sum_b / half)) and torch.equal(x2, torch.where(is_cross, sum_b / half, zero))
    assert torch.equal(xij, h[e[0]] * h[e[1]])


def test_eval_caches_follow_parameter_updates(case):
    """The eval fast path caches weight panels and mix coefficients; optimiser-style in-place updates
    (version bump) and .data edits followed by a mode switch must both be seen."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    H = 32
    torch.manual_seed(1)
    x = torch.randn(case.n, H, device=DEV)
    pred = predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    e = case.e.to(DEV)

    def run():
        with torch.no_grad():
            return pred(x, case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e)

    def run_modules():                       # autograd on -> plain torch modules, no caches involved
        return pred(x, case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e).detach()

    assert close(run(), run_modules())
    with torch.no_grad():
        pred.alpha.add_(0.7); pred.beta.mul_(-2.0); pred.xcn1lin[0].weight.mul_(1.5); pred.lin[8].weight.add_(0.1)
    assert close(run(), run_modules())
    pred.xijlin[0].weight.data.mul_(0.5)     # bypasses the version counter ...
    pred.train(); pred.eval()                # ... but every pass starts with a mode switch
    assert close(run(), run_modules())


# ---- cn6: the 3-hop predictor (model.py:2445-2951) ------------------------------
@pytest.fixture(scope="module")
def case3(hiplib):
    from ocn_amd.sparse import SparseTensor
    n, B = 1500, 1200
    oadj = make_graph(n, 5, 60, 31, isolated=7)
    oadj2 = O.adj2_sparse(oadj)
    oadj3 = O.adj3_sparse(oadj, oadj2)
    e = batch(oadj, B, 91)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    adj3 = SparseTensor.from_torch_sparse_coo_tensor(adj2.to_torch_sparse_coo_tensor() @ adj.to_torch_sparse_coo_tensor(), False)
    return SimpleNamespace(n=n, B=B, oadj=oadj, oadj2=oadj2, oadj3=oadj3, e=e, adj=adj, adj2=adj2, adj3=adj3,
                           ocn=[O.adjoverlap(oadj, t, e) for t in (oadj, oadj2, oadj3)])


def test_cn6_counts_and_pools(case3):
    from ocn_amd.utils import CNState3, adjoverlap
    c = case3
    assert c.adj3.nnz() == c.oadj3.nnz and spm_equal(c.adj3, c.oadj3)
    ed = c.e.to(DEV)
    h3 = adjoverlap(c.adj, c.adj3, ed)
    assert h3.counts().cpu().tolist() == torch.bincount(c.ocn[2].row, minlength=c.B).tolist()
    assert spm_equal(h3.materialize(), c.ocn[2])
    x = torch.randn(c.n, 64, generator=torch.Generator().manual_seed(2))
    for ip in (0.0, 0.37):
        r1, r2, r3, aux = O.cn6_pool(x, *c.ocn, torch.tensor([ip]))
        st = CNState3(c.adj, c.adj2, c.adj3, ed)
        assert st.cnt3.cpu().tolist() == torch.bincount(c.ocn[2].row, minlength=c.B).tolist()
        wa, wb, nip = st.weights(torch.tensor([ip], device=DEV))
        assert nip.item() == pytest.approx(float(aux["nip"]), rel=1e-6, abs=1e-12)
        g1, g2, g3, gx = st.gather(wa, wb, nip, x.to(DEV))
        assert torch.equal(gx.cpu(), x[c.e[0]] * x[c.e[1]])
        # every pooled vector in the oracle's order, bit for bit (innerprod != 0: S2 / S3 summed entry by entry)
        assert torch.equal(wa[:, 2].cpu()[torch.unique(torch.cat([c.ocn[0].col, c.ocn[1].col]))],
                           (1 / aux["S2"])[torch.unique(torch.cat([c.ocn[0].col, c.ocn[1].col]))])
        colsu3 = torch.unique(torch.cat([o.col for o in c.ocn]))
        assert torch.equal(wb[:, 0].cpu()[colsu3], (1 / aux["S3"])[colsu3])
        assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2) and torch.equal(g3.cpu(), r3)


@pytest.mark.parametrize("H,ln", [(32, True), (256, True), (64, False)])
def test_cn6_scores(case3, H, ln):
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    c = case3
    torch.manual_seed(13)
    x = torch.randn(c.n, H)
    pred = predictor_dict["cn6"](H, H, 1, 3, 0.1, 0.0, ln, use_xlin=True, tailact=True, beta=0.7).eval()
    with torch.no_grad():
        pred.alpha.copy_(torch.tensor([0.3, -0.2, 0.9]))
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    assert "xcn3lin.0.weight" in sd and not any(k.startswith("xcn4lin") for k in sd)
    ref = O.cn6_forward(sd, x, *c.ocn, c.e, ln, True)
    ed = c.e.to(DEV)
    with torch.no_grad():
        out = pred.to(DEV)(x.to(DEV), c.adj, adjoverlap(c.adj, c.adj, ed), adjoverlap(c.adj, c.adj2, ed),
                           adjoverlap(c.adj, c.adj3, ed), ed, None)
    assert out.shape == (c.B, 1) and close(out, ref), (out.cpu() - ref).abs().max()
    with pytest.raises(NotImplementedError):
        pred.train()(x.to(DEV), c.adj, adjoverlap(c.adj, c.adj, ed), adjoverlap(c.adj, c.adj2, ed),
                     adjoverlap(c.adj, c.adj3, ed), ed, None)


# ---- heads on class-major rows: skipping the all-zero pooled rows changes nothing ------------------
@pytest.mark.parametrize("name,H,tailact,two", [("cn5", 256, True, False), ("cn7", 64, False, False), ("cn5", 128, True, True)])
def test_zero_row_skipping_is_bitwise_neutral(hiplib, name, H, tailact, two):
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import CNState, adjoverlap
    n, B = 6000, 5003                                       # ragged: B is no multiple of the 128-row tile
    oadj = make_graph(n, 6, 80, 41, isolated=30)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    e = batch(oadj, B, 17).to(DEV)
    st = CNState(adj, adj, adj2, e)
    order2, inv, r = ops.class_order(st.cnt1, st.cnt2, st.order)
    c1, c2 = st.cnt1.cpu() > 0, st.cnt2.cpu() > 0
    cls = c1.long() * 2 + c2.long()
    n3, n2, n1 = [(cls == c).sum().item() for c in (3, 2, 1)]
    assert n3 > 0 and n1 > 0 and B - n3 - n2 - n1 > 0
    assert r.cpu().tolist() == [[0, n3 + n2], [0, n3], [n3 + n2, n3 + n2 + n1], [0, n3 + n2 + n1], [n3 + n2 + n1, B],
                                [n3, n3 + n2], [0, B]]
    o = order2.cpu()
    assert sorted(o.tolist()) == list(range(B)) and torch.equal(inv.cpu()[o], torch.arange(B))
    assert cls[o].tolist() == sorted(cls.tolist(), reverse=True)            # class-major ...
    where = torch.empty(B, dtype=torch.long)                                 # position in the incoming (source-sorted) order
    where[st.order.cpu()] = torch.arange(B)
    for c in (3, 1, 0):                                                      # ... and stable inside a class
        pos = where[o[cls[o] == c]]
        assert torch.equal(pos, pos.sort().values)
    torch.manual_seed(3)
    x = torch.randn(n, H, device=DEV)
    pred = predictor_dict[name](H, H, 1, 3, 0.1, 0.0, True, use_xlin=True, tailact=tailact, twolayerlin=two, beta=0.6).to(DEV).eval()
    if two:
        pred.innerprod.fill_(7.5)        # trained cn5: cn1-only entries then carry weight in xcn2
    args = SimpleNamespace(sum=1.3)
    outs = []
    for skip in (False, True):
        ops.skip_zero_rows = skip
        try:
            with torch.no_grad():
                outs.append(pred(x, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, args))
        finally:
            ops.skip_zero_rows = True
    assert outs[0].shape == (B, 1) and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("H", [128, 256])
@pytest.mark.parametrize("ln", [True, False])
def test_fused_heads_every_width(hiplib, H, ln, monkeypatch):
    """ocn_heads_fused (one launch for the whole head; the product takes it from ``ops.fused_heads_min_width`` up) at
    every width it is built for, class ranges on (ragged batch, all four classes present): against the grouped launches
    and against the torch modules."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    n, B = 3000, 2311
    oadj = make_graph(n, 5, 60, 43, isolated=20)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    e = batch(oadj, B, 19).to(DEV)
    torch.manual_seed(5)
    x = torch.randn(n, H, device=DEV)
    pred = partial(predictor_dict["cn5"], cndeg=-1)(H, H, 1, 3, 0.0, 0.0, ln).to(DEV).eval()     # the drivers' head
    with torch.no_grad():
        pred.innerprod.fill_(0.5)

    def run():
        with torch.no_grad():
            return pred(x, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, None)

    monkeypatch.setattr(ops, "fused_heads_min_width", 0)
    assert pred._fused_plan(H) is not None
    fused = run()
    monkeypatch.setattr(ops, "fused_heads", False)
    grouped = run()
    modules = pred(x, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, None).detach()   # autograd on: plain torch
    assert fused.shape == (B, 1)
    assert close(fused, modules), (fused - modules).abs().max()
    assert close(grouped, modules), (grouped - modules).abs().max()


@pytest.mark.parametrize("H", [128, 256])
@pytest.mark.parametrize("ln", [True, False])
@pytest.mark.parametrize("on_union", [True, False])
def test_heads_small_batch_kernel_is_bit_equal(hiplib, H, ln, on_union):
    """ocn_heads_fused has two forms (heads.hip): 128 candidates per workgroup with a candidate's eight layers on one wave,
    and — up to ``ops.heads_small_batch()`` candidates — 32 per workgroup with every layer's output features split over the
    four waves.  Same panels, same k order, same sums: the scores must be the same BITS, whatever the batch size, with every
    class boundary inside a tile, with rows of mixed magnitude and all-zero rows, and with a destination row map."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    torch.manual_seed(31 + H)
    pred = partial(predictor_dict["cn5"], cndeg=-1)(H, H, 1, 3, 0.0, 0.0, ln).to(DEV).eval()
    with torch.no_grad():
        for p in pred.parameters():
            p.mul_(1.0 + 0.5 * torch.rand_like(p))
    pack = pred._fused_pack(H, DEV)
    scratch = ops.buf(pred._ws, "heads_scratch", int(ops._lib.lib().ocn_heads_scratch_bytes(H)) // 4, torch.float32, DEV)
    prev = ops.heads_small_batch()
    try:
        for B, (n3, n2, n1) in ((1, (1, 0, 0)), (31, (5, 9, 3)), (32, (0, 0, 0)), (33, (33, 0, 0)), (257, (100, 0, 57)),
                                (1152, (300, 211, 97)), (4099, (1, 70, 2000)), (9000, (4000, 33, 31)), (40000, (9000, 7001, 5003))):
            x1, x2, xij = (torch.randn(B, H, device=DEV) for _ in range(3))
            x1[::3] *= 1e-3
            x2[:, ::5] *= 1e3
            x1[B // 2].zero_()
            xij[B // 3].zero_()
            r = torch.tensor([[0, n3 + n2], [0, n3], [n3 + n2, n3 + n2 + n1], [0, n3 + n2 + n1], [n3 + n2 + n1, B], [n3, n3 + n2], [0, B]],
                             dtype=torch.int64, device=DEV)
            rowmap = torch.randperm(B, device=DEV)
            for ranges, rm in ((None, None), (r, None), (r, rowmap)):
                ys = []
                for bound in (0, 1 << 40):                              # 128 candidates per workgroup | 32
                    ops.heads_small_batch(bound)
                    with torch.no_grad():
                        ys.append(ops.heads_fused(x1, x2, xij, pack, ranges, rm, on_union, scratch))
                torch.cuda.synchronize()
                assert torch.isfinite(ys[0]).all()
                assert torch.equal(ys[0], ys[1]), (B, ranges is not None, rm is not None, (ys[0] - ys[1]).abs().max().item())
    finally:
        ops.heads_small_batch(prev)


@pytest.mark.parametrize("scale", [1.0, 3.0e4, 1.0e-6])
def test_heads_product_accuracy(hiplib, scale):
    """The fused heads evaluate an f32 product as three f16 MFMAs on hi/lo splits with per-row power-of-two scaling
    (heads.hip).  Against an fp64 evaluation of the same head its error must be no worse than that of torch's own fp32
    evaluation (x 1.5 for noise) — also on pooled inputs far outside f16's range (raw walk-count pools reach 1e4, and a
    row scaled down to 1e-6 must keep its relative accuracy), and on rows that mix magnitudes."""
    import copy
    from ocn_amd.model import predictor_dict
    H, B = 256, 1024
    torch.manual_seed(11)
    pred = partial(predictor_dict["cn5"], cndeg=-1)(H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    with torch.no_grad():
        for p in pred.parameters():                      # trained-looking weights: not the symmetric init
            p.mul_(1.0 + 0.5 * torch.rand_like(p))
    x1, x2, xij = (torch.randn(B, H, device=DEV) * scale for _ in range(3))
    x1[::3] *= 1e-3                                        # rows of very different magnitude inside one wave's 32
    x2[:, ::5] *= 1e3                                      # columns of very different magnitude inside one row
    x1[7].zero_()
    with torch.no_grad():
        got = pred._heads_fused(x1.contiguous(), x2.contiguous(), xij.contiguous(), None).double()
        p64 = copy.deepcopy(pred).double()
        a = torch.sigmoid(p64.alpha).cumprod(-1)

        def head(m, dt):
            z = a.to(dt)[0] * m.xcn1lin(x1.to(dt)) + a.to(dt)[1] * m.xcn2lin(x2.to(dt)) + m.beta.to(dt) * m.xijlin(xij.to(dt))
            return m.lin(z)
        ref64 = head(p64, torch.float64)
        ref32 = head(pred, torch.float32).double()
    e_got, e_32 = (got - ref64).abs(), (ref32 - ref64).abs()
    assert torch.isfinite(got).all()
    assert e_got.max() <= 1.5 * e_32.max() + 1e-7 * ref64.abs().max(), (e_got.max().item(), e_32.max().item())
    assert e_got.pow(2).mean().sqrt() <= 1.5 * e_32.pow(2).mean().sqrt() + 1e-8, (e_got.pow(2).mean().sqrt().item(), e_32.pow(2).mean().sqrt().item())


@pytest.mark.parametrize("name,fin,H", [("cn5", 64, 128), ("cn7", 48, 32), ("cn5", 32, 40)])
def test_predictor_with_input_width_different_from_hidden(case, name, fin, H):
    """in_channels != hidden_channels (an encoder narrower than the predictor), and a hidden width none of
    the MFMA kernels take: the heads then walk branch by branch / through the torch modules — same scores."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    torch.manual_seed(21)
    x = torch.randn(case.n, fin)
    pred = predictor_dict[name](fin, H, 1, 3, 0.1, 0.0, True, use_xlin=True, tailact=True).eval()
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    args = SimpleNamespace(sum=0.5)
    ref = (O.cn5_forward(sd, x, case.ocn1, case.ocn2, case.e, True, True) if name == "cn5"
           else O.cn7_forward(sd, x, case.ocn1, case.ocn2, case.e, args.sum, True, True))
    e = case.e.to(DEV)
    with torch.no_grad():
        out = pred.to(DEV)(x.to(DEV), case.adj, adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e), e, args)
    assert close(out, ref), (out.cpu() - ref).abs().max()


def test_walk_route_on_a_dense_graph(hiplib):
    """Half of all pairs connected: nearly every swept element is a hit, so the LDS hit queue overflows and
    the in-place resolution path of the sweep runs; counts and values still equal the oracle's."""
    from ocn_amd.utils import CNState
    n, B = 400, 700
    g = torch.Generator().manual_seed(12)
    up = torch.triu(torch.rand(n, n, generator=g) < 0.5, diagonal=1)
    ei = up.nonzero().t().contiguous()
    oadj = O.to_symmetric(O.from_edge_index(ei, n))
    adj = to_product(oadj, DEV)
    e = torch.randint(0, n, (2, B), generator=g)
    oc1, oc2 = O.get_cn1_cn2(oadj, e)
    for nds in (None, adj.neighbor_degree_sum()):
        flags, wc, hc, c1, c2 = _walk_raw(adj, e.to(DEV), nds)
        assert c1.tolist() == torch.bincount(oc1.row, minlength=B).tolist()
        assert c2.tolist() == torch.bincount(oc2.row, minlength=B).tolist()
        assert hc[:, 3].tolist() == torch.zeros(n, dtype=torch.long).index_add_(0, oc2.col, oc2.val.long()).tolist()
    st = CNState(adj, None, None, e.to(DEV), walk=True)
    m2 = st.materialize(2)
    assert spm_equal(m2, oc2) and m2.storage.value().cpu().tolist() == oc2.val.tolist()


@pytest.mark.parametrize("H,B,lnnn", [(64, 20000, True), (64, 1500, False), (256, 4096, True), (32, 3000, True)])
def test_cn7_full_rows_share_the_row_sum(hiplib, monkeypatch, H, B, lnnn):
    """ogbl-ddi's regime (dense graph, A² full): a candidate whose whole source row is cn2 takes xcn2 = (A·h)[source]
    from ONE SpMM instead of summing the row again (ocn_cn_gather `rowsum`) — the same additions in the same order, so
    pooled vectors and scores are bit-equal to the plain evaluation; on every pooling kernel (packed, one wave per row,
    hub rows longer than 1024); the oracle agrees on the small batches."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import CNState, adjoverlap
    n = 1600
    oadj = make_graph(n, 500, 1500, seed=11, clique_frac=0.2, isolated=5)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    assert adj.max_rowcount() > 1024 and adj2.nnz() * 2 > n * n
    e = batch(oadj, B, 3).to(DEV)
    torch.manual_seed(H)
    x = torch.randn(n, H, device=DEV)
    pred = predictor_dict["cn7"](H, H, 1, 3, 0.0, 0.0, lnnn).eval().to(DEV)
    args = SimpleNamespace(sum=2.74)
    st = CNState(adj, adj, adj2, e)
    rows = adj._rowptr[1:] - adj._rowptr[:-1]
    full = (st.cnt2.long() == rows[e[0]]) & (rows[e[0]] > 0)
    assert full.float().mean().item() > 0.5                      # the regime the shortcut is for
    w = st.weights_cn7(args.sum)
    plain = st.gather(w, x)
    shared = st.gather(w, x, rowsum=ops.spmm_csr(adj._rowptr, adj._col, x))
    for a, b in zip(plain, shared):
        assert torch.equal(a, b)

    def score():
        with torch.no_grad():
            return pred(x, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, args)
    monkeypatch.setattr(ops, "share_full_rows", False)
    ref = score()
    assert getattr(pred, "_rowsum_cache", None) is None
    monkeypatch.setattr(ops, "share_full_rows", True)
    got = score()
    assert pred._rowsum_cache is not None and torch.equal(got, ref)
    if B <= 3000:                                                # (the column weights are the whole batch's: no sub-batch check)
        ec = e.cpu()
        oref = O.cn7_forward({k: v.detach().cpu() for k, v in pred.state_dict().items()}, x.cpu(), O.adjoverlap(oadj, oadj, ec),
                             O.adjoverlap(oadj, O.adj2_sparse(oadj), ec), ec, args.sum, lnnn)
        assert close(got, oref)


# ---- row f3: the Chebyshev bases of cn7 and cn6 under autograd ---------------------------------------
@pytest.mark.parametrize("k1,k2", [(0, 0), (1, 0), (4, 1), (10, 4), (0, 10)])
@pytest.mark.parametrize("route", ["pattern", "walk"])
def test_cn7_chebyshev_bases(case, k1, k2, route):
    """model.py:2958-3019, 3141-3165, 3186-3209: cn1 <- ncn1 * diag(T_k1(linspace(-1, 1, N))), cn2 <- cn2 * diag(T_k2(...)).
    The reference hard-wires k = 0 (its --polyfirst / --polysecond flags are dead); the predictor exposes the index
    as `polyfirst` / `polysecond` (default 0).  Pooled vectors bit-equal to the oracle for every basis tried, on both CN routes
    (the walk route's cn2 values are walk counts: value * T_k2 is still one product), scores within 1e-5."""
    from ocn_amd.model import chebyshev_diag, predictor_dict
    from ocn_amd.utils import CNState, adjoverlap, get_cn1_cn2
    H = 64
    torch.manual_seed(case.seed + 3)
    x = torch.randn(case.n, H)
    e = case.e.to(DEV)
    if route == "walk":
        ocn1, ocn2 = O.get_cn1_cn2(case.oadj, case.e)
        st = CNState(case.adj, None, None, e, walk=True)
    else:
        ocn1, ocn2 = case.ocn1, case.ocn2
        st = CNState(case.adj, case.adj, case.adj2, e)
    r1, r2, _ = O.cn7_pool(x, ocn1, ocn2, 2.74, polyfirst=k1, polysecond=k2)
    w = st.weights_cn7(2.74, chebyshev_diag(case.n, k1, DEV), chebyshev_diag(case.n, k2, DEV))
    g1, g2, _ = st.gather(w, x.to(DEV))
    assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2)
    pred = predictor_dict["cn7"](H, H, 1, 3, 0.0, 0.0, True).eval()
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    ref = O.cn7_forward(sd, x, ocn1, ocn2, case.e, 2.74, True, polyfirst=k1, polysecond=k2)
    pred = pred.to(DEV)
    assert pred.polyfirst == 0 and pred.polysecond == 0          # the reference's hard-wired basis is the default
    pred.polyfirst, pred.polysecond = k1, k2
    handles = get_cn1_cn2(case.adj, e) if route == "walk" else (adjoverlap(case.adj, case.adj, e), adjoverlap(case.adj, case.adj2, e))
    with torch.no_grad():
        out = pred(x.to(DEV), case.adj, *handles, e, SimpleNamespace(sum=2.74))
    tol = 1e-5 if route == "pattern" else 1e-5 * max(1.0, ref.abs().max().item())      # (raw walk-count pools: relative)
    assert (out.cpu() - ref).abs().max().item() <= tol + 1e-5 * ref.abs().max().item()
    with pytest.raises(ValueError):
        chebyshev_diag(case.n, 11, DEV)


def test_cn7_chebyshev_basis_switches_the_row_sum_shortcut_off(hiplib):
    """The dense-graph shortcut copies (A h)[source] for a candidate whose whole source row is cn2 — only valid while every
    cn2 weight is 1: with a non-trivial second basis the pooling must sum the weighted entries itself."""
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    n, H, B = 900, 64, 4096
    oadj = make_graph(n, 300, 850, seed=5, clique_frac=0.2)
    adj = to_product(oadj, DEV)
    adj2 = product_adj2(adj)
    assert adj2.nnz() * 2 > n * n
    e = batch(oadj, B, 7)
    torch.manual_seed(0)
    x = torch.randn(n, H)
    pred = predictor_dict["cn7"](H, H, 1, 3, 0.0, 0.0, True).eval()
    sd = {k: v.detach().clone() for k, v in pred.state_dict().items()}
    pred = pred.to(DEV)
    pred.polysecond = 3
    ed = e.to(DEV)
    with torch.no_grad():
        out = pred(x.to(DEV), adj, adjoverlap(adj, adj, ed), adjoverlap(adj, adj2, ed), ed, SimpleNamespace(sum=1.0))
    oadj2 = O.adj2_sparse(oadj)
    ref = O.cn7_forward(sd, x, O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e), e, 1.0, True, polysecond=3)
    assert close(out, ref)


@pytest.mark.parametrize("ip", [0.0, 0.37])
def test_cn6_backward_matches_oracle_autograd(case3, ip):
    """cn6 (model.py:2535-2951) under autograd in eval mode: gradients with respect to the embeddings (the transposed
    three-pool gather, ocn_cn_gather3_backward) and every used parameter against torch autograd through the oracle."""
    from ocn_amd.model import predictor_dict
    from ocn_amd.utils import adjoverlap
    c = case3
    H = 32
    torch.manual_seed(23)
    x = torch.randn(c.n, H)
    pred = predictor_dict["cn6"](H, H, 1, 3, 0.0, 0.0, True).eval()
    with torch.no_grad():
        pred.alpha.copy_(torch.tensor([0.3, -0.2, 0.9]))
        pred.innerprod.fill_(ip)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k != "innerprod") for k, v in pred.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    ref = O.cn6_forward(sd, xr, *c.ocn, c.e, True)
    wgt = torch.randn(c.B, 1, generator=torch.Generator().manual_seed(1))
    (ref * wgt).sum().backward()
    pred = pred.to(DEV)
    ed = c.e.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = pred(xd, c.adj, adjoverlap(c.adj, c.adj, ed), adjoverlap(c.adj, c.adj2, ed), adjoverlap(c.adj, c.adj3, ed), ed, None)
    assert out.requires_grad and close(out, ref)
    (out * wgt.to(DEV)).sum().backward()
    scale = xr.grad.abs().max().item()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= 2e-5 * max(1.0, scale)
    seen = 0
    for k, p in pred.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            continue
        g = sd[k].grad
        seen += 1
        assert (p.grad.cpu() - g).abs().max().item() <= 2e-5 * max(1.0, g.abs().max().item()), k
    assert seen >= 20 and any(k.startswith("xcn3lin") for k, p in pred.named_parameters() if p.grad is not None)
