"""CPU suite, part 1: the oracle against its only pins — the hand-derived Appendix C answers and the
independent set-based model.  (The reference holds no tests or golden vectors: "parity unpinned".)"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import naive_model as NM
from oracle import ocn_oracle as O
from ocn_amd.synth import chung_lu_graph, sample_edges

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def appc():
    return json.load(open(os.path.join(GOLD, "appendix_c.json")))


def test_smoke_inputs_spmoverlap(appc):
    s = appc["smoke_inputs"]
    a1 = O.from_edge_index(torch.tensor(s["adj1"]), s["n"])
    a2 = O.from_edge_index(torch.tensor(s["adj2"]), s["n"])
    out = O.spmoverlap_(a1, a2)
    assert [out.row.tolist(), out.col.tolist()] == s["spmoverlap"]
    assert out.val.tolist() == s["spmoverlap_values"] and out.val.dtype == torch.float32


def test_path_graph_known_answers(appc):
    g = appc["path_graph"]
    adj = O.to_symmetric(O.from_edge_index(torch.tensor(g["undirected_edges"]).t(), g["n"]))
    for i in range(4):
        assert adj.col[adj.row == i].tolist() == g["rows"][str(i)]
    a2 = O.adj2_sparse(adj)
    for i in range(4):
        assert a2.col[a2.row == i].tolist() == g["a2_rows"][str(i)]
    e = torch.tensor(g["batch"]).t().contiguous()
    cn1, cn2 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e)
    assert [cn1.col[cn1.row == r].tolist() for r in range(3)] == g["cn1_rows"]
    assert [cn2.col[cn2.row == r].tolist() for r in range(3)] == g["cn2_rows"]
    assert O.col_sum(cn1).tolist() == g["S1"] and O.col_sum(cn2).tolist() == g["cn2_colsum"]
    x = torch.randn(4, 8)
    for key, ip in (("cn5_innerprod_0", 0.0), ("cn5_innerprod_0.37", 0.37)):
        _, _, aux = O.cn5_pool(x, cn1, cn2, torch.tensor([ip]))
        pat = list(map(list, zip(aux["ncn2"].row.tolist(), aux["ncn2"].col.tolist())))
        assert pat == g["union_pattern"]
        assert aux["S2"].tolist() == pytest.approx(g[key]["S2"], rel=1e-6)
        assert aux["ncn2"].val.tolist() == pytest.approx(g[key]["ncn2"], rel=1e-6, abs=1e-7)
    assert aux["scale"] == pytest.approx(g["cn5_innerprod_0"]["scale"], rel=1e-7)
    # Q2: singleton column is zeroed (cn5) / filled with args.sum (cn7)
    e1 = torch.tensor([[0], [1]])
    c1, c2 = O.adjoverlap(adj, adj, e1), O.adjoverlap(adj, a2, e1)
    xc1, _, aux = O.cn5_pool(x, c1, c2, torch.tensor([0.0]))
    assert aux["inv1"][2].item() == 0.0 and xc1.abs().max().item() == 0.0
    xc1, _, aux = O.cn7_pool(x, c1, c2, 2.74)
    assert aux["inv1"][2].item() == pytest.approx(2.74) and torch.allclose(xc1[0], 2.74 * x[2])


def test_path_graph_walk_route_cn7_and_cn6_known_answers(appc):
    """Hand-derived answers (tests/golden/make_golden.py) for the pygho route, cn7 on it, and cn6."""
    g = appc["path_graph"]
    adj = O.to_symmetric(O.from_edge_index(torch.tensor(g["undirected_edges"]).t(), g["n"]))
    e = torch.tensor(g["batch"]).t().contiguous()
    c1, c2 = O.get_cn1_cn2(adj, e)
    w = g["walk_route"]
    assert [c1.col[c1.row == r].tolist() for r in range(3)] == g["cn1_rows"]
    assert [c2.col[c2.row == r].tolist() for r in range(3)] == w["cn2_rows"]
    assert [c2.val[c2.row == r].tolist() for r in range(3)] == w["cn2_values"]
    assert O.col_sum(c2).tolist() == w["walk_colsum"]
    eye = torch.eye(4)
    x1, x2, _ = O.cn7_pool(eye, c1, c2, 1.0)
    assert torch.allclose(x1, torch.tensor(g["cn7_walk"]["xcn1"]), rtol=1e-6, atol=0)
    assert torch.equal(x2, torch.tensor(g["cn7_walk"]["xcn2"]))
    a2 = O.adj2_sparse(adj)
    a3 = O.adj3_sparse(adj, a2)
    k6 = g["cn6_innerprod_0"]
    for r_, cols in k6["a3_rows"].items():
        assert a3.col[a3.row == int(r_)].tolist() == cols
    cn3 = O.adjoverlap(adj, a3, e)
    assert [cn3.col[cn3.row == r].tolist() for r in range(3)] == k6["cn3_rows"]
    _, _, x3, aux = O.cn6_pool(eye, O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e), cn3, torch.tensor([0.0]))
    assert aux["S3"].tolist() == k6["S3"]
    assert torch.allclose(x3, torch.tensor(k6["xcn3"]), rtol=1e-6, atol=0)


@pytest.mark.parametrize("seed", range(4))
def test_oracle_vs_naive_model(seed):
    n, B, H = 150 + 40 * seed, 120, 12
    ei = chung_lu_graph(n, 7, 50, seed=seed, clique_frac=0.5)
    adj = O.to_symmetric(O.from_edge_index(ei, n))
    a2 = O.adj2_sparse(adj)
    e = sample_edges(adj.row, adj.col, n, B, seed=seed)
    nb = NM.neighbours(n, ei.t().tolist())
    nb2 = NM.two_hop(nb)
    s1, s2 = NM.cn_sets(nb, nb2, e.t().tolist())
    cn1, cn2 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e)
    assert [c for r in s1 for c in r] == cn1.col.tolist()
    assert [c for r in s2 for c in r] == cn2.col.tolist()
    assert sorted((j, k) for j in range(n) for k in nb2[j]) == list(zip(a2.row.tolist(), a2.col.tolist()))
    x = torch.randn(n, H)
    for ip in (0.0, 0.37, -1.5):
        a, b, aux = O.cn5_pool(x, cn1, cn2, torch.tensor([ip]))
        na, nbb, naux = NM.cn5_pool(n, x.numpy(), s1, s2, ip)
        assert aux["scale"] == pytest.approx(naux["scale"], rel=1e-6)
        assert np.abs(a.numpy() - na).max() < 1e-5
        assert np.abs(b.numpy() - nbb).max() < 1e-4 * max(1.0, np.abs(nbb).max())
    w = NM.walk_counts(nb, e.t().tolist())
    c1, c2 = O.get_cn1_cn2(adj, e)
    assert c1.col.tolist() == cn1.col.tolist()
    assert [k for r in w for k in r] == c2.col.tolist()
    assert [v for r in w for v in r.values()] == c2.val.tolist()
    a, b, _ = O.cn7_pool(x, cn1, c2, 1.0)
    na, nbb, _ = NM.cn7_pool(n, x.numpy(), s1, w, 1.0)
    assert np.abs(a.numpy() - na).max() < 1e-5 and np.abs(b.numpy() - nbb).max() < 1e-4


@pytest.mark.parametrize("seed", range(3))
def test_cn6_oracle_vs_naive_model(seed):
    """3-hop predictor (model.py:2535-2951): the op-for-op restatement against the set-based model."""
    n, B, H = 120 + 30 * seed, 90, 10
    ei = chung_lu_graph(n, 5, 30, seed=seed + 20, clique_frac=0.4)
    adj = O.to_symmetric(O.from_edge_index(ei, n))
    a2 = O.adj2_sparse(adj)
    a3 = O.adj3_sparse(adj, a2)
    e = sample_edges(adj.row, adj.col, n, B, seed=seed + 3)
    nb = NM.neighbours(n, ei.t().tolist())
    nb2 = NM.two_hop(nb)
    nb3 = NM.three_hop(nb, nb2)
    assert sorted((j, k) for j in range(n) for k in nb3[j]) == list(zip(a3.row.tolist(), a3.col.tolist()))
    batch = e.t().tolist()
    s1, s2 = NM.cn_sets(nb, nb2, batch)
    s3 = [sorted(nb[i] & nb3[j]) for i, j in batch]
    cn1, cn2, cn3 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e), O.adjoverlap(adj, a3, e)
    assert [c for r in s3 for c in r] == cn3.col.tolist()
    x = torch.randn(n, H)
    for ip in (0.0, 0.37, -1.5):
        a, b, c, aux = O.cn6_pool(x, cn1, cn2, cn3, torch.tensor([ip]))
        na, nbb, nc, naux = NM.cn6_pool(n, x.numpy(), s1, s2, s3, ip)
        assert np.abs(a.numpy() - na).max() < 1e-5
        assert np.abs(b.numpy() - nbb).max() < 1e-4 * max(1.0, np.abs(nbb).max())
        assert np.abs(c.numpy() - nc).max() < 2e-4 * max(1.0, np.abs(nc).max())
        if ip == 0.0:       # fresh model: S3 = column counts of cn3
            assert aux["S3"].tolist() == np.where(naux["S3"] == 0, 1.0, naux["S3"]).tolist()
            assert aux["S3"].tolist() == torch.bincount(cn3.col, minlength=n).clamp(min=1).float().tolist()


def test_adjoverlap_edge_cases():
    n = 30
    adj = O.to_symmetric(O.from_edge_index(torch.tensor([[0, 1, 2, 3], [1, 2, 3, 0]]), n))
    a2 = O.adj2_sparse(adj)
    empty = torch.zeros(2, 0, dtype=torch.long)
    out = O.adjoverlap(adj, a2, empty)
    assert out.nnz == 0 and out.n_rows == 0
    e = torch.tensor([[29, 0, 0], [0, 29, 0]])             # isolated endpoints, self pair
    cn1, cn2 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e)
    assert torch.bincount(cn1.row, minlength=3).tolist() == [0, 0, 2]
    assert torch.bincount(cn2.row, minlength=3).tolist() == [0, 0, 0]     # N(0)∩N²(0): N² of a 4-cycle node = {0, 2}
    x = torch.randn(n, 4)
    xc1, xc2, aux = O.cn5_pool(x, cn1, cn2, torch.tensor([0.37]))
    assert torch.isfinite(xc1).all() and torch.isfinite(xc2).all()


def test_adj2_by_block_matches_sparse_and_fold_quirk():
    n = 70
    ei = chung_lu_graph(n, 9, 40, seed=9)
    adj = O.to_symmetric(O.from_edge_index(ei, n))
    ref = O.adj2_sparse(adj)
    blk = O.adj2_by_block(adj, block_size=32)
    assert blk.row.tolist() == ref.row.tolist() and blk.col.tolist() == ref.col.tolist()
    dense = adj.to_dense()
    assert torch.equal(blk.to_dense(), dense @ dense)       # walk-count values, offset-correct tiles
    folded = O.adj2_by_block(adj, block_size=32, fold_quirk=True)
    assert int(folded.row.max()) < 32 and int(folded.col.max()) < 32


def test_perm_batches_keeps_ragged_tail():
    b = O.perm_batches(10, 4)
    assert [x.tolist() for x in b] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_oracle_reproduces_its_committed_vectors():
    """tests/golden/oracle_vectors.json are ORACLE outputs (regression vectors for the HIP path): the oracle
    of today must still produce them."""
    import json
    import os
    recs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))
    for rec in recs[:2]:
        n, H = rec["n"], rec["H"]
        adj = O.to_symmetric(O.from_edge_index(torch.tensor(rec["edge_index"]), n))
        adj2 = O.adj2_sparse(adj)
        e = torch.tensor(rec["batch"])
        x = torch.randn(n, H, generator=torch.Generator().manual_seed(rec["x_seed"]))
        cn1, cn2 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, adj2, e)
        assert torch.bincount(cn1.row, minlength=e.shape[1]).tolist() == rec["cn1_counts"]
        a, b, _ = O.cn5_pool(x, cn1, cn2, torch.tensor([0.37]))
        assert a[0].tolist() == rec["cn5_ip0.37"]["xcn1_row0"] and b[0].tolist() == rec["cn5_ip0.37"]["xcn2_row0"]
        adj3 = O.adj3_sparse(adj, adj2)
        cn3 = O.adjoverlap(adj, adj3, e)
        assert adj3.nnz == rec["a3_nnz"] and torch.bincount(cn3.row, minlength=e.shape[1]).tolist() == rec["cn3_counts"]
        c = O.cn6_pool(x, cn1, cn2, cn3, torch.tensor([0.37]))[2]
        assert c[0].tolist() == rec["cn6_ip0.37"]["xcn3_row0"]


def test_order_sensitive_column_sum_known_answer():
    """A column whose fp32 sum depends on the order of its five entries (tests/golden/order_sensitive_colsum.json, derived
    by hand): the reference's index_add_ adds them in ascending batch-row order -> 2^25 exactly; adding the small
    entries first, or rounding the exact sum once, gives 2^25 + 4."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "order_sensitive_colsum.json")))
    adj = O.to_symmetric(O.from_edge_index(torch.tensor(g["undirected_edges"]).t(), g["n"]))
    a2 = O.adj2_sparse(adj)
    e = torch.tensor(g["batch"]).t().contiguous()
    cn1, cn2 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e)
    k = g["column"]
    assert cn1.row.tolist() == [0, 4] and cn1.col.tolist() == [k, k]
    assert cn2.row.tolist() == [1, 2, 3] and cn2.col.tolist() == [k, k, k]
    _, _, aux = O.cn5_pool(torch.randn(g["n"], 4), cn1, cn2, torch.tensor([g["innerprod"]]))
    assert aux["scale"] == 0.5 and aux["nip"] == -2.0 ** 25
    v = aux["ncn2"].val * aux["S2"][k]                                   # the union values, in batch-row order
    assert v.tolist() == g["values_in_batch_row_order"]
    assert aux["S2"][k].item() == g["S2_reference_order"]
    ones_first = torch.tensor([1.0, 1.0, 1.0, 2.0 ** 24, 2.0 ** 24])
    acc = torch.tensor(0.0)
    for t in ones_first:
        acc = acc + t
    assert acc.item() == g["S2_ones_first_or_rounded_once"] != g["S2_reference_order"]
    assert float(torch.tensor(sum(g["values_in_batch_row_order"]), dtype=torch.float64).float()) == g["S2_ones_first_or_rounded_once"]


def test_order_sensitive_pooling_known_answer():
    """tests/golden/order_sensitive_pooling.json (hand-derived): a pooled value that is 2^23 only when a row's entries are
    added in ascending column order, 2^23 + 2 in descending order."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "order_sensitive_pooling.json")))
    adj = O.to_symmetric(O.from_edge_index(torch.tensor(g["undirected_edges"]).t(), g["n"]))
    e = torch.tensor(g["batch"]).t().contiguous()
    cn1, cn2 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, O.adj2_sparse(adj), e)
    assert cn1.col.tolist() == [2, 3, 4, 5, 2, 3, 4, 5]
    x = torch.zeros(g["n"], 4)
    for k, v in g["x_rows"].items():
        x[int(k)] = v
    xcn1, _, aux = O.cn5_pool(x, cn1, cn2, torch.tensor([0.0]))
    assert aux["inv1"][2:6].tolist() == [0.5] * 4
    assert xcn1.unique().tolist() == [g["xcn1_ascending_column_order"]]
    acc = torch.tensor(0.0)
    for v in (1.0, 1.0, 1.0, 2.0 ** 24):
        acc = acc + torch.tensor(0.5) * v
    assert acc.item() == g["xcn1_descending_column_order"]


def _path_adj2_patterns(n, bs):
    """A^2 of the path 0-1-...-(n-1), derived by hand: (i, i) for every i (degree >= 1) and (i, i +- 2).  Folded by blocks
    of `bs` (block-local indices added without the block offset): every entry (i, j) lands on (i % bs, j % bs)."""
    full = {(i, i) for i in range(n)} | {(i, i + 2) for i in range(n - 2)} | {(i + 2, i) for i in range(n - 2)}
    return full, {(i % bs, j % bs) for i, j in full}


def test_block_route_known_answers_on_a_path():
    # the 4-node path with 2 x 2 blocks, written out: A^2 = [[1,0,1,0],[0,2,0,1],[1,0,2,0],[0,1,0,1]]; its four blocks are
    # diag(1,2), diag(1,1), diag(1,1), diag(2,1): folded onto the corner they add up to diag(5, 5)
    adj = O.to_symmetric(O.from_edge_index(torch.tensor([[0, 1, 2], [1, 2, 3]]), 4))
    a2 = O.adj2_by_block(adj, block_size=2)
    assert a2.to_dense().tolist() == [[1, 0, 1, 0], [0, 2, 0, 1], [1, 0, 2, 0], [0, 1, 0, 1]]
    folded = O.adj2_by_block(adj, block_size=2, fold_quirk=True)
    assert folded.to_dense()[:2, :2].tolist() == [[5, 0], [0, 5]] and folded.row.tolist() == [0, 1] and folded.col.tolist() == [0, 1]
    # a 70-node path with 32-wide blocks (ragged last block): patterns from the closed form above
    n, bs = 70, 32
    adj = O.to_symmetric(O.from_edge_index(torch.stack([torch.arange(n - 1), torch.arange(1, n)]), n))
    full, fold = _path_adj2_patterns(n, bs)
    a2, fq = O.adj2_by_block(adj, block_size=bs), O.adj2_by_block(adj, block_size=bs, fold_quirk=True)
    assert set(zip(a2.row.tolist(), a2.col.tolist())) == full
    assert set(zip(fq.row.tolist(), fq.col.tolist())) == fold


def test_chebyshev_diagonals_of_cn7():
    """model.py:2958-3019: T_k(linspace(-1, 1, n)) — the oracle's restatement against numpy's Chebyshev evaluation (fp64, so
    to rounding), T0 = ones, the index check's error, and k = 0 leaving cn7's pools untouched."""
    import numpy as np
    from numpy.polynomial import chebyshev as C
    for n in (1, 2, 7, 1001):
        xs = np.linspace(-1, 1, n)
        for k in range(11):
            d = O.evaluate_polynomial(n, k)
            assert d.dtype == torch.float32 and d.shape == (n,)
            assert np.abs(d.numpy() - C.chebval(xs, [0] * k + [1])).max() <= 2e-4      # (|T_10| terms reach 1280: fp32 cancellation)
    assert torch.equal(O.evaluate_polynomial(9, 0), torch.ones(9))
    for bad in (-1, 11):
        try:
            O.evaluate_polynomial(5, bad)
        except ValueError:
            continue
        raise AssertionError("poly_index outside 0..10 must raise")
    oadj = O.to_symmetric(O.from_edge_index(torch.tensor([[0, 1, 2, 0], [1, 2, 3, 2]]), 4))
    e = torch.tensor([[0, 1, 0], [1, 3, 3]])
    cn1, cn2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, O.adj2_sparse(oadj), e)
    x = torch.arange(8, dtype=torch.float32).reshape(4, 2)
    a = O.cn7_pool(x, cn1, cn2, 1.0)
    b = O.cn7_pool(x, cn1, cn2, 1.0, polyfirst=0, polysecond=0)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # T1 on 4 columns = linspace(-1, 1, 4): the cn2 pool of edge (0, 1) = columns {1, 2} -> (-1/3) h[1] + (1/3) h[2]
    c = O.cn7_pool(x, cn1, cn2, 1.0, polysecond=1)
    d = torch.linspace(-1, 1, 4)
    assert torch.equal(c[1][0], (d[1] * x[1]) + (d[2] * x[2]))
