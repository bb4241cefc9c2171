"""Property tests (hypothesis) of the HIP path against the oracle on arbitrary tiny graphs: isolated nodes,
self pairs, duplicate candidates, empty intersections, stars and cliques.  Integer results bit-exact, cn7
pools bit-exact, cn5 pools within tolerance; pattern and walk-count routes, and cn6."""
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import ocn_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@st.composite
def graph_and_batch(draw):
    n = draw(st.integers(3, 40))
    m = draw(st.integers(1, 3 * n))
    a = draw(st.lists(st.integers(0, n - 1), min_size=m, max_size=m))
    b = draw(st.lists(st.integers(0, n - 1), min_size=m, max_size=m))
    edges = sorted({(min(x, y), max(x, y)) for x, y in zip(a, b) if x != y})
    if not edges:
        edges = [(0, 1)]
    B = draw(st.integers(1, 70))
    bi = draw(st.lists(st.integers(0, n - 1), min_size=B, max_size=B))
    bj = draw(st.lists(st.integers(0, n - 1), min_size=B, max_size=B))
    return n, edges, list(zip(bi, bj))


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(graph_and_batch())
def test_hip_path_matches_the_oracle_on_arbitrary_tiny_graphs(hiplib, gb):
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNState, CNState3
    n, edges, batch = gb
    ei = torch.tensor(edges).t().contiguous()
    oadj = O.to_symmetric(O.from_edge_index(ei, n))
    oadj2 = O.adj2_sparse(oadj)
    oadj3 = O.adj3_sparse(oadj, oadj2)
    e = torch.tensor(batch).t().contiguous()
    B = e.shape[1]
    oc = [O.adjoverlap(oadj, t, e) for t in (oadj, oadj2, oadj3)]
    adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n)).to_symmetric()
    adj2 = SparseTensor.from_torch_sparse_coo_tensor(adj.to_torch_sparse_coo_tensor() @ adj.to_torch_sparse_coo_tensor(), False)
    adj3 = SparseTensor.from_torch_sparse_coo_tensor(adj2.to_torch_sparse_coo_tensor() @ adj.to_torch_sparse_coo_tensor(), False)
    assert adj2.nnz() == oadj2.nnz and adj3.nnz() == oadj3.nnz
    ed = e.to(DEV)
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(n + B))
    xd = x.to(DEV)
    cnt = [torch.bincount(c.row, minlength=B).tolist() for c in oc]
    # pattern route: counts, histograms, cn7 pools (bit-exact), cn5 pools at innerprod 0 (bit-exact)
    st_ = CNState(adj, adj, adj2, ed)
    assert st_.cnt1.cpu().tolist() == cnt[0] and st_.cnt2.cpu().tolist() == cnt[1]
    hc = st_.hist_counts().cpu()
    assert hc[:, 0].tolist() == torch.bincount(oc[0].col, minlength=n).tolist()
    assert hc[:, 1].tolist() == torch.bincount(oc[1].col, minlength=n).tolist()
    r1, r2, _ = O.cn7_pool(x, oc[0], oc[1], 1.5)
    g1, g2, gx = st_.gather(st_.weights_cn7(1.5), xd)
    assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2) and torch.equal(gx.cpu(), x[e[0]] * x[e[1]])
    r1, r2, _ = O.cn5_pool(x, oc[0], oc[1], torch.tensor([0.0]))
    st_ = CNState(adj, adj, adj2, ed)
    g1, g2, _ = st_.gather(st_.weights_cn5(torch.zeros(1, device=DEV)), xd)
    assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2)
    # cn6 at innerprod 0
    r3 = O.cn6_pool(x, *oc, torch.tensor([0.0]))[2]
    s3 = CNState3(adj, adj2, adj3, ed)
    assert s3.cnt3.cpu().tolist() == cnt[2]
    assert torch.equal(s3.gather(*s3.weights(torch.zeros(1, device=DEV)), xd)[2].cpu(), r3)
    # walk-count route: counts, values, cn7 pools
    w1, w2 = O.get_cn1_cn2(oadj, e)
    sw = CNState(adj, None, None, ed, walk=True)
    assert sw.cnt1.cpu().tolist() == torch.bincount(w1.row, minlength=B).tolist()
    assert sw.cnt2.cpu().tolist() == torch.bincount(w2.row, minlength=B).tolist()
    assert sw.hist_counts().cpu()[:, 3].tolist() == torch.zeros(n, dtype=torch.long).index_add_(0, w2.col, w2.val.long()).tolist()
    r1, r2, _ = O.cn7_pool(x, w1, w2, 0.0)
    g1, g2, _ = sw.gather(sw.weights_cn7(0.0), xd)
    assert torch.equal(g1.cpu(), r1) and torch.equal(g2.cpu(), r2)
