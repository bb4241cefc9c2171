"""Property tests (hypothesis) of the oracle against the independent set-based model on arbitrary tiny
graphs — isolated nodes, self pairs, duplicate candidates, empty intersections, star and clique shapes all
fall out of the generator.  CPU only."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import naive_model as NM
from oracle import ocn_oracle as O


@st.composite
def graph_and_batch(draw):
    n = draw(st.integers(3, 11))
    pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    edges = draw(st.lists(st.sampled_from(pairs), min_size=1, max_size=min(len(pairs), 24), unique=True))
    batch = draw(st.lists(st.tuples(st.integers(0, n - 1), st.integers(0, n - 1)), min_size=1, max_size=9))
    ip = draw(st.sampled_from([0.0, 0.37, -2.0]))
    return n, edges, batch, ip


@settings(max_examples=60, deadline=None)
@given(graph_and_batch())
def test_cn5_cn6_cn7_pools_match_the_set_model(gb):
    n, edges, batch, ip = gb
    ei = torch.tensor(edges).t().contiguous()
    adj = O.to_symmetric(O.from_edge_index(ei, n))
    a2 = O.adj2_sparse(adj)
    a3 = O.adj3_sparse(adj, a2)
    e = torch.tensor(batch).t().contiguous()
    nb = NM.neighbours(n, edges)
    nb2 = NM.two_hop(nb)
    nb3 = NM.three_hop(nb, nb2)
    s1, s2 = NM.cn_sets(nb, nb2, batch)
    s3 = [sorted(nb[i] & nb3[j]) for i, j in batch]
    cn1, cn2, cn3 = O.adjoverlap(adj, adj, e), O.adjoverlap(adj, a2, e), O.adjoverlap(adj, a3, e)
    assert [c for r in s1 for c in r] == cn1.col.tolist()
    assert [c for r in s2 for c in r] == cn2.col.tolist()
    assert [c for r in s3 for c in r] == cn3.col.tolist()
    x = torch.randn(n, 4, generator=torch.Generator().manual_seed(n))
    a, b, c, aux = O.cn6_pool(x, cn1, cn2, cn3, torch.tensor([ip]))
    na, nb_, nc, naux = NM.cn6_pool(n, x.numpy(), s1, s2, s3, ip)
    # tiny column sums can cancel (SURVEY Appendix C "conditioning hazard"): compare where the model is well conditioned
    well = np.abs(naux["S2"]).min() > 1e-3 and np.abs(naux["S3"]).min() > 1e-3
    assert np.abs(a.numpy() - na).max() < 1e-5
    if well:
        assert np.abs(b.numpy() - nb_).max() < 1e-3 * max(1.0, np.abs(nb_).max())
        assert np.abs(c.numpy() - nc).max() < 1e-3 * max(1.0, np.abs(nc).max())
    w = NM.walk_counts(nb, batch)
    c1, c2 = O.get_cn1_cn2(adj, e)
    assert [k for r in w for k in r] == c2.col.tolist() and [v for r in w for v in r.values()] == c2.val.tolist()
    p, q, _ = O.cn7_pool(x, c1, c2, 1.5)
    npq = NM.cn7_pool(n, x.numpy(), s1, w, 1.5)
    assert np.abs(p.numpy() - npq[0]).max() < 1e-5 and np.abs(q.numpy() - npq[1]).max() < 1e-4
