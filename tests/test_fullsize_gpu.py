"""Full-size runs of the BASELINE.json configurations on the GPU (`-m gpu`): sizes the oracle cannot finish in
seconds, so the checks are size-independent properties of the domain — the three sweep strategies of the walk
route agree bit for bit, histogram mass equals the per-edge counts, batch-permutation equivariance — plus the
oracle on the first 256 candidates of each batch (as a batch of their own: the normalisation is batch-coupled)."""
from types import SimpleNamespace

import pytest
import torch

from oracle import ocn_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _full_graph(name):
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like
    ei, n, shape = dataset_like(name, seed=0)
    adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n), trust_data=True).to_symmetric()
    return n, shape, adj


@pytest.fixture(scope="module")
def collab(hiplib):
    from tests.helpers import product_adj2
    n, shape, adj = _full_graph("collab")
    return SimpleNamespace(n=n, adj=adj, adj2=product_adj2(adj))


def test_graph_capture_of_a_65536_edge_step_replays_bitwise(collab):
    """One 65 536-edge cn5 step (source-sorted order, class-major heads, zero-row skipping: every path that only
    runs at B >= 4096) captured as a HIP graph and replayed on fresh candidate ids == the eager call, bit for bit."""
    import ocn_amd.model as M
    from ocn_amd.pipeline import GraphedScorer
    from ocn_amd.synth import sample_edges
    from ocn_amd.utils import adjoverlap
    c = collab
    H, B = 256, 65536
    torch.manual_seed(3)
    h = torch.randn(c.n, H, device=DEV)
    pred = M.predictor_dict["cn5"](H, H, 1, 3, 0.05, 0.4, True).to(DEV).eval()
    r, col, _ = c.adj.coo()
    rc, cc = r.cpu(), col.cpu()
    scorer = GraphedScorer(pred, h, c.adj, c.adj2, B, SimpleNamespace(sum=1.0))
    for seed in (1, 2, 3):
        e = sample_edges(rc, cc, c.n, B, seed=seed).to(DEV)
        got = scorer(e, check=True).clone()
        with torch.no_grad():
            want = pred(h, c.adj, adjoverlap(c.adj, c.adj, e), adjoverlap(c.adj, c.adj2, e), e, None)
        assert torch.equal(got, want)


# ---- the walk route at full size: ppa- and citation2-shaped graphs, B = 2048 -----------------------------
def _walk_raw(adj, e, nds):
    from ocn_amd import ops
    order, off, flags, wc, hist, c1, c2, status, _ = ops.cn_flags(
        adj._rowptr, adj._col, None, None, e[0].contiguous(), e[1].contiguous(), adj.size(1), adj.max_rowcount(),
        walk=True, nds=nds)
    n = int(off[-1])
    assert int(status[0]) == 0
    return flags[:n].clone(), wc[:n].clone(), ops.hist_counts(hist), c1.clone(), c2.clone()


def _oracle_rows(adj, rows):
    """Node set on which the oracle can reproduce a few candidates exactly: their endpoints and the endpoints'
    neighbours.  For (i, j) the oracle needs the rows of i, j and of every m in N(j) (Ej . A), and of those rows
    only the columns k in N(i) reach the result (Ei (.) ...): restricted to this set (ascending ids kept, so
    ascending-column sums keep their order) the rows are complete where it matters."""
    seeds = torch.unique(rows.reshape(-1))
    return torch.unique(torch.cat([seeds, adj[seeds].coo()[1]]))


@pytest.mark.parametrize("name,H,pname", [("ppa", 64, "cn5"), ("citation2", 32, "cn7")])
def test_walk_route_full_size(hiplib, name, H, pname):
    import ocn_amd.model as M
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import sample_edges
    from ocn_amd.utils import CNState, get_cn1_cn2
    n, shape, adj = _full_graph(name)
    B = 2048
    r, c, _ = adj.coo()
    e = sample_edges(r.cpu(), c.cpu(), n, B, seed=1).to(DEV)
    deg = adj.storage.rowcount()
    assert int(deg.max()) > 1024, "hub rows (> 1024 neighbours) must be present at full size"
    nds = adj.neighbor_degree_sum()
    # the three sweep strategies agree bit for bit: forward only, cost-based two-sided, forced reverse
    fwd = _walk_raw(adj, e, None)
    mix = _walk_raw(adj, e, nds)
    forced = torch.zeros_like(nds); forced[e[0]] = 1 << 50; forced[e[1]] = 0
    forced[e[0][e[0] == e[1]]] = 0
    rev = _walk_raw(adj, e, forced)
    for a, b, cc in zip(fwd, mix, rev):
        assert torch.equal(a, b) and torch.equal(a, cc)
    flags, wc, hist, cnt1, cnt2 = mix
    # histogram mass = sum of the per-edge counts; walk-count mass = sum of wc over the cn2 entries
    assert int(hist[:, 0].sum()) == int(cnt1.sum()) and int(hist[:, 1].sum()) == int(cnt2.sum())
    assert int(hist[:, 3].sum()) == int(wc[(flags & 2) != 0].sum())
    assert bool((wc[(flags & 2) == 0] == 0).all()) and bool((wc[(flags & 2) != 0] > 0).all())
    assert bool((hist[:, 2] <= hist[:, 0] + hist[:, 1]).all()) and bool((hist[:, 2] >= torch.maximum(hist[:, 0], hist[:, 1])).all())
    # cn1 is symmetric in (i, j); counts are equivariant under a permutation of the batch, histograms invariant
    st = CNState(adj, None, None, e, walk=True)
    assert torch.equal(st.cnt1, cnt1) and torch.equal(st.cnt2, cnt2)
    sw = CNState(adj, None, None, e.flip(0).contiguous(), walk=True)
    assert torch.equal(sw.cnt1, cnt1)
    perm = torch.randperm(B, device=DEV)
    pm = CNState(adj, None, None, e[:, perm].contiguous(), walk=True)
    assert torch.equal(pm.cnt1, cnt1[perm]) and torch.equal(pm.cnt2, cnt2[perm]) and torch.equal(pm.hist_counts(), hist)
    # candidates whose source is a hub are in the batch (positives are drawn by degree)
    assert int((deg[e[0]] > 1024).sum()) > 0
    # the oracle on the first 256 candidates, as a batch of their own, on the 2-hop closure of their endpoints
    sub = e[:, :256].contiguous()
    keep = _oracle_rows(adj, sub)
    relabel = torch.full((n,), -1, dtype=torch.long, device=DEV)
    relabel[keep] = torch.arange(keep.numel(), device=DEV)
    rr, cc2, _ = adj[keep].coo()                                   # rows of the kept nodes, columns global
    inside = relabel[cc2] >= 0
    osub = O.SpM(rr[inside].cpu(), relabel[cc2[inside]].cpu(), None, keep.numel(), keep.numel())
    esub = relabel[sub].cpu()
    oc1, oc2 = O.get_cn1_cn2(osub, esub)
    h1, h2 = get_cn1_cn2(adj, sub)
    assert h1.counts().cpu().tolist() == torch.bincount(oc1.row, minlength=256).tolist()
    assert h2.counts().cpu().tolist() == torch.bincount(oc2.row, minlength=256).tolist()
    m2 = h2.materialize()
    r2, c2, v2 = m2.coo()
    assert r2.cpu().tolist() == oc2.row.tolist() and relabel[c2].cpu().tolist() == oc2.col.tolist()
    assert v2.cpu().tolist() == oc2.val.tolist()
    # predictor scores of those 256 candidates against the oracle (same embeddings, drivers' head layout)
    torch.manual_seed(2)
    hfull = torch.randn(n, H, device=DEV)
    lnnn = name != "citation2"
    pred = M.predictor_dict[pname](H, H, 1, 3, 0.0, 0.0, lnnn).to(DEV).eval()
    sd = {k: v.detach().cpu().clone() for k, v in pred.state_dict().items()}
    hsub = hfull[keep].cpu()
    args = SimpleNamespace(sum=1.0)
    if pname == "cn5":
        ref = O.cn5_forward(sd, hsub, oc1, oc2, esub, lnnn)
        pool = O.cn5_pool(hsub, oc1, oc2, sd["innerprod"])[:2]
    else:
        ref = O.cn7_forward(sd, hsub, oc1, oc2, esub, args.sum, lnnn)
        pool = O.cn7_pool(hsub, oc1, oc2, args.sum)[:2]
    with torch.no_grad():
        out = pred(hfull, adj, h1, h2, sub, args).cpu()
    sd64 = {k: v.double() for k, v in sd.items()}
    ref64 = O._heads(sd64, hsub.double(), pool[0].double(), pool[1].double(), esub, lnnn, False, False)
    noise = (ref.double() - ref64).abs().max().item()
    err = (out.double() - ref64).abs().max().item()
    assert err <= 1e-5 + 1e-5 * ref.abs().max().item() + 2 * noise, (err, noise)


# ---- ddi at its batch size: B = 32 768 through the LDS-histogram intersection ------------------------------
def test_ddi_full_batch(hiplib):
    import ocn_amd.model as M
    from ocn_amd.synth import sample_edges
    from ocn_amd.utils import CNState, adjoverlap, sparse_tensor_multiply
    n, shape, adj = _full_graph("ddi")
    adj2 = sparse_tensor_multiply(adj, 1024)
    B = 32768
    r, c, _ = adj.coo()
    rc, cc = r.cpu(), c.cpu()
    e = sample_edges(rc, cc, n, B, seed=1).to(DEV)
    st = CNState(adj, adj, adj2, e)
    st.check_status()
    hist = st.hist_counts()
    assert int(hist[:, 0].sum()) == int(st.cnt1.sum()) and int(hist[:, 1].sum()) == int(st.cnt2.sum())
    assert bool((hist[:, 2] >= torch.maximum(hist[:, 0], hist[:, 1])).all()) and bool((hist[:, 2] <= hist[:, 0] + hist[:, 1]).all())
    sw = CNState(adj, adj, adj2, e.flip(0).contiguous())
    assert torch.equal(sw.cnt1, st.cnt1)
    perm = torch.randperm(B, device=DEV)
    pm = CNState(adj, adj, adj2, e[:, perm].contiguous())
    assert torch.equal(pm.cnt1, st.cnt1[perm]) and torch.equal(pm.cnt2, st.cnt2[perm]) and torch.equal(pm.hist_counts(), hist)
    # A² of the ddi shape is (nearly) full: every neighbour of a non-isolated source is a cn2 entry
    deg = adj.storage.rowcount()
    a2 = adj2.storage.rowcount()
    full = a2[e[1]] == n
    assert bool(full.any()) and torch.equal(st.cnt2[full], deg[e[0]][full].to(torch.int32))
    # the oracle on the first 256 candidates (whole graph: 4 267 nodes)
    oadj = O.SpM(rc, cc, None, n, n)
    oadj2 = O.adj2_by_block(oadj, 1024)
    sub = e[:, :256].contiguous()
    esub = sub.cpu()
    oc1, oc2 = O.adjoverlap(oadj, oadj, esub), O.adjoverlap(oadj, oadj2, esub)
    h1, h2 = adjoverlap(adj, adj, sub), adjoverlap(adj, adj2, sub)
    assert h1.counts().cpu().tolist() == torch.bincount(oc1.row, minlength=256).tolist()
    assert h2.counts().cpu().tolist() == torch.bincount(oc2.row, minlength=256).tolist()
    H = 64
    torch.manual_seed(4)
    hfull = torch.randn(n, H, device=DEV)
    pred = M.predictor_dict["cn7"](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
    sd = {k: v.detach().cpu().clone() for k, v in pred.state_dict().items()}
    ref = O.cn7_forward(sd, hfull.cpu(), oc1, oc2, esub, 2.74, True)
    with torch.no_grad():
        out = pred(hfull, adj, h1, h2, sub, SimpleNamespace(sum=2.74)).cpu()
    err = (out - ref).abs().max().item()
    assert err <= 1e-5 + 1e-5 * ref.abs().max().item(), err
