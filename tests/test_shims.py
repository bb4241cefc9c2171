"""CPU suite: the shim packages resolve the reference drivers' import block (NeighborOverlap_large.py:1-19) without
edits, and the torch_sparse surface added for it (SparseTensor.mul / +) matches dense arithmetic."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the import statements of NeighborOverlap_large.py:1-19 that do not come with Python / torch / sklearn
DRIVER_IMPORTS = [
    "from torch_sparse import SparseTensor",                                   # :6
    "import torch_geometric.transforms as T",                                  # :7
    "from model import predictor_dict, convdict, GCN, DropEdge",               # :8
    "from sklearn.metrics import roc_auc_score, average_precision_score",      # :10
    "from ogb.linkproppred import PygLinkPropPredDataset, Evaluator",          # :11
    "from torch_geometric.utils import negative_sampling",                     # :12
    "from torch.utils.tensorboard import SummaryWriter",                       # :13
    "from utils import PermIterator",                                          # :14
    "from ogbdataset import loaddataset",                                      # :16
    "from utils import adjoverlap",                                            # :18
    "from utils import sparse_tensor_multiply",                                # :19
    "from model import predictor_dict, convdict, GCN, DropEdge,GCN2,GCN3",     # NeighborOverlapCitation2.py:9
]


def test_driver_import_block_resolves_against_the_shims():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "shims"), ROOT]))
    prog = "\n".join(DRIVER_IMPORTS) + "\nw = SummaryWriter('x'); w.add_text('a', 'b')\nprint(sorted(predictor_dict))\n"
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "cn5" in r.stdout and "cn7" in r.stdout


@pytest.mark.skipif(not os.path.exists("/root/reference/NeighborOverlap_large.py"), reason="reference tree not present")
def test_every_import_of_the_reference_drivers_resolves():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_driver_imports.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]


def test_sparse_tensor_mul_and_add():
    from ocn_amd.sparse import SparseTensor
    g = torch.Generator().manual_seed(0)
    a = (torch.rand(7, 9, generator=g) < 0.3).float() * torch.randint(1, 4, (7, 9), generator=g).float()
    b = (torch.rand(7, 9, generator=g) < 0.3).float()
    sa, sb = SparseTensor.from_dense(a), SparseTensor.from_dense(b)
    col = torch.rand(1, 9, generator=g)
    row = torch.rand(7, 1, generator=g)
    assert torch.equal(sa.mul(col).to_dense(), a * col)                          # model.py:2272
    assert torch.equal((sa * row).to_dense(), a * row)
    assert torch.equal((sa + sb).to_dense(), a + b)                              # utils.py:318-321
    pat = SparseTensor(row=sb.coo()[0], col=sb.coo()[1], value=None, sparse_sizes=(7, 9))
    assert torch.equal(pat.mul(col).to_dense(), b * col)


def test_synthetic_dataset_needs_an_explicit_opt_in_and_announces_itself():
    """the unchanged driver must not print Hits@K on a silently substituted graph.  Without OCN_SYNTH=1 the
    loader raises; with it, one stderr line names shape, seed and scale; a dataset without a shape of its own is refused;
    the negative sampler returns exactly the number of pairs asked for."""
    base = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "shims"), ROOT]))
    prog = "from ogbdataset import loaddataset\nd, s = loaddataset('Cora', False)\nprint(d.num_nodes)\n"
    env = {k: v for k, v in base.items() if k != "OCN_SYNTH"}
    r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode != 0 and "OCN_SYNTH=1" in r.stderr
    r = subprocess.run([sys.executable, "-c", prog], env=dict(env, OCN_SYNTH="1", OCN_SYNTH_SCALE="0.2"), capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SYNTHETIC DATA" in r.stderr and "seed 0" in r.stderr and "scale 0.2" in r.stderr
    r = subprocess.run([sys.executable, "-c", prog.replace("Cora", "Pubmed")], env=dict(env, OCN_SYNTH="1"), capture_output=True, text=True, cwd="/tmp")
    assert r.returncode != 0 and "no synthetic shape" in r.stderr
    prog = ("import torch\nfrom torch_geometric.utils import negative_sampling\n"
            "ei = torch.randint(0, 50, (2, 400))\nprint(negative_sampling(ei, 50, 777).shape[1])\n")
    r = subprocess.run([sys.executable, "-c", prog], env=base, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip() == "777", r.stderr[-2000:]


# The text of the pygho drivers' local get_cn1_cn2 (NeighborOverlap_large_ppa.py:147-173 == NeighborOverlapCitation2.py:78-104),
# RESTATED here from the survey's description of it (call by call; nothing is read from /root/reference at run time).
PYGHO_GET_CN1_CN2 = """
import torch
import torch_sparse
from pygho import SparseTensor as pSparseTensor
from pygho.backend.Spspmm import spsphadamard, spspmm
from pygho.backend.Spmm import spmm

def wrap(adj_t):                       # …_ppa.py:83-90: the adjacency as a pygho tensor
    row, col, val = adj_t.coo()
    if val is None:
        val = torch.ones_like(row, dtype=torch.float)
    return pSparseTensor(torch.stack((row, col)), val, adj_t.sizes(), is_coalesced=True)

def get_cn1_cn2(adj, tedge):
    Ei = adj.index_select([0], tedge[0].unsqueeze(0))
    Ej = adj.index_select([0], tedge[1].unsqueeze(0))
    cn1 = spsphadamard(Ei, Ej)
    Ej2 = spspmm(Ej, 1, adj, 0)
    del Ej
    cn2 = spsphadamard(Ei, Ej2)
    del Ei, Ej2
    cn1 = cn1.to_torch_sparse_coo()
    cn2 = cn2.to_torch_sparse_coo()
    num_nodes = cn1.shape[1]
    num_edges = cn1.shape[0]
    row1, col1 = cn1.indices()
    row2, col2 = cn2.indices()
    value1 = cn1.values()
    value2 = cn2.values()
    cn1 = torch_sparse.SparseTensor(row=row1, col=col1, value=value1, sparse_sizes=(num_edges, num_nodes))
    cn2 = torch_sparse.SparseTensor(row=row2, col=col2, value=value2, sparse_sizes=(num_edges, num_nodes))
    return cn1, cn2
"""


def pygho_namespace():
    """The restated driver code, executed against shims/ (import order as under PYTHONPATH=shims:repo)."""
    shim = os.path.join(ROOT, "shims")
    added = shim not in sys.path
    if added:
        sys.path.insert(0, shim)
    try:
        ns = {}
        exec(compile(PYGHO_GET_CN1_CN2, "<restated get_cn1_cn2>", "exec"), ns)
        return ns
    finally:
        if added:
            sys.path.remove(shim)
            for m in [m for m in sys.modules if m in ("utils", "model", "ogbdataset", "_shimguard")]:
                if getattr(sys.modules[m], "__file__", "") and sys.modules[m].__file__.startswith(shim):
                    del sys.modules[m]


def test_pygho_pattern_of_the_drivers_collapses_to_the_cn_handles():
    """shims/pygho is a lazy algebra: the drivers' get_cn1_cn2, run as written, computes nothing and returns the two
    CNBatch handles of ocn_amd.utils.get_cn1_cn2 over the SAME adjacency object and the SAME candidate tensor (so the
    predictor fuses them into one intersection pass); the wrapped adjacency is an ocn_amd SparseTensor the encoders take;
    anything outside the pattern raises NotImplementedError."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import CNBatch
    ns = pygho_namespace()
    ei = torch.tensor([[0, 1, 0, 2, 1, 2, 2, 3], [1, 0, 2, 0, 2, 1, 3, 2]])
    adj_t = SparseTensor.from_edge_index(ei, sparse_sizes=(4, 4))
    adj = ns["wrap"](adj_t)
    assert isinstance(adj, SparseTensor) and adj.shape == (4, 4) and adj.nnz() == 8
    assert torch.equal(adj.indices, torch.stack(adj_t.coo()[:2])) and torch.equal(adj.values, torch.ones(8))
    tedge = torch.tensor([[0, 1, 0], [1, 3, 3]])
    cn1, cn2 = ns["get_cn1_cn2"](adj, tedge)
    assert isinstance(cn1, CNBatch) and isinstance(cn2, CNBatch)
    assert (cn1.mode, cn2.mode) == ("walk1", "walk2") and cn1.adj1 is adj and cn2.adj1 is adj
    assert cn1.tarei is tedge and cn2.tarei is tedge and cn1.sizes() == [3, 4]
    # a second call (the drivers' test() calls it twice per batch, …_ppa.py:201) gives handles that fuse with the first
    again = ns["get_cn1_cn2"](adj, tedge.clone())
    assert again[1].adj1 is adj and torch.equal(again[1].tarei, tedge)
    # outside the pattern
    from pygho.backend.Spspmm import spsphadamard, spspmm          # resolved from sys.modules (loaded above)
    Ei = adj.index_select([0], tedge[0].unsqueeze(0))
    with pytest.raises(NotImplementedError):
        adj.index_select([1], tedge[0].unsqueeze(0))
    with pytest.raises(NotImplementedError):
        spspmm(Ei, 0, adj, 0)
    with pytest.raises(NotImplementedError):
        spsphadamard(adj, adj)
    with pytest.raises(NotImplementedError):
        spspmm(Ei, 1, adj, 0).to_torch_sparse_coo()
    other = ns["wrap"](adj_t)
    with pytest.raises(NotImplementedError):
        spsphadamard(Ei, other.index_select([0], tedge[1].unsqueeze(0)))


def test_fold_quirk_switch_reaches_the_unchanged_ddi_call(monkeypatch):
    """`utils.sparse_tensor_multiply(spadj)` is what the unchanged ddi command calls (NeighborOverlap_large.py:116); which
    reading of utils.py:318-321 it gets is ops.adj2_fold_quirk (environment OCN_ADJ2_FOLD_QUIRK=1 at import)."""
    from ocn_amd import ops, utils
    seen = []
    monkeypatch.setattr(utils, "block_matrix_multiply", lambda spadj, bs, fold_quirk=False: seen.append((bs, fold_quirk)))
    utils.sparse_tensor_multiply("adj", 1024)
    monkeypatch.setattr(ops, "adj2_fold_quirk", True)
    utils.sparse_tensor_multiply("adj", 512)
    assert seen == [(1024, False), (512, True)]
    env = dict(os.environ, OCN_ADJ2_FOLD_QUIRK="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", "from ocn_amd import ops; print(ops.adj2_fold_quirk)"], env=env, capture_output=True, text=True)
    assert r.stdout.strip() == "True", r.stderr[-1000:]
