"""CPU suite, part 2: the C-ABI library loads and exports everything include/ocn_hip.h declares,
and the host-side mirror of the reference interface (names, signatures, state_dict keys, error
behaviour).  No compute calls — there is no GPU here."""
import inspect
import os
import re

import pytest
import torch

from oracle import ocn_oracle as O
from ocn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ocn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ocn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(hiplib):
    names = _declared()
    assert len(names) >= 13
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    for n in names:
        assert getattr(hiplib, n) is not None
    assert hiplib.ocn_abi_version() == _lib.ABI_VERSION == 9
    assert hiplib.ocn_scan_workspace_bytes(65536) >= 8 * (65536 // 2048 + 2)
    assert hiplib.ocn_spgemm_max_cols() >= 1_000_000


def test_entries_reject_bad_arguments_before_any_launch(hiplib):
    """Argument errors come back as OCN_EINVAL (-1) from the C entries themselves — checked here without a GPU, since
    every one of these returns before its first HIP call (ocn_hip.h: error behaviour)."""
    import ctypes
    E, NULL = -1, None
    one = ctypes.c_int64(0)
    p = ctypes.cast(ctypes.pointer(one), ctypes.c_void_p)          # a non-NULL host pointer: never dereferenced on these paths
    assert hiplib.ocn_heads_fused(NULL, NULL) == E
    args = _lib.OcnHeadsArgs()                                     # all-NULL pointers, H = 0
    assert hiplib.ocn_heads_fused(ctypes.byref(args), NULL) == E
    args.H, args.B = 64, 4                                         # a width the fused kernel is not built for
    assert hiplib.ocn_heads_fused(ctypes.byref(args), NULL) == E
    assert hiplib.ocn_heads_split_weight(NULL, 256, 256, 1.0, NULL, NULL) == E
    assert hiplib.ocn_heads_split_weight(p, 250, 256, 1.0, p, NULL) == E                    # N not a multiple of 32
    assert hiplib.ocn_heads_split_weight(p, 256, 256, 0.0, p, NULL) == E                    # scale must be positive
    assert hiplib.ocn_heads_panel_bytes(256, 256) == 256 * 256 * 4 and hiplib.ocn_heads_panel_bytes(250, 256) == 0
    assert hiplib.ocn_cn_colsum_exact(NULL, NULL, NULL, -1, NULL, NULL, NULL, NULL, 0, NULL, 0, NULL, NULL, NULL, NULL, NULL,
                                      NULL, NULL) == E
    assert hiplib.ocn_cn_colsum_exact(p, p, p, 4, p, p, NULL, NULL, 1 << 33, p, 4, p, p, NULL, p, NULL, p, NULL) == E   # flags_cap >= 2^32
    too_many = hiplib.ocn_walk_prep_max_batch() + 1
    assert hiplib.ocn_walk_prep(p, NULL, p, p, too_many, 2, p, p, p, NULL, p, p, p, p, p, p, p, p, NULL) == E
    assert hiplib.ocn_dense_from_csr(p, p, 100, 100, p, p, NULL) == E                       # row stride not a multiple of 64
    assert hiplib.ocn_dense_block_mm_bits(p, p, 128, 128, 8, 40, 0, 32, 0, p, 4, NULL) == E    # block start not a multiple of 32
    assert hiplib.ocn_bitrows_from_csr(p, p, -1, p, 4, NULL) == E
    assert hiplib.ocn_bitrows_count(p, 1, 4, 64, p, NULL) == E                              # stride shorter than the columns
    assert hiplib.ocn_batch_prep(p, p, -1, p, p, 10, NULL, NULL, NULL, 0, NULL) == E
    assert hiplib.ocn_zero_regions(NULL, NULL, 9, NULL) == E
    assert hiplib.ocn_check_edges(p, p, -1, 10, 10, p, NULL) == E
    assert hiplib.ocn_cn_flags(p, p, p, p, NULL, NULL, NULL, 0, NULL, 0, p, p, NULL, -1, 10, p, p, 0, p, p, NULL, p, NULL, NULL, NULL) == E
    assert hiplib.ocn_cn_flags(p, p, NULL, NULL, NULL, NULL, NULL, 0, NULL, 0, p, p, NULL, 4, 10, p, p, 0, p, p, NULL, p, NULL,
                               NULL, NULL) == E                                                    # neither a CSR nor bit rows for T1
    assert hiplib.ocn_cn_gather(p, p, p, p, NULL, 4, p, p, NULL, p, p, 0, 0, p, p, p, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == E   # H = 0
    assert hiplib.ocn_order_by_node_finish(p, 4, 0, p, p, NULL) == E


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "ocn_hip.h")).read()
    for cite in ("utils.py:162-183", "model.py:2261", "model.py:2426-2429", "model.py:42-55",
                 "NeighborOverlap_large.py:68-74"):
        assert cite in txt


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OcnHipError, match="no CPU fallback"):
        _lib.lib()


def test_product_path_refuses_cpu_tensors(hiplib):
    from ocn_amd import ops
    from ocn_amd.model import predictor_dict
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import adjoverlap
    adj = SparseTensor.from_edge_index(torch.tensor([[0, 1], [1, 0]]), sparse_sizes=(3, 3))
    e = torch.tensor([[0], [1]])
    pred = predictor_dict["cn5"](8, 8, 1, 3, 0.0).eval()
    with torch.no_grad(), pytest.raises(_lib.OcnHipError, match="no CPU path"):
        pred(torch.randn(3, 8), adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj, e), e)
    with pytest.raises(_lib.OcnHipError):
        ops.spmm_csr(adj._rowptr, adj._col, torch.randn(3, 16))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ocn_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} references the oracle"


def test_perm_iterator_matches_reference_semantics():
    from ocn_amd.utils import PermIterator
    it = PermIterator("cpu", 10, 4, training=False)
    assert len(it) == 3 and [b.tolist() for b in it] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert [b.tolist() for b in it] == [b.tolist() for b in O.perm_batches(10, 4)]
    torch.manual_seed(0)
    it = PermIterator("cpu", 10, 4, training=True)
    got = [b for b in it]
    assert len(it) == 2 and len(got) == 2 and all(b.numel() == 4 for b in got)     # drop-last
    assert len(set(torch.cat(got).tolist())) == 8


def test_predictor_signatures_and_state_dict_keys():
    from ocn_amd.model import predictor_dict
    assert {"cn5", "cn7"} <= set(predictor_dict)
    ctor = ["in_channels", "hidden_channels", "out_channels", "num_layers", "dropout", "edrop", "ln",
            "cndeg", "use_xlin", "tailact", "twolayerlin", "beta"]
    for name in ("cn5", "cn7"):
        cls = predictor_dict[name]
        assert list(inspect.signature(cls.__init__).parameters)[1:] == ctor            # model.py:2173-2185
        assert list(inspect.signature(cls.forward).parameters)[1:] == ["x", "adj", "cn1", "cn2", "tar_ei", "filled1"]
    assert list(inspect.signature(predictor_dict["cn5"].multidomainforward).parameters)[1:] == [
        "x", "adj", "cn1", "cn2", "tar_ei", "filled1", "cndropprobs"]
    assert list(inspect.signature(predictor_dict["cn7"].multidomainforward).parameters)[1:] == [
        "x", "adj", "cn1", "cn2", "tar_ei", "args", "filled1", "cndropprobs"]

    def keys(**kw):
        return set(predictor_dict["cn5"](16, 16, 1, 3, 0.1, 0.0, **kw).state_dict())

    c6 = predictor_dict["cn6"]                                                             # model.py:2445-2458, 2950
    assert list(inspect.signature(c6.__init__).parameters)[1:] == ctor
    assert list(inspect.signature(c6.forward).parameters)[1:] == ["x", "adj", "cn1", "cn2", "cn3", "tar_ei", "args"]
    assert list(inspect.signature(c6.multidomainforward).parameters)[1:] == [
        "x", "adj", "cn1", "cn2", "cn3", "tar_ei", "args", "cndropprobs"]
    k6 = set(c6(16, 16, 1, 3, 0.1, 0.0, True).state_dict())
    assert {"xcn3lin.0.weight", "xcn3lin.3.bias", "xcn3lin.4.weight", "xcn3lin.7.weight"} <= k6
    assert not any(k.startswith("xcn4lin") for k in k6) and "alpha" in k6 and "innerprod" in k6

    base = keys()
    lin = lambda p, idx: {f"{p}.{i}.{s}" for i in idx for s in ("weight", "bias")}
    want = ({"beta", "alpha", "innerprod", "dropadj.ratio"} | lin("xcnlin", (0, 3, 7)) | lin("xcn1lin", (0, 3, 7))
            | lin("xcn2lin", (0, 3, 7)) | lin("xcn4lin", (0, 3, 7)) | lin("xijlin", (0, 4)) | lin("lin", (0, 8)))
    assert base == want
    assert keys(ln=True) == want | lin("xcnlin", (4,)) | lin("xcn1lin", (4,)) | lin("xcn2lin", (4,)) \
        | lin("xcn4lin", (4,)) | lin("xijlin", (1,)) | lin("lin", (1,))
    assert keys(tailact=True) == want - lin("xcnlin", (7,)) - lin("xijlin", (4,))
    assert keys(twolayerlin=True, ln=True) >= lin("lin", (4, 5))
    assert keys(use_xlin=True) == want | lin("xlin", (0, 3))
    # same registration order as the reference => same init under the same seed; buffers checkpointed
    p = predictor_dict["cn7"](8, 8, 1, 3, 0.0, beta=0.33)
    assert p.beta.item() == pytest.approx(0.33) and p.alpha.tolist() == [1, 1, 1] and p.innerprod.tolist() == [0.0]
    p.load_state_dict({k: v for k, v in p.state_dict().items() if "xcn4lin" not in k}, strict=False)


def test_encoder_state_dict_keys():
    from ocn_amd.model import GCN, GCN2, GCN3, convdict, convdict2, convdict3
    names = {"gcn", "gcn_cached", "sage", "gin", "max", "puremax", "puresum", "puremean", "puregcn", "none"}
    assert set(convdict) == set(convdict2) == set(convdict3) == names
    sig = list(inspect.signature(GCN.__init__).parameters)[1:]
    assert sig == ["in_channels", "hidden_channels", "out_channels", "num_layers", "dropout", "ln", "res",
                   "max_x", "conv_fn", "jk", "edrop", "xdropout", "taildropout", "noinputlin"]
    g = GCN(128, 256, 256, 1, 0.05, True, False, -1, "gin", True, 0.0, xdropout=0.7, taildropout=0.3)
    assert set(g.state_dict()) == {"adjdrop.ratio", "jkparams", "convs.0.lin.weight", "convs.0.bias",
                                   "lins.0.0.weight", "lins.0.0.bias"}
    g = GCN(1433, 256, 256, 1, 0.05, True, False, -1, "puregcn", True)
    assert set(g.state_dict()) == {"adjdrop.ratio", "jkparams", "xemb.1.weight", "xemb.1.bias"}
    g = GCN2(58, 64, 64, 1, 0.0, True, False, 57, "gcn", True)
    assert set(g.state_dict()) == {"adjdrop.ratio", "jkparams", "xemb.0.weight", "convs.0.lin.0.weight",
                                   "lins.0.0.weight", "lins.0.0.bias"}
    g = GCN3(128, 32, 32, 5, 0.28, True, True, -1, "gcn", True)
    assert {f"convs.{i}.lin.0.weight" for i in range(5)} <= set(g.state_dict())
    assert "lins.4.0.weight" not in g.state_dict() and "lins.3.0.weight" in g.state_dict()


def test_sparse_container_matches_oracle_formats():
    """Format conversions are plain torch ops and run on CPU: compare with the oracle's SpM."""
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import chung_lu_graph
    n = 120
    ei = chung_lu_graph(n, 6, 30, seed=4)
    dup = torch.cat([ei, ei[:, :10]], dim=1)                       # duplicates + one direction only
    sp = SparseTensor.from_edge_index(dup, sparse_sizes=(n, n))
    o = O.from_edge_index(dup, n)
    assert sp.nnz() == o.nnz and sp.coo()[1].tolist() == o.col.tolist()
    s, os_ = sp.to_symmetric(), O.to_symmetric(o)
    assert s.coo()[0].tolist() == os_.row.tolist() and s.coo()[1].tolist() == os_.col.tolist()
    assert s.sizes() == [n, n] and s.sparse_sizes() == (n, n) and s.size(1) == n
    idx = torch.tensor([5, 5, 0, n - 1])
    sel, osel = s[idx], O.row_select(os_, idx)
    assert sel.coo()[0].tolist() == osel.row.tolist() and sel.coo()[1].tolist() == osel.col.tolist()
    assert s.storage.rowcount().tolist() == os_.rowcount().tolist()
    assert s.max_rowcount() == int(os_.rowcount().max())
    assert torch.equal(s.to_dense(), os_.to_dense())
    assert torch.equal(s.sum(dim=0), os_.to_dense().sum(0))
    rt = SparseTensor.from_torch_sparse_coo_tensor(s.to_torch_sparse_coo_tensor(), False)
    assert rt.coo()[1].tolist() == os_.col.tolist()
    with pytest.raises(IndexError):
        SparseTensor.from_edge_index(torch.tensor([[0], [n]]), sparse_sizes=(n, n))


def test_adjoverlap_signature_and_unsupported_branches():
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.utils import adjoverlap, sparse_tensor_multiply
    assert list(inspect.signature(adjoverlap).parameters) == [
        "adj1", "adj2", "tarei", "filled1", "calresadj", "cnsampledeg", "ressampledeg"]
    assert list(inspect.signature(sparse_tensor_multiply).parameters) == ["spadj", "block_size"]
    adj = SparseTensor.from_edge_index(torch.tensor([[0, 1], [1, 0]]), sparse_sizes=(3, 3))
    h = adjoverlap(adj, adj, torch.tensor([[0, 1], [1, 2]]))
    assert h.sizes() == [2, 3]
    with pytest.raises(NotImplementedError):
        adjoverlap(adj, adj, torch.tensor([[0], [1]]), calresadj=True)


def test_library_knobs_have_their_documented_defaults(hiplib):
    """Host-only entries: the bound of the small-batch head (include/ocn_hip.h) and the small-graph bound of the intersection pass."""
    assert hiplib.ocn_heads_small_batch(-1) == 16384
    assert hiplib.ocn_heads_small_batch(100) == 16384 and hiplib.ocn_heads_small_batch(16384) == 100
    assert hiplib.ocn_cn_flags_small_graph_cols() == 8192


def test_heads_kernel_isa_audit():
    """The fused heads' k-step is hand-placed inline asm whose waits are counted by hand, and the small-batch form requests its
    weight fragments by asm and waits for them by count (heads.hip).  tools/check_heads_asm.py compiles both and checks the ISA: no instruction touches the destination of an LDS read the lgkmcnt ladder has not
    retired, no VALU result feeds an MFMA within two wait states, no reader of an MFMA result within twelve, no scratch."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("check_heads_asm", os.path.join(os.path.dirname(__file__), "..", "tools", "check_heads_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    problems, stats = mod.audit(mod.compile_asm())
    assert len(stats) == 8 and all(s["mfma"] > 0 and s["ds_read"] > 0 for s in stats.values())      # two forms x H in {128, 256} x LayerNorm on / off
    assert not problems, problems[:5]


def test_training_and_sort_kernels_do_not_spill(tmp_path):
    """The intersection / pooling / scan / column-sum kernels and the round-3 kernels of the training loop (COO -> CSR build with its row sorts, weight gradient, deterministic pooling
    backward): compiled for gfx950, none keeps a register in scratch and the weight-gradient kernel holds its MFMAs."""
    import os
    import re
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    seen = {}
    for name in ("coo_csr", "wgrad", "pool_bwd", "cn_stage", "scan", "colsum"):       # (cn_stage: round 3 shipped a 104-register spill in the H = 512 hub-row kernel)
        out = tmp_path / (name + ".s")
        subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                        "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "ocn_amd", "csrc"),
                        os.path.join(root, "ocn_amd", "csrc", name + ".hip"), "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
        for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.vgpr_spill_count:\s+(\d+)", text, re.S):
            kernel, body, spills = m.group(1), m.group(2), int(m.group(3))
            if ".private_segment_fixed_size" in body:
                scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
                seen[kernel] = (spills, scratch)
        if name == "wgrad":
            assert text.count("v_mfma_f32_32x32x16_bf16") >= 24            # four tiles x six cross terms
    assert len(seen) >= 10, sorted(seen)
    assert all(v == (0, 0) for v in seen.values()), {k: v for k, v in seen.items() if v != (0, 0)}


def test_overlapped_steps_host_order_without_a_gpu():
    """pipeline.overlapped_steps on one stream (no HIP here): phase A of batch t + 1 is enqueued before phase B of batch t,
    results come back in batch order, the hooks bracket every step, and no scratch set is begun again before the batch
    that used it has finished."""
    from ocn_amd.pipeline import overlapped_steps
    log = []

    def begin(it):
        log.append(("A", it))
        return it

    def finish(tok):
        log.append(("B", tok))
        return tok * 10

    outs = list(overlapped_steps(begin, finish, 5, before_step=lambda it: log.append(("pre", it)),
                                 after_step=lambda it: log.append(("post", it)), overlap=False))
    assert outs == [0, 10, 20, 30, 40]
    order = [e for e in log if e[0] in "AB"]
    assert order == [("A", 0), ("A", 1), ("B", 0), ("A", 2), ("B", 1), ("A", 3), ("B", 2), ("A", 4), ("B", 3), ("B", 4)]
    for it in range(5):
        assert log.index(("pre", it)) < log.index(("B", it)) < log.index(("post", it))
    assert list(overlapped_steps(begin, finish, 0, overlap=False)) == []


def test_the_shipped_library_was_built_without_experiment_macros(hiplib):
    """the kernels carry ~30 `OCN_X_*` timing-experiment switches, several of which give wrong results when
    defined.  The product build never defines one: `_lib.build_flags()` holds none, the library records the compile command of
    every translation unit (-frecord-command-line -> .GCC.command.line) and none of them names one, and no `ocn_debug_*` entry
    (they only exist under those switches) is exported."""
    import re
    from ocn_amd import _lib
    assert not any("OCN_X_" in f for f in _lib.build_flags())
    blob = open(_lib.LIB_PATH, "rb").read()
    cmds = re.findall(rb"[^\x00]*-frecord-command-line[^\x00]*", blob)
    units = {os.path.basename(s) for s in _lib.sources()}
    seen = {u for u in units for c in cmds if u.encode() in c}
    assert seen == units, f"compile commands recorded for {sorted(seen)} of {sorted(units)}"
    assert all(b"-ffp-contract=off" in c for c in cmds) and sum(b"gfx950" in c for c in cmds) >= len(units)
    assert b"OCN_X_" not in blob
    assert b"ocn_debug_" not in blob


def test_sparse_tensor_is_freed_without_the_cycle_collector():
    """A training loop builds a masked adjacency and its A² (7 GB of bit rows at the collab shape) PER BATCH: they must go when
    the step drops them, not when Python's cycle collector runs (round 4: `adj.storage` holding its owner back made every
    SparseTensor a reference cycle; with fewer Python objects per step the collector ran rarely and the allocator grew by 6.5 GB
    every few steps — 0.2 s stalls)."""
    import gc
    import weakref
    from ocn_amd.sparse import SparseTensor
    was = gc.isenabled()
    gc.disable()
    try:
        a = SparseTensor(rowptr=torch.tensor([0, 1, 2]), col=torch.tensor([1, 0]), sparse_sizes=(2, 2))
        assert a.storage.rowcount().tolist() == [1, 1] and a.storage.col().tolist() == [1, 0]
        v = a.to_torch_sparse_coo_tensor()
        r = weakref.ref(a)
        del a, v
        assert r() is None
    finally:
        if was:
            gc.enable()
