"""Writes the fixtures under tests/golden/.

Two kinds, kept apart on purpose:

* ``appendix_c.json`` — HAND-DERIVED known answers, worked
  out from reading /root/reference/utils.py:146-183,248-285 and
  model.py:2252-2440,3102-3226.  They are literals below: nothing computes them.
  They are the only pins the oracle has (the reference holds no tests or golden
  vectors of its own and cannot run in the build container — SURVEY.md §8c).
* ``oracle_vectors.json`` — seeded small cases with outputs produced by
  ``oracle/ocn_oracle.py`` (regression vectors for the HIP path on the GPU box,
  where the oracle is also re-run live).  These are ORACLE outputs, not
  reference outputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

APPENDIX_C = {
    "source": "SURVEY.md Appendix C (hand-derived from reference source reading)",
    "smoke_inputs": {
        # utils.py:332-335 — the reference's own unasserted smoke inputs
        "adj1": [[0, 0, 1, 2, 3], [0, 1, 1, 2, 3]],
        "adj2": [[0, 3, 1, 2, 3], [0, 1, 1, 2, 3]],
        "n": 4,
        "spmoverlap": [[0, 1, 2, 3], [0, 1, 2, 3]],
        "spmoverlap_values": [1.0, 1.0, 1.0, 1.0],
    },
    "path_graph": {
        "n": 4,
        "undirected_edges": [[0, 1], [1, 2], [2, 3], [0, 2]],
        "rows": {"0": [1, 2], "1": [0, 2], "2": [0, 1, 3], "3": [2]},
        "a2_rows": {"0": [0, 1, 2, 3], "1": [0, 1, 2, 3], "2": [0, 1, 2], "3": [0, 1, 3]},
        "batch": [[0, 1], [1, 3], [0, 3]],
        "cn1_rows": [[2], [2], [2]],
        "cn2_rows": [[1, 2], [0], [1]],
        "cn1_counts": [1, 1, 1],
        "cn2_counts": [2, 1, 1],
        "S1": [0, 0, 3, 0],
        "cn2_colsum": [1, 2, 1, 0],
        "union_pattern": [[0, 1], [0, 2], [1, 0], [1, 2], [2, 1], [2, 2]],
        "cn5_innerprod_0": {
            "scale": 0.33333334,
            "S2": [1.0, 2.0, 1.0, 1.0],
            "ncn2": [0.5, 1.0, 1.0, 0.0, 0.5, 0.0],
        },
        "cn5_innerprod_0.37": {
            "nip": 1.11,
            "S2": [1.0, 2.0, -0.11000001, 1.0],
            "ncn2": [0.5, -5.727272, 1.0, 3.363636, 0.5, 3.363636],
        },
        "q2_single_edge_batch": {"batch": [[0, 1]], "S1_col2": 1, "cn5_inv1_col2": 0.0},
        # Hand-derived from NeighborOverlap_large_ppa.py:147-173: cn2[e,k] = |N(k) ∩ N(j_e)| for k in N(i_e),
        # zeros dropped.  e0 = (0,1): N(1)∩N(1) = {0,2} -> 2, N(2)∩N(1) = {0} -> 1; e1 = (1,3): N(0)∩N(3) = {2}
        # -> 1, N(2)∩N(3) = {} dropped; e2 = (0,3): N(1)∩N(3) = {2} -> 1, N(2)∩N(3) = {} dropped.
        "walk_route": {"cn2_rows": [[1, 2], [0], [1]], "cn2_values": [[2.0, 1.0], [1.0], [1.0]],
                       "walk_colsum": [1.0, 3.0, 1.0, 0.0]},
        # model.py:3114-3126, 3186-3216 with x[k] = one-hot(k): S1 = [0,0,3,0] so ncn1 = 1/3 on column 2 of every
        # row; cn2 is pooled raw (the walk counts above)
        "cn7_walk": {"xcn1": [[0, 0, 0.33333334, 0]] * 3, "xcn2": [[0, 2.0, 1.0, 0], [1.0, 0, 0, 0], [0, 1.0, 0, 0]]},
        # model.py:2846-2933 at innerprod = 0: N^3(1) = {0,1,2,3}, N^3(3) = N(0) ∪ N(1) ∪ N(3) = {0,1,2};
        # cn3 = N(i) ∩ N^3(j); S3 = its column counts [1,2,3,0]; ncn3 = 1/S3
        "cn6_innerprod_0": {"a3_rows": {"1": [0, 1, 2, 3], "3": [0, 1, 2]}, "cn3_rows": [[1, 2], [0, 2], [1, 2]],
                            "S3": [1.0, 2.0, 3.0, 1.0],
                            "xcn3": [[0, 0.5, 0.33333334, 0], [1.0, 0, 0.33333334, 0], [0, 0.5, 0.33333334, 0]]},
    },
}


def oracle_vectors():
    from oracle import ocn_oracle as O
    from ocn_amd.synth import chung_lu_graph, sample_edges

    out = []
    for seed, (n, avg, B, H) in enumerate([(64, 6, 40, 8), (300, 10, 257, 32), (1000, 12, 512, 64)]):
        g = torch.Generator().manual_seed(100 + seed)
        ei = chung_lu_graph(n, avg_deg=avg, max_deg=n // 3, seed=100 + seed)
        adj = O.to_symmetric(O.from_edge_index(ei, n))
        adj2 = O.adj2_sparse(adj)
        edges = sample_edges(adj.row, adj.col, n, B, seed=200 + seed)
        x = torch.randn(n, H, generator=g)
        cn1 = O.adjoverlap(adj, adj, edges)
        cn2 = O.adjoverlap(adj, adj2, edges)
        rec = dict(n=n, H=H, edge_index=ei.tolist(), batch=edges.tolist(), x_seed=100 + seed,
                   cn1_counts=torch.bincount(cn1.row, minlength=B).tolist(),
                   cn2_counts=torch.bincount(cn2.row, minlength=B).tolist(),
                   a2_nnz=adj2.nnz)
        for ip in (0.0, 0.37):
            xcn1, xcn2, aux = O.cn5_pool(x, cn1, cn2, torch.tensor([ip]))
            rec[f"cn5_ip{ip}"] = dict(xcn1_sum=xcn1.double().sum().item(),
                                      xcn2_sum=xcn2.double().sum().item(),
                                      xcn1_row0=xcn1[0].tolist(), xcn2_row0=xcn2[0].tolist(),
                                      scale=float(aux["scale"]))
        xcn1, xcn2, _ = O.cn7_pool(x, cn1, cn2, 2.74)
        rec["cn7_sum2.74"] = dict(xcn1_row0=xcn1[0].tolist(), xcn2_row0=xcn2[0].tolist())
        if n <= 300:                                   # the 3-hop predictor (cn6): A³ stays small here
            adj3 = O.adj3_sparse(adj, adj2)
            cn3 = O.adjoverlap(adj, adj3, edges)
            rec["a3_nnz"] = adj3.nnz
            rec["cn3_counts"] = torch.bincount(cn3.row, minlength=B).tolist()
            for ip in (0.0, 0.37):
                _, _, xcn3, aux = O.cn6_pool(x, cn1, cn2, cn3, torch.tensor([ip]))
                rec[f"cn6_ip{ip}"] = dict(xcn3_sum=xcn3.double().sum().item(), xcn3_row0=xcn3[0].tolist())
        out.append(rec)
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "appendix_c.json"), "w") as f:
        json.dump(APPENDIX_C, f, indent=1)
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(oracle_vectors(), f)
    print("wrote fixtures")
