"""Independent set-based model of the CN stage — TEST INFRASTRUCTURE ONLY.

Written from the *mathematical* statement in SURVEY.md Appendix A (not from the
op sequence the oracle follows), with Python sets / dicts and float64
arithmetic, so that an error in reading the reference's tensor code in
``ocn_oracle.py`` does not silently repeat here.  Small graphs only.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Set, Tuple

import numpy as np


def neighbours(n: int, und_edges: Sequence[Tuple[int, int]]) -> List[Set[int]]:
    nb: List[Set[int]] = [set() for _ in range(n)]
    for a, b in und_edges:
        nb[a].add(b)
        nb[b].add(a)
    return nb


def two_hop(nb: List[Set[int]]) -> List[Set[int]]:
    """Row pattern of A·A: k ∈ N²(j) iff ∃m: m ∈ N(j) ∧ k ∈ N(m)."""
    return [set().union(*[nb[m] for m in nb[j]]) if nb[j] else set() for j in range(len(nb))]


def cn_sets(nb, nb2, batch):
    cn1 = [sorted(nb[i] & nb[j]) for i, j in batch]
    cn2 = [sorted(nb[i] & nb2[j]) for i, j in batch]
    return cn1, cn2


def walk_counts(nb, batch):
    """pygho route: cn2 value at (e,k) = |N(k) ∩ N(j_e)| for k ∈ N(i_e), zero dropped."""
    out = []
    for i, j in batch:
        row = {}
        for k in sorted(nb[i]):
            c = len(nb[k] & nb[j])
            if c:
                row[k] = float(c)
        out.append(row)
    return out


def cn5_pool(n: int, h: np.ndarray, cn1, cn2, innerprod: float):
    """Appendix A.3 steps 1-6 in float64."""
    h = h.astype(np.float64)
    S1 = np.zeros(n)
    for row in cn1:
        for k in row:
            S1[k] += 1
    inv1 = np.where(S1 >= 2, 1.0 / np.maximum(S1, 1), 0.0)
    present = [inv1[k] for row in cn1 for k in row]
    union_nonempty = any(len(r) for r in cn1) or any(len(r) for r in cn2)
    scale = (max(present) if present else 0.0) if union_nonempty else 1.0
    nip = innerprod / scale if scale > 0 else innerprod
    v: List[Dict[int, float]] = []
    for r1, r2 in zip(cn1, cn2):
        row = {}
        for k in sorted(set(r1) | set(r2)):
            row[k] = (1.0 if k in r2 else 0.0) - nip * (inv1[k] if k in r1 else 0.0)
        v.append(row)
    S2 = np.zeros(n)
    for row in v:
        for k, val in row.items():
            S2[k] += val
    S2 = np.where(S2 == 0, 1.0, S2)
    xcn1 = np.stack([sum((inv1[k] * h[k] for k in r), np.zeros(h.shape[1])) for r in cn1])
    xcn2 = np.stack([sum((val / S2[k] * h[k] for k, val in r.items()), np.zeros(h.shape[1]))
                     for r in v])
    return xcn1, xcn2, dict(S1=S1, inv1=inv1, scale=scale, nip=nip, S2=S2, v=v)


def cn7_pool(n: int, h: np.ndarray, cn1, cn2_vals, sum_fill: float):
    """Appendix A.4: cn1 column-normalised with ``sum_fill`` for singleton columns, cn2 raw."""
    h = h.astype(np.float64)
    S1 = np.zeros(n)
    for row in cn1:
        for k in row:
            S1[k] += 1
    inv1 = np.where(S1 >= 2, 1.0 / np.maximum(S1, 1), sum_fill)
    xcn1 = np.stack([sum((inv1[k] * h[k] for k in r), np.zeros(h.shape[1])) for r in cn1])
    xcn2 = np.stack([sum((val * h[k] for k, val in r.items()), np.zeros(h.shape[1]))
                     for r in cn2_vals])
    return xcn1, xcn2, dict(S1=S1, inv1=inv1)


def three_hop(nb: List[Set[int]], nb2: List[Set[int]]) -> List[Set[int]]:
    """Row pattern of A²·A: k ∈ N³(j) iff ∃m ∈ N²(j): k ∈ N(m)."""
    return [set().union(*[nb[m] for m in nb2[j]]) if nb2[j] else set() for j in range(len(nb))]


def cn6_pool(n: int, h: np.ndarray, cn1, cn2, cn3, innerprod: float):
    """3-hop predictor pooling: stage 1 as cn5; stage 2: cn3' = cn3 − nip·ncn1 − nip·ncn2' on the union
    of the three patterns (ncn2' lives on cn1 ∪ cn2), column-normalised (zero sums -> 1)."""
    xcn1, xcn2, aux = cn5_pool(n, h, cn1, cn2, innerprod)
    h = h.astype(np.float64)
    inv1, S2, v2, nip = aux["inv1"], aux["S2"], aux["v"], aux["nip"]
    any1 = any(len(r) for r in cn1)
    nonempty = any1 or any(len(r) for r in cn2) or any(len(r) for r in cn3)
    present = [inv1[k] for row in cn1 for k in row]
    scale = (max(present) if present else 0.0) if nonempty else 1.0
    nip_b = innerprod / scale if scale > 0 else innerprod
    v3: List[Dict[int, float]] = []
    for r1, r2v, r3 in zip(cn1, v2, cn3):
        row = {}
        for k in sorted(set(r1) | set(r2v) | set(r3)):
            row[k] = ((1.0 if k in r3 else 0.0) - nip_b * (inv1[k] if k in r1 else 0.0)
                      - nip_b * (r2v[k] / S2[k] if k in r2v else 0.0))
        v3.append(row)
    S3 = np.zeros(n)
    for row in v3:
        for k, val in row.items():
            S3[k] += val
    S3 = np.where(S3 == 0, 1.0, S3)
    xcn3 = np.stack([sum((val / S3[k] * h[k] for k, val in r.items()), np.zeros(h.shape[1])) for r in v3])
    return xcn1, xcn2, xcn3, dict(aux, S3=S3, nip=nip)
