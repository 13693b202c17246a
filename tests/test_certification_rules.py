"""The walk's order test (tests/cert_model.py = the rules of csrc/tn_trace_walk.hip, statement for statement) against the
oracle's LITERAL dedupe / pairing (oracle/tn_oracle.c: post_process_row, a restatement of optix_trace_rays.cu:110-266) on crafted
chains: whenever the rules certify a chain, "pair hit k-1 with hit k, drop the short pairs (and hit 2's segment under rule C)"
must give exactly the reference's segments.  Chains are 1..20 tets long with tiny gaps, exact ties and inversions injected at
the entry, at the exit and in between -- including the short chains on which get_common_tetrahedra's EMPTY == EMPTY match of the
two hull faces bites.  CPU only."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import cert_model  # noqa: E402

EMPTY = 0xFFFFFFFF


def _chain_world(n):
    """n tets 0..n-1 along a line, faces f_0..f_n: tet k = vertices k..k+3, face k = (k, k+1, k+2); f_0 and f_n are hull faces"""
    faces = np.zeros((n + 1, 3), np.uint32)
    ft = np.zeros((n + 1, 2), np.uint32)
    for k in range(n + 1):
        faces[k] = (k, k + 1, k + 2)
        a, b = (k - 1 if k >= 1 else None), (k if k < n else None)
        ft[k] = (b, EMPTY) if a is None else ((a, EMPTY) if b is None else (min(a, b), max(a, b)))
    return faces, ft


def _chain_t(rng, n1):
    gaps = rng.uniform(2e-6, 1e-3, n1).astype(np.float32)
    t = (np.float32(1.0) + np.cumsum(gaps, dtype=np.float32)).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):
        r = rng.random()
        s = (int(rng.integers(1, min(5, n1))) if r < 0.3 else
             (int(n1 - 1 - rng.integers(0, min(4, n1 - 1))) if r < 0.6 else int(rng.integers(1, n1))))
        for k in range(s, min(n1, s + int(rng.integers(1, 4)))):
            d = np.float32(0) if rng.random() < 0.12 else np.float32(rng.choice([-1, 1]) * rng.uniform(0, 1.6e-6))
            t[k:] = (t[k:] + np.float32((t[k - 1] + d) - t[k])).astype(np.float32)
        e = min(n1 - 1, s + 3)
        if rng.random() < 0.4:          # the gap behind the cluster barely long / barely short
            t[e:] = (t[e:] + np.float32((t[:e].max() + np.float32(rng.uniform(0.7e-6, 1.4e-6))) - t[e])).astype(np.float32)
    return t


@pytest.mark.parametrize("model", ["clusters", "pairwise"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_certified_chains_reduce_to_plain_pairing(oracle, seed, model):
    rng = np.random.default_rng(seed)
    M = 64
    used = {"A": 0, "B": 0, "C": 0, "D": 0}
    certified = 0
    trials = 40000
    for _ in range(trials):
        n1 = int(rng.integers(2, 22))
        t = _chain_t(rng, n1)
        if not np.all(t > 0):
            continue
        faces, ft = _chain_world(n1 - 1)
        fid = (rng.permutation(n1) if rng.random() < 0.5 else np.arange(n1)).astype(np.uint32)   # face ids unrelated to chain order
        faces_p, ft_p = np.zeros_like(faces), np.zeros_like(ft)
        faces_p[fid], ft_p[fid] = faces, ft
        ok, drop2, rules = (cert_model.certify if model == "clusters" else cert_model.certify_pairwise)(t, fid)
        if not ok:
            continue
        certified += 1
        for k in used:
            used[k] += rules[k]
        order = np.lexsort((fid, t))                       # the reference's input: hits sorted on (t, face id)
        ids, ts, uv = np.zeros((1, M), np.uint32), np.zeros((1, M), np.float32), np.zeros((1, M, 2), np.float32)
        ids[0, :n1], ts[0, :n1] = fid[order], t[order]
        res = oracle.postprocess(faces_p, ft_p, np.array([n1], np.uint32), ids, ts, uv)
        nv = int(res["num_visited_cells"][0])
        lit = [(int(res["visited_cells"][0, j]), float(res["hit_distances"][0, j, 0]), float(res["hit_distances"][0, j, 1])) for j in range(nv)]
        assert cert_model.plain_pairing(t, drop2) == lit, (t.tolist(), fid.tolist(), drop2, lit)
    if model == "pairwise":
        used.pop("D")
    assert certified > 0.4 * trials and min(used.values()) > 100, (certified, used)      # every rule of the model is exercised
