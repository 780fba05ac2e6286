"""Regenerates the committed golden fixture (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be run, so these are OUR fixtures: a tiny synthetic index in
the reference's file formats plus the outputs of the CPU oracle on it.  They pin (a) the oracle against
accidental change and (b) the HIP path (which must reproduce them bit for bit, loading the files through
bang_load).  The 14-node / D=2 / L=4 / 2-chunk toy shape mirrors the SIFT1BTOY define of the reference
(BANG_Inmemory/parANN.h:70-79); `tiny` is a slightly larger uint8 case."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

from bang_amd import formats, synth  # noqa: E402
from oracle import oracle as O       # noqa: E402


def main():
    out = {}
    # --- toy: N=14, D=2, float, R=4, 2 chunks of 1 dim, medoid 6 (shape of SIFT1BTOY) -------------------
    rng = np.random.default_rng(14)
    vec = rng.normal(size=(14, 2)).astype(np.float32)
    adj = np.zeros((14, 4), np.uint32)
    deg = np.zeros(14, np.uint32)
    for i in range(14):
        d = ((vec - vec[i]) ** 2).sum(1)
        d[i] = np.inf
        nb = np.sort(np.argsort(d)[: 3 + (i % 2)]).astype(np.uint32)      # ragged degrees 3/4, sorted ascending
        deg[i] = len(nb)
        adj[i, : len(nb)] = nb
    piv = np.zeros((256, 2), np.float32)
    piv[:, 0] = np.linspace(-3, 3, 256)
    piv[:, 1] = np.linspace(3, -3, 256) ** 3 / 9
    cen = vec.mean(0).astype(np.float32)
    codes = np.stack([np.abs(piv[None, :, j] - (vec[:, j] - cen[j])[:, None]).argmin(1) for j in range(2)], 1).astype(np.uint8)
    toy = formats.Index("float", 14, 2, 4, 2, 6, formats.pack_graph(vec, deg, adj), codes, piv, cen,
                        np.array([0, 1, 2], np.uint32))
    formats.write_index(os.path.join(HERE, "toy"), toy)
    tq = rng.normal(size=(5, 2)).astype(np.float32)
    formats.write_bin(os.path.join(HERE, "toy_query.bin"), tq)
    ids, dists, st = O.Oracle(toy).search(tq, 3, 4, with_stats=True)
    out.update(toy_ids=ids, toy_dists=dists, toy_stats=st)

    # --- tiny: N=600, D=32, uint8, R=16, m=12 (chunks of 3 and 2 dims -> psz 4), k=5 ----------------------
    ix, q, gt_i, gt_d = synth.make_index(600, 32, "uint8", 16, 12, 24, K=5, n_clusters=8, seed=99, pq_iters=3)
    formats.write_index(os.path.join(HERE, "tiny"), ix)
    formats.write_bin(os.path.join(HERE, "tiny_query.bin"), q)
    formats.write_truthset(os.path.join(HERE, "tiny_gt.bin"), gt_i, gt_d)
    for L in (5, 17, 40):
        ids, dists, st = O.Oracle(ix).search(q, 5, L, with_stats=True)
        out[f"tiny_ids_L{L}"] = ids
        out[f"tiny_dists_L{L}"] = dists
        out[f"tiny_stats_L{L}"] = st
    np.savez(os.path.join(HERE, "expected.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
