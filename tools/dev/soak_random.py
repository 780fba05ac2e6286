#!/usr/bin/env python3
"""One-off soak: random index shapes / search parameters on the GPU against the oracle, all three loop forms (device graph, pull, walker).
    python tools/dev/soak_random.py [n_cases] [seed]
Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import bang_amd  # noqa: E402
from bang_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    bang_amd.build()
    O.build()
    bad = 0
    for c in range(n_cases):
        dtype = str(rng.choice(["uint8", "int8", "float"]))
        D = int(rng.choice([16, 24, 32, 37, 48, 64, 96, 100, 128, 140, 200, 256]))
        divs = [m for m in (4, 5, 6, 8, 12, 16, 25, 32, 37, 48, 50, 64, 70, 74, 96, 100, 128) if m <= D]
        m = int(rng.choice(divs))
        R = int(rng.choice([8, 16, 24, 32, 48, 64]))
        N = int(rng.integers(300, 4000))
        Q = int(rng.choice([1, 2, 7, 16, 33, 64, 100]))
        L = int(rng.choice([3, 5, 10, 20, 33, 64, 70, 100, 152, 200, 300, 512]))
        k = int(rng.integers(1, min(L, 20) + 1))
        mips = bool(rng.integers(0, 5) == 0) and dtype == "float"
        try:
            ix, q, _, _ = synth.make_index(N, D, dtype, R, m, Q, K=min(10, k), n_clusters=8, seed=int(rng.integers(1, 1 << 30)), pq_iters=2)
        except Exception as e:
            print(f"case {c}: skipped ({type(e).__name__}: {e})")
            continue
        fn = True if mips else None
        qq = np.ascontiguousarray(q[:, :D - 1]) if fn is not None else q
        ids_o, dists_o, st_o = O.Oracle(ix).search(qq, k, L, mips=bool(fn), with_stats=True)
        for graph in (1, 0, "walker"):
            opts = dict(graph=0, pull=0) if graph == "walker" else dict(graph=graph)
            with bang_amd.Engine(dtype, **opts) as e:
                e.load_index(ix)
                if fn is not None:
                    e.set_searchparams(k, L, bang_amd.DIST_MIPS)
                else:
                    e.set_searchparams(k, L)
                e.alloc(Q)
                for rep in range(2):
                    e.init(Q)
                    ids, dists = e.query(qq)
                    st = e.stats()
                    ok = (np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
                          and st["dist_evals"] == int(st_o[:, 2].sum()) and st["candidates"] == int(st_o[:, 1].sum()))
                    if not ok:
                        bad += 1
                        print(f"MISMATCH case {c} graph={graph} rep={rep}: N={N} D={D} {dtype} R={R} m={m} Q={Q} k={k} L={L} mips={fn is not None} "
                              f"ids_equal={np.array_equal(ids, ids_o)} evals {st['dist_evals']} vs {int(st_o[:, 2].sum())}", flush=True)
                e.free()
                e.unload()
        if c % 10 == 9:
            print(f"... {c + 1} cases, {bad} mismatches", flush=True)
    print(f"soak: {n_cases} cases x 3 loop forms x 2 batches, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
