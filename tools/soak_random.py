#!/usr/bin/env python3
"""Soak: random index shapes / search parameters on the GPU against the oracle, all three loop forms (device graph, pull, walker), with
the launch policies of the search kernel (BANG_SUMM_ITERS, BANG_SEARCH_MAX_WGS, BANG_SPEC_ROWS, BANG_SEARCH_PRIO: instruction order and query hand-out,
never results) drawn at random per case.
    python tools/soak_random.py [n_cases] [seed] [budget_seconds]
Prints one line per failure and a summary; exit code 1 on any mismatch.  tests/test_gpu_policies.py runs a 60-second slice of it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

POLICY_KEYS = ("BANG_SUMM_ITERS", "BANG_SEARCH_MAX_WGS", "BANG_SPEC_ROWS", "BANG_SEARCH_PRIO")


def run(n_cases=40, seed=7, budget_s=None, log=print):
    """-> (cases run, engine batches compared, mismatches)"""
    import bang_amd
    from bang_amd import synth
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    bang_amd.build()
    O.build()
    saved = {k: os.environ.get(k) for k in POLICY_KEYS}
    bad = done = batches = 0
    t0 = time.time()
    try:
        for c in range(n_cases):
            if budget_s is not None and time.time() - t0 > budget_s:
                break
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
            policy = {"BANG_SUMM_ITERS": str(int(rng.choice([-1, 0, 1, 3, 40]))),
                      "BANG_SEARCH_MAX_WGS": str(int(rng.choice([0, 0, 1, 3]))), "BANG_SPEC_ROWS": str(int(rng.integers(0, 3))),
                      "BANG_SEARCH_PRIO": str(rng.choice(["0", "1", "auto"]))}
            try:
                ix, q, _, _ = synth.make_index(N, D, dtype, R, m, Q, K=min(10, k), n_clusters=8, seed=int(rng.integers(1, 1 << 30)), pq_iters=2)
            except Exception as e:
                log(f"case {c}: skipped ({type(e).__name__}: {e})")
                continue
            qq = np.ascontiguousarray(q[:, :D - 1]) if mips else q
            ids_o, dists_o, st_o = O.Oracle(ix).search(qq, k, L, mips=mips, with_stats=True)
            os.environ.update(policy)
            for graph in (1, 0, "walker"):
                opts = dict(graph=0, pull=0) if graph == "walker" else dict(graph=graph)
                with bang_amd.Engine(dtype, **opts) as e:
                    e.load_index(ix)
                    if mips:
                        e.set_searchparams(k, L, bang_amd.DIST_MIPS)
                    else:
                        e.set_searchparams(k, L)
                    e.alloc(Q)
                    for rep in range(2):
                        e.init(Q)
                        ids, dists = e.query(qq)
                        st = e.stats()
                        batches += 1
                        ok = (np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
                              and st["dist_evals"] == int(st_o[:, 2].sum()) and st["candidates"] == int(st_o[:, 1].sum()))
                        if not ok:
                            bad += 1
                            log(f"MISMATCH case {c} graph={graph} rep={rep}: N={N} D={D} {dtype} R={R} m={m} Q={Q} k={k} L={L} mips={mips} {policy} "
                                f"ids_equal={np.array_equal(ids, ids_o)} evals {st['dist_evals']} vs {int(st_o[:, 2].sum())}")
                    e.free()
                    e.unload()
            done += 1
            if c % 10 == 9:
                log(f"... {c + 1} cases, {bad} mismatches")
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    return done, batches, bad


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else None
    done, batches, bad = run(n_cases, seed, budget, log=lambda *a: print(*a, flush=True))
    print(f"soak: {done} cases, {batches} batches in three loop forms, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
