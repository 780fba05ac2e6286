#!/usr/bin/env python3
"""What one rank's shard of the 10 K-query batch costs on a GPU of its own -- ONE load of the workload, then every batch size and
engine-option variant measured on the same engine (bench.py's own measure()).

    python tools/shard_sweep.py [--workload sift1b_shape] [--queries 10000,5000,2500,1250] [--variants BANG_SUMM_ITERS=0,BANG_SUMM_ITERS=1 | rows_hbm=0 | ...] [--steps 6]
                                [--shape-n N] [--out gpurun_out/shard_sweep.md] [--check]

--check: the ids of every variant must equal those of the first variant at the same batch size (launch policies never change results).
Prints a markdown table (and writes it to --out)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sift1b_shape")
    ap.add_argument("--queries", default="10000,5000,2500,1250")
    ap.add_argument("--variants", default="default", help="comma list of option settings; '+' joins several options of one variant (BANG_SPEC_ROWS=1+rows_hbm=0); 'default' = none")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape-n", type=int, default=0)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--graph", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    import torch
    import bench
    import bang_amd
    from oracle import oracle as O  # noqa: F401  (bench.run_config's checker)
    bang_amd.build()
    ctx = bench.Ctx()
    ctx.rank, ctx.local_rank, ctx.world, ctx.k, ctx.weak, ctx.live_traffic = 0, 0, 1, 10, False, None
    torch.cuda.set_device(0)
    ctx.dev = ctx.cdev = torch.device("cuda", 0)
    qs = [int(x) for x in a.queries.split(",")]
    wl = bench.build_workload(a.workload, ctx, Q=max(qs), shape_n=a.shape_n, stream=True)
    structured = wl["gt_i"] is not None
    graph = a.graph or ("auto" if structured else wl["graph"])
    eng = bench.make_engine(wl, graph, ctx, timing=1)
    L = a.L or 152
    args = argparse.Namespace(no_events=False)
    rows, ref_ids = [], {}
    for q in qs:
        my_q = np.ascontiguousarray(wl["queries"][:q])
        ctx.Q_total = q
        for var in a.variants.split(","):
            env_set = []
            for kv in ([] if var == "default" else var.split("+")):
                key, val = kv.split("=")
                if key.startswith("BANG_"):              # an environment switch (read when used; dropped again behind the variant)
                    os.environ[key] = val
                    env_set.append(key)
                elif key == "rows_frac":                 # this fraction of the adjacency rows in HBM (bang_rows_slice_e: rows [0, frac N)) -- what peer rows
                    eng.rows_slice(0, int(float(val) * wl["ix"].N))          # give a rank of a W-GPU node, with local HBM standing in for the peers'
                else:
                    eng.set_option(key, int(val))
            eng.set_searchparams(ctx.k, L)
            eng.alloc(q)
            res = bench.measure(eng, wl, my_q, L, a.steps, a.warmup, ctx, graph)
            st = res["agg"]
            same = None
            if a.check:
                if q in ref_ids:
                    same = bool(np.array_equal(ref_ids[q][0], res["ids"]) and np.array_equal(ref_ids[q][1].view(np.uint32), res["dists"].view(np.uint32)))
                else:
                    ref_ids[q] = (res["ids"].copy(), res["dists"].copy())
            eng.free()
            for key in env_set:
                os.environ.pop(key, None)
            r = res["roofline"] or {}
            rows.append(dict(queries=q, variant=var, qps=res["queries_per_s"], ms=res["ms_per_step"], launch_us=r.get("avg_launch_us"),
                             qps_incl_init=res["qps_incl_init"],
                             skip_ctr=int(st.get("filter_loads_skipped", 0)) // max(1, a.steps), iters=int(st["iterations"]), step_ms=res["step_ms"], same_as_first=same))
            print(json.dumps(rows[-1]), flush=True)
    eng.unload(); eng.close()
    wl["release"]()
    base = {r["variant"]: r["ms"] for r in rows if r["queries"] == qs[0]}
    lines = [f"# {wl['name']}", "",
             f"`tools/shard_sweep.py --workload {a.workload} --queries {a.queries} --variants {a.variants}`: one load, L = {L}, {a.steps} timed steps per cell "
             f"(bang_query wall time; bang_init outside).  speed-up = time of the {qs[0]}-query batch of the same variant / this time.", "",
             "| queries | variant | QPS | ms per batch | search launch us | speed-up of the batch | ids equal first variant |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append(f"| {r['queries']} | {r['variant']} | {r['qps']:.0f} | {r['ms']:.3f} | {r['launch_us']} | {base[r['variant']] / r['ms']:.2f} x | {r['same_as_first']} |")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(txt)


if __name__ == "__main__":
    main()
