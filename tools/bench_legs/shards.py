"""bench.py legs on the PRIMARY engine (one load of the headline index): what one rank's shard of the 10 K batch costs on a GPU of its own
(`shards`: the evidence behind the projected multi-GPU speed-ups, gather included) and the north-star data flow on the same index
(`walker`: C++ walker threads serving the host-paced kernel from the same 256-byte rows the kernel otherwise pulls itself)."""
import os
import time

import numpy as np

from .common import check_properties, leg_summary, log, measure


def gather_ms_world1(ctx, Q, k, reps=20):
    """The job's ONE collective -- all_gather_into_tensor of the [Q / W][k] int64 id blocks from device buffers (bang_amd/shard.py) -- executed
    by RCCL with ONE rank, followed by what rank 0 of the N > 1 job does with it (one D2H copy of the [Q][k] block): what the tail of a
    sharded step costs (launch + completion + copy), the floor of its cost on W ranks (the xGMI transfer of <= 100 KB per rank is a few
    microseconds on top).  Returns (ms, note)."""
    import torch
    import torch.distributed as dist
    from bang_amd import shard
    own = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=ctx.dev)
            own = True
        dg = shard.DeviceGather(Q, k, 0, 1, ctx.dev, coll_device=ctx.dev)
        for _ in range(3):
            dg.gather()
            dg.batch_ids(copy=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dg.gather()
            dg.batch_ids(copy=False)                                   # (rank 0 of the N > 1 job copies the gathered [Q][k] block to the host: part of its step)
        ms = 1e3 * (time.perf_counter() - t0) / reps
        return round(ms, 4), ("RCCL all_gather_into_tensor from device buffers + the D2H copy of the gathered [Q][k] block, world size 1, "
                              "launch to completion (mean of %d)" % reps)
    except Exception as ex:                                  # noqa: BLE001  (no RCCL here: the projection then says so)
        return None, "RCCL gather not measurable here: " + repr(ex)[:160]
    finally:
        if own:
            try:
                dist.destroy_process_group()
            except Exception:                                # noqa: BLE001
                pass


def leg_shards(b, prim, L, full_ms, steps=8, warm=2):
    """Times Q / 2, Q / 4, Q / 8 queries of the primary batch on the primary engine -- bang_query wall time, H2D / K1 / search / re-rank / D2H
    included, bang_init outside -- and projects the speed-up of a W-GPU job as  t(Q) / (t(Q / W) + gather)."""
    cfg, ctx, k = b.cfg, b.ctx, b.k
    eng, wl = prim["eng"], prim["wl"]
    Q = prim["Qr"]
    out = {"full_batch_ms": full_ms, "queries": Q}
    g_ms, g_note = gather_ms_world1(ctx, Q, k)
    out["gather_ms_world1"], out["gather_note"] = g_ms, g_note
    cfg["gather_ms_world1"] = g_ms
    for W in (2, 4, 8):
        q = Q // W
        my_q = np.ascontiguousarray(wl["queries"][:q])
        eng.free()
        eng.set_searchparams(k, L)
        eng.alloc(q)
        ctx.Q_total = q
        r = measure(eng, wl, my_q, L, steps, warm, ctx, "host")
        ok = check_properties(wl["ix"], my_q, r["ids"], r["dists"], k) if wl["gt_i"] is None else None
        a = r["agg"]
        out[f"shard_{q}"] = {"ms": r["ms_per_step"], "launch_us": (r["roofline"] or {}).get("avg_launch_us"), "rerank_fused": int(a.get("rerank_fused", 0)),
                             "step_ms_min_max": [min(r["step_ms"]), max(r["step_ms"])], "result_properties_ok": ok, "iterations": a["iterations"]}
        cfg[f"shard_ms_{q}"] = r["ms_per_step"]
        proj = full_ms / (r["ms_per_step"] + (g_ms or 0.0))
        cfg[f"projected_speedup_{W}"] = round(proj, 3)
        out[f"projected_speedup_{W}"] = round(proj, 3)
        log(f"[bench] shard of {q} queries: {r['ms_per_step']:.3f} ms per batch -> projected speed-up on {W} GPUs {proj:.2f} x (gather {g_ms} ms included)")
    out["note"] = ("one rank's shard of the batch timed on THIS GPU (same index, same engine load); projected_speedup_W = full_batch_ms / (shard_ms + gather_ms_world1): "
                   "no multi-GPU hardware was involved -- contention of W GPUs for one host's DRAM is not in it")
    cfg["at_shards"] = out
    ctx.Q_total = Q


def leg_walker_rows(b, prim, L, steps=5, warm=1):
    """The north-star data flow at the headline's N, on the primary engine: option walker = 1 -- the C++ walker threads hand the parents'
    adjacency rows to the host-paced search kernel through the PCIe BAR (bang_search.cu:771-813), reading the 256-byte pull rows."""
    cfg, ctx, k = b.cfg, b.ctx, b.k
    eng, wl = prim["eng"], prim["wl"]
    Q = prim["Qr"]
    eng.free()
    eng.set_option("walker", 1)
    try:
        eng.set_searchparams(k, L)
        eng.alloc(Q)
        r = measure(eng, wl, prim["my_q"], L, steps, warm, ctx, "host")
        ok = check_properties(wl["ix"], prim["my_q"], r["ids"], r["dists"], k) if wl["gt_i"] is None else None
        same = bool(np.array_equal(r["ids"], prim["res"]["ids"]))
        leg = leg_summary(r, wl, "host", props=ok, extra={"ids_equal_pulled_run": same, "walker_reads": "256-byte pull rows (no graph image resident)",
                                                         "N": int(wl["ix"].N)})
        cfg["at_sift1b_shape_walker"] = leg
        cfg["walker_qps"], cfg["walker_ms"], cfg["walker_N"] = leg["queries_per_s"], leg["ms_per_batch"], int(wl["ix"].N)
        cfg["walker_ids_equal_pulled_run"] = same
        rf = r["roofline"] or {}
        if "pcie_h2d" in rf:
            cfg["walker_bar_GBps"] = rf["pcie_h2d"]["achieved_GBps"]
            leg["pcie_h2d"] = rf["pcie_h2d"]
        cfg["walker_leg_threads"] = r["agg"]["walker_threads"]
        cfg["walker_step_ms_min"], cfg["walker_step_ms_max"] = min(r["step_ms"]), max(r["step_ms"])
    finally:
        eng.free()
        eng.set_option("walker", 0)
