"""bench.py legs at N = 1: the other BASELINE configurations, each timed like the primary on its own index.  Every leg writes a nested
object `config.at_<leg>` and -- the driver's record keeps scalars only -- its headline numbers as `config.<leg>_qps`, `_recall`, ..."""
import numpy as np

from .common import flat, leg_summary, make_engine, measure, release_config, run_config
from .cpu import cpu_baseline_structured
from .k2 import k2_alone


class Run:
    """What the legs share: ctx, args, the oracle module O, the config dict of the output line, the K2-alone results, k, steps."""
    def __init__(self, ctx, args, O, cfg, k2, k, steps=5, warm=1):
        self.ctx, self.args, self.O, self.cfg, self.k2, self.k, self.steps, self.warm = ctx, args, O, cfg, k2, k, steps, warm


def leg_L200(b, prim, graph):
    """BASELINE.json configs[1] also names L = 200: the same batch at that worklist length (structured primary only)."""
    e = prim["eng"]
    e.free(); e.set_searchparams(b.k, 200); e.alloc(prim["Qr"])
    r2 = measure(e, prim["wl"], prim["my_q"], 200, 3, 1, b.ctx, graph)
    ids_o, _ = prim["orc"].search(prim["my_q"][:64], b.k, 200)
    b.cfg["at_L200"] = leg_summary(r2, prim["wl"], graph, recall=b.O.recall(prim["wl"]["gt_i"], prim["wl"]["gt_d"], r2["ids"], b.k),
                                   extra={"parity_vs_oracle_first_64": bool(np.array_equal(r2["ids"][:64], ids_o))})
    flat(b.cfg, "L200", b.cfg["at_L200"])


def leg_sift1m(b):
    """configs[1]: SIFT1M-like, recall-verified; engine default placement (HBM), then the host placement in both loop forms."""
    cfg, k, O, ctx, args = b.cfg, b.k, b.O, b.ctx, b.args
    r1 = run_config("sift1m", ctx, args, O, steps=b.steps, warmup=b.warm, keep=True)
    wl1 = r1["wl"]
    cfg["at_sift1m"] = leg_summary(r1["res"], wl1, r1["graph"], recall=r1["recall"], extra={"parity_vs_oracle_first_64": r1["ok"]})
    flat(cfg, "sift1m", cfg["at_sift1m"])
    ids1, L1, q1_ = r1["res"]["ids"], r1["L"], r1["my_q"]
    e = r1["eng"]
    e.free(); e.set_searchparams(k, 200); e.alloc(r1["Qr"])
    r2 = measure(e, wl1, q1_, 200, 3, 1, ctx, r1["graph"])
    ids_o, _ = r1["orc"].search(q1_[:64], k, 200)
    cfg["at_sift1m_L200"] = leg_summary(r2, wl1, r1["graph"], recall=O.recall(wl1["gt_i"], wl1["gt_d"], r2["ids"], k),
                                        extra={"parity_vs_oracle_first_64": bool(np.array_equal(r2["ids"][:64], ids_o))})
    flat(cfg, "sift1m_L200", cfg["at_sift1m_L200"])
    if not args.no_cpu_baseline:
        cb = cpu_baseline_structured(r1, O, k, wl1["queries"])
        cfg["at_sift1m"]["cpu_baseline"] = cb
        cfg["sift1m_cpu_qps"] = cb["value"]
    b.k2[f"m{wl1['ix'].m}"] = k2_alone(wl1["ix"].D, wl1["ix"].m, wl1["ix"].dtype, ctx)
    e.free(); e.unload(); e.close()
    r1["eng"] = None
    for key, opts in (("sift1m_host_pull", {}), ("sift1m_host_walker", {"walker": 1})):
        e2 = make_engine(wl1, "host", ctx, timing=0 if args.no_events else 1)
        for kk, vv in opts.items():
            e2.set_option(kk, vv)
        e2.set_searchparams(k, L1)
        e2.alloc(q1_.shape[0])
        r3 = measure(e2, wl1, q1_, L1, b.steps, b.warm, ctx, "host", traffic_key=("sift1m_host" if not opts else None))
        cfg["at_" + key] = leg_summary(r3, wl1, "host", recall=O.recall(wl1["gt_i"], wl1["gt_d"], r3["ids"], k),
                                       extra={"ids_equal_device_run": bool(np.array_equal(r3["ids"], ids1))})
        flat(cfg, key, cfg["at_" + key])
        cfg[key + "_ids_equal_device_run"] = cfg["at_" + key]["ids_equal_device_run"]
        e2.free(); e2.unload(); e2.close()
    release_config(r1)


def leg_deep100m(b):
    """configs[2]: DEEP100M-shape (f32, m = 74), graph in HBM."""
    r = run_config("deep100m_shape", b.ctx, b.args, b.O, steps=b.steps, warmup=b.warm, keep=True)
    b.cfg["at_deep100m_shape"] = leg_summary(r["res"], r["wl"], r["graph"], props=r["ok"])
    flat(b.cfg, "deep100m_shape", b.cfg["at_deep100m_shape"])
    ixd = r["wl"]["ix"]
    mm, DD, dt = ixd.m, ixd.D, ixd.dtype
    cs_d = int(r["res"]["agg"]["code_stride"]) or mm
    release_config(r)
    b.k2[f"m{mm}"] = k2_alone(DD, mm, dt, b.ctx, stride=(cs_d if cs_d != mm else 0))


def leg_sift1b(b):
    """configs[3] as a leg (when another workload is the primary)."""
    r = run_config("sift1b_shape", b.ctx, b.args, b.O, steps=b.steps, warmup=b.warm, keep=True)
    b.cfg["at_sift1b_shape"] = leg_summary(r["res"], r["wl"], r["graph"], props=r["ok"])
    flat(b.cfg, "sift1b_shape", b.cfg["at_sift1b_shape"])
    release_config(r)


def leg_structured(b, name):
    """SIFT1B's PQ layout and placement (graph in host RAM, rows pulled over PCIe) on a structured, recall-verified index (sift300m: three
    tenths of the headline's size -- "QPS @ recall >= 0.9" at 3e8 points; sift100m on request), the first 64 queries against the oracle."""
    r = run_config(name, b.ctx, b.args, b.O, graph="host", steps=b.steps, warmup=b.warm, traffic=False, keep=True)
    b.cfg["at_" + name] = leg_summary(r["res"], r["wl"], "host", recall=r["recall"], extra={"parity_vs_oracle_first_64": r["ok"]})
    flat(b.cfg, name, b.cfg["at_" + name])
    b.cfg[name + "_hops_max"] = b.cfg["at_" + name]["hops_p50_p99_max"][2]
    b.cfg[name + "_N"] = int(r["wl"]["ix"].N)
    release_config(r)


def leg_sift10m(b):
    """A structured index beyond the Infinity Cache (N = 10 M: 320 MB of codes, 3.9 GB graph), recall-verified, both placements."""
    cfg, k, O, ctx, args = b.cfg, b.k, b.O, b.ctx, b.args
    r = run_config("sift10m", ctx, args, O, graph="host", steps=b.steps, warmup=b.warm, keep=True)
    w3 = r["wl"]
    cfg["at_sift10m"] = leg_summary(r["res"], w3, "host", recall=r["recall"], extra={"parity_vs_oracle_first_64": r["ok"]})
    flat(cfg, "sift10m", cfg["at_sift10m"])
    e = r["eng"]
    e.free(); e.unload(); e.close()
    r["eng"] = None
    e4 = make_engine(w3, "device", ctx, timing=0 if args.no_events else 1)
    e4.set_searchparams(k, r["L"])
    e4.alloc(r["Qr"])
    r5 = measure(e4, w3, r["my_q"], r["L"], b.steps, b.warm, ctx, "device")
    cfg["at_sift10m_device_graph"] = leg_summary(r5, w3, "device", recall=O.recall(w3["gt_i"], w3["gt_d"], r5["ids"], k),
                                                 extra={"ids_equal_host_run": bool(np.array_equal(r5["ids"], r["res"]["ids"]))})
    flat(cfg, "sift10m_device", cfg["at_sift10m_device_graph"])
    e4.free(); e4.unload(); e4.close()
    release_config(r)
