"""bench.py: roofline.traffic measured live -- two child runs of the primary configuration under rocprofv3 --pmc."""
import os
import sys
import time

from .common import ROOT

def live_traffic(args, log):
    """roofline.traffic measured in THIS run: two child runs of this very command (primary workload only, 3 timed steps) under
    `rocprofv3 --pmc` -- separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes:
        pass A: FETCH_SIZE + TCC_EA0_RDREQ_DRAM_32B_sum      pass B: WRITE_SIZE + TCC_EA0_RDREQ_IO_32B_sum + TCC_EA0_WRREQ_sum
    Calibration on known byte counts in this path's access shapes (tools/traffic_calib.hip, profiles/r04_traffic_calibration.md):
    EVERY read request of gfx950's L2 to memory is a 128-byte line -- a 4-byte filter probe as much as a code row or a streamed read --
    and FETCH_SIZE tallies each at 64 bytes (exactly half, for every shape: the guide's x2 holds throughout), while
    TCC_EA0_RDREQ_DRAM_32B x 32 and WRITE_SIZE x 1024 (32 bytes per scattered 4-byte store) are byte-exact.  `bytes` = HBM reads
    (DRAM_32B x 32) + HBM writes (WRITE_SIZE x 1024); reads over PCIe (the pulled adjacency rows, IO_32B x 32) are listed apart.
    The children run and exit BEFORE this process initialises the GPU (they need the HBM the parent would hold).
    Returns {bytes per launch, parts, note} or {None, why}."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"bytes": None, "note": "rocprofv3 is not on PATH"}
    steps = 3
    # the program behind `--` is the interpreter itself (no PATH look-up, no shim, no `env` hop: the profiler's preloaded library
    # initialises the GPU before the program starts, and an exec from such a process takes the machine down on this pool)
    child = [os.path.realpath(sys.executable), os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(steps), "--warmup", "1", "--no-legs",
             "--no-cpu-baseline", "--pull", str(args.pull)]
    for flag, val in (("--graph", args.graph), ("--L", args.L), ("--queries", args.queries), ("--shape-n", args.shape_n),
                      ("--lanes", args.lanes), ("--threads", args.threads)):
        if val:
            child += [flag, str(val)]
    if args.resident_graph:
        child.append("--resident-graph")
    env = dict(os.environ, BANG_BENCH_NO_TRAFFIC="1", TMPDIR="/tmp")
    got = {}
    passes = (("A", ("FETCH_SIZE", "TCC_EA0_RDREQ_DRAM_32B_sum")), ("B", ("WRITE_SIZE", "TCC_EA0_RDREQ_IO_32B_sum", "TCC_EA0_WRREQ_sum")))
    for tag, counters in passes:
        d = tempfile.mkdtemp(prefix="bang_pmc_", dir="/tmp")
        t0 = time.time()
        try:
            pr = subprocess.Popen(["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child,
                                  cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                so, _ = pr.communicate(timeout=180)                 # (a pass takes ~30 s -- up to two minutes more on a box that has never imported torch; the run stays within minutes whatever the profiler does)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)            # (the session this call started: nothing else is in it)
                pr.communicate()
                return {"bytes": None, "note": f"PMC pass {tag} did not finish in 180 s"}
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for r in csv.DictReader(open(f[0]))] if f else []
            sel = [r for r in rows if "search_kernel" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in sel})[-steps:]                # the timed launches
            if pr.returncode != 0 or len(ids) < steps:
                log(f"[bench] live PMC pass {tag} FAILED: rc {pr.returncode}, {len(ids)} launches of the search kernel seen; child stdout tail: {so[-300:]!r}")
                return {"bytes": None, "note": f"PMC pass {tag} failed (rc {pr.returncode}, {len(ids)} launches of the search kernel seen)"}
            for c in counters:
                got[c] = sum(float(r["Counter_Value"]) for r in sel if int(r["Dispatch_Id"]) in ids and r["Counter_Name"] == c) / steps
            log(f"[bench] live PMC pass {tag} ({time.time() - t0:.0f}s): " + ", ".join(f"{c} = {got[c]:.4g}" for c in counters) + " per launch of the search kernel")
        except Exception as e:                                   # (a profiler problem must not cost the bench line)
            return {"bytes": None, "note": f"PMC pass {tag} raised {type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd = int(got["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32)
    wr = int(got["WRITE_SIZE"] * 1024)
    io = int(got["TCC_EA0_RDREQ_IO_32B_sum"] * 32)
    return {"bytes": rd + wr, "hbm_read": rd, "hbm_write": wr, "pcie_read": io, "fetch_size_raw": int(got["FETCH_SIZE"] * 1024),
            "write_requests": int(got["TCC_EA0_WRREQ_sum"]),
            "note": f"live rocprofv3 --pmc, {steps} launches: rd {rd / 1e9:.2f} GB (RDREQ_DRAM_32B x 32) + wr {wr / 1e9:.2f} GB (WRITE_SIZE); PCIe rd {io / 1e9:.2f} GB. "
                    f"One pass per counter group on the same command; TCC_EA0_RDREQ_DRAM_32B is byte-exact on known byte counts "
                    f"(profiles/r04_traffic_calibration.md), FETCH_SIZE tallies every 128-byte request at 64 (raw {got['FETCH_SIZE'] * 1024 / 1e9:.3f} GB); "
                    f"a scattered 4-byte store counts 32 B"}
