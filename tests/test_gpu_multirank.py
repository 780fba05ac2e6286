"""The N > 1 path of bench.py on a 1-GPU box: two ranks (one process each, launched exactly as the driver launches them) share
GPU 0 (BANG_BENCH_SHARE_GPU) and ONE host graph -- rank 0 writes the index files into tmpfs, both ranks load them through
bang_load, which maps `_disk.bin` shared.  Each rank searches its shard with the real engine; the result ids stay in device
buffers (bang_query_dev_e) and are gathered with the job's single collective (gloo here, staged through the host: RCCL refuses two
ranks on one device).  Every rank checks its shard against the oracle and rank 0 checks the gathered block of the whole batch."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(nproc, extra, timeout=380, workload="tiny", L=46, launcher=True):
    """launcher=True: the ranks are started as the driver starts them (torch.distributed.run); False: plain `python bench.py --gpus N`,
    which starts its N ranks itself (tools/bench_legs/launch.py)."""
    env = dict(os.environ, BANG_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for kk in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BANG_BENCH_SELF_LAUNCHED"):
        env.pop(kk, None)
    base = ["bench.py", "--gpus", str(nproc), "--workload", workload, "--L", str(L), "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--no-legs", "--backend", "gloo"] + extra
    if nproc == 1 or not launcher:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + base
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("graph", ["host", "host-nopeer", "host-walker", "device"])
def test_two_ranks_one_gpu_one_host_graph(libbang, graph, monkeypatch):
    """host: the default of the host placement -- ONE rows file in the shared directory (rank 0 builds it, rank 1 maps it:
    BANG_PULL_ROWS_DIR) and, N > 1, the rows of this small index entirely in the two ranks' HBM slices (peer rows: nothing is pulled over
    PCIe any more); host-nopeer: both ranks pull every row from the shared file (rounds 2-4); host-walker: both ranks' walker threads
    read ONE mapped graph file."""
    walker = graph == "host-walker"
    nopeer = graph == "host-nopeer"
    if nopeer:
        monkeypatch.setenv("BANG_BENCH_NO_PEER_ROWS", "1")
    graph = "host" if (walker or nopeer) else graph
    extra = ["--graph", graph] + (["--pull", "0"] if walker else [])
    one = _bench(1, extra)
    two = _bench(2, extra)
    if graph == "host":
        for r in (one, two):
            assert ("pulled" in r["config"]["host_loop"]) == (not walker), r["config"]["host_loop"]
        assert (one["config"]["pcie_pulled_bytes_per_step"] > 0) == (not walker)
        if walker:
            assert two["config"]["pcie_pulled_bytes_per_step"] == 0 and not two["config"]["peer_rows"]
        elif nopeer:
            assert two["config"]["pcie_pulled_bytes_per_step"] > 0 and two["config"]["peer_rows"] is None
        else:
            assert two["config"]["peer_rows"]["fraction"] == 1.0 and two["config"]["rows_from_peer_hbm_per_step"] > 0, two["config"]["peer_rows"]
            assert two["config"]["pcie_pulled_bytes_per_step"] == 0
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["parity_vs_oracle_first_64"] is True and two["config"]["parity_vs_oracle_first_64"] is True
    assert two["config"]["graph"] == graph and two["config"]["L"] == 46
    # what the collective delivered on rank 0 -- the [Q][k] block of the WHOLE batch, gathered from the device buffers the engine
    # left the shards' ids in (bang_query_dev_e) -- equals the oracle's answer for every query
    assert two["config"]["gathered_ids_equal_oracle_whole_batch"] is True
    # the sharded job answers the same 1000 queries: the batch recall (shard recalls weighted by shard size) is the single-process one
    assert abs(two["config"]["recall_at_10"] - one["config"]["recall_at_10"]) < 2e-3
    assert two["value"] > 0 and two["ms_per_step"] > 0
    assert two["config"]["search_ms_per_step_max_over_ranks"] > 0 and two["config"]["gather_ms_per_step_max_over_ranks"] > 0


def test_plain_bench_gpus_2_starts_its_two_ranks_itself(libbang):
    """`python bench.py --gpus 2` with NO launcher (no RANK / WORLD_SIZE in the environment) used to run one rank and print
    "n_gpus": 1.  Now the process starts two rank processes itself before touching the GPU: the line says 2 GPUs, the job's backend counted two
    ranks (an all-reduce of ones), every rank reports its own search / gather time, and the gathered block equals the oracle's answer."""
    two = _bench(2, ["--graph", "host"], launcher=False)
    assert two["n_gpus"] == 2 and two["world_seen"] == 2 and two["config"]["rccl_world_seen"] == 2
    assert two["ranks_started_by"] == "self" and two["scaling"] == "strong"
    assert two["config"]["gathered_ids_equal_oracle_whole_batch"] is True and two["config"]["parity_vs_oracle_first_64"] is True
    assert len(two["search_ms_per_rank"]) == 2 and len(two["gather_ms_per_rank"]) == 2 and min(two["search_ms_per_rank"]) > 0
    assert two["peer_rows_fallback"] is False and two["config"]["peer_rows"]["fraction"] == 1.0
    # the same job under the launcher says so
    two_l = _bench(2, ["--graph", "host"])
    assert two_l["ranks_started_by"] == "launcher" and two_l["world_seen"] == 2


def test_plain_bench_gpus_2_refuses_a_one_gpu_box():
    """Fewer visible devices than ranks, no BANG_BENCH_SHARE_GPU: exit status 2 and no line -- never a silent 1-rank run."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {kk: v for kk, v in os.environ.items() if kk not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BANG_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "tiny", "--no-legs", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "only 1 HIP device" in r.stderr, (r.returncode, r.stderr[-500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_two_ranks_throughput_mode(libbang):
    """--batches 2: every rank streams two WHOLE batches per step, no collective on the data path ("scaling": "weak")."""
    two = _bench(2, ["--graph", "device", "--batches", "2"])
    assert two["scaling"] == "weak" and two["config"]["batches_per_step"] == 2
    assert two["config"]["parity_vs_oracle_first_64"] is True


def test_two_ranks_streamed_sift1b_shape_share_one_rows_file(libbang):
    """configs[4] in small: the SIFT1B-shape index loaded by STREAMING on every rank (no graph image anywhere), ONE pull-rows file
    for the node -- rank 0 builds it while its entries stream through, rank 1 maps it and checks its signature at the end of its
    own stream -- and the sharded batch answers with the properties the single-rank run has."""
    args = ["--shape-n", "12000000", "--queries", "512"]
    one = _bench(1, args, workload="sift1b_shape", L=40)
    two = _bench(2, args, workload="sift1b_shape", L=40)
    for r in (one, two):
        c = r["config"]
        assert c["result_properties_ok"] is True and c["graph"] == "host" and "pulled" in c["host_loop"], c
        assert "STREAMED" in c["workload"] and "N=12000000" in c["workload"]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    # (rank 1 never read an index entry: it received the vectors from rank 0's device buffer and mapped the rows file -- bang_load_shared_e;
    # the per-rank stream of round 2 is still reachable)
    env_old = os.environ.get("BANG_BENCH_NO_VECTOR_BROADCAST")
    os.environ["BANG_BENCH_NO_VECTOR_BROADCAST"] = "1"
    try:
        two_b = _bench(2, args, workload="sift1b_shape", L=40)
    finally:
        if env_old is None:
            os.environ.pop("BANG_BENCH_NO_VECTOR_BROADCAST", None)
        else:
            os.environ["BANG_BENCH_NO_VECTOR_BROADCAST"] = env_old
    assert two_b["config"]["result_properties_ok"] is True


def test_two_ranks_sift1b_shape_gathered_batch_equals_oracle(libbang):
    """configs[4] with the oracle in the loop: the SIFT1B-shape layout (uint8, m = 70, rows pulled from ONE shared rows file) at an
    N whose PQ codes also fit host memory, sharded over two ranks; the gathered [Q][k] block must equal the oracle's ids for EVERY
    query of the batch, and so must each rank's shard (first 64)."""
    args = ["--shape-n", "3000000", "--queries", "512", "--host-codes"]
    one = _bench(1, args, workload="sift1b_shape", L=40)
    two = _bench(2, args, workload="sift1b_shape", L=40)
    assert one["config"]["result_properties_ok"] is True and two["config"]["result_properties_ok"] is True
    assert two["config"]["gathered_ids_equal_oracle_whole_batch"] is True, two["config"]
    assert two["n_gpus"] == 2 and "pulled" in two["config"]["host_loop"]


def test_rccl_all_gather_from_device_buffers_single_rank(libbang):
    """The data path of the N > 1 job with the REAL collective: bang_query_dev_e leaves the ids in a device buffer and RCCL's
    all_gather_into_tensor reads it in place (one rank: RCCL refuses two ranks on one device, and this pool has one GPU per box);
    the gathered block equals the oracle's answer for the whole batch."""
    env = dict(os.environ, BANG_BENCH_FORCE_GATHER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--workload", "tiny", "--L", "46", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-legs", "--backend", "nccl"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert c["parity_vs_oracle_first_64"] is True and c["gathered_ids_equal_oracle_whole_batch"] is True
    assert c["gather_ms_per_step_max_over_ranks"] > 0 and c["search_ms_per_step_max_over_ranks"] > 0
    # the same on the north-star layout (streamed SIFT1B-shape index at reduced N, rows pulled)
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--workload", "sift1b_shape", "--shape-n", "6000000", "--queries", "1024", "--L", "40",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-legs", "--backend", "nccl"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["result_properties_ok"] is True and d["config"]["gather_ms_per_step_max_over_ranks"] > 0


def test_two_ranks_peer_rows_over_ipc(libbang):
    """PEER ROWS (include/bang_c.h, bang_amd/shard.py share_rows): two ranks -- two processes on GPU 0 -- each keep one slice of the adjacency
    rows in HBM, export it with hipIpcGetMemHandle and map the sibling's allocation with hipIpcOpenMemHandle; the search kernel reads a
    parent's row from its own slice, from the SIBLING's allocation, or (beyond the two slices) from the shared rows file in host memory.  The
    gathered [Q][k] block equals the oracle's answer for the whole batch, rows did come from the peer, and some still came over PCIe."""
    args = ["--shape-n", "3000000", "--queries", "512", "--host-codes"]
    env_old = os.environ.get("BANG_BENCH_PEER_SLICE_ROWS")
    os.environ["BANG_BENCH_PEER_SLICE_ROWS"] = "1000000"        # 2 x 1e6 of 3e6 rows in "the node's HBM", the last third on the host
    try:
        two = _bench(2, args, workload="sift1b_shape", L=40)
    finally:
        if env_old is None:
            os.environ.pop("BANG_BENCH_PEER_SLICE_ROWS", None)
        else:
            os.environ["BANG_BENCH_PEER_SLICE_ROWS"] = env_old
    c = two["config"]
    assert c["peer_rows"] and c["peer_rows"].get("slice_rows") == 1000000 and abs(c["peer_rows"]["fraction"] - 2 / 3) < 1e-6, c["peer_rows"]
    assert c["result_properties_ok"] is True and c["gathered_ids_equal_oracle_whole_batch"] is True, c
    assert c["rows_from_peer_hbm_per_step"] > 0 and c["rows_from_own_hbm_per_step"] > 0 and c["pcie_pulled_bytes_per_step"] > 0, c
    # the whole graph in the two slices: nothing is pulled over PCIe any more
    two = _bench(2, args, workload="sift1b_shape", L=40)
    c = two["config"]
    assert c["peer_rows"]["fraction"] == 1.0 and c["gathered_ids_equal_oracle_whole_batch"] is True, c
    assert c["rows_from_peer_hbm_per_step"] > 0 and c["pcie_pulled_bytes_per_step"] == 0, c
