"""bench.py --gpus N > 1 WITHOUT a launcher: the process starts its N rank processes itself.

The reference is one process on one GPU (BANG_Base/test_driver.cpp:564-599); the N > 1 job is ours (one process per
GPU, queries sharded, one RCCL gather).  The driver launches it through `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`; a plain `python bench.py --gpus N` used to run ONE rank and print "n_gpus": 1.  Now:

* no RANK / WORLD_SIZE in the environment and --gpus N > 1  ->  `self_launch()`: N fresh children of this interpreter, one per GPU, with
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, started BEFORE this process (or anything it imported) has
  touched the GPU; the parent never initialises HIP, never re-execs itself, relays rank 0's JSON line and exits with the worst status of
  its children.  A rank that dies takes the others with it after a grace period (by PID: nothing is killed by pattern).
* fewer visible devices than ranks  ->  exit status 2 with a message, never a silent smaller run (BANG_BENCH_SHARE_GPU=1 is the dry run
  of the N > 1 logic on a 1-GPU box: every rank on device 0).
* under a launcher, WORLD_SIZE != --gpus  ->  exit status 2 (`check_world()`).
"""
import os
import signal
import socket
import subprocess
import sys
import time


def visible_gpus() -> int:
    """Devices this process would see, WITHOUT initialising HIP: torch.cuda.device_count() does not create a context on this image
    (it honours HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)."""
    import torch
    return int(torch.cuda.device_count())


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def under_launcher() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def check_world(gpus: int, world: int) -> None:
    """A rank count that contradicts --gpus is an error, whoever started us."""
    if world != gpus:
        sys.stderr.write(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: refusing to report a run of {world} rank(s) as {gpus} GPU(s)\n")
        raise SystemExit(2)


def check_devices(gpus: int) -> None:
    if os.environ.get("BANG_BENCH_SHARE_GPU"):
        return
    n = visible_gpus()
    if n < gpus:
        sys.stderr.write(f"bench.py: --gpus {gpus} but only {n} HIP device(s) visible: refusing to run fewer ranks than asked for "
                         f"(BANG_BENCH_SHARE_GPU=1 runs the N > 1 logic with every rank on device 0)\n")
        raise SystemExit(2)


def self_launch(gpus: int, argv, script: str, grace_s: float = 30.0) -> int:
    """Starts `gpus` rank processes of `script argv`, waits for them, returns the exit status for the parent."""
    check_devices(gpus)
    port = free_port()
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BANG_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # (dmabuf IPC: peer rows and RCCL across processes need it on this driver)
        # rank 0 owns stdout (the ONE JSON line); whatever another rank prints goes to stderr
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=(None if r == 0 else sys.stderr)))
    rcs = [None] * gpus
    t_fail = None
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and t_fail is None:
                    t_fail = time.time()
                    sys.stderr.write(f"bench.py: rank {i} exited with status {rcs[i]}; the other ranks get {grace_s:.0f} s to follow\n")
        if t_fail is not None and time.time() - t_fail > grace_s:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.send_signal(signal.SIGTERM)
            time.sleep(3.0)
            for i, p in enumerate(procs):
                if rcs[i] is None and p.poll() is None:
                    p.kill()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.wait()
            break
        time.sleep(0.05)
    bad = [rc for rc in rcs if rc != 0]
    if not bad:
        return 0
    pos = [rc for rc in bad if rc > 0]
    return max(pos) if pos else 1
