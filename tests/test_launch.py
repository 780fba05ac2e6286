"""bench.py --gpus N is the rank count (tools/bench_legs/launch.py): the checks that need no GPU, and the mechanics of the self-launch on a
stand-in script (environment of the ranks, rank 0 owns stdout, status relay, a dead rank takes the others along)."""
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.bench_legs import launch  # noqa: E402


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BANG_BENCH_SHARE_GPU",
                                                          "BANG_BENCH_SELF_LAUNCHED")}
    env.update(kw)
    return env


def _bench(args, **kw):
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=_env(**kw), capture_output=True, text=True, timeout=300)


def test_gpus_2_without_devices_is_an_error_not_a_smaller_run():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs here")
    r = _bench(["--gpus", "2", "--workload", "tiny", "--no-legs", "--no-cpu-baseline"])
    assert r.returncode == 2 and "HIP device(s) visible" in r.stderr and "--gpus 2" in r.stderr, (r.returncode, r.stderr[-400:])
    assert "{" not in r.stdout


def test_world_size_must_equal_gpus():
    r = _bench(["--gpus", "2", "--workload", "tiny"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
    r = _bench(["--gpus", "1", "--workload", "tiny"], RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    r = _bench(["--gpus", "0"])
    assert r.returncode == 2


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def _launch(tmp_path, n, body, argv=(), grace="1.0"):
    code = ("import sys; sys.path.insert(0, %r); from tools.bench_legs import launch; "
            "sys.exit(launch.self_launch(%d, %r, %r, grace_s=%s))" % (ROOT, n, list(argv), _script(tmp_path, body), grace))
    return subprocess.run([sys.executable, "-c", code], env=_env(BANG_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=120)


def test_self_launch_starts_n_ranks_with_their_environment(tmp_path):
    r = _launch(tmp_path, 4, """
        import os, sys
        e = os.environ
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["BANG_BENCH_SELF_LAUNCHED"] == "1" and e["LOCAL_RANK"] == e["RANK"]
        assert sys.argv[1:] == ["--gpus", "4", "--x"]
        print("rank", e["RANK"], "of", e["WORLD_SIZE"], "port", e["MASTER_PORT"], flush=True)
        """, argv=["--gpus", "4", "--x"])
    assert r.returncode == 0, r.stderr
    out = r.stdout.strip().splitlines()
    assert len(out) == 1 and out[0].startswith("rank 0 of 4 port ")          # rank 0 owns stdout ...
    err = [l for l in r.stderr.splitlines() if l.startswith("rank ")]
    assert sorted(l.split()[1] for l in err) == ["1", "2", "3"]               # ... the others' output goes to stderr
    assert len({l.split()[-1] for l in err + out}) == 1                       # one rendezvous port for the job


def test_self_launch_relays_a_failing_rank_and_ends_the_others(tmp_path):
    t0 = time.time()
    r = _launch(tmp_path, 3, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)          # (a rank waiting in a collective for the one that died)
        """)
    assert r.returncode == 7 and time.time() - t0 < 60, (r.returncode, r.stderr)
    assert "rank 1 exited with status 7" in r.stderr


def test_self_launch_refuses_without_devices(tmp_path):
    import torch
    n = torch.cuda.device_count() + 1
    code = "import sys; sys.path.insert(0, %r); from tools.bench_legs import launch; sys.exit(launch.self_launch(%d, [], 'x.py'))" % (ROOT, n)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr


def test_under_launcher_and_checks(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert not launch.under_launcher()
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert launch.under_launcher()
    launch.check_world(2, 2)
    import pytest
    with pytest.raises(SystemExit) as ei:
        launch.check_world(8, 1)
    assert ei.value.code == 2
