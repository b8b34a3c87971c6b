"""CPU-only: `python bench.py --gpus N` without a torchrun environment starts N fresh worker processes (one per GPU),
relays rank 0's JSON line and fails when a worker fails.  The workers run in --dry-run mode (gloo rendezvous, barrier,
MAX all-reduce of the wall time, the step replaced by a sleep): the launcher plumbing is what is under test."""
import json
import os
import subprocess
import sys

from conftest import REPO


def _run(*extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--dry-run", *extra],
                          capture_output=True, text=True, timeout=300, env=e)


def test_plain_multi_gpu_launch_spawns_workers():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "2 independent samples"
    assert line["ms_per_step"] >= 1.0                                   # the 1 ms sleeps, max over ranks
    assert "DRY RUN" in line["data"] and line["value"] == 0.0          # never mistaken for a measurement


def test_failed_worker_fails_the_launch():
    r = _run("--gpus", "2", "--dry-run-fail-rank", "1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_under_torchrun_environment_no_respawn():
    """WORLD_SIZE set (torch.distributed.run launched us): run as that rank, do not spawn."""
    r = _run("--gpus", "1", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    r = _run("--gpus", "2", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_switches_are_recorded():
    r = _run("--gpus", "1", env={"S3D_WINO": "2", "S3D_VCAT": "0"})
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["s3d_switches"] == {"S3D_WINO": "2", "S3D_VCAT": "0"}


def test_peer_that_dies_before_rendezvous_ends_the_launch_within_seconds():
    """Rank 1 exits before init_process_group: rank 0 would sit in the store rendezvous for minutes.  The launcher polls
    all children, stops the survivor and fails fast."""
    import time
    t0 = time.time()
    r = _run("--gpus", "2", "--dry-run-fail-rank", "1", "--dry-run-fail-early")
    dt = time.time() - t0
    assert r.returncode != 0
    assert dt < 60, f"launcher took {dt:.0f} s to notice a dead peer"
    assert "rank(s) [1]" in r.stderr and "stopped" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_line_reports_every_rank():
    r = _run("--gpus", "2")
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert [x["rank"] for x in line["ranks"]] == [0, 1] and len(line["per_rank_ms"]) == 2
    assert line["rccl_world_size"] == 2 and line["dist_backend"] == "gloo"     # (nccl = RCCL on the GPU box)


def test_launch_deadline(tmp_path):
    """A hung rank (sleeps forever) is bounded by S3D_LAUNCH_TIMEOUT."""
    import time
    from sin3dm_amd.launcher import spawn_ranks
    script = tmp_path / "hang.py"
    script.write_text("import time\ntime.sleep(600)\n")
    import io
    err = io.StringIO()
    t0 = time.time()
    rc = spawn_ranks(str(script), [], 2, timeout=2.0, err=err)
    assert rc == 1 and time.time() - t0 < 30 and "no result after" in err.getvalue()


def test_config_c3_line_over_two_ranks():
    """`bench.py --config c3 --gpus 2` (BASELINE configs[2]'s per-GPU share: DDIM-100, batch 8 per GPU) goes through the same
    self-spawning launch; the line names the workload, counts batch x ranks samples and keeps the contract's keys."""
    r = _run("--gpus", "2", "--config", "c3")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["name"] == "c3" and line["config"]["batch_per_gpu"] == 8 and line["config"]["steps_per_sample"] == 100
    assert "DDIM-100" in line["metric"] and "configs[2]" in line["config"]["workload"]
    assert line["config"]["parallelism"] == "2 x 8 independent samples" and line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["devices_verified_distinct"] is False        # the CPU dry run has no bus id / uuid to verify: said, not assumed
    r = _run("--gpus", "1", "--config", "c5")
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["name"] == "c5" and "(256,256,128)" in line["metric"]
    # configs[3]'s diffusion stage: the one workload whose N > 1 path carries a collective (the flat-gradient all-reduce)
    r = _run("--gpus", "2", "--config", "c4")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["name"] == "c4" and line["config"]["batch_per_gpu"] == 4 and line["config"]["steps_per_sample"] == 1
    assert "training" in line["metric"] and "configs[3]" in line["config"]["workload"] and line["config"]["parallelism"].startswith("dp2")


def test_launcher_stops_its_ranks_when_it_is_terminated(tmp_path):
    """SIGTERM to the launcher (a harness `timeout`, a cancelled gpurun call): the ranks — each leading its own session — are
    stopped with it instead of staying behind inside a collective with their GPUs held."""
    import signal
    import time
    child = tmp_path / "rank.py"
    child.write_text("import os, sys, time\nopen(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    parent = tmp_path / "parent.py"
    parent.write_text(
        "import sys\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "from sin3dm_amd.launcher import spawn_ranks\n"
        f"sys.exit(spawn_ranks({str(child)!r}, [{str(tmp_path / 'pid')!r}], 2, timeout=500))\n")
    p = subprocess.Popen([sys.executable, str(parent)], stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not all(os.path.exists(f"{tmp_path}/pid.{r}") and open(f"{tmp_path}/pid.{r}").read() for r in (0, 1)):
        assert time.time() - t0 < 60 and p.poll() is None
        time.sleep(0.1)
    pids = [int(open(f"{tmp_path}/pid.{r}").read()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    _, err = p.communicate(timeout=60)
    assert p.returncode == 1 and "SIGTERM" in err
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, f"rank process {pid} outlived its launcher"


def _launch_two_ranks(tmp_path, rank_body):
    import time
    child = tmp_path / "rank.py"
    child.write_text(rank_body)
    parent = tmp_path / "parent.py"
    parent.write_text(
        "import sys\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "from sin3dm_amd.launcher import spawn_ranks\n"
        f"sys.exit(spawn_ranks({str(child)!r}, [{str(tmp_path / 'pid')!r}], 2, timeout=500))\n")
    p = subprocess.Popen([sys.executable, str(parent)], stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not all(os.path.exists(f"{tmp_path}/pid.{r}") and open(f"{tmp_path}/pid.{r}").read() for r in (0, 1)):
        assert time.time() - t0 < 60 and p.poll() is None
        time.sleep(0.1)
    return p, [int(open(f"{tmp_path}/pid.{r}").read()) for r in (0, 1)]


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                                      # a zombie waiting for its (dead) parent's wait() is not a live rank
        return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def test_second_signal_cannot_cut_the_clean_up_short(tmp_path):
    """GNU `timeout` signals the launcher's pid and then its process group, a double Ctrl-C does the same: the second delivery
    used to raise inside the clean-up and abort the SIGKILL escalation.  Ranks that IGNORE SIGTERM need that
    escalation; a second and third SIGTERM arrive while the launcher waits out their grace period."""
    import signal
    import time
    p, pids = _launch_two_ranks(tmp_path, "import os, signal, sys, time\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                                          "open(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    p.send_signal(signal.SIGTERM)
    time.sleep(0.5)
    p.send_signal(signal.SIGTERM)
    time.sleep(0.5)
    p.send_signal(signal.SIGINT)
    _, err = p.communicate(timeout=60)
    assert p.returncode == 1 and "SIGTERM" in err and "Traceback" not in err, err
    time.sleep(0.5)
    assert not any(_alive(pid) for pid in pids), "a rank that ignores SIGTERM outlived an interrupted clean-up"


def test_ranks_die_with_a_killed_launcher(tmp_path):
    """SIGKILL to the launcher leaves it no chance to clean up; the ranks lead their own sessions, so nothing else reaches
    them: each carries PR_SET_PDEATHSIG = SIGKILL."""
    import signal
    import time
    p, pids = _launch_two_ranks(tmp_path, "import os, sys, time\nopen(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    p.send_signal(signal.SIGKILL)
    p.communicate(timeout=60)
    t0 = time.time()
    while any(_alive(pid) for pid in pids) and time.time() - t0 < 10:
        time.sleep(0.1)
    assert not any(_alive(pid) for pid in pids), "ranks outlived a SIGKILLed launcher"
