"""One-process-per-GPU launcher used by `bench.py --gpus N` and `tools/bench_train.py --gpus N` when no torchrun
environment is present.  Deliberately torch-free: the parent never touches a GPU, never exec()s, only starts N fresh
children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) and watches ALL of them:

  * the first child that exits non-zero ends the run: the others are terminated (SIGTERM, then SIGKILL) and the launch
    fails within seconds — a rank that dies before the RCCL rendezvous must not leave its peers inside
    `init_process_group` / a barrier until the store or collective timeout (minutes);
  * an overall deadline (S3D_LAUNCH_TIMEOUT seconds, default 3600) bounds a hung collective the same way;
  * the children never outlive the parent's watch: they run in their own sessions (process groups), the poll loop sits in a
    try/finally that stops every group, and SIGTERM / SIGINT / SIGHUP to the parent (a harness `timeout`, a cancelled gpurun
    call, Ctrl-C) are turned into that same clean-up instead of leaving N ranks inside an RCCL collective with their GPUs held.
    The FIRST such signal is only recorded (the poll loop sees the flag) and every later one is ignored, so the clean-up —
    SIGTERM, grace period, SIGKILL, the waits — cannot be cut short by a second delivery (GNU `timeout` signals the pid and
    then the group; a double Ctrl-C does the same).  Children are started with these signals blocked, so none can arrive
    between `Popen` and the book-keeping that makes the child stoppable.  Should the launcher itself be SIGKILLed, every rank
    carries a parent-death signal (prctl PR_SET_PDEATHSIG = SIGKILL, set between fork and exec) and dies with it;
  * rank 0's stdout is drained by a thread (no pipe dead-lock) and its last JSON line is relayed.

The reference is single-process (src/utils/dist_util.py:29-42 is commented out); this is the launch side of SURVEY.md §8e.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time

_RDZV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _signal_group(p, sig):
    """Signal the child's whole process group (it leads its own session: helpers it started go with it)."""
    try:
        os.killpg(p.pid, sig)
    except (OSError, ProcessLookupError):
        try:
            p.send_signal(sig)
        except OSError:
            pass


def _stop(procs, grace=5.0):
    live = [p for p in procs if p.poll() is None]
    for p in live:
        _signal_group(p, signal.SIGTERM)
    t_end = time.time() + grace
    for p in live:
        try:
            p.wait(timeout=max(0.05, t_end - time.time()))
        except subprocess.TimeoutExpired:
            _signal_group(p, signal.SIGKILL)
            p.wait()
    for p in procs:                       # anything a finished rank left behind in its group
        if p not in live and _group_alive(p):
            _signal_group(p, signal.SIGKILL)


def _group_alive(p):
    try:
        os.killpg(p.pid, 0)
        return True
    except (OSError, ProcessLookupError):
        return False


_PR_SET_PDEATHSIG = 1


def _libc():
    try:
        import ctypes
        return ctypes.CDLL(None, use_errno=True)
    except OSError:
        return None


def _rank_preexec(libc, parent_pid, blocked):
    die = _die_with_parent(libc, parent_pid)

    def hook():
        if blocked:
            signal.pthread_sigmask(signal.SIG_UNBLOCK, blocked)      # the launcher blocked them around the fork: the rank must not inherit that
        die()
    return hook


def _die_with_parent(libc, parent_pid):
    """preexec hook of a rank (runs between fork and exec): SIGKILL me when the launcher goes away — the ranks lead their own
    sessions, so neither the terminal's SIGHUP nor a group kill of the launcher reaches them otherwise."""
    def hook():
        if libc is None:
            return
        libc.prctl(_PR_SET_PDEATHSIG, int(signal.SIGKILL), 0, 0, 0)
        if os.getppid() != parent_pid:        # the launcher died before the prctl took effect
            os._exit(1)
    return hook


def spawn_ranks(script, argv, n, timeout=None, poll_s=0.1, out=sys.stdout, err=sys.stderr, module=False, relay_json=True) -> int:
    """Run `python script *argv` (module=True: `python -m script *argv`) as n ranks.  relay_json: rank 0's stdout is
    captured and its last line that starts with '{' is relayed (the bench contract: ONE JSON line); otherwise the children
    inherit stdout (training / sampling CLIs that log as they go).  Returns a process exit code."""
    timeout = float(os.environ.get("S3D_LAUNCH_TIMEOUT", "3600")) if timeout is None else float(timeout)
    port = free_port()
    base = {k: v for k, v in os.environ.items() if k not in _RDZV_KEYS}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    chunks = []
    failed, why = None, None
    t0 = time.time()

    got = []                              # the first stop signal's name; later ones are swallowed (see the module text)

    def on_signal(signum, frame):
        if not got:
            got.append(signal.Signals(signum).name)

    handled = [sg for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP) if threading.current_thread() is threading.main_thread()]
    previous = {sg: signal.signal(sg, on_signal) for sg in handled}
    reader = None
    libc, me = _libc(), os.getpid()
    try:
        for r in range(n):
            if got:
                break
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            cmd = [sys.executable] + (["-m", script] if module else [os.path.abspath(script)]) + list(argv)
            stdout = None if not relay_json else (subprocess.PIPE if r == 0 else subprocess.DEVNULL)
            # a stop signal must not land between the fork and the append that makes the child stoppable: block, start, note, unblock
            # (the child gets an empty mask back from subprocess: restore_signals / its own exec)
            old = signal.pthread_sigmask(signal.SIG_BLOCK, handled) if handled else None
            try:
                procs.append(subprocess.Popen(cmd, env=env, stdout=stdout, text=True, start_new_session=True,
                                              preexec_fn=_rank_preexec(libc, me, handled)))
            finally:
                if old is not None:
                    signal.pthread_sigmask(signal.SIG_SETMASK, old)
        if not got:
            reader = threading.Thread(target=lambda: chunks.extend(procs[0].stdout) if relay_json else None, daemon=True)
            reader.start()
        while not got:
            rcs = [p.poll() for p in procs]
            bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                failed, why = bad, f"rank(s) {bad} exited with {[rcs[i] for i in bad]}"
                break
            if all(rc == 0 for rc in rcs):
                break
            if time.time() - t0 > timeout:
                failed, why = [i for i, rc in enumerate(rcs) if rc is None], f"no result after {timeout:.0f} s"
                break
            time.sleep(poll_s)
        if got:
            failed, why = list(range(len(procs))), f"the launcher received {got[0]}"
    except KeyboardInterrupt:             # (only when this is not the main thread's handler: e.g. raised by a caller's own handler)
        failed, why = list(range(len(procs))), "the launcher was interrupted"
    finally:
        # whatever ended the watch — a failed rank, the deadline, a signal, an exception in this function — no rank survives it
        if any(p.poll() is None for p in procs) or failed is not None:
            _stop(procs)
        for sg, h in previous.items():
            signal.signal(sg, h)
    if reader is None:
        err.write(f"{os.path.basename(script)}: launch of {n} ranks failed: {why}\n")
        return 1
    reader.join(timeout=5.0)
    text = "".join(chunks)
    line = next((l for l in reversed(text.splitlines()) if l.startswith("{")), None)
    if failed is None and not relay_json:
        return 0
    if failed is not None or line is None:
        err.write(f"{os.path.basename(script)}: launch of {n} ranks failed: {why or 'rank 0 printed no result line'}; "
                  f"the remaining ranks were stopped after {time.time() - t0:.1f} s; rank 0 stdout tail: {text[-400:]!r}\n")
        return 1
    out.write(line + "\n")
    out.flush()
    return 0


def device_identity(local: int) -> dict:
    """What a rank reports so that the JSON line shows N DISTINCT devices took part (index, PCI bus id, uuid, name)."""
    import torch
    p = torch.cuda.get_device_properties(local)
    bus = None
    if hasattr(p, "pci_bus_id"):
        bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
    return {"device_index": int(local), "pci_bus_id": bus, "uuid": str(getattr(p, "uuid", "")) or None, "name": p.name}


def maybe_spawn_module(module: str, argv=None) -> int | None:
    """`S3D_GPUS=N python -m <module> ...` without a torchrun environment: run the module as N ranks (one per GPU) and return
    the exit code; None when this process should simply run as the single (or torchrun-launched) rank it is."""
    n = int(os.environ.get("S3D_GPUS", "1") or 1)
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return None
    return spawn_ranks(module, sys.argv[1:] if argv is None else argv, n, module=True, relay_json=False)
