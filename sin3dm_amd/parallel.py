"""Multi-GPU sharding of the sampling path: independent samples striped over ranks, no data-path collective.

The reference is single-process (its DDP/dist code is commented out: src/diffusion/train_util.py:8-9,
src/utils/dist_util.py:29-42).  Samples are independent (src/sample.py:33-47 loops over batches; GroupNorm
statistics and rollout means are per sample), so N GPUs = N processes that each take a stripe of the sample
indices.  Seeds are a function of the SAMPLE index, never of the rank or world size, so the set of results is
identical for any N.  One process per GPU, launched with `python -m torch.distributed.run`; the backend is
"nccl" (= RCCL over xGMI) on GPUs and "gloo" in the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None, timeout_s=None, force=None):
    """Initialise torch.distributed from the torchrun environment.  A single process stays without a process group unless
    `force` (or S3D_FORCE_DIST=1): then the group is created at world size 1 as well and EVERY collective below really runs —
    RCCL initialisation, the parameter broadcast, the AVG all-reduce and the communication-stream exchange execute on a one-GPU
    box exactly as they do on eight (tests/test_rccl_world1.py; MASTER_ADDR / MASTER_PORT default to a local rendezvous)."""
    rank, local, world = env_rank_world()
    if force is None:
        force = os.environ.get("S3D_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend, **kw)
    return rank, local, world


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def active():
    """True when a process group exists — at ANY world size: the collectives below run whenever it does (a forced world-size-1
    group exercises the same RCCL calls as a multi-GPU launch), and are skipped only in a plain single process."""
    return dist.is_available() and dist.is_initialized()


def shutdown():
    """Barrier, then destroy the process group (when there is one).  Every entry point that called init() ends with this:
    a rank that exits with a live RCCL / gloo group can abort in the group's destructor ('terminate called without an active
    exception', rc -6) after its work is done, and the launcher reports that finished run as failed."""
    if active():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()


def launch_token():
    """A string every rank of ONE launch derives identically and two launches do not share: the rendezvous port plus the
    launcher's pid (torchrun's run id when it provides one).  The pid makes it a SINGLE-NODE token (ranks on another node have
    another launcher): sin3dm_amd.train refuses multi-node launches instead of waiting for a name nobody writes."""
    return "%s_%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.environ.get("MASTER_PORT", "0"), os.getppid())


def wait_for_file(path, poll_s=0.5, log_every_s=60.0, what="a file"):
    """Block until `path` exists — the hand-off for work ONE rank does for minutes before the process group is created (the
    auto-encoder stage of train.py).  No collective and no timeout: a rank parked in an RCCL barrier for the length of stage 1
    would be aborted by the collective watchdog whenever the stage outlasts it.  Ends early (SystemExit) only when the
    launcher is gone (the parent pid changed: nobody is left to stop this rank), otherwise the launcher ends the wait by
    terminating the rank when rank 0 fails."""
    import sys
    import time
    parent = os.getppid()
    t0 = last = time.time()
    while not os.path.exists(path):
        if os.getppid() != parent:
            raise SystemExit(f"waiting for {what}: the launcher is gone, giving up")
        time.sleep(poll_s)
        now = time.time()
        if now - last >= log_every_s:
            print(f"[rank {os.environ.get('RANK', '?')}] still waiting for {what} ({now - t0:.0f} s)", file=sys.stderr, flush=True)
            last = now


def shard_indices(n_samples, rank, world):
    """Sample indices of this rank: i with i % world == rank (round-robin keeps per-rank counts within one)."""
    return list(range(rank, n_samples, world))


def sample_seed(base_seed, sample_index):
    """Per-sample generator seed (1000 + sample index)."""
    return int(base_seed) + int(sample_index)


def batches(indices, batch_size):
    for i in range(0, len(indices), batch_size):
        yield indices[i:i + batch_size]


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (used for the wall time of a timed region)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    """All ranks' python objects on every rank (file paths of the written samples; host side only)."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


# ---- data-parallel training: replicas with one exchange per step
def broadcast_flat_(flat, src=0):
    """Identical start on every rank: overwrite `flat` with rank `src`'s copy (the role of the reference's dead
    sync_params, src/utils/dist_util.py:62-68)."""
    if active():
        dist.broadcast(flat, src=src)
    return flat


def average_flat_(flat):
    """Mean over ranks of a flat gradient vector, in place: ONE all-reduce per training step (RCCL over xGMI on GPUs:
    28 MB at 64 channels, 112 MB at 128).  loss = mean over the GLOBAL batch => average the per-rank gradients."""
    if active():
        if _has_avg():
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)       # RCCL scales inside the collective: one pass over the vector, not two
        else:
            dist.all_reduce(flat)
            flat.mul_(1.0 / dist.get_world_size())
    return flat


def _has_avg():
    """ReduceOp.AVG exists in RCCL (backend "nccl", also inside a composite backend string), not in gloo.  RCCL's fp32 average
    scales every rank's operand by 1/world before the sum; for a power-of-two world size that is exact, i.e. the same bits as
    sum-then-scale — for any other world size (DP 3 / 5 / 6 / 7) the two differ in the last bit, so those keep SUM + one scaling
    pass: a run's arithmetic does not depend on which backend carried it."""
    world = dist.get_world_size()
    return "nccl" in str(dist.get_backend()) and (world & (world - 1)) == 0


# ---- gradient exchange overlapped with the backward pass ("comm can simply overlap the tail of backward")
def grad_ready_groups(layout):
    """layout: [(parameter name, offset, numel)] of the flat vector in state-dict order.  The backward pass runs from the
    output head towards the input, and the library marks two points (s3d_unet_backward_marked): returns three lists of merged
    (begin, end) ranges — final at mark 0 (out.*, output_blocks.* except emb_layers), at mark 1 (input_blocks.* except
    emb_layers, in_conv.*), and at the end (time_embed.*, every emb_layers.*: the stacked timestep linears come last)."""
    groups = ([], [], [])
    for name, off, numel in layout:
        if "emb_layers" in name or name.startswith("time_embed"):
            k = 2
        elif name.startswith("out.") or name.startswith("output_blocks."):
            k = 0
        elif name.startswith("input_blocks.") or name.startswith("in_conv."):
            k = 1
        else:
            k = 2
        g = groups[k]
        if g and g[-1][1] == off:
            g[-1] = (g[-1][0], off + numel)
        else:
            g.append((off, off + numel))
    assert sum(e - b for g in groups for b, e in g) == sum(n for _, _, n in layout)
    return groups


def _bucketed(flat, ranges, min_elems=1 << 16):
    """Ranges as all-reduce calls: a large contiguous range is reduced in place; runs of small ones (norm weights, biases
    between the emb_layers gaps) go through one packed staging buffer each, copied back afterwards."""
    big = [(b, e) for b, e in ranges if e - b >= min_elems]
    small = [(b, e) for b, e in ranges if e - b < min_elems]
    return big, small


class GroupStaging:
    """The packed buffers of the small ranges of each group, allocated ONCE (per flat vector and grouping) and reused by every
    step: a per-step torch.cat on the communication stream would allocate there every step (and its block would be recycled
    across streams by the caching allocator)."""

    def __init__(self, flat, groups):
        self.key = (flat.data_ptr(), flat.numel(), tuple(tuple(g) for g in groups))
        self.bufs = []
        for ranges in groups:
            _, small = _bucketed(flat, ranges)
            n = sum(e - b for b, e in small)
            self.bufs.append(torch.empty(n, dtype=flat.dtype, device=flat.device) if n else None)

    def matches(self, flat, groups):
        return self.key == (flat.data_ptr(), flat.numel(), tuple(tuple(g) for g in groups))


def average_flat_groups_(flat, groups, marks=None, comm_stream=None, staging=None):
    """Mean over ranks of `flat`, group by group.  At world size 2 this is bit-identical to average_flat_ (a two-term sum does
    not depend on how the vector is cut); from three ranks on the ring / tree reduction order of an element depends on where its
    chunk boundaries fall, so the cut vector and the whole vector agree only to fp32 round-off — S3D_OVERLAP_ALLREDUCE=0 and =1
    are then two different (equally valid) training trajectories.
    On a GPU, with `marks` (the events the backward pass recorded) and `comm_stream`: each group's all-reduces are enqueued on
    comm_stream behind its event, so they run while the main stream is still in the backward pass; the caller's stream waits for
    comm_stream at the end.  Without them (CPU / gloo): sequential.  staging: a GroupStaging kept by the caller across steps
    (None: buffers are allocated for this call)."""
    if not active():
        return flat
    world = dist.get_world_size()
    on_gpu = flat.is_cuda and comm_stream is not None
    main = torch.cuda.current_stream(flat.device) if on_gpu else None
    if staging is None or not staging.matches(flat, groups):
        staging = GroupStaging(flat, groups)

    avg = _has_avg()
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM

    def reduce_group(k, ranges):
        big, small = _bucketed(flat, ranges)
        for b, e in big:
            dist.all_reduce(flat[b:e], op=op)
        if small:
            stage = staging.bufs[k]
            o = 0
            for b, e in small:
                stage[o:o + e - b].copy_(flat[b:e])
                o += e - b
            dist.all_reduce(stage, op=op)
            o = 0
            for b, e in small:
                flat[b:e].copy_(stage[o:o + e - b])
                o += e - b

    for k, ranges in enumerate(groups):
        if not ranges:
            continue
        if on_gpu:
            if marks is not None and k < len(marks):
                comm_stream.wait_event(marks[k])
            else:
                comm_stream.wait_stream(main)             # the last group: everything the main stream has been given so far
            with torch.cuda.stream(comm_stream):
                reduce_group(k, ranges)
        else:
            reduce_group(k, ranges)
    if on_gpu:
        main.wait_stream(comm_stream)
    if not avg:
        flat.mul_(1.0 / world)
    return flat
