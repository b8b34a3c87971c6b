"""CPU-only, world_size 2 over gloo: the N>1 sampling path (striping, per-sample seeds, max-over-ranks timing,
path gather) behaves as the single-process path."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, os.environ["S3D_REPO"])
import torch
from sin3dm_amd import parallel
rank, local, world = parallel.init(backend="gloo")
assert world == 2 and rank == int(os.environ["RANK"])
mine = parallel.shard_indices(7, rank, world)
# per-sample noise depends on the sample index only
noise = {i: torch.randn(4, generator=torch.Generator().manual_seed(parallel.sample_seed(1000, i))).tolist() for i in mine}
parallel.barrier()
t = parallel.max_over_ranks(0.25 + rank)            # rank 1 is the slow one
allp = parallel.gather_objects({"rank": rank, "mine": mine, "noise": noise})
if rank == 0:
    print("RESULT " + json.dumps({"t": t, "all": allp}))
parallel.shutdown()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RDZV_TROUBLE = ("Address already in use", "EADDRINUSE", "Connection refused", "Connection reset", "connectFullMesh", "Gloo connect")


def _spawn2(script, extra=None, timeout=180):
    """Two gloo ranks of `script` on a fresh rendezvous port.  A port handed out by the kernel can be taken again between the probe
    and the store's bind (or a full-mesh connect can be reset on a loaded box): such launches are retried on another port — what is
    under test is the ranks' logic, not the port lottery."""
    for attempt in range(3):
        env = dict(os.environ, S3D_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
        env.update(extra or {})
        procs = []
        for r in range(2):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=timeout) for p in procs]
        bad = [err for p, (_, err) in zip(procs, outs) if p.returncode != 0]
        if bad and attempt < 2 and any(t in err for err in bad for t in _RDZV_TROUBLE):
            continue
        return procs, outs


def test_two_rank_gloo_striping(tmp_path):
    import json
    import torch
    from sin3dm_amd import parallel
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs, outs = _spawn2(script)
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[len("RESULT "):])
    assert res["t"] == pytest.approx(1.25)                      # MAX over ranks
    got = sorted(i for r in res["all"] for i in r["mine"])
    assert got == list(range(7))                                 # every sample exactly once
    assert res["all"][0]["mine"] == [0, 2, 4, 6] and res["all"][1]["mine"] == [1, 3, 5]
    # identical to what one process would draw for the same sample indices
    for r in res["all"]:
        for i, v in r["noise"].items():
            ref = torch.randn(4, generator=torch.Generator().manual_seed(parallel.sample_seed(1000, int(i)))).tolist()
            assert v == ref


def test_shard_helpers():
    from sin3dm_amd import parallel
    for n in (0, 1, 5, 64):
        for world in (1, 2, 4, 8):
            parts = [parallel.shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert list(parallel.batches(list(range(5)), 2)) == [[0, 1], [2, 3], [4]]


TRAIN_WORKER = r"""
import json, os, sys
sys.path.insert(0, os.environ["S3D_REPO"])
import numpy as np, torch
from sin3dm_amd import parallel
from sin3dm_amd.diffusion.resample import LossSecondMomentResampler
rank, local, world = parallel.init(backend="gloo")
# data-parallel step on a toy objective: loss_b = 0.5*|p - x_b|^2, global batch of 4 split 2 / 2
x = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 6) * 0.1
p = torch.full((6,), float(rank + 1))                  # ranks start different ...
parallel.broadcast_flat_(p, src=0)                     # ... and are made identical (rank 0's copy)
mine = x[rank * 2:(rank + 1) * 2]
g = (p[None] - mine).mean(0)                           # gradient of the LOCAL mean
parallel.average_flat_(g)                              # -> gradient of the GLOBAL mean
p -= 0.5 * g
# the loss-aware timestep sampler sees every rank's (t, loss) pairs in rank order
class D: num_timesteps = 4
s = LossSecondMomentResampler(D(), history_per_term=2)
s.update_with_local_losses(torch.tensor([rank, 3]), torch.tensor([1.0 + rank, 5.0 + rank]))
out = parallel.gather_objects({"p": p.tolist(), "g": g.tolist(), "counts": s._loss_counts.tolist(), "hist": s._loss_history.tolist()})
if rank == 0:
    print("RESULT " + json.dumps(out))
parallel.shutdown()
"""


def test_two_rank_gloo_data_parallel_step(tmp_path):
    import json
    import torch
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    procs, outs = _spawn2(script)
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    # what ONE process with the whole batch of 4 computes
    x = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 6) * 0.1
    p = torch.full((6,), 1.0)
    g = (p[None] - x).mean(0)
    p = p - 0.5 * g
    for r in res:
        assert torch.allclose(torch.tensor(r["g"]), g, atol=1e-6)
        assert torch.allclose(torch.tensor(r["p"]), p, atol=1e-6)
        assert r["counts"] == [1, 1, 0, 2]                        # t=0 (rank 0), t=1 (rank 1), t=3 twice
        assert r["hist"][3] == [5.0, 6.0]                         # rank order
    assert res[0] == res[1]


CHUNK_WORKER = r"""
import json, os, sys
sys.path.insert(0, os.environ["S3D_REPO"])
import numpy as np, torch
from sin3dm_amd import parallel, testing as T
rank, local, world = parallel.init(backend="gloo")
# the flat-vector layout of a real denoiser (state-dict order, as s3d_unet_param_offset lays it out)
layout, off = [], 0
for name, shp in T.unet_param_shapes(model_channels=32).items():
    n = int(np.prod(shp)); layout.append((name, off, n)); off += n
groups = parallel.grad_ready_groups(layout)
g = torch.Generator().manual_seed(77 + rank)
flat = torch.randn(off, generator=g) * torch.logspace(-6, 3, off)          # every magnitude: rounding differences would show
whole = parallel.average_flat_(flat.clone())
chunked = parallel.average_flat_groups_(flat.clone(), groups)
covered = torch.zeros(off, dtype=torch.int32)
for grp in groups:
    for b, e in grp:
        covered[b:e] += 1
res = {"equal": bool(torch.equal(whole, chunked)), "covered_once": bool((covered == 1).all()), "n": off,
       "ngroups": [len(x) for x in groups], "first": [x[0] if x else None for x in groups],
       "sum": float(whole.double().sum())}
out = parallel.gather_objects(res)
if rank == 0:
    print("RESULT " + json.dumps(out))
parallel.shutdown()
"""


def test_grad_ready_groups_cover_the_flat_vector_in_backward_order():
    import numpy as np
    from sin3dm_amd import parallel, testing as T
    layout, off = [], 0
    for name, shp in T.unet_param_shapes(model_channels=32).items():
        n = int(np.prod(shp)); layout.append((name, off, n)); off += n
    g0, g1, g2 = parallel.grad_ready_groups(layout)
    names = {k: [n for n, o, m in layout if any(b <= o < e for b, e in g)] for k, g in enumerate((g0, g1, g2))}
    assert all(n.startswith(("out.", "output_blocks.")) and "emb_layers" not in n for n in names[0]) and names[0]
    assert all(n.startswith(("input_blocks.", "in_conv.")) and "emb_layers" not in n for n in names[1]) and names[1]
    assert all("emb_layers" in n or n.startswith("time_embed") for n in names[2]) and names[2]
    assert sorted(names[0] + names[1] + names[2]) == sorted(n for n, _, _ in layout)
    # the early groups carry nearly all the bytes (the 3x3 convolution weights): that is what overlaps with the backward pass
    early = sum(e - b for g in (g0, g1) for b, e in g)
    assert early > 0.8 * off


def test_two_rank_gloo_chunked_allreduce_equals_whole_vector(tmp_path):
    """The gradient all-reduce cut into the groups the backward pass finishes in order (parallel.average_flat_groups_, used by
    TrainLoop.forward_backward with the library's progress marks on a GPU) gives bit for bit what ONE all-reduce of the whole
    flat vector gives (world size 2, gloo)."""
    import json
    script = tmp_path / "chunk_worker.py"
    script.write_text(CHUNK_WORKER)
    procs, outs = _spawn2(script)
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    for r in res:
        assert r["equal"] and r["covered_once"], r
    assert res[0]["sum"] == res[1]["sum"]


LOOP_WORKER = r"""
import json, os, sys
sys.path.insert(0, os.environ["S3D_REPO"])
import numpy as np, torch
from sin3dm_amd import parallel, testing as T
from sin3dm_amd.diffusion.train_util import TrainLoop
rank, local, world = parallel.init(backend="gloo")

# a stand-in denoiser with the REAL flat-vector layout (state-dict order of a 32-channel UNet) and a deterministic gradient,
# so that TrainLoop's multi-rank control flow (broadcast, exchange, optimizer step, annealing) runs without a GPU
layout, off = [], 0
for name, shp in T.unet_param_shapes(model_channels=32).items():
    n = int(np.prod(shp)); layout.append((name, off, n)); off += n

class Toy:
    def __init__(self, seed):
        self.flat_parameters = torch.randn(off, generator=torch.Generator().manual_seed(seed))
    def mark_parameters_changed(self): pass
    def grad_ready_groups(self): return parallel.grad_ready_groups(layout)
    def named_parameters(self): return iter(())
    def parameters(self): return iter([self.flat_parameters])

class ToyDiffusion:
    num_timesteps = 1000
    def training_losses_and_grads(self, model, micro, t, weights, cond, grad_out=None, grad_marks=None):
        assert grad_marks is None                        # CPU: no events
        g = torch.tanh(model.flat_parameters * 3.0) * micro.mean() + torch.sin(model.flat_parameters * (1.0 + t.float().mean() / 1000))
        g = g * torch.logspace(-4, 2, off)               # every magnitude
        grad_out.copy_(g)
        z = torch.zeros(micro.shape[0])
        return {"loss": z, "mse_xy": z, "mse_xz": z, "mse_yz": z}, grad_out

class Sgd:
    def __init__(self, model, lr): self.model, self.lr = model, lr
    def step(self, g): self.model.flat_parameters.add_(g, alpha=-self.lr)

def data():
    k = 0
    while True:
        yield torch.full((2, 12, 4, 4), float(rank + 1 + k)), {}
        k += 1

def run(overlap):
    os.environ["S3D_OVERLAP_ALLREDUCE"] = "1" if overlap else "0"
    torch.manual_seed(100 + rank)                        # the timestep sampler's stream: per rank, identical for both runs
    np.random.seed(100 + rank)
    model = Toy(seed=5 + rank)                           # ranks start DIFFERENT: the loop must broadcast rank 0's copy
    loop = TrainLoop(model=model, diffusion=ToyDiffusion(), data=data(), batch_size=2, microbatch=-1, lr=1e-2, ema_rate="0.99",
                     log_interval=1000, save_interval=10**9, resume_checkpoint=False, lr_anneal_steps=10, log_dir=None,
                     optimizer=Sgd(model, 1e-2))
    assert loop.overlap_allreduce == overlap and loop.world == 2
    it = data()
    for _ in range(2):
        batch, cond = next(it)
        loop.run_step(batch, cond)
        loop.step += 1
    return model.flat_parameters.clone(), loop

a, la = run(False)
b, lb = run(True)
assert lb._staging is not None and la._staging is None
res = {"equal": bool(torch.equal(a, b)), "sum": float(a.double().sum()), "moved": float((a - Toy(5).flat_parameters).abs().max()),
       "default_off": os.environ.pop("S3D_OVERLAP_ALLREDUCE") is not None and TrainLoop.__init__.__defaults__ is not None}
out = parallel.gather_objects(res)
if rank == 0:
    print("RESULT " + json.dumps(out))
parallel.shutdown()
"""


def test_two_rank_gloo_trainloop_overlap_on_off_same_parameters(tmp_path):
    """TrainLoop.run_step twice on two gloo ranks from the same seeds, once with the single all-reduce (the default) and once with
    S3D_OVERLAP_ALLREDUCE=1 (the exchange cut into the backward pass's finishing groups, persistent staging buffers): the flat
    parameters end bit for bit the same, on both ranks, and equal across ranks (the broadcast at start + identical averaged
    gradients).  The denoiser / optimizer are stand-ins with the real flat layout — the HIP kernels need a GPU; what is covered
    here is the loop's multi-rank control flow (src/diffusion/train_util.py:163-247)."""
    import json
    script = tmp_path / "loop_worker.py"
    script.write_text(LOOP_WORKER)
    procs, outs = _spawn2(script)
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])
    assert all(r["equal"] for r in res), res
    assert res[0]["sum"] == res[1]["sum"] and res[0]["moved"] > 1e-4


STAGE1_WORKER = r"""
import os, sys, time
sys.path.insert(0, os.environ["S3D_REPO"])
from sin3dm_amd import parallel, train
rank = int(os.environ["RANK"])
only_enc = os.environ["ONLY_ENC"] == "1"
calls = []
def slow_stage1(args):                                    # stands in for ShapeAutoEncoder.train (minutes on a real shape)
    time.sleep(float(os.environ["FAKE_STAGE1_S"]))
    calls.append("ae")
    open(os.path.join(args.tag, "encoding", "feat.npz"), "wb").close()
train.train_ae = slow_stage1
train.train_diffusion = lambda args, rank=0: calls.append("diffusion")      # stage 2 needs a GPU: what is under test comes before it
train.dist_util.dev = lambda: None
# the process group is created after stage 1, with a timeout SHORTER than stage 1 took: had the ranks been parked in a
# collective (or in the rendezvous) during stage 1, this would have expired
real_init = parallel.init
parallel.init = lambda device=None: real_init(backend="gloo", timeout_s=float(os.environ["FAKE_STAGE1_S"]) * 0.75)
t0 = time.time()
train.main(["--tag", os.environ["EXP"], "--data_path", "unused.npz"] + (["--only_enc"] if only_enc else []), confirm=lambda q: "y")
waited = time.time() - t0
if only_enc:
    assert calls == (["ae"] if rank == 0 else [])         # nothing follows stage 1: the other ranks neither wait nor run anything
else:
    assert os.path.exists(os.path.join(os.environ["EXP"], "encoding", "feat.npz"))       # every rank comes out AFTER stage 1
    assert calls == (["ae", "diffusion"] if rank == 0 else ["diffusion"])
print("RESULT %d %.2f" % (rank, waited))
"""


def _run_stage1(tmp_path, only_enc, nproc_env=None):
    script = tmp_path / "stage1_worker.py"
    script.write_text(STAGE1_WORKER)
    exp = tmp_path / "EXP"
    env = dict(os.environ, S3D_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", EXP=str(exp),
               FAKE_STAGE1_S="4", ONLY_ENC="1" if only_enc else "0")
    env.pop("LOCAL_WORLD_SIZE", None)
    env.update(nproc_env or {})
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    return exp, procs, outs


def test_train_cli_ranks_wait_for_stage1_outside_any_collective(tmp_path):
    """python -m sin3dm_amd.train on N ranks: rank 0 runs the auto-encoder stage (src/train.py:8-29) before torch.distributed is
    initialised; the other ranks wait for its marker file — not in a barrier whose watchdog would abort the job when stage 1
    outlasts the collective timeout (25 000 iterations x 6.3 ms is already 160 s).  The marker is removed once every
    rank is past its wait, and a stale one of an earlier launch — one whose launcher pid is gone — is cleared by rank 0;
    the marker of a launch whose launcher is still alive (a second job on the same --tag) is left alone."""
    os.makedirs(tmp_path / "EXP")
    gone = subprocess.Popen([sys.executable, "-c", "pass"])
    gone.wait()
    stale = tmp_path / "EXP" / f".stage1_done_run_1_{gone.pid}"
    stale.write_text("")
    live = tmp_path / "EXP" / f".stage1_done_run_2_{os.getpid()}"
    live.write_text("")
    exp, procs, outs = _run_stage1(tmp_path, only_enc=False)
    assert live.exists()
    live.unlink()
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    waits = {int(l.split()[1]): float(l.split()[2]) for o, _ in outs for l in o.splitlines() if l.startswith("RESULT ")}
    assert waits[0] >= 4.0 and waits[1] >= 3.0, waits            # rank 1 really waited for rank 0's stage
    assert not [f for f in os.listdir(exp) if f.startswith(".stage1_done_")], os.listdir(exp)


def test_train_cli_only_enc_leaves_no_marker_and_parks_nobody(tmp_path):
    """--only_enc on N ranks: stage 1 is rank 0's alone and nothing follows it — the other ranks return at once, no marker file is
    written (it used to stay behind for ever: ADVICE r4)."""
    exp, procs, outs = _run_stage1(tmp_path, only_enc=True)
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    waits = {int(l.split()[1]): float(l.split()[2]) for o, _ in outs for l in o.splitlines() if l.startswith("RESULT ")}
    assert waits[0] >= 4.0 and waits[1] < 3.0, waits
    assert not [f for f in os.listdir(exp) if f.startswith(".stage1_done_")]


def test_train_cli_refuses_a_multi_node_launch(tmp_path):
    """The marker's name contains the launcher's pid: ranks of another node would wait for a name nobody writes.  Fail fast."""
    exp, procs, outs = _run_stage1(tmp_path, only_enc=False, nproc_env={"LOCAL_WORLD_SIZE": "1"})
    for p, (o, err) in zip(procs, outs):
        assert p.returncode != 0 and "one node only" in err
