"""-m gpu, ONE device: RCCL (torch.distributed backend "nccl") really executes — at world size 1.  Every call of the multi-GPU
path that does not need a second device to be exercised runs here on the driver's one-GPU box: process-group creation on
cuda:0, the parameter broadcast, ReduceOp.AVG, the bucketed / staged group exchange on a communication stream behind the REAL
progress marks of s3d_unet_backward_marked, TrainLoop.run_step with the exchange on and off, and bench.py --force-dist.
At world size 1 every collective is the identity, so each result is compared BIT FOR BIT with the same computation without a
process group.  What still needs N > 1 is named in DESIGN.md section 6 (cross-rank sums, xGMI bandwidth, the scaling curve).
Each test body runs in a fresh child process (a process group is created once per process; nothing re-execs after touching
the GPU).  SURVEY.md section 8e; the reference's DDP is dead code (src/diffusion/train_util.py:8-9, 98-99,
src/utils/dist_util.py:29-42, 62-68)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

_STRIP = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "S3D_OVERLAP_ALLREDUCE", "S3D_FORCE_DIST")


def _child(tmp_path, code, timeout=900):
    script = tmp_path / "worker.py"
    script.write_text(code)
    env = {k: v for k, v in os.environ.items() if k not in _STRIP}
    env.update(S3D_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])


COLLECTIVES = r"""
import json, os, sys
sys.path.insert(0, os.environ["S3D_REPO"])
import torch, torch.distributed as dist
from sin3dm_amd import parallel
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
assert not parallel.active()
rank, local, world = parallel.init(backend="nccl", device=dev, force=True)
assert parallel.active() and (rank, world) == (0, 1) and dist.get_world_size() == 1 and "nccl" in str(dist.get_backend())
g = torch.Generator(device=dev).manual_seed(5)
n = (7 << 20) + 12345                                   # 28 MB of fp32: the flat gradient of the 64-channel UNet, odd length
flat = torch.randn(n, device=dev, generator=g)
flat[::1000] = 1e-42                                    # denormals and signed zeros must come back untouched as well
flat[1::1000] = -0.0
ref = flat.clone()
res = {}
parallel.broadcast_flat_(flat, src=0); torch.cuda.synchronize()
res["broadcast_bits"] = bool(torch.equal(flat.view(torch.int32), ref.view(torch.int32)))
assert parallel._has_avg()                              # RCCL: the average is taken inside the collective
parallel.average_flat_(flat); torch.cuda.synchronize()
res["avg_bits"] = bool(torch.equal(flat.view(torch.int32), ref.view(torch.int32)))
# the cut exchange: big ranges reduced in place, runs of small ones through a packed staging buffer, on a communication stream
# behind events recorded on the main stream, joined at the end
bounds = [0, 100, 70000, 70010, 70500, 1 << 20, (1 << 20) + 7, 3 << 20, n]
ranges = list(zip(bounds[:-1], bounds[1:]))
groups = ([ranges[0], ranges[2], ranges[5]], [ranges[1], ranges[3], ranges[7]], [ranges[4], ranges[6]])
comm = torch.cuda.Stream(device=dev)
marks = [torch.cuda.Event(), torch.cuda.Event()]
staging = parallel.GroupStaging(flat, groups)
for it in range(3):
    flat.mul_(1.0)                                      # main-stream work in front of the marks
    marks[0].record(); marks[1].record()
    parallel.average_flat_groups_(flat, groups, marks, comm, staging=staging)
torch.cuda.synchronize()
res["groups_bits"] = bool(torch.equal(flat.view(torch.int32), ref.view(torch.int32)))
res["max"] = parallel.max_over_ranks(1.25, device=dev)
res["gather"] = parallel.gather_objects({"rank": rank})
parallel.barrier()
res["backend"] = str(dist.get_backend()); res["world"] = dist.get_world_size()
res["device"] = torch.cuda.get_device_properties(0).name
parallel.shutdown()
assert not parallel.active()
print("RESULT " + json.dumps(res))
"""


def test_rccl_world1_collectives_are_bitwise_identities(tmp_path):
    """init("nccl") on cuda:0 at world size 1, then broadcast / AVG all-reduce / the staged group exchange on a communication
    stream / MAX / object gather / barrier / shutdown: every one executes in RCCL and returns its operand bit for bit
    (denormals and -0.0 included)."""
    res = _child(tmp_path, COLLECTIVES)
    assert res["broadcast_bits"] and res["avg_bits"] and res["groups_bits"], res
    assert res["max"] == 1.25 and res["gather"] == [{"rank": 0}] and "nccl" in res["backend"] and res["world"] == 1


LOOP = r"""
import json, os, sys
sys.path.insert(0, os.environ["S3D_REPO"])
import numpy as np, torch, torch.distributed as dist
from sin3dm_amd import parallel, testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.train_util import TrainLoop
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
H, W, D = 20, 28, 12
calls = {"all_reduce": 0, "broadcast": 0}
_ar, _bc = dist.all_reduce, dist.broadcast
def ar(*a, **k):
    calls["all_reduce"] += 1
    return _ar(*a, **k)
def bc(*a, **k):
    calls["broadcast"] += 1
    return _bc(*a, **k)
dist.all_reduce, dist.broadcast = ar, bc

def data():
    x0 = torch.from_numpy(T.synthetic_noise((12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
    while True:
        yield x0.unsqueeze(0).expand(2, -1, -1, -1), dict(H=H, W=W, D=D)

def run(overlap, expect_dist):
    os.environ["S3D_OVERLAP_ALLREDUCE"] = "1" if overlap else "0"
    torch.manual_seed(100); np.random.seed(100)
    model = TriplaneUNetModelSmall(12, 32, 12, channel_mult=(1, 2), use_scale_shift_norm=True)
    model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 5))
    model.to(dev)
    loop = TrainLoop(model=model, diffusion=create_gaussian_diffusion(steps=1000, predict_xstart=True), data=data(), batch_size=2,
                     microbatch=-1, lr=1e-3, ema_rate="0.99", log_interval=10 ** 9, save_interval=10 ** 9, resume_checkpoint=False,
                     lr_anneal_steps=10, log_dir=None)
    assert loop.overlap_allreduce == overlap and loop.dist_on == expect_dist and loop.world == 1
    it = data()
    for _ in range(3):
        batch, cond = next(it)
        loop.run_step(batch, cond)
        loop.step += 1
    torch.cuda.synchronize()
    assert (loop._comm is not None) == (overlap and expect_dist)
    return model.flat_parameters.clone(), loop.opt.ema[0].clone()

plain, plain_ema = run(False, False)                     # no process group: the single-process arithmetic
n0 = dict(calls)
parallel.init(backend="nccl", device=dev, force=True)
assert "nccl" in str(dist.get_backend())
one, one_ema = run(False, True)                          # ONE RCCL all-reduce of the flat gradient per step (+ the start broadcast)
n1 = dict(calls)
cut, cut_ema = run(True, True)                           # the exchange cut at s3d_unet_backward_marked's marks, on the communication stream
n2 = dict(calls)
eq = lambda a, b: bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))
res = {"one_eq_plain": eq(one, plain), "cut_eq_plain": eq(cut, plain), "ema_one": eq(one_ema, plain_ema), "ema_cut": eq(cut_ema, plain_ema),
       "finite": bool(torch.isfinite(plain).all()),
       "calls_plain": n0, "calls_one": {k: n1[k] - n0[k] for k in n1}, "calls_cut": {k: n2[k] - n1[k] for k in n2}}
parallel.shutdown()
print("RESULT " + json.dumps(res))
"""


def test_rccl_world1_trainloop_exchange_on_off_and_no_group_same_bits(tmp_path):
    """TrainLoop.run_step x 3 with the REAL denoiser (HIP forward / backward with progress marks, fused AdamW + EMA): without a process
    group, with one RCCL all-reduce per step, and with the exchange cut into the backward pass's finishing groups on the
    communication stream behind the library's events (S3D_OVERLAP_ALLREDUCE=1) — parameters and EMA bit-equal in all three
    (the world-size-1 twin of tests/test_multi_gpu.py::test_two_rank_rccl_trainloop_overlap_on_off_same_parameters).  The call
    counts prove RCCL was entered: 1 broadcast + 3 all-reduces, then 1 broadcast + >= 3 x 3 group all-reduces."""
    res = _child(tmp_path, LOOP)
    assert res["finite"] and res["one_eq_plain"] and res["cut_eq_plain"] and res["ema_one"] and res["ema_cut"], res
    assert res["calls_plain"] == {"all_reduce": 0, "broadcast": 0}, res
    assert res["calls_one"] == {"all_reduce": 3, "broadcast": 1}, res
    assert res["calls_cut"]["broadcast"] == 1 and res["calls_cut"]["all_reduce"] >= 9, res


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in _STRIP}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--prewarm", "20",
                        "--profile-every", "0", "--no-cpu-baseline", "--traffic", "off", "--chains", "0", "--overlap", "off", *extra],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("config", ["c4", "c2"])
def test_bench_force_dist_runs_rccl_at_one_gpu(config):
    """bench.py --gpus 1 --force-dist: the line carries rccl_world_size 1 / dist_backend nccl; in c4 (TrainLoop) every step now
    contains a real RCCL all-reduce of the 28-MB flat gradient — and costs what the step without a process group costs
    (a loose gate here; the measured pair is kept in profiles/r06_rccl_world1.txt)."""
    plain = _bench("--config", config)
    forced = _bench("--config", config, "--force-dist")
    assert plain["rccl_world_size"] is None and plain["dist_backend"] is None
    assert forced["rccl_world_size"] == 1 and "nccl" in forced["dist_backend"] and forced["n_gpus"] == 1
    assert len(forced["ranks"]) == 1 and forced["ranks"][0]["rank"] == 0
    assert forced["ms_per_step"] < plain["ms_per_step"] * 1.15 + 0.05, (plain["ms_per_step"], forced["ms_per_step"])
    print(f"\n[rccl world 1] {config}: ms_per_step without a group {plain['ms_per_step']:.4f}, with RCCL at world 1 {forced['ms_per_step']:.4f}")
