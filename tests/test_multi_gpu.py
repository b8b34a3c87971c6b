"""-m gpu, and SKIPPED below two visible devices: the RCCL twins of the gloo world-size-2 tests (tests/test_parallel.py) — the
first thing to run on a multi-GPU MI355X node (tools/first_multi_gpu.sh does).  One process per GPU, backend "nccl" (= RCCL),
rendezvous on 127.0.0.1.  SURVEY.md section 8e; the reference's DDP is dead code (src/diffusion/train_util.py:8-9, 98-99,
src/utils/dist_util.py:29-42, 62-68)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs (a multi-GPU MI355X node)")]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


LOOP_WORKER = r"""
import json, os, sys
sys.path.insert(0, os.environ["S3D_REPO"])
import numpy as np, torch
from sin3dm_amd import parallel, testing as T
from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.diffusion.train_util import TrainLoop
from sin3dm_amd.diffusion.unet_triplane import TriplaneUNetModelSmall
rank, local, world = parallel.env_rank_world()
torch.cuda.set_device(local)
dev = torch.device(f"cuda:{local}")
parallel.init(device=dev)
assert torch.distributed.get_backend() == "nccl"
H, W, D = 20, 28, 12

def data():
    x0 = torch.from_numpy(T.synthetic_noise((12, H + D, W + D), 400)).clamp(-1, 1).to(dev)
    while True:
        yield x0.unsqueeze(0).expand(2, -1, -1, -1), dict(H=H, W=W, D=D)

def run(overlap):
    os.environ["S3D_OVERLAP_ALLREDUCE"] = "1" if overlap else "0"
    torch.manual_seed(100 + rank); np.random.seed(100 + rank)          # timestep / noise streams: per rank, the same for both runs
    model = TriplaneUNetModelSmall(12, 32, 12, channel_mult=(1, 2), use_scale_shift_norm=True)
    model.load_state_dict(T.synthetic_state_dict(T.unet_param_shapes(model_channels=32), 5 + rank))   # ranks start DIFFERENT
    model.to(dev)
    loop = TrainLoop(model=model, diffusion=create_gaussian_diffusion(steps=1000, predict_xstart=True), data=data(), batch_size=2,
                     microbatch=-1, lr=1e-3, ema_rate="0.99", log_interval=10 ** 9, save_interval=10 ** 9, resume_checkpoint=False,
                     lr_anneal_steps=10, log_dir=None)
    assert loop.overlap_allreduce == overlap and loop.world == world
    it = data()
    for _ in range(3):
        batch, cond = next(it)
        loop.run_step(batch, cond)
        loop.step += 1
    torch.cuda.synchronize()
    return model.flat_parameters.clone()

a = run(False)
b = run(True)
res = {"equal": bool(torch.equal(a, b)), "finite": bool(torch.isfinite(a).all()), "sum": float(a.double().sum()),
       "digest": float((a.double() * torch.arange(a.numel(), device=dev).double().cos()).sum()),
       "device": torch.cuda.get_device_properties(local).name, "local": local}
out = parallel.gather_objects(res)
if rank == 0:
    print("RESULT " + json.dumps(out))
parallel.barrier()
torch.distributed.destroy_process_group()
"""


def _run_ranks(tmp_path, code, n=2, extra_env=None):
    script = tmp_path / "worker.py"
    script.write_text(code)
    env = dict(os.environ, S3D_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("S3D_OVERLAP_ALLREDUCE", None)
    env.update(extra_env or {})
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    return json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0][len("RESULT "):])


def test_two_rank_rccl_trainloop_overlap_on_off_same_parameters(tmp_path):
    """TrainLoop.run_step three times on two RCCL ranks with the REAL denoiser (HIP forward / backward, fused AdamW + EMA, device
    repack): the flat parameters end bit for bit the same with ONE all-reduce after the backward pass and with the exchange cut
    into the pass's finishing groups on a communication stream behind the library's progress marks (S3D_OVERLAP_ALLREDUCE=1) —
    a two-term sum does not depend on how the vector is cut — and are equal across the ranks (broadcast at start, averaged
    gradients)."""
    res = _run_ranks(tmp_path, LOOP_WORKER)
    assert all(r["equal"] and r["finite"] for r in res), res
    assert res[0]["sum"] == res[1]["sum"] and res[0]["digest"] == res[1]["digest"], res
    assert sorted(r["local"] for r in res) == [0, 1]


def test_bench_line_over_two_rccl_ranks():
    """`python bench.py --gpus 2` (independent samples, no data-path collective; barrier + MAX over RCCL): one line, two distinct
    devices, whole-job value = 2 samples' worth."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--prewarm", "30"],
                       capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["dist_backend"] == "nccl"
    assert line["devices_verified_distinct"] is True and len({(x["pci_bus_id"], x["uuid"]) for x in line["ranks"]}) == 2
    assert abs(line["value"] - 2 * 1.0 / (line["ms_per_step"] * 1e-3) / 1000) < 1e-6 * line["value"]


def test_sample_cli_two_ranks_write_the_single_rank_files(tmp_path):
    """S3D_GPUS=2 python -m sin3dm_amd.sample: sample indices striped over the ranks, per-sample generators — the files equal
    the ones a single rank writes."""
    import numpy as np
    from test_cli_gpu import make_experiment
    tag = make_experiment(str(tmp_path), hwd=(12, 16, 10), mc=32)
    outs = {}
    for name, gpus in (("one", "1"), ("two", "2")):
        e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run([sys.executable, "-m", "sin3dm_amd.sample", "--tag", tag, "--n_samples", "4", "--output", name, "--use_ddim", "True",
                            "--timestep_respacing", "10", "--vox", "--reso", "16"], cwd=REPO, env=dict(e, S3D_GPUS=gpus),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = [dict(np.load(os.path.join(tag, name, f"{i:03d}", "feat.npz"))) for i in range(4)]
    for a, b in zip(outs["one"], outs["two"]):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
