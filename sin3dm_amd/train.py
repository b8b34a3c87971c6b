"""Training CLI with the reference's flags and on-disk layout (src/train.py): auto-encoder stage, then diffusion stage.

    python -m sin3dm_amd.train --tag EXP --data_path shape.npz [--only_enc] [--enc_n_iters 25000 --diff_n_iters 25000 ...]
    python -m sin3dm_amd.train --tag EXP --enc_log PATH/TO/encoding            # reuse an existing encoding
    python -m torch.distributed.run --nproc-per-node 8 -m sin3dm_amd.train --tag EXP --data_path ... --diff_batch_size 4
    S3D_GPUS=8 python -m sin3dm_amd.train --tag EXP --data_path ... --diff_batch_size 4     # the same, self-launched (sin3dm_amd/launcher.py)

Stage 1 (src/train.py:8-29): ShapeAutoEncoder.train on the preprocessed shape, then EXP/encoding/{args.json, feat.npz,
ckpt_final.pth, eval_stat.json}.  It is a single-shape fit of a few minutes: rank 0 runs it BEFORE the process group exists and
the other ranks wait on a marker file (no collective is open while it runs, so neither --enc_n_iters nor a slow box can trip
the RCCL watchdog); torch.distributed is initialised after it, for stage 2 only.
Stage 2 (:32-75): the triplane diffusion UNet (sin3dm_amd/diffusion/train_util.py) -> EXP/diffusion/{args.json,
ema_<rate>_<step>.pt, opt<step>.pt, progress.jsonl} — the files sample.py of either implementation reads.
Multi-GPU: --diff_batch_size is PER GPU (the reference's 32 = 8 x 4); gradients are averaged with one all-reduce per
step.  The reference's reconstruction mesh export after stage 1 (decode_texmesh: PyMCubes/xatlas/nvdiffrast) is out of
scope.
"""
from __future__ import annotations

import os

from . import parallel
from .utils import dist_util
from .utils.common_util import seed_all
from .utils.parser_util import diffusion_log_dir, encoding_feat_path, encoding_log_dir, train_args


def train_ae(args):
    """Reference: src/train.py:8-29."""
    from .encoding.model import ShapeAutoEncoder
    from .utils.triplane_util import save_triplane_data

    print("[Training autoencoder]")
    assert args.data_path is not None
    ae = ShapeAutoEncoder(encoding_log_dir(args.tag), args, device=dist_util.dev())
    ae.train(args.data_path)
    fm = ae.encode()
    print("feat maps shape:", [tuple(f.shape) for f in fm])
    fm = [f.squeeze(0).detach().cpu().numpy() for f in fm]
    save_triplane_data(encoding_feat_path(args.tag), fm[0], fm[1], fm[2])


def train_diffusion(args, rank=0):
    """Reference: src/train.py:32-75."""
    from .diffusion.resample import create_named_schedule_sampler
    from .diffusion.script_util import create_model_and_diffusion_from_args
    from .diffusion.train_util import TrainLoop
    from .utils.triplane_util import get_data_iterator, load_triplane_data

    log_dir = diffusion_log_dir(args.tag)
    if rank == 0:
        print("[Training diffusion]\ncreating data loader...")
    src_data, sizes = load_triplane_data(encoding_feat_path(args.tag), device=dist_util.dev())
    data_iter = get_data_iterator(src_data, sizes, args.diff_batch_size)
    if rank == 0:
        print("creating model and diffusion...")
    model, diffusion = create_model_and_diffusion_from_args(args)
    model.to(dist_util.dev())
    schedule_sampler = create_named_schedule_sampler(args.schedule_sampler, diffusion)
    if rank == 0:
        print("training...")
    TrainLoop(model=model, diffusion=diffusion, data=data_iter, batch_size=args.diff_batch_size, microbatch=-1,
              lr=args.diff_lr, ema_rate=args.ema_rate, log_interval=args.log_interval, save_interval=args.save_interval,
              resume_checkpoint=False, use_fp16=args.use_fp16, schedule_sampler=schedule_sampler,
              weight_decay=args.weight_decay, lr_anneal_steps=args.diff_n_iters, log_dir=log_dir).run_loop()


def stage1_marker(tag):
    """Written by rank 0 once the experiment directory and the encoding are in place (one name per launch)."""
    return os.path.join(tag, ".stage1_done_" + parallel.launch_token())


def _marker_launcher_alive(path):
    """The marker's name ends in the launcher's pid (parallel.launch_token): a marker whose launcher still runs belongs to a live
    launch on the same --tag whose ranks may still be waiting for it — only the markers of dead launchers are stale."""
    try:
        pid = int(os.path.basename(path).rsplit("_", 1)[-1])
    except ValueError:
        return False
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


def main(argv=None, confirm=input):
    rank, local, world = parallel.env_rank_world()
    args = train_args(argv, confirm=confirm, write=rank == 0)        # rank 0 alone creates directories / symlink / args.json
    seed_all(0 + rank)                               # per-rank timestep / noise streams; weights are broadcast from rank 0
    dist_util.setup_dist(local if world > 1 else args.gpu_id)
    # stage 1 belongs to one process and takes minutes: no process group yet, the other ranks wait for its marker file
    if world > 1 and int(os.environ.get("LOCAL_WORLD_SIZE", world)) != world:
        # the marker's name contains the launcher's pid (what makes it unique per launch), which differs from node to node
        raise SystemExit("sin3dm_amd.train: the stage-1 hand-off is a marker file named per launcher: one node only "
                         f"(WORLD_SIZE={world}, LOCAL_WORLD_SIZE={os.environ.get('LOCAL_WORLD_SIZE')})")
    if args.only_enc and rank != 0:
        return                                       # stage 1 is rank 0's alone and nothing follows it: no marker, no wait
    marker = stage1_marker(args.tag)
    if rank == 0:
        if world > 1:                                # markers of launches that died between writing and removing theirs
            import glob
            for stale in glob.glob(os.path.join(args.tag, ".stage1_done_*")):
                if stale != marker and not _marker_launcher_alive(stale):     # (a second launch on this --tag must not take a live launch's marker)
                    try:
                        os.remove(stale)
                    except OSError:
                        pass
        if args.enc_log is None:
            train_ae(args)
        if args.only_enc:
            return
        if world > 1:
            open(marker, "w").close()                # the experiment directory + encoding exist: everyone may read them
    else:
        parallel.wait_for_file(marker, what="rank 0's auto-encoder stage")
    parallel.init(device=dist_util.dev())
    parallel.barrier()
    if rank == 0 and world > 1:
        os.remove(marker)                            # every rank is past its wait
    train_diffusion(args, rank)
    parallel.shutdown()                              # barrier + destroy_process_group: no rank exits with a live group


if __name__ == "__main__":
    from .launcher import maybe_spawn_module
    rc = maybe_spawn_module("sin3dm_amd.train")       # S3D_GPUS=N: N fresh ranks, one per GPU (this process touches none)
    if rc is not None:
        raise SystemExit(rc)
    main()
