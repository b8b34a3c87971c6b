"""Training CLI with the reference's flags and on-disk layout (src/train.py) — diffusion stage.

    python -m sin3dm_amd.train --tag EXP --enc_log PATH/TO/encoding [--diff_batch_size 32 --diff_n_iters 25000 ...]
    python -m torch.distributed.run --nproc-per-node 8 -m sin3dm_amd.train --tag EXP --enc_log ... --diff_batch_size 4

Reads <enc_log>/{args.json,feat.npz} (the triplane latent written by the reference's auto-encoder stage), trains the
triplane diffusion UNet on the MI355X (sin3dm_amd/diffusion/train_util.py) and writes EXP/diffusion/{args.json,
ema_<rate>_<step>.pt, opt<step>.pt, progress.jsonl} — the files sample.py of either implementation reads.
Multi-GPU: --diff_batch_size is PER GPU (the reference's 32 = 8 x 4); gradients are averaged with one all-reduce
per step.  The auto-encoder stage (src/train.py:8-29, ShapeAutoEncoder.train) is the next tier:
without --enc_log this CLI stops with a message instead of training it.
"""
from __future__ import annotations

import os

from . import parallel
from .utils import dist_util
from .utils.common_util import seed_all
from .utils.parser_util import diffusion_log_dir, encoding_feat_path, train_args


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


def main(argv=None, confirm=input):
    rank, local, world = parallel.env_rank_world()
    args = train_args(argv, confirm=confirm if rank == 0 else (lambda _: "y"))
    seed_all(0 + rank)                               # per-rank timestep / noise streams; weights are broadcast from rank 0
    dist_util.setup_dist(local if world > 1 else args.gpu_id)
    parallel.init(device=dist_util.dev())
    if args.only_enc or args.enc_log is None:
        raise NotImplementedError(
            "the auto-encoder stage (ShapeAutoEncoder.train) is not implemented on this path yet (SURVEY.md §8f rank 3): "
            "train it with the reference and pass its folder as --enc_log")
    train_diffusion(args, rank)
    parallel.barrier()


if __name__ == "__main__":
    main()
