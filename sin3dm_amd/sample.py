"""Sampling CLI with the reference's flags and on-disk layout (src/sample.py):

    python -m sin3dm_amd.sample --tag EXP --n_samples N [--use_ddim True --timestep_respacing 100] [--resize a b c] [--vox]
    python -m torch.distributed.run --nproc-per-node 8 -m sin3dm_amd.sample --tag EXP --n_samples 64 ...

Reads EXP/encoding/{args.json,feat.npz}, EXP/diffusion/{args.json,ema_<rate>_<iters>.pt} and the AE checkpoint
EXP/encoding/ckpt_final.pth written by the reference's train.py; writes EXP/<output>/NNN/feat.npz (and
r<reso>_voxel.npz with --vox).  Multi-GPU: sample indices are striped over the ranks (sin3dm_amd/parallel.py).
Without --vox the decode stage extracts the iso-surface on the device (marching cubes) and writes a vertex-coloured
object.obj; the reference's UV atlas / baked texture (xatlas, nvdiffrast) is out of scope.
"""
from __future__ import annotations

import os

import torch

from . import parallel
from .utils import dist_util
from .utils.parser_util import (diffusion_model_path, encoding_feat_path, encoding_log_dir, sample_args)


def sample_diffusion(args, rank=0, world=1, base_seed=1000):
    """Reference: src/sample.py:6-48, with per-sample seeds and rank striping."""
    from .diffusion.script_util import create_model_and_diffusion_from_args
    from .utils.triplane_util import decompose_featmaps, load_triplane_data, save_triplane_data

    dev = dist_util.dev()
    src_data, sizes = load_triplane_data(encoding_feat_path(args.tag), device=dev)
    model, diffusion = create_model_and_diffusion_from_args(args)
    model.load_state_dict(dist_util.load_state_dict(diffusion_model_path(args.tag, args.ema_rate, args.diff_n_iters),
                                                    map_location="cpu"))
    model.to(dev).eval()
    sample_fn = diffusion.ddim_sample_loop if args.use_ddim else diffusion.p_sample_loop

    result_dir = os.path.join(args.tag, args.output)
    os.makedirs(result_dir, exist_ok=True)
    C = src_data.shape[0]
    H, W, D = (int(s * r) for s, r in zip(sizes, args.resize))
    if rank == 0:
        print("H, W, D:", H, W, D)

    paths = []
    mine = parallel.shard_indices(args.n_samples, rank, world)
    groups = list(parallel.batches(mine, args.diff_batch_size))
    # x_T and every step's eps of sample i come from ITS generator (seed = base + i): a sample does not depend on the batch it
    # is in, the chain it runs on or the number of GPUs
    gens = [[torch.Generator(device=dev).manual_seed(parallel.sample_seed(base_seed, i)) for i in idx] for idx in groups]
    kw = {"H": H, "W": W, "D": D}
    nch = sample_chains(len(groups), args.diff_batch_size)
    full = [g for g in range(len(groups)) if len(groups[g]) == args.diff_batch_size]
    outs = {}
    if nch > 1 and len(full) > 1:
        # several batches of equal shape on this GPU: up to `nch` of them in flight as independent chains (one stream + one
        # workspace lane each) instead of one after the other (src/sample.py:33-47)
        res = diffusion.sample_loop_chains(model, [args.diff_batch_size, C, H + D, W + D], len(full), chains=nch,
                                           ddim=bool(args.use_ddim), generators=[gens[g] for g in full], device=dev, model_kwargs=kw)
        outs = dict(zip(full, res))
    for g, idx in enumerate(groups):
        samples = outs[g] if g in outs else sample_fn(model, [len(idx), C, H + D, W + D], progress=rank == 0, model_kwargs=kw,
                                                      generator=gens[g])
        xy, xz, yz = (t.detach().cpu().numpy() for t in decompose_featmaps(samples, (H, W, D)))
        for j, i in enumerate(idx):
            path = os.path.join(result_dir, f"{i:03d}", "feat.npz")
            save_triplane_data(path, xy[j], xz[j], yz[j])
            paths.append(path)
    return paths


def sample_chains(n_batches, batch):
    """How many of a rank's batches run at once as independent chains (GaussianDiffusion.sample_loop_chains).  S3D_SAMPLE_CHAINS
    sets it (1 = off).  Default: chains pay where a step's launches are a round or two of blocks each — measured on one MI355X
    at 128^3 (profiles/r05_two_chains.txt) — so three at batch 1, two at batch 2, none from batch 4 on."""
    env = os.environ.get("S3D_SAMPLE_CHAINS")
    if env:
        return max(1, min(int(env), n_batches))
    return max(1, min(3 if batch == 1 else 2 if batch == 2 else 1, n_batches))


def decode(args, paths):
    """Reference: src/sample.py:51-78 (voxel branch; mesh export is out of scope)."""
    from .encoding.model import ShapeAutoEncoder
    from .utils.triplane_util import load_triplane_data

    ae = ShapeAutoEncoder(encoding_log_dir(args.tag), args, device=dist_util.dev())
    ae.load_ckpt("final")
    for path in paths:
        fm = [f.unsqueeze(0) for f in load_triplane_data(path, device=dist_util.dev(), compose=False)]
        if args.vox:
            ae.decode_voxel(os.path.dirname(path), fm, args.reso)
        else:
            # iso-surface on the device, vertex-coloured object.obj (the UV-atlas / baked-texture export of the reference
            # needs xatlas + nvdiffrast and stays out of scope)
            ae.decode_mesh(os.path.dirname(path), fm, args.reso)


def main(argv=None):
    args = sample_args(argv)
    rank, local, world = parallel.env_rank_world()
    dist_util.setup_dist(local if world > 1 else args.gpu_id)
    parallel.init(device=dist_util.dev())
    paths = sample_diffusion(args, rank, world)
    decode(args, paths)
    all_paths = sorted(p for ps in parallel.gather_objects(paths) for p in ps)
    if rank == 0:
        print(f"wrote {len(all_paths)} samples under {os.path.join(args.tag, args.output)}")
    parallel.shutdown()                              # barrier + destroy_process_group
    return all_paths


if __name__ == "__main__":
    from .launcher import maybe_spawn_module
    rc = maybe_spawn_module("sin3dm_amd.sample")      # S3D_GPUS=N: N fresh ranks, one per GPU (this process touches none)
    if rc is not None:
        raise SystemExit(rc)
    main()
