"""Independent sample chains (GaussianDiffusion.sample_loop_chains, TriplaneUNetModelSmall.lane, the loops' `generator`):
several runs of the reference's p_sample_loop / ddim_sample_loop (src/diffusion/gaussian_diffusion.py:442-536, 640-734; the
caller loops batch after batch, src/sample.py:33-47) in flight at once, each on its own HIP stream and workspace lane of ONE
denoiser handle.  Bar: every run's result is BIT-IDENTICAL to the same run alone on the default stream and lane."""
import os

import numpy as np
import pytest
import torch

from sin3dm_amd import testing as T
from test_hip_parity import cu, dev, make_diffusion, make_model

pytestmark = pytest.mark.gpu


def gens(seeds):
    return [torch.Generator(device=dev()).manual_seed(int(s)) for s in seeds]


@pytest.mark.parametrize("ddim,resp,B,mc,hwd", [(False, "20", 1, 32, (10, 14, 6)), (True, "10", 2, 64, (12, 8, 10)), (False, "10", 1, 128, (32, 32, 32))])
def test_chains_equal_single_runs_bit_for_bit(ddim, resp, B, mc, hwd):
    H, W, D = hwd
    kw = dict(H=H, W=W, D=D)
    model = make_model(mc)
    diff = make_diffusion(resp)
    shape = (B, 12, H + D, W + D)
    n = 5
    seeds = [[100 * r + b for b in range(B)] for r in range(n)]
    loop = diff.ddim_sample_loop if ddim else diff.p_sample_loop
    alone = [loop(model, shape, model_kwargs=kw, generator=gens(s)) for s in seeds]
    assert not torch.equal(alone[0], alone[1]) and all(torch.isfinite(a).all() for a in alone)
    for chains in (2, 3):
        got = diff.sample_loop_chains(model, shape, n, chains=chains, ddim=ddim, generators=[gens(s) for s in seeds], model_kwargs=kw)
        torch.cuda.synchronize()
        assert len(got) == n
        for r in range(n):
            assert torch.equal(got[r], alone[r]), (chains, r)
    # chains = 1 is the plain sequence of runs
    got = diff.sample_loop_chains(model, shape, 2, chains=1, ddim=ddim, generators=[gens(s) for s in seeds[:2]], model_kwargs=kw)
    assert torch.equal(got[0], alone[0]) and torch.equal(got[1], alone[1])
    assert model._lane == 0                                          # the default lane is selected again afterwards


def test_per_sample_generators_make_a_sample_independent_of_its_batch():
    """One generator per batch element: sample b's x_T and every eps come from generator b in calls whose size does not depend on
    B, and the kernels' results do not depend on the batch — so a sample is the same alone, in a batch of 3 and on a chain."""
    H, W, D = 10, 14, 6
    kw = dict(H=H, W=W, D=D)
    model = make_model(32)
    diff = make_diffusion("20")
    seeds = (7, 8, 9)
    batch = diff.p_sample_loop(model, (3, 12, H + D, W + D), model_kwargs=kw, generator=gens(seeds))
    for b, s in enumerate(seeds):
        one = diff.p_sample_loop(model, (1, 12, H + D, W + D), model_kwargs=kw, generator=gens([s]))
        assert torch.equal(one[0], batch[b]), b
    # a single torch.Generator is one call for the whole batch: reproducible, but another stream of values
    a = diff.p_sample_loop(model, (3, 12, H + D, W + D), model_kwargs=kw, generator=gens([7])[0])
    b = diff.p_sample_loop(model, (3, 12, H + D, W + D), model_kwargs=kw, generator=gens([7])[0])
    assert torch.equal(a, b) and not torch.equal(a, batch)


def test_default_noise_path_equals_two_kernel_path_with_the_same_eps():
    """the loops' DEFAULT path (noise_fn None: eps drawn for many steps by one randn, the fused denoise step) against
    the forward + sampler-kernel pair of the single-step API fed the very same eps — drawn here the way the loop draws it."""
    H, W, D = 10, 14, 6
    kw = dict(H=H, W=W, D=D)
    model = make_model(32)
    diff = make_diffusion("20")
    shape = (2, 12, H + D, W + D)
    g = gens([42])[0]
    got = diff.p_sample_loop(model, shape, model_kwargs=kw, generator=g)
    g = gens([42])[0]
    x = torch.randn(*shape, device=dev(), generator=g)
    chunk = max(1, min(diff.num_timesteps, diff._NOISE_AHEAD_BYTES // (4 * int(np.prod(shape)))))
    assert chunk >= diff.num_timesteps                               # (tiny planes: one draw covers the run)
    eps = torch.randn((diff.num_timesteps,) + shape, device=dev(), generator=g)
    it = iter(eps)
    diff.noise_fn = lambda z: next(it)
    with torch.no_grad():
        for i in range(diff.num_timesteps - 1, -1, -1):
            x = diff.p_sample(model, x, torch.full((2,), i, device=dev(), dtype=torch.int64), model_kwargs=kw)["sample"]
    diff.noise_fn = None
    assert torch.equal(got, x)
    # chunked draws: a small look-ahead gives another (documented) stream; per-step draws (= 0) are randn_like per step
    diff._NOISE_AHEAD_BYTES = 0
    per_step = diff.p_sample_loop(model, shape, model_kwargs=kw, generator=gens([42])[0])
    g = gens([42])[0]
    x = torch.randn(*shape, device=dev(), generator=g)
    diff.noise_fn = lambda z: torch.randn(z.shape, device=z.device, generator=g)
    with torch.no_grad():
        for i in range(diff.num_timesteps - 1, -1, -1):
            x = diff.p_sample(model, x, torch.full((2,), i, device=dev(), dtype=torch.int64), model_kwargs=kw)["sample"]
    assert torch.equal(per_step, x)


def test_lanes_are_independent_workspaces_of_one_handle():
    """Forwards of different shapes on different lanes / streams, interleaved: each equals the same forward on lane 0."""
    model = make_model(32)
    cases = [((10, 14, 6), 2, 71), ((12, 8, 10), 1, 72), ((9, 13, 7), 3, 73)]
    want, xs, ts = [], [], []
    for (H, W, D), B, seed in cases:
        x = cu(T.synthetic_noise((B, 12, H + D, W + D), seed))
        t = torch.full((B,), 17.0 + seed, device=dev())
        with torch.no_grad():
            want.append(model(x, t, H=H, W=W, D=D))
        xs.append(x); ts.append(t)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in cases]
    got = [None] * len(cases)
    for rep in range(3):
        for k, ((H, W, D), B, seed) in enumerate(cases):
            with torch.cuda.stream(streams[k]), model.lane(k + 1), torch.no_grad():
                got[k] = model(xs[k], ts[k], H=H, W=W, D=D)
    torch.cuda.synchronize()
    for k in range(len(cases)):
        assert torch.equal(got[k], want[k]), k
    assert model._lane == 0
    with pytest.raises(AssertionError):
        with model.lane(99):
            pass
    # training stays on lane 0
    from sin3dm_amd import _lib
    H, W, D = cases[0][0]
    with model.lane(1):
        with pytest.raises(AssertionError):
            model.forward_train(xs[0], ts[0], H, W, D)


def test_sample_cli_chains_write_the_same_files(tmp_path):
    """python -m sin3dm_amd.sample on a synthetic experiment: with chains (default at --diff_batch_size 1) and without
    (S3D_SAMPLE_CHAINS=1), and with the samples batched — the feat.npz of every sample is identical."""
    from test_cli_gpu import make_experiment
    from sin3dm_amd import sample
    from sin3dm_amd.utils.parser_util import sample_args
    tag = make_experiment(str(tmp_path), hwd=(12, 16, 10), mc=32)
    outs = {}
    for name, env, bs in (("chains", None, 1), ("serial", "1", 1), ("batched", None, 4)):
        if env is None:
            os.environ.pop("S3D_SAMPLE_CHAINS", None)
        else:
            os.environ["S3D_SAMPLE_CHAINS"] = env
        try:
            args = sample_args(["--tag", tag, "--n_samples", "4", "--output", name, "--use_ddim", "True", "--timestep_respacing", "10"])
            args.diff_batch_size = bs                  # (the reference's sample.py reads it from the experiment's diffusion args, src/sample.py:33)
            paths = sample.sample_diffusion(args)
        finally:
            os.environ.pop("S3D_SAMPLE_CHAINS", None)
        outs[name] = [dict(np.load(p)) for p in sorted(paths)]
        assert len(outs[name]) == 4
    for name in ("serial", "batched"):
        for a, b in zip(outs["chains"], outs[name]):
            assert sorted(a) == sorted(b)
            for k in a:
                assert np.array_equal(a[k], b[k]), (name, k)


def test_chains_at_the_scored_size_are_bit_identical():
    """BASELINE configs[1]'s model and planes (128-ch, 128^3; respaced to 100 ancestral steps to keep the test short): three samples
    as three chains, as two chains and alone — the same bits; and the whole run's state stays finite with an exactly zero corner."""
    H = W = D = 128
    kw = dict(H=H, W=W, D=D)
    model = make_model(128)
    diff = make_diffusion("100")
    shape = (1, 12, H + D, W + D)
    seeds = [[1000], [1001], [1002]]
    alone = [diff.p_sample_loop(model, shape, model_kwargs=kw, generator=gens(s)) for s in seeds]
    for chains in (3, 2):
        got = diff.sample_loop_chains(model, shape, 3, chains=chains, generators=[gens(s) for s in seeds], model_kwargs=kw)
        torch.cuda.synchronize()
        for r in range(3):
            assert torch.equal(got[r], alone[r]), (chains, r)
    assert all(torch.isfinite(a).all() and float(a[..., H:, W:].abs().max()) == 0.0 for a in alone)
    assert not torch.equal(alone[0], alone[1])
