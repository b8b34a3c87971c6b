"""CPU: host-side logic of the independent-chain sampler (GaussianDiffusion.sample_loop_chains) with the per-step work replaced
by a recording stand-in — issue order, lane / stream pairing, result order, the sequential fallback — and of the noise source
(`_randn`: one generator per batch element makes a sample's draws independent of its batch)."""
import contextlib

import torch

from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
from sin3dm_amd.sample import sample_chains


class FakeModel(torch.nn.Module):
    def __init__(self, lanes=True):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.cur = 0
        if lanes:
            self.lane = self._lane

    def _lane(self, k):
        @contextlib.contextmanager
        def ctx():
            prev, self.cur = self.cur, k
            try:
                yield self
            finally:
                self.cur = prev
        return ctx()


def fake_loops(diff, log, steps):
    def loop(model, shape, noise=None, device=None, generator=None, **kw):
        for s in range(steps):
            log.append((generator, s, getattr(model, "cur", None), STREAM[0]))
            yield {"sample": torch.full(shape, float(generator) * 100 + s), "pred_xstart": None}
    diff.p_sample_loop_progressive = loop
    diff.ddim_sample_loop_progressive = loop


STREAM = [None]


def fake_streams(n):
    @contextlib.contextmanager
    def ctx(k):
        prev, STREAM[0] = STREAM[0], k
        try:
            yield
        finally:
            STREAM[0] = prev
    return [ctx_factory(ctx, k) for k in range(n)]


class ctx_factory:                                   # a re-enterable context object, like torch.cuda.stream(s)
    def __init__(self, f, k):
        self.f, self.k, self.c = f, k, None

    def __enter__(self):
        self.c = self.f(self.k)
        return self.c.__enter__()

    def __exit__(self, *a):
        return self.c.__exit__(*a)


def test_round_robin_issue_lane_stream_pairing_and_result_order():
    diff = create_gaussian_diffusion(steps=1000)
    log = []
    fake_loops(diff, log, steps=3)
    m = FakeModel()
    res = diff.sample_loop_chains(m, (1, 2), 5, chains=2, generators=[0, 1, 2, 3, 4], device="cpu", streams=fake_streams(2))
    assert [float(r.flatten()[0]) for r in res] == [2.0, 102.0, 202.0, 302.0, 402.0]        # final step of run r, ordered by run
    # chain c always works with lane c on stream c; the two chains alternate step by step
    assert all(lane == stream for _, _, lane, stream in log)
    assert [(g, s) for g, s, _, _ in log[:6]] == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2)]
    by_run = {}
    for g, s, lane, _ in log:
        by_run.setdefault(g, set()).add(lane)
    assert all(len(v) == 1 for v in by_run.values())                    # a run never changes chain
    assert by_run[0] == {0} and by_run[1] == {1} and by_run[4] in ({0}, {1})
    assert m.cur == 0 and STREAM[0] is None


def test_without_lanes_or_with_one_chain_runs_follow_each_other():
    for model, chains in ((FakeModel(lanes=False), 3), (FakeModel(), 1)):
        diff = create_gaussian_diffusion(steps=1000)
        log = []
        fake_loops(diff, log, steps=2)
        res = diff.sample_loop_chains(model, (1, 2), 3, chains=chains, generators=[0, 1, 2], device="cpu")
        assert [(g, s) for g, s, _, _ in log] == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]
        assert [float(r.flatten()[0]) for r in res] == [1.0, 101.0, 201.0]


def test_randn_per_element_generators_do_not_depend_on_the_batch():
    from sin3dm_amd.diffusion.gaussian_diffusion import GaussianDiffusion as GD
    g = lambda s: torch.Generator().manual_seed(s)
    both = GD._randn((2, 3, 4, 5), "cpu", [g(1), g(2)])
    one = GD._randn((1, 3, 4, 5), "cpu", [g(2)])
    assert both.shape == (2, 3, 4, 5) and torch.equal(both[1], one[0])
    lead = GD._randn((2, 3, 4, 5), "cpu", [g(1), g(2)], lead=7)
    lead1 = GD._randn((1, 3, 4, 5), "cpu", [g(2)], lead=7)
    assert lead.shape == (7, 2, 3, 4, 5) and lead.is_contiguous() and lead[3].is_contiguous() and torch.equal(lead[:, 1], lead1[:, 0])
    assert lead1[3].is_contiguous()
    whole = GD._randn((2, 3, 4, 5), "cpu", g(1), lead=7)
    assert whole.shape == (7, 2, 3, 4, 5) and torch.equal(whole, torch.randn((7, 2, 3, 4, 5), generator=g(1)))


def test_sample_cli_chain_count_rule(monkeypatch):
    monkeypatch.delenv("S3D_SAMPLE_CHAINS", raising=False)
    assert sample_chains(8, 1) == 3 and sample_chains(2, 1) == 2 and sample_chains(1, 1) == 1
    assert sample_chains(8, 2) == 2 and sample_chains(8, 4) == 1 and sample_chains(8, 32) == 1
    monkeypatch.setenv("S3D_SAMPLE_CHAINS", "1")
    assert sample_chains(8, 1) == 1
    monkeypatch.setenv("S3D_SAMPLE_CHAINS", "4")
    assert sample_chains(8, 8) == 4 and sample_chains(3, 1) == 3


def test_model_timesteps_for_host_known_indices_are_the_device_path_floats():
    """TrainLoop uploads the denoiser's timestep values computed on the HOST (_model_timesteps_host: the timestep_map lookup and the
    fp32 scaling of respace.py:123-128) instead of gathering them on the device in every step: bit-equal to _model_timesteps."""
    import numpy as np
    import torch
    from sin3dm_amd.diffusion.script_util import create_gaussian_diffusion
    for kw in (dict(), dict(rescale_timesteps=True), dict(timestep_respacing="100"), dict(timestep_respacing="ddim50", rescale_timesteps=True)):
        d = create_gaussian_diffusion(steps=1000, predict_xstart=True, **kw)
        idx = np.array([0, 1, d.num_timesteps // 3, d.num_timesteps - 1, 5], dtype=np.int64)
        dev_path = d._model_timesteps(torch.from_numpy(idx)).float().numpy()
        host = d._model_timesteps_host(idx)
        assert host.dtype == np.float32 and np.array_equal(host, dev_path), kw


def test_chains_fail_loudly_when_the_shared_weights_change_mid_run():
    """Chains on several lanes share the denoiser's ONE packed weight image; a repack would be ordered on one chain's stream only
    (ADVICE r5).  The loop snapshots model.parameter_stamp() and raises as soon as it moves while chains are in flight."""
    import pytest
    diff = create_gaussian_diffusion(steps=1000)
    log = []
    fake_loops(diff, log, steps=4)
    m = FakeModel()
    m.stamp = 0
    m.parameter_stamp = lambda: m.stamp
    it = diff.sample_loop_chains_progressive(m, (1, 2), 2, chains=2, generators=[1, 2], device="cpu", streams=fake_streams(2))
    next(it)
    m.stamp = 1                                        # an optimizer step / load_state_dict between two rounds
    with pytest.raises(RuntimeError, match="parameters changed"):
        next(it)
    # one chain at a time: nothing is shared, nothing is checked
    it = diff.sample_loop_chains_progressive(m, (1, 2), 2, chains=1, generators=[1, 2], device="cpu")
    next(it); m.stamp = 2; next(it)


def test_loop_sets_the_in_conv_carry_flags_by_identity_and_version():
    """GaussianDiffusion._loop (gaussian_diffusion.py:488-536 in the reference) asks the denoiser to carry the next step's in_conv
    (s3d_unet_step_film_carry): CARRY_OUT on every step but the last; CARRY_IN only when the tensor it continues from still IS the
    previous step's sample — same object, same version counter.  A consumer that edits or replaces out["sample"] between two steps
    gets a step without CARRY_IN (the library would otherwise use an in_conv of the unedited tensor)."""
    from sin3dm_amd import _lib
    diff = create_gaussian_diffusion(steps=1000, timestep_respacing="5")
    seen = []

    def fake_step(mode, model, x, t, clip, dfn, kw, fuse=False, noise=None, carry=0, **rest):
        seen.append((carry, x))
        return torch.zeros_like(x), torch.zeros_like(x), None
    diff._step = fake_step
    m = FakeModel()

    def run(action):
        seen.clear()
        for n, out in enumerate(diff.p_sample_loop_progressive(m, (1, 2, 3, 3), noise=torch.zeros(1, 2, 3, 3), device="cpu")):
            if n == 1 and action == "edit":
                out["sample"].add_(1.0)
            if n == 1 and action == "replace":
                out["sample"] = out["sample"].clone()
        return [c for c, _ in seen]

    O, I = _lib.CARRY_OUT, _lib.CARRY_IN
    assert run(None) == [O, O | I, O | I, O | I, I]
    assert run("edit") == [O, O | I, O, O | I, I]           # step 2 continues from an edited tensor: no CARRY_IN there
    assert run("replace") == [O, O | I, O, O | I, I]
