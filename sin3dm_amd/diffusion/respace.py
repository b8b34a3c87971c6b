"""Timestep respacing (reference: src/diffusion/respace.py)."""
from __future__ import annotations

import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion, HostTimesteps, host_values_of


def space_timesteps(num_timesteps, section_counts):
    """Which original steps to keep (reference: src/diffusion/respace.py:7-60).

    `section_counts`: "ddimN" (fixed integer stride giving exactly N steps), a comma-separated string or a
    list of per-section counts; each equally sized section is covered by evenly spaced, rounded indices
    that include its first and last step.
    """
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                steps = range(0, num_timesteps, stride)
                if len(steps) == want:
                    return set(steps)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    n_sec = len(section_counts)
    base, extra = divmod(num_timesteps, n_sec)
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process over a subset of the base steps (reference: src/diffusion/respace.py:63-113).

    New betas are 1 - abar_i / abar_(previous kept) so the kept marginals are unchanged; the model still
    sees ORIGINAL step indices through `_WrappedModel`.
    """

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        self.timestep_map = [i for i in range(self.original_num_steps) if i in self.use_timesteps]
        last, new_betas = 1.0, []
        for i in self.timestep_map:
            new_betas.append(1 - base.alphas_cumprod[i] / last)
            last = base.alphas_cumprod[i]
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        # the device copy of the timestep map is shared by all wrappers of this diffusion: building it per step would be
        # a pageable host-to-device copy, i.e. a host/GPU synchronisation in every denoising step
        if not hasattr(self, "_map_cache"):
            self._map_cache = {}
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps, self._map_cache)

    def _prepare_loop(self, model, batch, device):
        self._wrap_model(model).prepare_loop(self.num_timesteps, batch, device)

    def _step(self, mode, model, *args, **kwargs):
        return super()._step(mode, self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        """Reference :93-96: training conditions the denoiser on ORIGINAL step indices too."""
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def _scale_timesteps(self, t):
        return t          # scaling happens inside the wrapped model

    def _model_timesteps(self, t):
        """What `_WrappedModel` feeds the denoiser, for the graph-free training path (training_losses_and_grads)."""
        key = str(t.device)
        if not hasattr(self, "_map_cache"):
            self._map_cache = {}
        m = self._map_cache.get(key)
        if m is None:
            m = self._map_cache[key] = th.tensor(self.timestep_map, device=t.device, dtype=th.float32)
        ts = m[t]
        return ts * (1000.0 / self.original_num_steps) if self.rescale_timesteps else ts

    def _model_timesteps_host(self, idx):
        """The same floats for indices known on the host (numpy): the map lookup and the fp32 x fp32 scaling of _mapped."""
        v = np.asarray(self.timestep_map, dtype=np.float32)[np.asarray(idx)]
        return v * np.float32(1000.0 / self.original_num_steps) if self.rescale_timesteps else v


class _WrappedModel:
    """Maps respaced indices back to original ones before calling the denoiser (:116-128)."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps, map_cache=None):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self._maps = map_cache if map_cache is not None else {}

    def parameters(self):
        return self.model.parameters()

    def _mapped(self, v):
        # the reference multiplies an fp32 tensor by the python scalar (respace.py:124-128): fp32 x fp32(scalar), rounded
        # once — done the same way here so that the host-known path feeds the denoiser the identical float
        m = np.float32(self.timestep_map[v])
        return float(m * np.float32(1000.0 / self.original_num_steps)) if self.rescale_timesteps else float(m)

    def prepare_loop(self, num_timesteps, batch, device):
        """Called by the sampling loops before their first step: the mapped timesteps of the whole schedule as ONE device
        tensor (row i = the batch's timesteps of loop index i) and, if the denoiser offers it, its per-timestep tables in
        one batched launch — instead of a host-to-device copy and three small launches on the first visit of every index."""
        key = str(th.empty(0, device=device).device)      # 'cuda' -> 'cuda:0': the key __call__ derives from ts.device
        sched = ("schedule", key, batch)
        entry = self._maps.get(sched)
        if entry is None:
            vals = [self._mapped(i) for i in range(num_timesteps)]
            col = th.tensor(vals, device=device, dtype=th.float32)
            rows = col[:, None].expand(-1, batch).contiguous()
            for i, v in enumerate(vals):
                hv = (v,) * batch
                self._maps[(key, hv)] = HostTimesteps(rows[i], hv)
            entry = self._maps[sched] = (vals, col, rows)
        # the timestep tables themselves are recomputed for every loop (= every sample): batched, not carried over
        prep = getattr(self.model, "prepare_timesteps", None)
        if prep is not None:
            prep(entry[0], entry[1])

    def __getattr__(self, name):          # forward_train / backward_flat / flat_parameters of the wrapped denoiser
        return getattr(self.__dict__["model"], name)

    def _mapped_host(self, ts):
        """HostTimesteps of the ORIGINAL step values for host-known respaced indices (one cached device tensor per batch of values)."""
        mapped = tuple(self._mapped(v) for v in host_values_of(ts))
        ck = (str(ts.device), mapped)
        new_ts = self._maps.get(ck)
        if new_ts is None:
            new_ts = self._maps[ck] = HostTimesteps(th.tensor(mapped, device=ts.device, dtype=th.float32), mapped)
        return new_ts

    @property
    def denoise_step(self):
        """The denoiser's fused step (UNet + sampler update in the output head), with the same timestep mapping as __call__;
        None when the denoiser has none (GaussianDiffusion._step then runs the two-kernel path)."""
        inner = getattr(self.model, "denoise_step", None)
        if inner is None:
            return None
        return lambda x, ts, step, **kw: inner(x, self._mapped_host(ts), step, **kw)

    def __call__(self, x, ts, **kwargs):
        # the map is kept in float32: the denoiser embeds float timesteps anyway (nn.py:113), so one gather replaces the
        # reference's integer gather + cast (exact: indices < 2^24)
        key = str(ts.device)
        hv = host_values_of(ts)
        if hv is not None:
            # the loop told us the values: map them on the host, one cached device tensor per distinct batch of values
            return self.model(x, self._mapped_host(ts), **kwargs)
        m = self._maps.get(key)
        if m is None:
            m = self._maps[key] = th.tensor(self.timestep_map, device=ts.device, dtype=th.float32)
        new_ts = m[ts]
        if self.rescale_timesteps:
            new_ts = new_ts * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
