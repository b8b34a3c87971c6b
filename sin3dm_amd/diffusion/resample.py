"""Timestep samplers of the training loop (reference: src/diffusion/resample.py).

`sample(batch_size, device)` returns (timesteps int64 [B], importance weights float32 [B]) with the reference's
numpy generator semantics (np.random.choice), so `seed_all` reproduces the same timestep stream.
"""
from __future__ import annotations

import numpy as np
import torch as th
import torch.distributed as dist


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")


class ScheduleSampler:
    # True: the draw for step k + 1 does not depend on step k's losses — TrainLoop may make it (and upload it) while step k runs
    prefetchable = False

    def weights(self):
        raise NotImplementedError

    def sample_host(self, batch_size):
        """(timesteps int64 [B], importance weights float64 [B]) as numpy arrays: the reference's draw (resample.py:33-48)."""
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        return idx, 1 / (len(p) * p[idx])

    def sample(self, batch_size, device):
        idx, weights = self.sample_host(batch_size)
        t, w = th.from_numpy(idx).long(), th.from_numpy(weights).float()
        if th.device(device).type == "cuda":
            # pinned + non_blocking: a pageable host-to-device copy would drain the GPU queue in every training step
            return t.pin_memory().to(device, non_blocking=True), w.pin_memory().to(device, non_blocking=True)
        return t.to(device), w.to(device)


class UniformSampler(ScheduleSampler):
    prefetchable = True

    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


# ---------------------------------------------------------------------------------------------------------------------------
# OUTSIDE the hot-path scope (OUT OF SCOPE): everything below this line.  It exists only
# because the reference's CLI accepts `--schedule_sampler loss-second-moment` (src/diffusion/resample.py:17-21) and
# create_named_schedule_sampler must not fail on a flag the reference takes; nothing on the measured path uses it (the default and
# every benchmark / parity test run UniformSampler), it has no kernel, and no parity claim is made for it.
# ---------------------------------------------------------------------------------------------------------------------------
class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Share this rank's (t, loss) pairs with every rank, then update the history identically everywhere."""
        ts, losses = local_ts.detach().cpu().tolist(), local_losses.detach().cpu().tolist()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, (ts, losses))
            ts = [t for part in gathered for t in part[0]]
            losses = [v for part in gathered for v in part[1]]
        self.update_with_all_losses(ts, losses)

    def update_with_all_losses(self, ts, losses):
        raise NotImplementedError


class LossSecondMomentResampler(LossAwareSampler):
    """Importance-samples timesteps proportionally to the RMS of their last `history_per_term` losses, once every
    timestep has that many (uniform until then), mixed with `uniform_prob` of the uniform distribution."""

    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        T = diffusion.num_timesteps
        self._loss_history = np.zeros([T, history_per_term], dtype=np.float64)   # ring buffer per timestep
        self._loss_counts = np.zeros([T], dtype=np.int64)                       # losses seen, saturating at the window
        self._next = np.zeros([T], dtype=np.int64)                              # ring write position

    def _warmed_up(self):
        return bool(np.all(self._loss_counts == self.history_per_term))

    def weights(self):
        T = self.diffusion.num_timesteps
        if not self._warmed_up():
            return np.ones([T], dtype=np.float64)
        rms = np.sqrt((self._loss_history ** 2).mean(axis=-1))                  # order inside the window is irrelevant
        return (1 - self.uniform_prob) * rms / rms.sum() + self.uniform_prob / T

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            if self._loss_counts[t] < self.history_per_term:
                self._loss_history[t, self._loss_counts[t]] = loss               # fill phase keeps arrival order
                self._loss_counts[t] += 1
            else:
                self._loss_history[t, self._next[t]] = loss                      # overwrite the oldest entry
                self._next[t] = (self._next[t] + 1) % self.history_per_term
