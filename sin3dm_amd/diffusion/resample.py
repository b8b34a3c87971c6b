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
    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        weights = 1 / (len(p) * p[idx])
        t, w = th.from_numpy(idx).long(), th.from_numpy(weights).float()
        if th.device(device).type == "cuda":
            # pinned + non_blocking: a pageable host-to-device copy would drain the GPU queue in every training step
            return t.pin_memory().to(device, non_blocking=True), w.pin_memory().to(device, non_blocking=True)
        return t.to(device), w.to(device)


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


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
    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=np.int64)

    def weights(self):
        if not self._warmed_up():
            return np.ones([self.diffusion.num_timesteps], dtype=np.float64)
        w = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.uniform_prob
        w += self.uniform_prob / len(w)
        return w

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            if self._loss_counts[t] == self.history_per_term:
                self._loss_history[t, :-1] = self._loss_history[t, 1:]       # drop the oldest term
                self._loss_history[t, -1] = loss
            else:
                self._loss_history[t, self._loss_counts[t]] = loss
                self._loss_counts[t] += 1

    def _warmed_up(self):
        return bool((self._loss_counts == self.history_per_term).all())
