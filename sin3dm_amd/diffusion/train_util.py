"""Diffusion training loop on the MI355X (reference: src/diffusion/train_util.py, fp32 path).

Same constructor, step order and checkpoint files as the reference's TrainLoop —
    run_step = forward_backward -> optimizer step -> EMA update -> linear lr anneal          (:163-172)
    save     = ema_<rate>_<step:06d>.pt (state_dict by name) + opt<step:06d>.pt                (:258-279)
— but built for one process per GPU: the model's parameters are views of one flat device vector
(TriplaneUNetModelSmall.flat_parameters), so that
  * forward + loss + backward run without an autograd graph (GaussianDiffusion.training_losses_and_grads),
  * data-parallel training is ONE RCCL all-reduce of the flat gradient vector per step (28 MB at 64 channels),
  * AdamW + all EMA copies are ONE kernel over the flat vectors (FlatAdamW -> s3d_train_adamw_ema).
The reference is single-process (its DDP is commented out, :8-9, :98-99); with world_size 1 the arithmetic is the
reference's.  tensorboard / matplotlib logging is out of scope: scalars go to stdout and progress.jsonl.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time

import numpy as np
import torch as th
import torch.distributed as dist

from .. import _lib, parallel
from .resample import LossAwareSampler, UniformSampler


class FlatAdamW:
    """torch.optim.AdamW (betas 0.9/0.999, eps 1e-8, decoupled weight decay) + update_ema (src/diffusion/nn.py:55-65)
    on the model's flat parameter vector; `ema` holds one flat copy per rate, initialised to the parameters (:92-95)."""

    def __init__(self, model, lr, weight_decay=0.0, ema_rates=(), betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        flat = model.flat_parameters
        self.exp_avg = th.zeros_like(flat)
        self.exp_avg_sq = th.zeros_like(flat)
        self.ema_rates = [float(r) for r in ema_rates]
        self.ema = [flat.detach().clone() for _ in self.ema_rates]
        self.steps = 0

    def step(self, flat_grad):
        flat = self.model.flat_parameters
        assert flat_grad.shape == flat.shape and flat_grad.is_cuda
        self.steps += 1
        n = len(self.ema)
        ema_ptrs = (C.c_void_p * max(n, 1))(*[C.c_void_p(e.data_ptr()) for e in self.ema])
        rates = (C.c_float * max(n, 1))(*self.ema_rates)
        with th.cuda.device(flat.device):
            _lib.check(_lib.load().s3d_train_adamw_ema(
                _lib.ptr(flat), _lib.ptr(flat_grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), ema_ptrs, rates, n,
                flat.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps,
                _lib.stream_ptr()))
        self.model.mark_parameters_changed()

    # torch.optim.AdamW-compatible state (parameter order = model.parameters()), so opt*.pt files interchange
    def state_dict(self):
        names = [n for n, _ in self.model.named_parameters()]
        m, v = self.model.split_flat(self.exp_avg), self.model.split_flat(self.exp_avg_sq)
        state = {i: {"step": th.tensor(float(self.steps)), "exp_avg": m[n].clone(), "exp_avg_sq": v[n].clone()}
                 for i, n in enumerate(names)} if self.steps else {}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        names = [n for n, _ in self.model.named_parameters()]
        m, v = self.model.split_flat(self.exp_avg), self.model.split_flat(self.exp_avg_sq)
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            m[n].copy_(st["exp_avg"]); v[n].copy_(st["exp_avg_sq"])
            self.steps = int(float(st["step"]))
        g = sd["param_groups"][0]
        self.lr, self.weight_decay, self.eps, self.betas = float(g["lr"]), float(g["weight_decay"]), float(g["eps"]), tuple(g["betas"])


class TrainLoop:
    def __init__(self, *, model, diffusion, data, batch_size, microbatch, lr, ema_rate, log_interval, save_interval,
                 resume_checkpoint, use_fp16=False, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.0,
                 lr_anneal_steps=0, log_dir=None, optimizer=None):
        """optimizer: an object with .step(flat_grad) and .lr working on model.flat_parameters (default: FlatAdamW, the fused
        AdamW + EMA kernel; the CPU tests of the multi-rank control flow hand in a plain one)."""
        if use_fp16:
            raise NotImplementedError("use_fp16 is not runnable in the reference (see TriplaneUNetModelSmall) and not implemented")
        self.model, self.diffusion, self.data = model, diffusion, data
        self.batch_size = batch_size
        self.microbatch = microbatch if microbatch > 0 else batch_size
        assert self.microbatch == self.batch_size, "the reference eliminated the microbatch feature (:211-212)"
        self.lr = lr
        self.ema_rate = [ema_rate] if isinstance(ema_rate, float) else [float(x) for x in str(ema_rate).split(",")]
        self.log_interval, self.save_interval = log_interval, save_interval
        self.resume_checkpoint = resume_checkpoint
        self.schedule_sampler = schedule_sampler or UniformSampler(diffusion)
        self.weight_decay, self.lr_anneal_steps = weight_decay, lr_anneal_steps
        self.log_dir = log_dir
        self.step = 0
        self.resume_step = 0
        self.dist_on = parallel.active()             # a process group exists (any world size, a forced world-size-1 group included)
        self.world = dist.get_world_size() if self.dist_on else 1
        self.rank = dist.get_rank() if self.dist_on else 0
        self.global_batch = self.batch_size * self.world
        self._kvs = {}
        self._pending_losses = []
        self._t_last = None

        if resume_checkpoint:
            self.resume_step = parse_resume_step_from_filename(resume_checkpoint)
            self.model.load_state_dict(th.load(resume_checkpoint, map_location="cpu"))
        if self.dist_on:                             # identical start on every rank (the role of sync_params)
            parallel.broadcast_flat_(self.model.flat_parameters, src=0)
            self.model.mark_parameters_changed()
        self.opt = optimizer if optimizer is not None else FlatAdamW(model, lr=self.lr, weight_decay=self.weight_decay, ema_rates=self.ema_rate)
        if self.resume_step:
            self._load_optimizer_and_ema()
        self._grad = th.empty_like(self.model.flat_parameters)
        # Gradient exchange overlapped with the backward pass: OFF by default.  The communication-stream / event ordering runs
        # on hardware at world size 1 (tests/test_rccl_world1.py: real RCCL calls behind the real backward marks) but has never
        # run on more than one GPU (no multi-GPU node was available to this build), and from three ranks on the cut vector is
        # reduced in a different order than the whole one (parallel.average_flat_groups_): S3D_OVERLAP_ALLREDUCE=1 opts in
        # until a multi-GPU RCCL run has compared both.
        self.overlap_allreduce = os.environ.get("S3D_OVERLAP_ALLREDUCE", "0") == "1"
        self._marks = self._comm = self._groups = self._staging = None
        # Step inputs drawn AHEAD (S3D_PREFETCH_INPUTS=0: at the start of their step, as the reference): see _draw_inputs
        self.prefetch_inputs = os.environ.get("S3D_PREFETCH_INPUTS", "1") != "0"
        self._next_inputs = None
        self._stage = None

    # ------------------------------------------------------------------ resume
    def _load_optimizer_and_ema(self):
        base = os.path.dirname(self.resume_checkpoint)
        opt_path = os.path.join(base, f"opt{self.resume_step:06}.pt")
        if os.path.exists(opt_path):
            self.opt.load_state_dict(th.load(opt_path, map_location=self.model.flat_parameters.device))
        for k, rate in enumerate(self.ema_rate):
            path = os.path.join(base, f"ema_{rate}_{self.resume_step:06d}.pt")
            if os.path.exists(path):
                sd = th.load(path, map_location="cpu")
                for name, view in self.model.split_flat(self.opt.ema[k]).items():
                    view.copy_(sd[name])

    # ------------------------------------------------------------------ loop
    def run_loop(self):
        while not self.lr_anneal_steps or self.step + self.resume_step < self.lr_anneal_steps:
            batch, cond = next(self.data)
            self.run_step(batch, cond)
            if self.step % self.log_interval == 0:
                self.dumpkvs()
            if self.step % self.save_interval == 0 and self.step > 0:
                self.save()
                if os.environ.get("DIFFUSION_TRAINING_TEST", "") and self.step > 0:
                    return
            self.step += 1
        if (self.step - 1) % self.save_interval != 0:
            self.save()

    def run_step(self, batch, cond):
        self.forward_backward(batch, cond)
        self.opt.step(self._grad)
        self._anneal_lr()
        self.log_step()

    # ------------------------------------------------------------------ step inputs
    _STAGE_RING = 4

    def _draw_inputs(self, shape, dev):
        """One step's random inputs, with as little on the GPU's critical path as possible: timesteps (int64), the model's
        timestep values (float32: the timestep_map lookup + scaling done on the host) and the importance weights travel in ONE
        pinned staging row and one asynchronous copy — instead of two copies, a gather and a cast launched at the start of the
        step — and the noise is drawn with them.  Same values as the reference order (np.random for the timesteps, the device
        generator for the noise: each generator is still consumed once per step, in step order)."""
        B = int(shape[0])
        idx, w = self.schedule_sampler.sample_host(B)
        if self._stage is None or self._stage["B"] != B or self._stage["dev"] != dev:
            host = th.empty((self._STAGE_RING, 16 * B), dtype=th.uint8).pin_memory()
            self._stage = {"B": B, "dev": dev, "host": host, "dev_rows": th.empty((self._STAGE_RING, 16 * B), dtype=th.uint8, device=dev),
                           "events": [None] * self._STAGE_RING, "k": 0}
        st = self._stage
        k = st["k"]; st["k"] = (k + 1) % self._STAGE_RING
        if st["events"][k] is not None:
            st["events"][k].synchronize()                 # (the copy of four draws ago: long complete)
        row = st["host"][k]
        row[:8 * B].view(th.int64).copy_(th.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)))
        row[8 * B:12 * B].view(th.float32).copy_(th.from_numpy(np.ascontiguousarray(self.diffusion._model_timesteps_host(idx), dtype=np.float32)))
        row[12 * B:].view(th.float32).copy_(th.from_numpy(np.ascontiguousarray(w, dtype=np.float32)))
        drow = st["dev_rows"][k]
        drow.copy_(row, non_blocking=True)
        ev = th.cuda.Event(); ev.record(); st["events"][k] = ev
        noise = th.randn(tuple(shape), device=dev, dtype=th.float32)
        return {"shape": tuple(shape), "t": drow[:8 * B].view(th.int64), "t_model": drow[8 * B:12 * B].view(th.float32),
                "weights": drow[12 * B:].view(th.float32), "noise": noise}

    def forward_backward(self, batch, cond):
        dev = self.model.flat_parameters.device
        micro = batch.to(dev)
        ahead = (self.prefetch_inputs and micro.is_cuda and getattr(self.schedule_sampler, "prefetchable", False)
                 and hasattr(self.diffusion, "_model_timesteps_host"))
        extra = {}
        if ahead:
            inp = self._next_inputs
            if inp is None or inp["shape"] != tuple(micro.shape):
                inp = self._draw_inputs(micro.shape, dev)
            self._next_inputs = None
            t, weights = inp["t"], inp["weights"]

            def draw_next():
                # the next step's inputs, enqueued between this step's forward and backward pass: the host is a whole forward pass
                # ahead of the GPU there, so the two small launches (one copy, one randn) open no gap — at the end of the step they sat
                # between the last weight-gradient reduction and the optimizer kernel, at its start (round 5) around four torch
                # kernels with 114 us of launch gaps (profiles/r05_train_timeline.txt)
                self._next_inputs = self._draw_inputs(micro.shape, dev)
            extra = {"noise": inp["noise"], "t_model": inp["t_model"], "after_forward": draw_next}
        else:
            t, weights = self.schedule_sampler.sample(micro.shape[0], dev)
        if self.dist_on and self.overlap_allreduce:
            # loss = mean over the GLOBAL batch.  The backward pass fills the flat gradient from the output blocks towards
            # the input; the groups that are final early are all-reduced on a communication stream while the rest of it
            # runs (RCCL over xGMI on GPUs), joined before the optimizer step.  Bit-identical to the single all-reduce at
            # world size 2 (tests/test_parallel.py); on hardware it has run at world size 1 only (tests/test_rccl_world1.py).
            if self._groups is None:
                self._groups = self.model.grad_ready_groups()
                self._staging = parallel.GroupStaging(self._grad, self._groups)       # allocated once, not per step
                if self._grad.is_cuda:
                    self._marks = [th.cuda.Event(), th.cuda.Event()]
                    self._comm = th.cuda.Stream(device=dev)
            losses, grad = self.diffusion.training_losses_and_grads(self.model, micro, t, weights, cond, grad_out=self._grad,
                                                                    grad_marks=self._marks, **extra)
            parallel.average_flat_groups_(grad, self._groups, self._marks, self._comm, staging=self._staging)
        else:
            losses, grad = self.diffusion.training_losses_and_grads(self.model, micro, t, weights, cond, grad_out=self._grad, **extra)
            parallel.average_flat_(grad)                  # one all-reduce of the whole flat vector per step
        if ahead and self._next_inputs is None:           # (a diffusion object without the hook)
            self._next_inputs = self._draw_inputs(micro.shape, dev)
        if isinstance(self.schedule_sampler, LossAwareSampler):
            self.schedule_sampler.update_with_local_losses(t, losses["loss"].detach())
        if self.step % 10 == 0:
            # the values stay on the device until they are printed (dumpkvs): reading them here would drain the GPU queue
            self._pending_losses.append((t, weights, losses))

    def _anneal_lr(self):
        if not self.lr_anneal_steps:
            return
        frac_done = (self.step + self.resume_step) / self.lr_anneal_steps
        self.opt.lr = self.lr * (1 - frac_done)

    # ------------------------------------------------------------------ logging (stdout + progress.jsonl)
    def logkv(self, k, v):
        self._kvs[k] = v

    def log_step(self):
        self.logkv("step", self.step + self.resume_step)
        self.logkv("samples", (self.step + self.resume_step + 1) * self.global_batch)
        self.logkv("lr", self.opt.lr)
        if self.step % self.log_interval == 0:            # the steps whose values run_loop dumps
            # grad_norm / param_norm of MixedPrecisionTrainer._compute_norms (fp16_util.py:217-225): two reductions over
            # the flat vectors instead of 2 x 138 small ones (and only when they are going to be printed: .item() syncs)
            self.logkv("grad_norm", float(self._grad.norm()))
            self.logkv("param_norm", float(self.model.flat_parameters.norm()))

    def log_loss_dict(self, ts, losses):
        for key, values in losses.items():
            vals = values.detach().float().cpu().numpy()
            self.logkv(key, float(vals.mean()))
            for sub_t, sub_loss in zip(ts.cpu().numpy(), vals):
                quartile = int(4 * sub_t / self.diffusion.num_timesteps)
                self.logkv(f"{key}_q{quartile}", float(sub_loss))

    def dumpkvs(self):
        for t, weights, losses in self._pending_losses:           # (in step order: a later step's value of a key replaces an earlier one's, as before)
            self.log_loss_dict(t, {k: v * weights for k, v in losses.items()})
        self._pending_losses = []
        now = time.time()
        if self._t_last is not None:
            self.logkv("sec_per_step", (now - self._t_last) / max(self.log_interval, 1))
        self._t_last = now
        if self.rank == 0:
            print(" | ".join(f"{k} {v:.6g}" if isinstance(v, float) else f"{k} {v}" for k, v in sorted(self._kvs.items())), flush=True)
            if self.log_dir:
                with open(os.path.join(self.log_dir, "progress.jsonl"), "a") as f:
                    f.write(json.dumps(self._kvs) + "\n")
        self._kvs = {}

    # ------------------------------------------------------------------ checkpoints
    def save(self):
        if self.rank != 0 or not self.log_dir:
            return
        step = self.step + self.resume_step
        for rate, flat in zip(self.ema_rate, self.opt.ema):
            sd = {k: v.detach().cpu().clone() for k, v in self.model.split_flat(flat).items()}
            sd = {k: sd[k] for k in self.model.state_dict().keys()}          # the module's key order
            th.save(sd, os.path.join(self.log_dir, f"ema_{rate}_{step:06d}.pt"))
        th.save(self.opt.state_dict(), os.path.join(self.log_dir, f"opt{step:06d}.pt"))


def parse_resume_step_from_filename(filename):
    """path/to/modelNNNNNN.pt -> NNNNNN (0 when the name has no step)."""
    split = filename.split("model")
    if len(split) < 2:
        return 0
    try:
        return int(split[-1].split(".")[0])
    except ValueError:
        return 0
