"""TriplaneUNetModelSmall / TriplaneUNetModelSmallRaw with the reference's constructor, parameter names
and forward signature (src/diffusion/unet_triplane.py:315-510, 513-702) — executed by libsin3dm_hip.so.

The module owns ordinary nn.Parameters under the reference's state_dict names, so `load_state_dict`,
`.to(dev)`, `.eval()`, `.parameters()` and checkpoints written by the reference all work.  Their values are
mirrored into the HIP handle (repacked to the kernels' layouts) whenever they change.

Under no_grad (sampling, gaussian_diffusion.py:525) the forward is the inference kernel sequence.  With grad enabled
and trainable parameters it is an autograd node whose backward is the HIP backward pass (training tier): the
parameters are then re-homed as views of ONE flat device vector (state-dict order of the C ABI, PyTorch layouts) so
that the optimizer / EMA / all-reduce can work on a single buffer; `.grad` of every parameter is filled as
`loss.backward()` would.  The gradient with respect to the input x is not computed (training never needs it).
"""
from __future__ import annotations

import ctypes as C
import os

import torch as th
import torch.nn as nn

from .. import _lib
from ..testing import unet_param_shapes


class _Node(nn.Module):
    """Anonymous container used to rebuild the reference's dotted parameter names."""


def _register(root, dotted, tensor):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor))


class _TriplaneUNetBase(nn.Module):
    _ROLLOUT = True

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks=1, dropout=0,
                 channel_mult=(1, 2), use_checkpoint=False, use_fp16=False, use_scale_shift_norm=False):
        super().__init__()
        if isinstance(channel_mult, str):
            channel_mult = tuple(int(c) for c in channel_mult.split(","))
        if use_fp16:
            # the reference's fp16 path cannot run either: in_conv stays fp32 while x is cast to fp16
            # (unet_triplane.py:479, fp16_util only converts input_blocks/output_blocks)
            raise NotImplementedError("use_fp16 is not runnable in the reference and is not implemented here")
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.dropout = dropout              # the reference has its Dropout commented out (:242)
        self.channel_mult = tuple(channel_mult)
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.dtype = th.float32

        shapes = unet_param_shapes(in_channels, model_channels, out_channels, num_res_blocks, self.channel_mult,
                                   use_scale_shift_norm, self._ROLLOUT)
        self._param_names = list(shapes)
        gen = th.Generator().manual_seed(0)
        for name, shape in shapes.items():
            _register(self, name, self._init_tensor(name, shape, gen))

        self._handle = None
        self._synced = None
        self._plist = None
        self._film_cache = {}       # host-known timestep values -> FiLM table (dropped whenever the parameters change)
        self._film_sched = None     # the last schedule announced by a sampling loop (prepare_timesteps)
        self._lane = 0              # workspace lane the inference calls go to (see lane())
        self._flat = None           # training: flat master parameters on the device (see _ensure_flat)
        self._flat_layout = None    # [(name, offset, numel, shape)]
        self._flat_dirty = False
        self.last_flat_grad = None  # flat gradient vector of the most recent backward

    @staticmethod
    def _init_tensor(name, shape, gen):
        """Default initialisation in the spirit of the reference's modules: uniform(+-1/sqrt(fan_in)) for
        conv/linear, ones/zeros for norms, zeros for the zero_module'd convs (out_layers.2, out.2)."""
        leaf = name.rsplit(".", 1)[-1]
        if ".norm_" in name:
            return th.ones(shape) if leaf == "weight" else th.zeros(shape)
        if ".out_layers.2." in name or name.startswith("out.2."):
            return th.zeros(shape)
        fan_in = 1
        wshape = shape if leaf == "weight" else None
        if wshape is not None:
            for d in wshape[1:]:
                fan_in *= d
            bound = (1.0 / fan_in) ** 0.5
        else:
            bound = 0.05
        return (th.rand(shape, generator=gen) * 2 - 1) * bound

    # ------------------------------------------------------------------ HIP handle
    def _cfg(self):
        cm = list(self.channel_mult) + [0] * (8 - len(self.channel_mult))
        return _lib.UNetCfg(self.in_channels, self.model_channels, self.out_channels, self.num_res_blocks,
                            len(self.channel_mult), (C.c_int32 * 8)(*cm), int(bool(self.use_scale_shift_norm)),
                            int(self._ROLLOUT))

    def _ensure_handle(self):
        lib = _lib.load()
        if self._handle is None:
            h = C.c_void_p()
            cfg = self._cfg()
            _lib.check(lib.s3d_unet_create(C.byref(cfg), C.byref(h)))
            self._handle = h
        # cheap change detection (this runs in every denoising step): in-place updates bump ._version, and anything that
        # re-homes the tensors (.to / .cuda / .float) goes through _apply below, which drops the cached state
        if self._plist is None:
            named = dict(self.named_parameters())
            self._plist = [named[n] for n in self._param_names]
        stamp = tuple(p._version for p in self._plist)
        if stamp != self._synced or self._flat_dirty:
            self._film_cache.clear()
        if self._flat is not None:
            if stamp != self._synced or self._flat_dirty:          # parameters changed on the device: repack there
                params = dict(self.named_parameters())
                if not self._flat_intact(params):                   # somebody re-assigned .data: start over
                    self._flat = None
                    self._synced = None
                    return self._ensure_handle()
                with th.cuda.device(self._flat.device):
                    _lib.check(lib.s3d_unet_repack(self._handle, _lib.stream_ptr()))
                self._synced, self._flat_dirty = stamp, False
            return lib
        if stamp != self._synced:
            # only the tensors whose version moved travel to the host and into the handle (a stock optimizer that updates
            # every parameter still re-sends all of them each step: training belongs on the flat path, _ensure_flat)
            old = self._synced if self._synced is not None and len(self._synced) == len(stamp) else None
            for i, name in enumerate(self._param_names):
                if old is not None and old[i] == stamp[i]:
                    continue
                host = self._plist[i].detach().to("cpu", th.float32).contiguous()
                shape = (C.c_int64 * host.dim())(*host.shape)
                _lib.check(lib.s3d_unet_set_param(self._handle, name.encode(), C.c_void_p(host.data_ptr()), shape,
                                                  host.dim()))
            self._synced = stamp
        return lib

    # ------------------------------------------------------------------ workspace lanes (independent sample chains)
    def lane(self, k):
        """Context manager: the inference calls inside go to workspace lane `k` of the HIP handle (s3d_unet_select_lane).
        Independent sample chains — one lane and one HIP stream each — share this module and its packed weights; what a
        forward writes (activations, the timestep MLP's scratch, the cached FiLM tables) exists once per lane, so chains need no
        event edges between them (GaussianDiffusion.sample_loop_chains drives them).  One host thread; lane 0 is the default."""
        return _LaneCtx(self, int(k))

    def _select_lane(self, k):
        if not 0 <= k < _lib.MAX_LANES:
            raise AssertionError(f"lane must be in [0, {_lib.MAX_LANES})")
        # (entered and left around every step of a chain: the parameter sync of _ensure_handle is the callee's job, not this switch's)
        lib = _lib.load() if self._handle is not None else self._ensure_handle()
        _lib.check(lib.s3d_unet_select_lane(self._handle, k))
        self._lane = k

    def parameter_stamp(self):
        """What _ensure_handle compares to decide that the packed weight image is stale.  Chains in flight on several lanes share
        that image and only the issuing lane's stream would see a repack: GaussianDiffusion.sample_loop_chains_progressive
        snapshots this at its start and fails loudly if it moves before the last chain has finished."""
        if self._plist is None:
            named = dict(self.named_parameters())
            self._plist = [named[n] for n in self._param_names]
        return (tuple(p._version for p in self._plist), bool(self._flat_dirty))

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)      # .to(), .cuda(), .float(): the parameter storage moves
        self._lane = 0                                  # (a handle created after this starts on lane 0)
        self._film_cache = {}
        self._film_sched = None
        self._synced = None
        self._flat = None
        self._plist = None
        return out

    # ------------------------------------------------------------------ training tier: flat parameters
    def _flat_intact(self, params):
        base = self._flat.data_ptr()
        return all(params[n].data_ptr() == base + 4 * off and params[n].device == self._flat.device
                   for n, off, _, _ in self._flat_layout)

    def _ensure_flat(self):
        """Re-home every parameter as a view of one flat fp32 device vector and attach it to the handle."""
        lib = self._ensure_handle()
        if self._flat is not None:
            return lib
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        _lib.require_gpu(next(iter(params.values())))
        n = lib.s3d_unet_num_params(self._handle)
        layout = []
        for i in range(n):
            name, shape, nd, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int64()
            _lib.check(lib.s3d_unet_param_info(self._handle, i, C.byref(name), shape, C.byref(nd)))
            _lib.check(lib.s3d_unet_param_offset(self._handle, i, C.byref(off)))
            shp = tuple(shape[k] for k in range(nd.value))
            numel = 1
            for d in shp:
                numel *= d
            layout.append((name.value.decode(), off.value, numel, shp))
        total = lib.s3d_unet_param_numel(self._handle)
        flat = th.empty(total, device=dev, dtype=th.float32)
        with th.no_grad():
            for name, off, numel, shp in layout:
                p = params[name]
                assert tuple(p.shape) == shp, (name, tuple(p.shape), shp)
                view = flat[off:off + numel].view(shp)
                view.copy_(p.detach().to(dev, th.float32))
                p.data = view
        with th.cuda.device(dev):
            _lib.check(lib.s3d_unet_train_attach(self._handle, _lib.ptr(flat), total))
            _lib.check(lib.s3d_unet_repack(self._handle, _lib.stream_ptr()))
        self._flat, self._flat_layout = flat, layout
        self._plist = None
        named = dict(self.named_parameters())
        self._plist = [named[n] for n in self._param_names]
        self._synced = tuple(p._version for p in self._plist)
        self._flat_dirty = False
        return lib

    @property
    def flat_parameters(self):
        """The flat master vector (allocating / attaching it on first use)."""
        self._ensure_flat()
        return self._flat

    def mark_parameters_changed(self):
        """Call after writing to `flat_parameters` directly (a fused optimizer step): forces a device repack."""
        self._flat_dirty = True

    def forward_train(self, x, timesteps, H, W, D):
        """Forward pass that keeps the activations for one `backward_flat` (no autograd graph)."""
        lib = self._ensure_flat()
        self._ensure_handle()                      # repack if the parameters changed
        h, t, out = self._prep(x, timesteps, H, W, D)
        with th.cuda.device(h.device):
            _lib.check(lib.s3d_unet_forward_train(self._handle, _lib.ptr(h), _lib.ptr(t), h.shape[0], int(H), int(W), int(D),
                                                  _lib.ptr(out), _lib.stream_ptr()))
        self._train_inputs = (h, t)                # keep the tensors the tape points at alive until the backward
        return out

    def backward_flat(self, d_out, out=None, marks=None):
        """Backward of the last forward_train: returns the flat gradient vector (layout of flat_parameters).
        marks: up to two torch.cuda.Event that the library records on the current stream when a group of gradients is final
        (grad_ready_groups() says which ranges of the flat vector each one covers)."""
        lib = _lib.load()
        d_out = d_out.contiguous().float()
        g = out if out is not None else th.empty_like(self._flat)
        with th.cuda.device(d_out.device):
            if marks:
                st = th.cuda.current_stream(d_out.device)
                for ev in marks:
                    ev.record(st)                      # (creates the underlying HIP event; re-recorded by the library at its mark)
                evs = (C.c_void_p * len(marks))(*[C.c_void_p(ev.cuda_event) for ev in marks])
                _lib.check(lib.s3d_unet_backward_marked(self._handle, _lib.ptr(d_out), _lib.ptr(g), _lib.stream_ptr(), evs, len(marks)))
            else:
                _lib.check(lib.s3d_unet_backward(self._handle, _lib.ptr(d_out), _lib.ptr(g), _lib.stream_ptr()))
        self._train_inputs = None
        self.last_flat_grad = g
        return g

    def grad_ready_groups(self):
        """[ranges final at mark 0, at mark 1, at the end of backward_flat]: merged (begin, end) ranges of the flat vector."""
        from ..parallel import grad_ready_groups
        self._ensure_flat()
        return grad_ready_groups([(name, off, numel) for name, off, numel, _ in self._flat_layout])

    def split_flat(self, flat):
        """{name: view} of a vector laid out like flat_parameters."""
        self._ensure_flat()
        return {name: flat[off:off + numel].view(shp) for name, off, numel, shp in self._flat_layout}

    def __del__(self):
        h = self.__dict__.pop("_handle", None)      # not via nn.Module.__setattr__: it may run at interpreter exit
        if h is not None:
            try:
                _lib.load().s3d_unet_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ live kernel timing (bench.py)
    def profile(self, every, classes=7):
        """Record HIP events around the convolution launches of every `every`-th forward (0 = off); classes: bit mask of the
        launch classes that are bracketed (1: 3x3, 2: 1x1, 4: rank-1 tables)."""
        lib = self._ensure_handle()
        _lib.check(lib.s3d_unet_profile_classes(self._handle, int(classes)))
        _lib.check(lib.s3d_unet_profile(self._handle, int(every)))

    def profile_read(self, into=None):
        """Accumulate the recorded launch durations: dict of lists over (conv3x3, conv1x1, rank1)."""
        prof = into if into is not None else _lib.Profile()
        _lib.check(_lib.load().s3d_unet_profile_read(self._handle, C.byref(prof)))
        return prof

    def profile_kernel(self, cls=0):
        """Name of the kernel the most recent timed launch of class cls (0: 3x3, 1: 1x1, 2: rank-1) dispatched."""
        return (_lib.load().s3d_unet_profile_kernel(self._handle, int(cls)) or b"").decode()

    def convert_to_fp16(self):
        raise NotImplementedError("fp16 is not runnable in the reference (see __init__)")

    def convert_to_fp32(self):
        pass

    # ------------------------------------------------------------------ forward
    def forward(self, x, timesteps, H=None, W=None, D=None, y=None):
        """x: [N, C, H+D, W+D] composed triplane, timesteps: [N]; returns a tensor of x's shape
        (reference: unet_triplane.py:465-510)."""
        assert H is not None and W is not None and D is not None
        _lib.require_gpu(x)
        if y is not None:
            x = th.cat([x, y], dim=1)
        if th.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            self._ensure_flat()
            names = [n for n, _, _, _ in self._flat_layout]
            params = dict(self.named_parameters())
            return _UNetFn.apply(self, x, timesteps, int(H), int(W), int(D), *[params[n] for n in names])
        h, t, out = self._prep(x, timesteps, H, W, D)
        B = h.shape[0]
        lib = self._ensure_handle()
        hv = getattr(timesteps, "host_values", None)
        with th.cuda.device(h.device):
            if hv is not None and len(hv) == B:
                # timestep values known on the host (the sampling loops): the FiLM table of these values is computed once
                # per weight version and reused by every later step / sample with the same values
                film, stride = self._film_for(lib, hv, t)
                _lib.check(lib.s3d_unet_forward_film(self._handle, _lib.ptr(h), _lib.ptr(film), stride, B, int(H), int(W),
                                                     int(D), _lib.ptr(out), _lib.stream_ptr()))
            else:
                _lib.check(lib.s3d_unet_forward(self._handle, _lib.ptr(h), _lib.ptr(t), B, int(H), int(W), int(D),
                                                _lib.ptr(out), _lib.stream_ptr()))
        assert out.shape == x.shape or y is not None
        return out

    # denoise_step takes `carry` (GaussianDiffusion._loop asks); S3D_CARRY_IN_CONV=0: every step launches its own in_conv (A/B)
    carries_in_conv = os.environ.get("S3D_CARRY_IN_CONV", "1") != "0"

    def denoise_step(self, x, timesteps, step, H=None, W=None, D=None, carry=0):
        """One step of a sampling loop in one library call: this forward (host-known `timesteps`) with the sampler update
        `step` (_lib.SamplerArgs with x / noise / tables / outputs set; its model_out is ignored) applied by the output
        head's launch — src/diffusion/unet_triplane.py:465-510 + gaussian_diffusion.py:396-440 / 538-600.  The model output
        is not materialised.  Results equal forward() followed by s3d_sampler_step bit for bit.
        carry (_lib.CARRY_OUT | _lib.CARRY_IN, s3d_unet_step_film_carry): CARRY_OUT — this step's sample is the next step's input:
        the head also leaves the next step's in_conv (a pointwise TriplaneConv, :378, 482) in the workspace; CARRY_IN — `x` is the
        previous step's sample, untouched since: that in_conv is used instead of launching the kernel.  Same bits either way."""
        assert H is not None and W is not None and D is not None
        _lib.require_gpu(x)
        hv = getattr(timesteps, "host_values", None)
        B = x.shape[0]
        assert hv is not None and len(hv) == B, "denoise_step needs host-known timesteps (HostTimesteps)"
        assert x.is_contiguous() and x.dtype == th.float32 and x.data_ptr() == step.x
        assert x.shape[1] == self.in_channels == self.out_channels and x.shape[2] == H + D and x.shape[3] == W + D
        lib = self._ensure_handle()
        t = timesteps.to(device=x.device, dtype=th.float32).contiguous()
        with th.cuda.device(x.device):
            film, stride = self._film_for(lib, hv, t)
            if carry:
                _lib.check(lib.s3d_unet_step_film_carry(self._handle, _lib.ptr(film), stride, B, int(H), int(W), int(D), C.byref(step),
                                                        None, _lib.stream_ptr(), int(carry)))
            else:
                _lib.check(lib.s3d_unet_step_film(self._handle, _lib.ptr(film), stride, B, int(H), int(W), int(D), C.byref(step),
                                                  None, _lib.stream_ptr()))

    def prepare_timesteps(self, values, t_dev):
        """The FiLM tables of a whole schedule (host values + the same values on the device) in one batched launch (three
        linears over len(values) rows); the sampling loops call this at the start of EVERY loop, i.e. once per sample: the
        per-step work is batched and hoisted, not carried over between samples.  Remembered, so that a weight update in the
        middle of a loop refills the cache in one go."""
        self._film_sched = (tuple(values), t_dev)
        lib = self._ensure_handle()
        dkey = (str(t_dev.device), self._lane)
        width = lib.s3d_unet_film_width(self._handle)
        t = t_dev.to(th.float32).contiguous()
        film = th.empty((len(values), width), device=t.device, dtype=th.float32)
        with th.cuda.device(t.device):
            _lib.check(lib.s3d_unet_film(self._handle, _lib.ptr(t), len(values), _lib.ptr(film), _lib.stream_ptr()))
        # tables are produced on the stream that is current NOW; a later forward on another stream waits for this event
        # (s3d_unet_film's scratch exists once per workspace LANE: calls on one lane are expected to be issued from one stream at a
        # time; chains on different lanes need no edge between them — but they share ONE packed weight image: see parameter_stamp)
        mark = self._film_mark(t.device)
        for k, v in enumerate(values):
            self._film_cache[(dkey, v)] = (film[k:k + 1], mark)

    @staticmethod
    def _film_mark(device):
        st = th.cuda.current_stream(device)
        ev = th.cuda.Event()
        ev.record(st)
        return (st.cuda_stream, ev)

    @staticmethod
    def _film_wait(mark, device):
        st = th.cuda.current_stream(device)
        if st.cuda_stream != mark[0]:
            st.wait_event(mark[1])

    def _film_for(self, lib, hv, t):
        """(device FiLM table, row stride) for host-known timestep values: one row when the batch shares a value."""
        same = all(v == hv[0] for v in hv)
        key = ((str(t.device), self._lane), hv[0] if same else hv)
        hit = self._film_cache.get(key)
        if hit is None and same and self._film_sched is not None and hv[0] in self._film_sched[0] \
                and str(self._film_sched[1].device) == str(t.device):
            self.prepare_timesteps(*self._film_sched)          # (the weights changed: refill the schedule's tables at once)
            hit = self._film_cache.get(key)
        film = None
        if hit is not None:
            film = hit[0]
            self._film_wait(hit[1], t.device)
        if film is None:
            if len(self._film_cache) > 4096:
                self._film_cache.clear()
            n = 1 if same else len(hv)
            width = lib.s3d_unet_film_width(self._handle)
            film = th.empty((n, width), device=t.device, dtype=th.float32)
            _lib.check(lib.s3d_unet_film(self._handle, _lib.ptr(t[:n].contiguous()), n, _lib.ptr(film), _lib.stream_ptr()))
            self._film_cache[key] = (film, self._film_mark(t.device))
        return film, (0 if same else film.shape[1])

    def _prep(self, x, timesteps, H, W, D):
        h = x.contiguous().float()
        B, Cin, Hc, Wc = h.shape
        assert Cin == self.in_channels, f"expected {self.in_channels} input channels, got {Cin}"
        assert Hc == H + D and Wc == W + D, f"composed map {tuple(h.shape[-2:])} != (H+D, W+D) = {(H + D, W + D)}"
        assert timesteps.shape == (B,)
        t = timesteps.to(device=h.device, dtype=th.float32).contiguous()
        out = th.empty((B, self.out_channels, Hc, Wc), device=h.device, dtype=th.float32)
        return h, t, out


class _LaneCtx:
    def __init__(self, model, k):
        self.model, self.k, self.prev = model, k, 0

    def __enter__(self):
        self.prev = self.model._lane
        if self.k != self.prev:
            self.model._select_lane(self.k)
        return self.model

    def __exit__(self, *exc):
        if self.model._lane != self.prev:
            self.model._select_lane(self.prev)
        return False


class _UNetFn(th.autograd.Function):
    """model(x, t) as an autograd node: backward = the HIP backward pass.  Each backward writes a fresh flat gradient
    vector and hands autograd views of it, so accumulation into existing .grad tensors keeps its usual meaning."""

    @staticmethod
    def forward(ctx, model, x, t, H, W, D, *params):
        ctx.model = model
        return model.forward_train(x.detach(), t, H, W, D)

    @staticmethod
    def backward(ctx, d_out):
        model = ctx.model
        g = model.backward_flat(d_out)
        views = tuple(g[off:off + numel].view(shp) for _, off, numel, shp in model._flat_layout)
        return (None, None, None, None, None, None) + views


class TriplaneUNetModelSmall(_TriplaneUNetBase):
    """Triplane UNet with rollout convolutions (reference: unet_triplane.py:315-510)."""
    _ROLLOUT = True


class TriplaneUNetModelSmallRaw(_TriplaneUNetBase):
    """Same topology without rollout and without the skip-size resize (reference: :513-702)."""
    _ROLLOUT = False
