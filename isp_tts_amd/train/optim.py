"""The reference's `Optimizer` (experiments/optimizers.py:189-289: AdamW + gradient clipping + ExponentialLR) over flat
arenas, and its data-parallel form.

Reference step (optimizers.py:230-244): backward -> `clip_grad_norm_(param_groups[0]["params"], grad_clip)` (group 0 = the
weight-decay group when parameters are grouped, :34-40) -> `torch.optim.AdamW.step()` -> `zero_grad()`; DDP all-reduces the
gradients inside backward.  Here:

  * `FlatParameters` re-homes every parameter (and its .grad) as a view into ONE fp32 arena, weight-decay group first
    (the same order torch's two param_groups enumerate them in), so "the optimizer" is three launches over contiguous
    memory - `ispk_grad_sqnorm_f32` (2 launches), `ispk_adamw_f32` with the clip coefficient applied on the fly - and
    the gradient exchange is one collective over one buffer.
  * across N ranks (one process per GPU, RCCL over xGMI): reduce-scatter of the gradient arena (sum; the 1/N goes into
    the kernel's grad_scale), each rank updates ITS slice with its slice of the moments (optimizer state is sharded:
    8 B per parameter per rank instead of 8 N), one scalar all-reduce for the clip norm, all-gather of the parameters.
    Same arithmetic as DDP's averaged gradients + a replicated AdamW; 2 x 92.5 MB over the mesh per step.
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .. import runtime

ALIGN = 64   # floats: every tensor starts on a 256-byte boundary of the arena (16-byte loads, whole cache lines)


def group_weight_decayable_params(params: Iterable[nn.Parameter]):
    """optimizers.py:15-20: tensors that squeeze to fewer than 2 dimensions (biases, norm gains, log-slopes, 1-wide
    projections) take no weight decay."""
    wd, no_wd = [], []
    for p in params:
        (no_wd if p.squeeze().ndim < 2 else wd).append(p)
    return wd, no_wd


class FlatParameters:
    """Parameters and gradients of a model as views into two flat fp32 arenas: [decay group | rest | zero padding]."""

    def __init__(self, params: Iterable[nn.Parameter], group_wd_params: bool = True, multiple_of: int = 1):
        seen, plist, everything = set(), [], []
        for p in params:
            if id(p) in seen:
                continue
            seen.add(id(p))
            everything.append(p)               # frozen ones too: torch.optim numbers EVERY tensor it is handed
            if p.requires_grad:
                plist.append(p)
        assert plist, "no trainable parameters"
        # torch.optim.AdamW's index space (what a reference checkpoint's optimizer state is keyed by): the tensors as handed
        # over, decay group first (optimizers.py:34-40), frozen ones included - `freeze()` (row f3) must not shift it
        wd_all, no_wd_all = group_weight_decayable_params(everything) if group_wd_params else (everything, [])
        self.torch_index = {id(p): i for i, p in enumerate(wd_all + no_wd_all)}
        self.torch_group_sizes = (len(wd_all), len(no_wd_all))
        dev = plist[0].device
        assert all(p.dtype == torch.float32 and p.device == dev for p in plist), "fp32 master parameters on one device"
        wd, no_wd = group_weight_decayable_params(plist) if group_wd_params else (plist, [])
        self.params = wd + no_wd
        self.n_decay_tensors = len(wd)
        self.offsets, off = [], 0
        for i, p in enumerate(self.params):
            if i == len(wd):
                self.n_decay = off                  # (aligned) end of the decay group
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        if not no_wd:
            self.n_decay = off
        unit = ALIGN * multiple_of
        self.total = (off + unit - 1) // unit * unit
        self.data = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o:o + p.numel()].view(p.shape)
                view.copy_(p)
                p.data = view
                self._attach_grad(p, o)

    def _attach_grad(self, p, o) -> None:
        g = self.grad[o:o + p.numel()].view(p.shape)
        g._ispk_grad_arena = True        # backward kernels may write this buffer directly (train/stack.py `_deliver`)
        g._ispk_dirty = False            # ... overwriting while nothing has been delivered since the arena was zeroed
        p.grad = g

    def zero_grad(self) -> None:
        if self.grad.is_cuda:
            runtime.zero_(self.grad)
        else:
            self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # someone may have set .grad to None / another tensor
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o or not getattr(p.grad, "_ispk_grad_arena", False):
                self._attach_grad(p, o)
            else:
                p.grad._ispk_dirty = False

    def mark_updated(self) -> None:
        """The arena was written behind autograd's back: bump every parameter's version so that staged weight images
        (staging.StagedWeights) are rebuilt."""
        for p in self.params:
            torch.autograd.graph.increment_version(p)


class FlatAdamW:
    """`Optimizer` of experiments/optimizers.py:189-289 for a flat arena.  `step(loss)` = backward + clip + AdamW +
    zero_grad and returns the gradient norm of the clipped group (None when not finite, as :238-239)."""

    def __init__(self, params, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 grad_clip: Optional[float] = 1.0, gamma: Optional[float] = 0.995, group_wd_params: bool = True,
                 grad_accum_steps: int = 1, process_group=None, update: Optional[Callable] = None,
                 sqnorm: Optional[Callable] = None):
        if isinstance(params, nn.Module):
            params = params.parameters()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        # optimizers.py:34-40: grouping only when weight_decay > 0; otherwise ONE group (which is then what gets clipped)
        self.flat = FlatParameters(params, group_wd_params and weight_decay > 0., multiple_of=self.world)
        self.lr, self.base_lr, self.betas, self.eps, self.weight_decay = lr, lr, tuple(betas), eps, weight_decay
        self.grad_clip, self.gamma, self.grad_accum_steps = grad_clip, gamma, grad_accum_steps
        self.step_count, self.last_epoch = 0, 0
        self.shard = self.flat.total // self.world
        self.lo = self.rank * self.shard
        dev = self.flat.data.device
        self.exp_avg = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.grad_shard = torch.zeros(self.shard, dtype=torch.float32, device=dev) if self.world > 1 else None
        self.param_shard = torch.zeros(self.shard, dtype=torch.float32, device=dev) if self.world > 1 else None
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._update = update or runtime.adamw          # (hooks: the CPU gloo test drives the exchange with the oracle's math)
        self._sqnorm = sqnorm or runtime.grad_sqnorm
        self._one = None
        self._frozen: dict = {}       # arena index -> (parameter copy, exp_avg copy, exp_avg_sq copy) of tensors frozen after the build
        self.check_finite = True      # `step` returns None for a non-finite norm like the reference (:238-239): one host sync
        self._reduce_scatter = self.world > 1 and dist.get_backend(process_group) != "gloo"   # gloo has none

    # ------------------------------------------------------------------------------------------------------------ step
    def step(self, loss_value: Optional[Tensor] = None, step_optimizer: bool = True, *, args_dev: Optional[Tensor] = None,
             check_finite: Optional[bool] = None):
        """`args_dev` / `check_finite` are per CALL (a captured step, train/graph.py, passes the device record the AdamW factors
        are read from and skips the host read of the norm); the optimizer object itself never changes mode."""
        check_finite = self.check_finite if check_finite is None else check_finite
        if loss_value is not None:
            if self.grad_accum_steps == 1 and loss_value.is_cuda and loss_value.dtype == torch.float32:
                if self._one is None or self._one.device != loss_value.device:
                    self._one = torch.ones((), dtype=torch.float32, device=loss_value.device)
                loss_value.backward(self._one)          # (no division, no ones_like: the root gradient is a kept tensor)
            else:
                (loss_value / self.grad_accum_steps).backward()
        if not step_optimizer:
            return None
        flat, n_dec = self.flat, self.flat.n_decay
        self.step_count += 1
        if self.world > 1:
            if self._reduce_scatter:
                dist.reduce_scatter_tensor(self.grad_shard, flat.grad, op=dist.ReduceOp.SUM, group=self.group)
                g = self.grad_shard
            else:
                dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=self.group)
                g = flat.grad[self.lo:self.lo + self.shard]
            n_dec = min(max(n_dec - self.lo, 0), self.shard)
        else:
            g = flat.grad
        frozen = self._hold_frozen()
        clip = self.grad_clip is not None
        if clip:
            if n_dec > 0:
                self._sqnorm(g[:n_dec], self.sq)
            else:
                self.sq.zero_()
            if self.world > 1:
                dist.all_reduce(self.sq, op=dist.ReduceOp.SUM, group=self.group)
        if args_dev is not None:
            runtime.adamw_dev(flat.data[self.lo:self.lo + self.shard], g, self.exp_avg, self.exp_avg_sq, n_dec, args_dev,
                              self.sq if clip else None)
        else:
            self._update(flat.data[self.lo:self.lo + self.shard], g, self.exp_avg, self.exp_avg_sq, n_dec, self.lr, self.betas,
                         self.eps, self.weight_decay, self.step_count, self.sq if clip else None,
                         self.grad_clip if clip else 1.0, 1.0 / self.world)
        if self.world > 1:
            # (from a copy of the slice: input and output of the collective do not alias)
            self.param_shard.copy_(flat.data[self.lo:self.lo + self.shard])
            dist.all_gather_into_tensor(flat.data, self.param_shard, group=self.group)
        if frozen:
            self._restore_frozen(frozen)
        flat.mark_updated()
        flat.zero_grad()
        if not clip:
            return None
        # norm of the averaged gradients
        norm = runtime.sqrt_scale(self.sq, 1.0 / self.world) if self.sq.is_cuda else self.sq.sqrt() / self.world
        return norm if not check_finite or bool(torch.isfinite(norm)) else None

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat.zero_grad()

    # ------------------------------------------------- tensors frozen AFTER the arena was built (`model.freeze()`, row f3)
    def _shard_range(self, i: int):
        """The part of arena tensor i that lies in this rank's shard, in shard coordinates (None: nothing)."""
        o, n = self.flat.offsets[i], self.flat.params[i].numel()
        a, b = max(o, self.lo), min(o + n, self.lo + self.shard)
        return (a - self.lo, b - self.lo) if a < b else None

    def _hold_frozen(self) -> list:
        """torch.optim.AdamW skips a tensor whose .grad is None - no decay, no moment update.  A tensor frozen after the arena
        was built still lies in it (its gradient stays zero: `runtime.deliver_grads` drops what the backward produces), so the
        flat update would decay it and age its moments: keep a copy from the moment the freeze is seen and put it back after
        every update.  -> indices of the frozen tensors."""
        flat = self.flat
        idx = [i for i, p in enumerate(flat.params) if not p.requires_grad]
        for i in list(self._frozen):
            if i not in idx:
                del self._frozen[i]                      # trainable again
        for i in idx:
            if i not in self._frozen:
                o, n = flat.offsets[i], flat.params[i].numel()
                r = self._shard_range(i)
                self._frozen[i] = (flat.data[o:o + n].clone(),
                                   None if r is None else self.exp_avg[r[0]:r[1]].clone(),
                                   None if r is None else self.exp_avg_sq[r[0]:r[1]].clone())
        return idx

    def _restore_frozen(self, idx: list) -> None:
        flat, items = self.flat, []
        for i in idx:
            o, n = flat.offsets[i], flat.params[i].numel()
            pc, mc, vc = self._frozen[i]
            items.append((pc, flat.data[o:o + n], runtime.SEG_COPY))
            r = self._shard_range(i)
            if r is not None:
                items += [(mc, self.exp_avg[r[0]:r[1]], runtime.SEG_COPY), (vc, self.exp_avg_sq[r[0]:r[1]], runtime.SEG_COPY)]
        if flat.data.is_cuda:
            runtime.segments(items)
        else:
            for src, dst, _ in items:
                dst.copy_(src)

    # ------------------------------------------------------------------------------------------------- lr schedule
    def anneal_on_epoch_end(self, *args) -> None:
        """ExponentialLR stepped per epoch (optimizers.py:249-252; recipes/default.yaml:100-101 gamma 0.995)."""
        if self.gamma is not None:
            self.last_epoch += 1
            self.lr = self.base_lr * self.gamma ** self.last_epoch

    def anneal_on_step_end(self, *args) -> None:
        pass

    def get_last_lr(self):
        return [self.lr, self.lr] if self.flat.n_decay_tensors < len(self.flat.params) else [self.lr]

    # ------------------------------------------------------------------------------------------------- checkpoints
    def _full_moments(self):
        if self.world == 1:
            return self.exp_avg, self.exp_avg_sq
        out = []
        for t in (self.exp_avg, self.exp_avg_sq):
            full = torch.empty(self.flat.total, dtype=torch.float32, device=t.device)
            dist.all_gather_into_tensor(full, t, group=self.group)
            out.append(full)
        return out

    def state_dict(self) -> dict:
        """The layout the reference's checkpoints hold (`Optimizer.state_dict`, optimizers.py:266-270): torch.optim.AdamW's
        state_dict (per-parameter step / exp_avg / exp_avg_sq, two param_groups) + the scheduler's."""
        m, v = self._full_moments()
        f = self.flat
        state = {}
        for p, o in zip(f.params, f.offsets):     # keyed like torch.optim: index among ALL tensors handed over (frozen included)
            state[f.torch_index[id(p)]] = {"step": torch.tensor(float(self.step_count)),
                                           "exp_avg": m[o:o + p.numel()].view(p.shape).clone(),
                                           "exp_avg_sq": v[o:o + p.numel()].view(p.shape).clone()}
        common = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "amsgrad": False, "initial_lr": self.base_lr}
        n0, n1 = f.torch_group_sizes
        groups = [dict(common, weight_decay=self.weight_decay, params=list(range(n0)))]
        if n1:
            groups.append(dict(common, weight_decay=0., params=list(range(n0, n0 + n1))))
        return {"optimizer": {"state": state, "param_groups": groups},
                "lr_scheduler": {"gamma": self.gamma, "last_epoch": self.last_epoch, "base_lrs": [self.base_lr] * len(groups)}}

    def load_state_dict(self, state_dict: dict, restore_lr: bool = True) -> None:
        opt = state_dict["optimizer"]
        f = self.flat
        m = torch.zeros(f.total, dtype=torch.float32, device=f.data.device)
        v = torch.zeros_like(m)
        steps = set()
        for p, o in zip(f.params, f.offsets):
            st = opt["state"].get(f.torch_index[id(p)])
            if st is None:
                continue
            assert tuple(st["exp_avg"].shape) == tuple(p.shape), "optimizer state does not belong to this parameter"
            m[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
            v[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        assert len(steps) <= 1, "per-parameter step counts differ: not a state this optimizer can hold"
        self.step_count = steps.pop() if steps else 0
        self.exp_avg.copy_(m[self.lo:self.lo + self.shard])
        self.exp_avg_sq.copy_(v[self.lo:self.lo + self.shard])
        sched = state_dict.get("lr_scheduler")
        if restore_lr and sched is not None:
            self.last_epoch = int(sched.get("last_epoch", 0))
            self.lr = opt["param_groups"][0]["lr"]
            # ExponentialLR's closed form lr = initial_lr * gamma ** epoch restarts from the CHECKPOINT's initial_lr
            self.base_lr = float(opt["param_groups"][0].get("initial_lr", self.base_lr))

    def set_progress(self, iteration: int, epoch: int) -> None:
        self.step_count, self.last_epoch = iteration, epoch

    def __repr__(self) -> str:
        return (f"FlatAdamW(lr={self.lr}, betas={self.betas}, eps={self.eps}, weight_decay={self.weight_decay}, "
                f"grad_clip={self.grad_clip}, gamma={self.gamma}, parameters={self.flat.total}, world={self.world})")
