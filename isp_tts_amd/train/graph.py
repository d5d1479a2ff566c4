"""One training step - forward with dropout, the losses, backward, clip + AdamW, zeroing of the gradient arena, re-staging of
the updated weights - captured ONCE into a HIP graph and replayed (the reference's loop body, experiments/trainer.py:538-579).

A training step is ~750 launches of libispk.so and nothing else (tests/test_gpu_train.py::
test_training_step_issues_no_aten_compute_ops); eager, the host issues them more slowly than the device retires the small
ones.  What a replay cannot take from frozen launch arguments comes from device memory the host rewrites before each
replay: the dropout seed word (runtime.set_seed_source: every dropout kernel folds it into its seed when it runs) and the
AdamW factors of the step (bias corrections, lr: runtime.adam_args -> ispk_adamw_f32_dev).  The weight images the kernels
read (fused [to_q; to_kv], bf16 copies, transposes) are rebuilt INSIDE the captured step from the arena the optimizer
updates in place, into buffers of the graph's private pool: a replay always reads the weights of the previous replay.
"""
from __future__ import annotations

import gc

import torch
from torch import Tensor

from .. import runtime
from .model import acoustic_train_forward
from .optim import FlatAdamW


class GraphedTrainStep:
    """step = GraphedTrainStep(model, opt, batch); total, losses, norm = step(**next_batch).
    `batch`: text, text_len, mel, mel_len, pitch, energy (+ flow_x0, flow_t) on the device, at the shapes every later batch has.
    The `warmup` steps before the capture are REAL steps on `batch`.  Single rank, no gradient accumulation."""

    KEYS = ("text", "text_len", "mel", "mel_len", "pitch", "energy")

    def __init__(self, model, opt: FlatAdamW, batch: dict, amp: bool = True, train_aligner: bool = True, warmup: int = 2):
        assert opt.world == 1 and opt.grad_accum_steps == 1, "one rank, no accumulation"
        self.model, self.opt, self.amp, self.train_aligner = model, opt, amp, train_aligner
        dev = batch["text"].device
        self.static = {k: batch[k].clone() for k in self.KEYS}
        b, l = batch["text"].shape
        self.static["flow_x0"] = batch["flow_x0"].clone() if "flow_x0" in batch else torch.randn(b, l, 3, device=dev)
        self.static["flow_t"] = batch["flow_t"].clone() if "flow_t" in batch else torch.rand(b, device=dev)
        self.fresh_noise = "flow_x0" not in batch           # (no noise given: the caller wants new noise per step)
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.args_dev = torch.zeros(10, dtype=torch.float32, device=dev)
        self.graph = None
        runtime.set_seed_source(self.seed_dev)              # (process-wide: autograd's backward thread sees it too)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # warm-up on a side stream: reserves LDS, sizes workspaces, fills the allocator
                for _ in range(warmup):
                    self._push()
                    self.out = self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            gc.collect()                                    # (no cyclic collection inside a capture: isp_tts_amd/graph.py)
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                self._push()
                # Every weight image must be REBUILT inside the capture: a cache hit (no warm-up step, or an eager forward
                # since the last optimizer step) would record no staging launch and every replay would read capture-time weights.
                opt.flat.mark_updated()
                staged = runtime.stage_calls
                torch.cuda.synchronize()
                self._capture_stream_key = None
                with torch.cuda.graph(self.graph):
                    self._capture_stream_key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
                    self.out = self._body()
                assert runtime.stage_calls > staged, "the captured step re-staged no weight image: replays would train on stale weights"
            finally:
                if gc_was_on:
                    gc.enable()
            opt.step_count -= 1                             # the capture recorded the step, it did not run it
        finally:
            runtime.set_seed_source(None)

    def close(self) -> None:
        """Forget the graph and what the module-global caches still hold of its private pool (weight images staged during the
        capture, the capture stream's workspace), so that the pool's memory can be returned."""
        self.graph, self.out = None, None
        for m in self.model.modules():
            m.__dict__.pop("_train_images", None)
        if getattr(self, "_capture_stream_key", None) is not None:
            runtime.drop_workspace(self._capture_stream_key)
            self._capture_stream_key = None
        self.opt.flat.mark_updated()

    def __del__(self):
        try:
            if self.graph is not None:
                self.close()
        except Exception:      # (interpreter shutdown)
            pass

    def _body(self):
        s = self.static
        _, total, losses = acoustic_train_forward(self.model, s["text"], s["text_len"], s["mel"], s["mel_len"], s["pitch"], s["energy"],
                                                  flow_noise=s["flow_x0"], flow_time=s["flow_t"], amp=self.amp,
                                                  train_aligner=self.train_aligner)
        # (the finiteness check of an eager step is a host read of the norm; the AdamW factors come from the device record)
        norm = self.opt.step(total, args_dev=self.args_dev, check_finite=False)
        return total, losses, norm

    def _push(self) -> None:
        """Host state of the NEXT step into the two device records (pinned staging, stream-ordered copies)."""
        o = self.opt
        clip = o.grad_clip is not None
        args = runtime.adam_args(o.lr, o.betas, o.eps, o.weight_decay, o.step_count + 1, o.grad_clip if clip else 1.0, 1.0)
        self.args_dev.copy_(args, non_blocking=True)
        seed = torch.tensor([runtime.draw_seed()], dtype=torch.int64).pin_memory()
        self.seed_dev.copy_(seed, non_blocking=True)

    def __call__(self, **batch: Tensor):
        for k, v in batch.items():
            self.static[k].copy_(v, non_blocking=True)
        if self.fresh_noise and "flow_x0" not in batch:
            self.static["flow_x0"].normal_()
            self.static["flow_t"].uniform_()
        assert self.graph is not None, "GraphedTrainStep was closed"
        self._push()
        self.graph.replay()
        self.opt.step_count += 1
        self.opt.flat.mark_updated()        # eager users of the modules re-stage their weight images from the updated arena
        return self.out
