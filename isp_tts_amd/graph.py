"""HIP-graph replay of the forward path.

One `AcousticModel.forward` at the benchmark shape is ~300 kernel launches (7 per transformer layer x 16 layers, plus
the aligner front-end and the adaptor's tensor algebra).  At bf16 speeds the GPU finishes them faster than the host
can issue them, so the step is captured ONCE into a HIP graph (every launch of libispk.so goes to torch's current
stream, which is the capturing stream, and the library makes no non-stream API call inside a launch) and replayed.
Inputs are static buffers: `GraphedForward.__call__` copies new inputs in, replays, and returns the static outputs.
"""
from __future__ import annotations

import gc
from typing import Optional

import torch
from torch import Tensor

from . import staging


class GraphedCall:
    """`fn()` (any sequence of launches on the current stream over tensors that stay alive and in place) captured once
    into a HIP graph; `replay()` runs it again and returns the static outputs."""

    def __init__(self, fn, warmup: int = 2):
        self._fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up on a side stream: stages weights, reserves LDS, fills the allocator
            for _ in range(warmup):
                self.out = fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # No cyclic garbage collection while the stream is capturing: collecting an unreachable OLD graph there runs
        # hipGraphDestroy inside the capture ("operation not permitted when stream is capturing": the abort recorded in
        # round 2's r2_b7.err; the SIGSEGV of r2_b6.err was a replay AFTER such a capture had gone wrong in the same
        # process).  Collect first, at a point where the device is idle, then hold the collector off.  Destroying a graph
        # BETWEEN replays of other graphs, with the device idle, is safe on this runtime (ROCm 7.2) - both cases are held
        # by tests/test_gpu_model.py::test_graph_lifetime_old_graphs_die_outside_captures_and_replays.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.out = fn()
        finally:
            if gc_was_on:
                gc.enable()
        # the graph holds raw pointers to the modules' staged weight images: remember their generation
        self._staged_at_capture = staging.replacements()

    def replay(self):
        if staging.replacements() != self._staged_at_capture:
            raise RuntimeError("a staged weight image was rebuilt after this graph was captured (parameters changed: "
                               "load() / an optimizer step); the graph points at freed memory - capture a new one")
        self.graph.replay()
        return self.out


class GraphedForward(GraphedCall):
    def __init__(self, model, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor,
                 energy: Tensor, flow_noise: Optional[Tensor] = None, flow_time: Optional[Tensor] = None,
                 warmup: int = 2):
        self.model = model
        self.static = {"text": text.clone(), "text_len": text_len.clone(), "mel": mel.clone(), "mel_len": mel_len.clone(),
                       "pitch": pitch.clone(), "energy": energy.clone()}
        b, l = text.shape
        dev = text.device
        self.static["flow_noise"] = (flow_noise.clone() if flow_noise is not None
                                     else torch.randn(b, l, 3, device=dev))
        self.static["flow_time"] = flow_time.clone() if flow_time is not None else torch.rand(b, device=dev)
        super().__init__(self._run, warmup)

    def _run(self):
        s = self.static
        with torch.no_grad():      # (a replayed forward has no tape; with gradients enabled `model(...)` is the training forward)
            return self.model(s["text"], s["text_len"], s["mel"], s["mel_len"], s["pitch"], s["energy"],
                              flow_noise=s["flow_noise"], flow_time=s["flow_time"])

    def __call__(self, **inputs: Tensor):
        for k, v in inputs.items():
            self.static[k].copy_(v)
        return self.replay()


class SegmentedForward:
    r"""One forward as THREE HIP graphs on two streams (same kernels, same results as `GraphedForward`):

        main stream:  [front: embedding, text encoder, aligner front-end, soft averages] ........ [back: embedding stack,
                                                                       \                          length regulator,
        side stream:                                                    [side: MAS, flow predictor]  decoder, to_mel] join

    Inside ONE captured graph HIP replays the side branch late: MAS and the predictor's ~45 small launches end ~0.12 ms
    after to_mel with the chip otherwise idle (profiles/r02_bf16_timeline.txt); without them the step is 0.15 ms shorter.
    As a graph of its own, launched on its own stream as soon as the front piece is queued, the branch runs under the
    decoder.  Still ONE batch in flight: the next replay's front piece is ordered behind this replay's side piece."""

    def __init__(self, model, text: Tensor, text_len: Tensor, mel: Tensor, mel_len: Tensor, pitch: Tensor, energy: Tensor,
                 flow_noise: Optional[Tensor] = None, flow_time: Optional[Tensor] = None, warmup: int = 2, side_priority: int = 0):
        b, l = text.shape
        dev = text.device
        s = self.static = {"text": text.clone(), "text_len": text_len.clone(), "mel": mel.clone(), "mel_len": mel_len.clone(),
                           "pitch": pitch.clone(), "energy": energy.clone(),
                           "flow_noise": flow_noise.clone() if flow_noise is not None else torch.randn(b, l, 3, device=dev),
                           "flow_time": flow_time.clone() if flow_time is not None else torch.rand(b, device=dev)}
        self.model, self.side = model, torch.cuda.Stream(device=dev, priority=side_priority)
        frames = mel.shape[2]
        self.g_front = GraphedCall(lambda: model.forward_front(s["text"], s["text_len"], s["mel"], s["mel_len"], s["pitch"],
                                                               s["energy"]), warmup)
        st = self.g_front.out
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.g_side = GraphedCall(lambda: model.forward_side(st, s["text_len"], s["mel_len"], s["pitch"], s["energy"],
                                                                 s["flow_noise"], s["flow_time"]), warmup)
        torch.cuda.current_stream().wait_stream(self.side)
        self.g_back = GraphedCall(lambda: model.forward_back(st, s["text_len"], s["mel_len"], frames), warmup)
        self.out = model.assemble_output(st, self.g_side.out, self.g_back.out)
        self.front_done, self.side_done = torch.cuda.Event(), torch.cuda.Event()

    def replay(self):
        main = torch.cuda.current_stream()
        self.g_front.replay()
        self.front_done.record(main)
        self.side.wait_event(self.front_done)
        with torch.cuda.stream(self.side):
            self.g_side.replay()
            self.side_done.record(self.side)
        self.g_back.replay()
        main.wait_event(self.side_done)          # the caller's stream sees every output; one batch in flight
        return self.out

    def __call__(self, **inputs: Tensor):
        for k, v in inputs.items():
            self.static[k].copy_(v)
        return self.replay()


class GraphedForwardLanes:
    """Several `GraphedForward` instances ("lanes": own static buffers, own stream) replayed round-robin, so consecutive
    batches overlap on the GPU: while one batch is in its decoder (large, chip-filling launches) the next one runs its
    text-side stacks, aligner and MAS (small, latency-bound launches that leave most CUs idle).  Every replay is a
    complete forward on its own batch; only the latency of a single batch grows (about x 1.7 with two lanes) while
    throughput rises by about 20 %.

    HIP binds a stream to one of its few hardware queues at the stream's FIRST submission, round-robin.  The lanes'
    streams are therefore used back to back at construction, before any other new stream (e.g. RCCL's) submits; with
    another stream's first submission in between, two lanes were observed to share a queue and never overlap."""

    def __init__(self, model, *inputs, lanes: int = 2, calibrate: bool = True, **kw):
        self.dev = inputs[0].device
        self.lanes = [(GraphedForward(model, *inputs, **kw), None) for _ in range(max(1, lanes))]
        self.count = 0
        self._new_streams()
        # Which hardware queue a stream lands on cannot be chosen; so measure: alternating lanes must beat one lane
        # replayed back to back.  If it does not (the lanes share a queue), take fresh streams and try again.
        self.overlap = None
        if calibrate and len(self.lanes) > 1:
            for _ in range(6):
                one, alt = self._time(False), self._time(True)
                self.overlap = one / alt
                if alt < 0.93 * one:
                    break
                self._new_streams()

    def _new_streams(self) -> None:
        self.lanes = [(g, torch.cuda.Stream(device=self.dev)) for g, _ in self.lanes]
        for _, stream in self.lanes:
            with torch.cuda.stream(stream):
                torch.zeros(1, device=self.dev)
        torch.cuda.synchronize()

    def _time(self, alternate: bool, n: int = 6) -> float:
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            g, stream = self.lanes[k % len(self.lanes) if alternate else 0]
            with torch.cuda.stream(stream):
                g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def __len__(self) -> int:
        return len(self.lanes)

    def next_lane(self):
        """(GraphedForward, stream) of the next replay; the caller copies inputs / consumes outputs on that stream."""
        lane = self.lanes[self.count % len(self.lanes)]
        self.count += 1
        return lane

    def replay(self):
        """Replays the next lane on its stream (ordered after the caller's current stream) and returns its outputs; they
        are complete once that stream - or `torch.cuda.synchronize()` - has been waited for."""
        g, stream = self.next_lane()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            return g.replay()
