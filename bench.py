#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the acoustic-model forward path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            # starts N fresh worker processes itself when N > 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                # or under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set)

A "step" is one pass of the full `AcousticModel.forward` (text encoder -> aligner front-end -> MAS -> temporal adaptor
with one flow-matching evaluation -> decoder -> to_mel) over one batch of synthetic utterances already resident in HBM.
Workload = BASELINE config 3: B = 64 utterances x 100 phonemes x 512 mel frames, bf16 - per GPU with `--scaling weak`
(default: every rank owns its own 64 utterances), or 64 in total with `--scaling strong` (64 / N per rank).  Utterances
are independent end to end, so the only exchange is ONE RCCL gather of the mel outputs per step, inside the timed
region.  `value` is measured with ONE batch in flight per GPU; the double-buffered figure is a named extra.

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline"      the dominant kernel of the step (HIP events on the launch stream): achieved algorithmic FLOP/s or B/s
                  against the gfx950 peak, PMC traffic from profiles/, every other kernel under "kernels";
  "cpu_baseline"  the oracle (CPU restatement of the reference's algorithm, oracle/) on this host's cores, N = 1 only;
  extras          two_batches_in_flight, strong_scaling (N > 1), config4 (B = 256 variable-length, sharded by cost,
                  gathered and un-permuted), and at N = 1: infer_steps4, config2_fp32, f32_parity_path,
                  fp32_alignment_chain.
No work is skipped in any timed region; synthetic weights and inputs (isp_tts_amd.synth).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# gfx950 peaks from /opt/skills/guides/MI355X_MICROARCH.md ("Chip-level parameters", dense, no sparsity)
PEAK = {"hbm_GBs": 8000.0, "mfma_f32_TFs": 157.3, "mfma_bf16_TFs": 2500.0}
FLOP_PER_FRAME = 29.0e6          # full forward, SURVEY 8(d) / BASELINE.md section 4
FLOP_PER_FRAME_INFER = 30.8e6    # infer(steps=4)
FLOP_PER_FRAME_C2 = 26.8e6       # encoder + decoder + to_mel


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch utterances per GPU; strong: --batch utterances in total (batch / N per GPU)")
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU (weak) or in total (strong)")
    ap.add_argument("--text-len", type=int, default=100)
    ap.add_argument("--mel-len", type=int, default=512)
    ap.add_argument("--dtype", choices=["f32", "bf16", "split"], default="bf16",
                    help="bf16 = BASELINE config 3 (throughput path); f32 = the 1e-4 parity path on exact-fp32 MFMAs; "
                         "split = the 1e-4 parity path on split-fp16 products (three fp16 MFMAs per fp32-grade product)")
    ap.add_argument("--alignment", choices=["auto", "f32", "split"], default="auto",
                    help="precision of the chain upstream of MAS (text encoder + aligner front-end): auto = --dtype; "
                         "f32 = fp32 whatever --dtype is (MAS paths identical to the fp32 path); split = split-fp16 products "
                         "(fp32-grade logits at a third of the fp32 chain's cost)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="batches in flight per GPU for the headline value (graph instances replayed round-robin on their "
                         "own streams); the 2-in-flight figure is always reported as an extra")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (+ roofline)")
    ap.add_argument("--three-graphs", action="store_true",
                    help="headline step as three HIP graphs on two streams (graph.SegmentedForward) instead of one")
    ap.add_argument("--linear-graph", action="store_true",
                    help="headline graph without the forward's side-stream branches (experiments: graph turnaround)")
    ap.add_argument("--extras", default="", help="comma-separated names: run only these extra lines (default: all)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (no roofline object)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=64, help="utterances in the CPU-baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--c4-batch", type=int, default=256, help="config 4: utterances in total")
    ap.add_argument("--c4-frames", type=int, default=32768, help="config 4: padded frames per micro-batch (n x M_pad)")
    ap.add_argument("--gather-dtype", choices=["f32", "bf16"], default="f32",
                    help="dtype of the mel message of the per-step RCCL gather (bf16 halves the xGMI bytes; the lengths stay int64)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ parent: start N ranks
def launch_workers(args) -> int:
    """`python bench.py --gpus N` without a launcher: this process touches no GPU API (torch is not even imported) and
    starts N fresh workers, one per GPU, with the rendezvous in the environment; rank 0's JSON line is relayed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:], "--worker"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # Rank 0's stdout (the JSON line: tens of KB with the extras, more than a pipe buffer may hold) is drained by a reader
    # thread while we poll, so rank 0 never blocks in write().  Every rank is polled: the first non-zero exit (a rank that
    # dies before the rendezvous would otherwise leave the others blocked until the process group's 300-s timeout) takes
    # the siblings down with it.
    chunks: list = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(lambda: procs[0].stdout.read(65536), "")), daemon=True)
    reader.start()
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = rc or code
    for p in live:           # only reached with rc != 0: siblings of a failed rank
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    reader.join(timeout=30)
    sys.stdout.write("".join(chunks))
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------------------------------------ helpers (worker side)
def host_cores() -> int:
    """Cores this process may actually use: affinity mask, cgroup CPU quota, and the GPU box's per-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, sd):
    """Oracle forward (the reference's algorithm restated on PyTorch-CPU + C MAS) on a bounded sample of the SAME workload:
    cpu_batch utterances of the config-3 shape, 2 warm-up + cpu_iters timed passes, median."""
    import torch
    from isp_tts_amd import synth
    from oracle import acoustic_oracle as orc      # the checker, timed as the reported baseline - never the product
    torch.set_num_threads(host_cores())
    os.environ["OMP_NUM_THREADS"] = str(host_cores())
    inp = synth.make_inputs(args.cpu_batch, args.text_len, args.mel_len)
    a = (inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"], inp["flow_x0"],
         inp["flow_t"])
    for _ in range(2):
        orc.acoustic_forward(sd, *a)
    ts = []
    for _ in range(args.cpu_iters):
        t0 = time.perf_counter()
        orc.acoustic_forward(sd, *a)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    # BASELINE config 1 beside it ("B = 64; also B = 1 for C1"): ONE utterance through the same oracle forward
    one = synth.make_inputs(1, args.text_len, args.mel_len)
    a1 = (one["text"], one["text_len"], one["mel"], one["mel_len"], one["pitch"], one["energy"], one["flow_x0"], one["flow_t"])
    for _ in range(2):
        orc.acoustic_forward(sd, *a1)
    t1 = []
    for _ in range(max(5, args.cpu_iters)):
        t0 = time.perf_counter()
        orc.acoustic_forward(sd, *a1)
        t1.append(time.perf_counter() - t0)
    med1 = sorted(t1)[len(t1) // 2]
    return {"value": round(args.cpu_batch * args.mel_len / med, 1), "unit": "mel-frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle acoustic_forward fp32, B={args.cpu_batch} x L={args.text_len} x M={args.mel_len}, "
                      f"2 warm-up + {args.cpu_iters} timed, median {med * 1e3:.0f} ms",
            "config1_single_utterance": {"value": round(args.mel_len / med1, 1), "unit": "mel-frames/s",
                                         "sample": f"BASELINE config 1: the same oracle forward, B=1 x L={args.text_len} x "
                                                   f"M={args.mel_len}, 2 warm-up + {len(t1)} timed, median {med1 * 1e3:.0f} ms"}}


def roofline(prof_summary: dict, steps: int, event_floor_us: float):
    """Per kernel label: average launch duration (HIP events on the launch stream), achieved algorithmic FLOP/s and B/s
    (both always reported), and the fraction of the roofline that BINDS it: time floor = max(FLOPs / MFMA peak,
    bytes / HBM peak).  The dominant kernel (largest summed time) is the headline object."""
    if not prof_summary:
        return None
    kernels = {}
    for label, d in sorted(prof_summary.items(), key=lambda kv: -kv[1]["total_ms"]):
        us = d["avg_us"]   # raw event-pair interval (the empty-pair interval is reported beside it, not subtracted)
        sec = us * 1e-6
        flops, nbytes = d["flops"] / d["launches"], d["bytes"] / d["launches"]
        peak_tf = PEAK["mfma_bf16_TFs"] if ("bf16" in label or "split_f16" in label) else PEAK["mfma_f32_TFs"]
        t_mfma, t_hbm = flops / (peak_tf * 1e12), nbytes / (PEAK["hbm_GBs"] * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        if label.startswith("mas_kernel"):
            # one wavefront per utterance: 64 waves on a 1,024-SIMD chip.  Its HBM traffic IS the algorithmic 6 B per cell, but
            # the time is one wave's instruction issue over M x L cells, far from either roof - labelled so
            bound = "issue"
        kernels[label] = {"launches_per_step": round(d["launches"] / steps, 2), "avg_us": round(us, 2),
                          "ms_per_step": round(us * d["launches"] / steps / 1e3, 4),
                          "TFLOPs": round(flops / sec / 1e12, 2), "mfma_frac": round(t_mfma / sec, 4),
                          "GBs": round(nbytes / sec / 1e9, 1), "hbm_frac": round(t_hbm / sec, 4), "bound": bound,
                          "frac": round(max(t_mfma, t_hbm) / sec, 4)}
    top = next(iter(kernels))
    k = kernels[top]
    if k["bound"] == "mfma":
        peak = PEAK["mfma_bf16_TFs"] if ("bf16" in top or "split_f16" in top) else PEAK["mfma_f32_TFs"]
        rl = {"kernel": top, "bound": "mfma", "achieved": k["TFLOPs"], "peak": peak, "unit": "TFLOP/s"}
    else:
        rl = {"kernel": top, "bound": "hbm", "achieved": k["GBs"], "peak": PEAK["hbm_GBs"], "unit": "GB/s"}
    rl.update({"frac": k["frac"], "traffic": traffic_from_profiles(top), "avg_us": k["avg_us"],
               "event_floor_us": round(event_floor_us, 2), "kernels": kernels, "other_kernels": other_from_profiles()})
    return rl


def _profile_json(name: str):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def traffic_from_profiles(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/traffic.json, written by
    tools/pmc_traffic.py from separate rocprofv3 --pmc runs; FETCH_SIZE doubled per the gfx950 note of the guide)."""
    entry = (_profile_json("traffic.json") or {}).get(kernel)
    return entry["hbm_bytes_per_launch"] if entry else None


def other_from_profiles():
    """Kernels of the timed region that are NOT libispk launches (ATen element-wise glue, Tensile GEMMs, copies): they
    carry no HIP-event label here, so their per-step time comes from the committed rocprofv3 kernel trace of this same
    command (profiles/other_kernels.json, written by tools/collect_profiles.py)."""
    return _profile_json("other_kernels.json")


# ------------------------------------------------------------------------------------------------ worker
def worker(args) -> int:
    import torch
    from isp_tts_amd import runtime, synth
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    from isp_tts_amd.dist import MelGatherPipeline, plan_micro_batches, unshard
    from isp_tts_amd.graph import GraphedCall, GraphedForward, GraphedForwardLanes, SegmentedForward

    # host threads = this process's CPU share (torch defaults to every hardware thread of the box: 128 OpenMP threads on a
    # 16-core quota turn each host-side tensor copy into tens of milliseconds of throttled spinning)
    torch.set_num_threads(host_cores())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # stdout carries exactly ONE line, rank 0's JSON.  Libraries write there too (RCCL prints a five-line version banner
    # when its first communicator comes up, on every rank): from here on descriptor 1 IS stderr, and the JSON line goes to
    # a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ISPK_BENCH_FORCE_DIST=1: take the distributed path (RCCL init, gather, barriers) even with one rank, so the
    # multi-GPU code can be rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get("ISPK_BENCH_FORCE_DIST") == "1"
    dist = None
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,     # nccl == RCCL on ROCm
                                timeout=datetime.timedelta(seconds=300))

    dims = AcousticDims()
    sd = synth.make_state_dict()
    model = AcousticModel.init(dims.model_config()).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.requires_grad_(False)        # an inference deployment (with trainable parameters and gradients enabled `model(...)` is the training forward)
    cdt = {"f32": torch.float32, "bf16": torch.bfloat16, "split": torch.float16}[args.dtype]
    align_dt = {"f32": torch.float32, "split": torch.float16}.get(args.alignment)
    model.set_compute_dtype(cdt, alignment_dtype=align_dt)

    L, M = args.text_len, args.mel_len
    if args.scaling == "strong":
        assert args.batch % world == 0, "--scaling strong needs --batch divisible by --gpus"
    B = args.batch if args.scaling == "weak" else args.batch // world
    gather_root = None if os.environ.get("ISPK_BENCH_ALLGATHER") == "1" else 0

    def to_dev(inp):
        return {k: v.to(dev) for k, v in inp.items()}

    def fwd_args(d):
        return (d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"])

    def fence(pipes=()):
        for p in pipes:
            p.wait()               # every submitted gather has completed
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds: float) -> float:
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(step_fn, steps, warmup, pipes=()):
        """warmup untimed steps, then EXACTLY `steps` steps between barrier + synchronize pairs; MAX over ranks."""
        for _ in range(warmup):
            step_fn()
        fence(pipes)
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        fence(pipes)
        return max_over_ranks(time.perf_counter() - t0)

    def fixed_batch_runner(batch, in_flight, seed_shift=0):
        """Config-3 style stream of fixed-shape batches: graph instance(s) + the overlapped gather. -> (step_fn, pipes, lanes)"""
        d = to_dev(synth.make_inputs(batch, L, M, seed=synth.SEED + rank + seed_shift))   # each rank: its own utterances
        pipe = MelGatherPipeline(batch, dims.mel_dim, M, dev, root=gather_root,
                                 dtype=torch.bfloat16 if args.gather_dtype == "bf16" else torch.float32) if use_dist else None
        if args.no_graph:
            def step():
                out = model(*fwd_args(d)[:6], flow_noise=d["flow_x0"], flow_time=d["flow_t"])
                if pipe is not None:
                    pipe.submit(out.mel, out.adaptor_output.dec_lengths)
                return out
            return step, ([pipe] if pipe else []), None, d
        if args.three_graphs and in_flight == 1:
            seg = SegmentedForward(model, *fwd_args(d))

            def step():
                out = seg.replay()
                if pipe is not None:
                    pipe.submit(out.mel, out.adaptor_output.dec_lengths)
                return out
            return step, ([pipe] if pipe else []), "three", d
        # several graph instances in flight: LINEAR graphs (no side-stream branches inside a forward).  Two branchy graphs
        # do not overlap under HIP graph replay (measured gain 1.03); two linear ones on their own streams do - the other
        # batch's launches then play the part of the side branches.
        branches = model.overlap_streams
        model.overlap_streams = in_flight == 1 and not args.linear_graph
        try:
            lanes = GraphedForwardLanes(model, *fwd_args(d), lanes=in_flight, calibrate=in_flight > 1)
        finally:
            model.overlap_streams = branches

        def step():
            g, stream = lanes.next_lane()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                out = g.replay()
                if pipe is not None:   # the one exchange of the path: mel outputs over xGMI (RCCL gather), overlapped
                    pipe.submit(out.mel, out.adaptor_output.dec_lengths)
            return out
        return step, ([pipe] if pipe else []), lanes, d

    # ---------------------------------------------------------------------------------------------- headline (config 3)
    step, pipes, lanes, d = fixed_batch_runner(B, max(1, args.in_flight))
    elapsed = measure(step, args.steps, args.warmup, pipes)
    out = step()
    fence(pipes)
    assert torch.isfinite(out.mel).all()
    frames = world * B * M * args.steps
    value = frames / elapsed
    line = None
    if rank == 0:
        name, cus = runtime.device_info()
        line = {
            "metric": "mel-frames/s (whole node), batch=64 x 512-frame utterances"
                      + (" per GPU" if args.scaling == "weak" else " in total") + ", full forward incl. MAS",
            "value": round(value, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 3: AcousticModel.forward (TextEncoder + Aligner/MAS + TemporalAdaptor "
                                   "1 flow eval + MelDecoder + to_mel), fixed-length synthetic random-phoneme batch, "
                                   "random-init weights of the recipe architecture (23.2 M params)",
                       "batch_per_gpu": B, "global_batch": world * B, "text_len": L, "mel_len": M,
                       "alignment_chain": "fp32 (text encoder + aligner front-end: MAS paths identical to the fp32 path)"
                       if args.alignment == "f32" or args.dtype == "f32" else
                       "text encoder + aligner front-end in the compute dtype (see accuracy_vs_fp32_path / fp32_alignment_chain)",
                       "parallelism": (f"dp{world} (utterances sharded, one RCCL "
                                       f"{'all-gather' if gather_root is None else 'gather to rank 0'} of mel per step, "
                                       "overlapped with the next step)") if world > 1 else "single GPU",
                       "device": name, "compute_units": cus,
                       "launch": ("eager" if lanes is None else "three HIP graphs per step on two streams" if lanes == "three"
                                  else "HIP graph replay"),
                       "batches_in_flight": 1 if lanes is None or lanes == "three" else len(lanes)},
            "model_TFLOPs": round(value * FLOP_PER_FRAME / 1e12, 2),
        }

    # ---------------------------------------------------------------------------------------------- roofline (rank 0)
    if rank == 0 and not args.no_kernel_events:
        # graph replays cannot carry timing events: time the SAME kernels, launched eagerly, right after the timed region
        prof = runtime.LaunchProfiler()
        runtime.set_profiler(prof)
        prof_steps = min(args.steps, 5)
        for _ in range(prof_steps):
            model(*fwd_args(d)[:6], flow_noise=d["flow_x0"], flow_time=d["flow_t"])
        torch.cuda.synchronize()
        runtime.set_profiler(None)
        line["roofline"] = roofline(prof.summary(), prof_steps, event_floor(torch))
        line["roofline"]["timing"] = (f"HIP events around every launch of {prof_steps} eager passes of the same step "
                                      "(the timed region replays them as one HIP graph)")
    if use_dist:
        dist.barrier()

    def extra(name, fn):
        """Extras never take the headline down: an exception is reported in place of the figure.  (Every rank runs the
        same extras in the same order; the ones with collectives use only what the headline already exercised.)"""
        if args.extras and name not in args.extras.split(","):
            return
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001
            res = {"error": f"{type(e).__name__}: {e}"[:300]}
        if line is not None and res is not None:
            line[name] = res
        # an extra's graph instances die HERE, with the device idle - not whenever the cyclic collector next runs (inside
        # a later capture or replay, where destroying a graph aborts / crashes the HIP runtime)
        torch.cuda.synchronize()
        gc.collect()

    if not args.no_extras and not args.no_graph:
        # ------------------------------------------------------------------------- two batches in flight (all ranks)
        lanes = pipes = step = None          # release the headline's graph instance before the next capture
        torch.cuda.synchronize()
        gc.collect()

        def two_in_flight():
            st, pp, ln, _ = fixed_batch_runner(B, 2)
            el = measure(st, args.steps, args.warmup, pp)
            return {"value": round(world * B * M * args.steps / el, 1), "unit": "mel-frames/s",
                    "ms_per_step": round(1e3 * el / args.steps, 3), "steps": args.steps,
                    "lane_overlap_gain": None if ln.overlap is None else round(ln.overlap, 3),
                    "note": "two HIP-graph instances (linear: no branches inside a forward) replayed alternately on two streams: "
                            "one batch's small text-side launches overlap the other's decoder; every step is still a complete "
                            "forward of its own batch"}
        if max(1, args.in_flight) == 1:
            extra("two_batches_in_flight", two_in_flight)

        # ------------------------------------------------------------------------- strong scaling (N > 1, weak headline)
        if world > 1 and args.scaling == "weak" and args.batch % world == 0:
            def strong():
                bs = args.batch // world
                st, pp, _, _ = fixed_batch_runner(bs, 1, seed_shift=100)
                el = measure(st, args.steps, args.warmup, pp)
                return {"value": round(args.batch * M * args.steps / el, 1), "unit": "mel-frames/s", "scaling": "strong",
                        "global_batch": args.batch, "batch_per_gpu": bs, "ms_per_step": round(1e3 * el / args.steps, 3),
                        "steps": args.steps}
            extra("strong_scaling", strong)

        # ------------------------------------------------------------------------- strong scaling predicted on ONE GPU
        if world == 1:
            def per_rank_steps():
                """The metric's "batch = 64 at 1 / 2 / 4 / 8 GPUs" is a FIXED global batch: 64 / N utterances per rank, no
                collective inside a forward and one overlapped gather of mel per step.  So an N-GPU step costs what ONE GPU
                takes for 64 / N utterances: time exactly those steps here (same graph replay, one batch in flight)."""
                res, t = {}, {}
                for n in (1, 2, 4, 8):
                    bs = args.batch // n
                    if bs < 1 or args.batch % n:
                        continue
                    st, pp, _, _ = fixed_batch_runner(bs, 1, seed_shift=200 + n)
                    el = measure(st, args.steps, args.warmup, pp)
                    t[n] = el / args.steps
                    res[f"B{bs}"] = {"gpus": n, "batch_per_gpu": bs, "ms_per_step": round(1e3 * t[n], 3),
                                     "predicted_speedup": round(t[1] / t[n], 2),
                                     "predicted_mel_frames_per_s": round(args.batch * M / t[n], 1)}
                    st = pp = None
                    torch.cuda.synchronize()
                    gc.collect()
                res["note"] = ("per-rank step times of fixed-global-batch (strong) scaling measured on ONE MI355X; the 8-GPU "
                               "prediction ignores the per-step RCCL gather (0.66 MB per rank over a direct xGMI link, overlapped "
                               "with the next step); north_star's mark is >= 3.5x at 8 GPUs")
                return res
            extra("per_rank_step_ms", per_rank_steps)

        # ------------------------------------------------------------------------- BASELINE config 4 (all ranks)
        def config4():
            Bt, Lm, Mm = args.c4_batch, 200, 1024
            full = synth.make_inputs(Bt, Lm, Mm, variable=True)
            shards, plans = plan_micro_batches(full["mel_len"].tolist(), full["text_len"].tolist(), world, args.c4_frames)
            slot = max(len(s) for s in shards)
            block = torch.zeros((slot, dims.mel_dim, Mm), dtype=torch.float32, device=dev)
            lens = torch.full((slot,), -1, dtype=torch.int64, device=dev)
            graphs, off = [], 0
            for idx, m_pad, l_pad in plans[rank]:
                ii = torch.tensor(idx)
                mb = {"text": full["text"][ii, :l_pad], "text_len": full["text_len"][ii], "mel": full["mel"][ii, :, :m_pad],
                      "mel_len": full["mel_len"][ii], "pitch": full["pitch"][ii, :m_pad], "energy": full["energy"][ii, :m_pad],
                      "flow_x0": full["flow_x0"][ii, :l_pad], "flow_t": full["flow_t"][ii]}
                g = GraphedForward(model, *fwd_args(to_dev(mb)))
                graphs.append((g, off, len(idx), m_pad))
                off += len(idx)
            pipe = MelGatherPipeline(slot, dims.mel_dim, Mm, dev, root=gather_root) if use_dist else None
            result = {}

            def st():
                for g, o, n, m_pad in graphs:
                    out_ = g.replay()
                    block[o:o + n, :, :m_pad].copy_(out_.mel)
                    lens[o:o + n].copy_(out_.adaptor_output.dec_lengths)
                if pipe is not None:
                    pipe.submit(block, lens)

            def finish():   # root: wait for the exchange, restore the original utterance order
                if pipe is not None:
                    got = pipe.wait()
                    if got is not None:
                        result["mel"], result["len"] = unshard(got[0], got[1], shards)
                else:
                    result["mel"], result["len"] = unshard(block[None], lens[None], shards)

            for _ in range(2):
                st()
            finish()
            fence()
            steps4 = max(3, args.steps // 4)
            t0 = time.perf_counter()
            for _ in range(steps4):
                st()
                finish()
            fence()
            el = max_over_ranks(time.perf_counter() - t0)
            valid = int(full["mel_len"].sum())
            ok = None
            if "len" in result:
                ok = bool(torch.equal(result["len"].cpu(), full["mel_len"]))
            return {"value": round(valid * steps4 / el, 1), "unit": "valid mel-frames/s", "scaling": "strong",
                    "global_batch": Bt, "mel_len": "U{128..1024}", "text_len": "clamp(round(mel/5.12), 25, 200)",
                    "valid_frames": valid, "padded_frames_per_rank": [sum(len(i) * m for i, m, _ in p) for p in plans],
                    "micro_batches_per_rank": [len(p) for p in plans], "ms_per_step": round(1e3 * el / steps4, 3),
                    "steps": steps4, "gathered_lengths_match": ok,
                    "workload": "BASELINE config 4: B=256 variable-length forward; cost-balanced contiguous length ranges per "
                                "rank -> micro-batches padded to their own maximum -> one gather of [slot, 80, 1024] blocks "
                                "-> original order restored on rank 0"}
        extra("config4", config4)

    # ---------------------------------------------------------------------------------------------- N = 1 extras
    if world == 1 and rank == 0 and not args.no_extras and not args.no_graph:
        def timed_graph(g, n):
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        def infer_line():
            d2 = to_dev(synth.make_inputs(B, L, M))
            dur = torch.full((B, L), M // L, dtype=torch.int64, device=dev)
            dur[:, : M - (M // L) * L] += 1                                      # every row sums to M frames
            call = lambda: model.infer(d2["text"], text_lengths=d2["text_len"], duration_target=dur, steps=4,   # noqa: E731
                                       flow_noise=d2["flow_x0"], max_dec_len=M)
            g = GraphedCall(call)
            sec = timed_graph(g, 10)
            mel, ao = g.out
            assert mel.shape == (B, dims.mel_dim, M) and bool((ao.dec_lengths == M).all())
            mel = mel.clone()
            acc = None
            if cdt != torch.float32:      # the same call on the exact-fp32 path of this build, for the accuracy figure
                model.set_compute_dtype(torch.float32)
                try:
                    ref32 = call()[0]
                    torch.cuda.synchronize()
                finally:
                    model.set_compute_dtype(cdt, alignment_dtype=align_dt)
                err = (mel - ref32).abs()
                acc = {"mel_linf_vs_fp32": float(f"{err.max().item():.3e}"),
                       "mel_rel_rms_vs_fp32": float(f"{(err.pow(2).mean().sqrt() / ref32.pow(2).mean().sqrt()).item():.3e}")}
            v = B * M / sec
            return {"value": round(v, 1), "unit": "mel-frames/s", "ms_per_step": round(sec * 1e3, 3), "steps": 10,
                    "accuracy_vs_fp32_path": acc,
                    "model_TFLOPs": round(v * FLOP_PER_FRAME_INFER / 1e12, 2),
                    "workload": f"AcousticModel.infer(steps=4, duration_target sum={M}), B={B}, {args.dtype}: 4 Euler steps of "
                                "the flow predictor, no aligner / MAS (model.py:177-238)"}
        extra("infer_steps4", infer_line)

        def alignment_chain_line():
            if args.dtype != "bf16" or args.alignment == "f32":
                return None
            model.set_compute_dtype(torch.bfloat16, alignment_dtype=torch.float32)
            try:
                g = GraphedForward(model, *fwd_args(d))
                sec = timed_graph(g, 10)
                paths = g.out.aligner_output.attn_hard.clone()
            finally:
                model.set_compute_dtype(cdt, alignment_dtype=align_dt)
            return {"value": round(B * M / sec, 1), "unit": "mel-frames/s", "ms_per_step": round(sec * 1e3, 3), "steps": 10,
                    "note": "bf16 decoder / adaptor with the text encoder in fp32 as well: everything upstream of MAS is the "
                            "fp32 path's arithmetic, so alignments and durations are bit-identical to the fp32 path's"}, paths

        def split_chain_line():
            """bf16 decoder / adaptor with the chain upstream of MAS on split-fp16 products."""
            if args.dtype != "bf16" or args.alignment != "auto":
                return None
            model.set_compute_dtype(torch.bfloat16, alignment_dtype=torch.float16)
            try:
                g = GraphedForward(model, *fwd_args(d))
                sec = timed_graph(g, 10)
                paths = g.out.aligner_output.attn_hard.clone()
            finally:
                model.set_compute_dtype(cdt, alignment_dtype=align_dt)
            return {"value": round(B * M / sec, 1), "unit": "mel-frames/s", "ms_per_step": round(sec * 1e3, 3), "steps": 10,
                    "note": "bf16 decoder / adaptor; text encoder + aligner front-end on split-fp16 products (fp32-grade logits: "
                            "the MAS paths of the parity path at a fraction of the exact-fp32 chain's cost)"}, paths

        def f32_line():
            model.set_compute_dtype(torch.float32)
            try:
                g = GraphedForward(model, *fwd_args(d))
                sec = timed_graph(g, 5)
                res = g.out
                paths, mel32 = res.aligner_output.attn_hard.clone(), res.mel.clone()
            finally:
                model.set_compute_dtype(cdt, alignment_dtype=align_dt)
            return {"value": round(B * M / sec, 1), "unit": "mel-frames/s", "ms_per_step": round(sec * 1e3, 3),
                    "steps": 5, "note": "exact-fp32 MFMA everywhere: the path that holds mel L-inf < 1e-4 vs the oracle"}, paths, mel32

        if args.dtype == "bf16":
            state = {}

            def run_f32():
                res, state["p32"], state["mel32"] = f32_line()
                return res
            extra("f32_parity_path", run_f32)

            def parity_path_line():
                """The parity-grade FAST path: every product of the forward as three fp16 MFMAs over hi / lo terms
                (`set_compute_dtype(torch.float16)`, csrc/split.hip) - holds the same bars as the exact-fp32 path."""
                model.set_compute_dtype(torch.float16)
                try:
                    g = GraphedForward(model, *fwd_args(d))
                    sec = timed_graph(g, 10)
                    o = g.out
                    res = {"value": round(B * M / sec, 1), "unit": "mel-frames/s", "ms_per_step": round(sec * 1e3, 3), "steps": 10,
                           "dtype": "split-fp16 (fp32-grade: hi/lo fp16 terms, three MFMAs per product, fp32 accumulation)",
                           "model_TFLOPs": round(B * M / sec * FLOP_PER_FRAME / 1e12, 2)}
                    if "p32" in state:
                        same = (o.aligner_output.attn_hard == state["p32"]).flatten(1).all(1)
                        res["mel_linf_vs_fp32_path"] = float(f"{(o.mel - state['mel32']).abs().max().item():.3e}")
                        res["mas_paths_identical_to_fp32_path"] = f"{int(same.sum())}/{B}"
                    prof = runtime.LaunchProfiler()
                    runtime.set_profiler(prof)
                    for _ in range(3):
                        model(*fwd_args(d)[:6], flow_noise=d["flow_x0"], flow_time=d["flow_t"])
                    torch.cuda.synchronize()
                    runtime.set_profiler(None)
                    rl = roofline(prof.summary(), 3, event_floor(torch))
                    rl.pop("other_kernels", None)
                    rl["traffic"] = None
                    rl["note"] = ("FLOPs are the algorithmic ones (2 M N K per Linear, counted ONCE - the kernel issues three "
                                  "fp16 MFMAs per product, so frac = 1/3 is this arithmetic's ceiling against the 2.5 PF peak)")
                    res["roofline"] = rl
                finally:
                    runtime.set_profiler(None)
                    model.set_compute_dtype(cdt, alignment_dtype=align_dt)
                return res
            extra("parity_path", parity_path_line)

            def run_chain():
                r = alignment_chain_line()
                if r is None:
                    return None
                res, pc = r
                if "p32" in state:
                    same = (pc == state["p32"]).flatten(1).all(1)
                    res["mas_paths_identical_to_fp32_path"] = f"{int(same.sum())}/{B}"
                return res
            extra("fp32_alignment_chain", run_chain)

            def run_split_chain():
                r = split_chain_line()
                if r is None:
                    return None
                res, pc = r
                if "p32" in state:
                    same = (pc == state["p32"]).flatten(1).all(1)
                    res["mas_paths_identical_to_fp32_path"] = f"{int(same.sum())}/{B}"
                return res
            extra("split_fp16_alignment_chain", run_split_chain)

            def accuracy():
                """the headline path's outputs against the fp32 path of the same build on the same batch"""
                if "p32" not in state:
                    return None
                g = GraphedForward(model, *fwd_args(d))
                o = g.replay()
                torch.cuda.synchronize()
                same = (o.aligner_output.attn_hard == state["p32"]).flatten(1).all(1)
                err = (o.mel - state["mel32"]).abs()
                rel = (err.pow(2).mean().sqrt() / state["mel32"].pow(2).mean().sqrt()).item()
                return {"mel_linf_vs_fp32": round(err.max().item(), 6), "mel_rel_rms_vs_fp32": round(rel, 6),
                        "mas_paths_identical_to_fp32_path": f"{int(same.sum())}/{B}",
                        "dec_lengths_equal": bool(torch.equal(o.adaptor_output.dec_lengths, d["mel_len"]))}
            extra("accuracy_vs_fp32_path", accuracy)

        def config2_line():
            """BASELINE config 2: B=32 fixed-length, TextEncoder + MelDecoder + to_mel on given activations, fp32."""
            b2 = 32
            tok = synth._normal("bench/c2/tok", (b2, L, dims.text_dim)).to(dev)
            dec_in = synth._normal("bench/c2/dec", (b2, M, dims.text_dim)).to(dev)
            model.set_compute_dtype(torch.float32)
            try:
                def run():
                    enc = model.encoder(tok).out
                    dec = model.decoder(dec_in).out
                    return enc, model._to_mel(dec, None)
                g = GraphedCall(run)
                sec = timed_graph(g, 5)
            finally:
                model.set_compute_dtype(cdt, alignment_dtype=align_dt)
            v = b2 * M / sec
            return {"value": round(v, 1), "unit": "mel-frames/s", "ms_per_step": round(sec * 1e3, 3), "steps": 5,
                    "dtype": "f32", "model_TFLOPs": round(v * FLOP_PER_FRAME_C2 / 1e12, 2),
                    "frac_of_fp32_mfma_peak": round(v * FLOP_PER_FRAME_C2 / 1e12 / PEAK["mfma_f32_TFs"], 4),
                    "workload": "BASELINE config 2: B=32 x L=100 / M=512, ALiBi-MHA + FFN kernels of TextEncoder + MelDecoder "
                                "+ to_mel, fp32"}
        extra("config2_fp32", config2_line)

        def ingest_line():
            """PCIe-inclusive rate: collated HOST batches (the reference collator's dict, pageable memory) through
            `ingest.BatchIngest` (pinned ring + copy stream) into the graphed forward; never the headline `value`."""
            from isp_tts_amd import ingest
            hb = []
            for k in range(3):
                i_ = synth.make_inputs(B, L, M, seed=synth.SEED + 50 + k)
                hb.append({"text_vector": i_["text"], "text_vector_len": i_["text_len"], "mel": i_["mel"],
                           "mel_len": i_["mel_len"], "pitch": i_["pitch"], "energy": i_["energy"]})
            nbytes = sum(t.numel() * t.element_size() for t in hb[0].values())
            g = GraphedForward(model, *fwd_args(d))
            n = max(10, args.steps)

            ing = ingest.BatchIngest(dev, B, L, M, dims.mel_dim, slots=2)     # pinned allocation: once, outside the loop

            def overlapped():
                ing.submit(hb[0])
                for k in range(n):
                    if k + 1 < n:
                        ing.submit(hb[(k + 1) % 3])
                    g(**ingest.model_inputs(ing.get()))
                    ing.done()
                torch.cuda.synchronize()

            def synchronous():          # what accelerate's device placement does: pageable .to(device) on the compute stream
                for k in range(n):
                    g(**ingest.model_inputs({k_: v.to(dev) for k_, v in hb[k % 3].items()}))
                torch.cuda.synchronize()
            res = {}
            for name, fn in (("overlapped", overlapped), ("synchronous_pageable", synchronous)):
                fn()
                t0 = time.perf_counter()
                fn()
                res[name] = (time.perf_counter() - t0) / n
            return {"value": round(B * M / res["overlapped"], 1), "unit": "mel-frames/s",
                    "ms_per_step": round(res["overlapped"] * 1e3, 3), "steps": n, "host_bytes_per_batch": nbytes,
                    "synchronous_pageable_copy_ms_per_step": round(res["synchronous_pageable"] * 1e3, 3),
                    "note": "host batches in pageable memory -> pinned slot (host memcpy on this thread) -> copy stream -> "
                            "graph's input buffers; the copy of batch k+1 runs under the forward of batch k"}
        extra("host_batches_over_pcie", ingest_line)

        if not args.no_cpu_baseline:
            extra("cpu_baseline", lambda: cpu_baseline(args, sd))
            if isinstance(line.get("cpu_baseline"), dict) and "value" in line["cpu_baseline"]:
                line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)

    # ---------------------------------------------------------------------------------------------- training step (all ranks)
    if not args.no_extras:
        def train_line():
            """one optimizer step of BASELINE config 5 under the reference's total loss (mel + flow + CTC +
            binarisation, loss.py:140-182) over every parameter of the model - text embedding, TextEncoder, aligner
            front-end, the adaptor's embedding module and flow predictor, MelDecoder, to_mel (train.acoustic_train_forward)
            - with the recipes' dropout, bf16 AMP for the Linear GEMMs, sharded flat AdamW; B utterances x M frames per
            GPU of the headline's synthetic batch."""
            from isp_tts_amd import train
            model.train()                       # training mode: the recipes' dropout on attention and feed-forward
            trained = list(model.parameters())
            for p in trained:
                p.requires_grad_(True)
            model.set_compute_dtype(torch.float32)
            opt = train.FlatAdamW(trained, lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
            opt.check_finite = False
            prof = runtime.LaunchProfiler() if rank == 0 else None

            amp = [True]

            def st():
                _, total, _ = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"],
                                                           d["energy"], flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=amp[0])
                opt.step(total)
            try:
                n = max(3, args.steps // 4)
                for _ in range(2):
                    st()
                fence()
                t0 = time.perf_counter()
                for _ in range(n):
                    st()
                fence()
                el = max_over_ranks(time.perf_counter() - t0)
                amp[0] = False                       # the all-fp32 step beside it
                for _ in range(2):
                    st()
                fence()
                t0 = time.perf_counter()
                for _ in range(n):
                    st()
                fence()
                el32 = max_over_ranks(time.perf_counter() - t0)
                amp[0] = True
                elg = None
                if world == 1:                       # the step as ONE HIP graph (train.GraphedTrainStep; single rank)
                    batch = {k: d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy", "flow_x0", "flow_t")}
                    gstep = train.GraphedTrainStep(model, opt, batch, amp=True, warmup=2)
                    for _ in range(2):
                        gstep()
                    fence()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        gstep()
                    fence()
                    elg = time.perf_counter() - t0
                    gstep.close()
                    del gstep
                kern = None
                if prof is not None:
                    runtime.set_profiler(prof)
                st()                       # EVERY rank takes this step (its collectives must match); rank 0 records it
                torch.cuda.synchronize()
                if prof is not None:
                    runtime.set_profiler(None)
                    summ = prof.summary()
                    kern = {k: {"ms": round(v["total_ms"], 3), "launches": v["launches"]} for k, v in
                            sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}
            finally:
                model.set_compute_dtype(cdt, alignment_dtype=align_dt)
                model.eval()
                model.requires_grad_(False)
            frames = world * B * M
            # forward 2 N K per Linear + attention 4 N^2 64 H; backward = 2 x the Linears' + 2.5 x the attention's
            best = el if elg is None else min(el, elg)
            res = {"value": round(frames * n / best, 1), "unit": "mel-frames/s", "ms_per_step": round(1e3 * best / n, 3), "steps": n,
                   "ms_per_step_eager": round(1e3 * el / n, 3),
                   "ms_per_step_graph": None if elg is None else round(1e3 * elg / n, 3),
                   "dtype": "bf16 operands for the Linear GEMMs (AMP), fp32 everything else and master weights",
                   "ms_per_step_all_fp32": round(1e3 * el32 / n, 3), "global_batch": world * B, "parameters": opt.flat.total,
                   "optimizer": f"flat AdamW, clip 1.0, {'reduce-scatter + all-gather over RCCL, moments sharded' if world > 1 else 'single rank'}",
                   "workload": "BASELINE config 5: mel + flow + CTC + binarisation losses, every parameter trained (text embedding, "
                               "TextEncoder, aligner, adaptor embedding module + flow predictor, MelDecoder, to_mel); forward "
                               "with the recipes' dropout + backward + clip + AdamW; ms_per_step = the whole step replayed as one HIP graph "
                               "(single rank; eager launches beside it and on more ranks)"}
            if kern:
                res["ms_by_kernel_one_step"] = kern
            return res
        extra("train_step", train_line)

    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def event_floor(torch) -> float:
    """Median interval of an empty HIP-event pair on the launch stream, in us (reported beside per-launch timings)."""
    torch.cuda.synchronize()
    ts = []
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    return v[len(v) // 2]


def main() -> int:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.worker:
        return launch_workers(args)        # no GPU API has been touched in this process
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
