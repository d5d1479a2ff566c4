#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the acoustic-model forward path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the full `AcousticModel.forward` (encoder -> aligner front-end -> MAS -> temporal adaptor with
one flow-matching evaluation -> decoder -> to_mel) over one batch of synthetic fixed-length utterances that is already
resident in HBM: BASELINE config 3, B = 64 utterances x 100 phonemes x 512 mel frames PER GPU (weak scaling: each rank
owns its own utterances — they are independent end to end — and the only exchange is one RCCL all-gather of the mel
outputs per step, inside the timed region).  Synthetic weights and inputs (isp_tts_amd.synth); no work is skipped.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     the dominant kernel of the timed region (by summed HIP-event time on the launch stream): achieved
                  algorithmic FLOP/s (or B/s) per launch vs the gfx950 peak; "kernels" lists every timed kernel;
  "cpu_baseline": the oracle (CPU restatement of the reference algorithm, oracle/) timed on this host's cores on a
                  bounded sample of the same workload (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from isp_tts_amd import runtime, synth  # noqa: E402
from isp_tts_amd.acoustic import AcousticModel  # noqa: E402
from isp_tts_amd.config import AcousticDims  # noqa: E402
from isp_tts_amd.dist import MelGatherPipeline  # noqa: E402
from isp_tts_amd.graph import GraphedForward, GraphedForwardLanes  # noqa: E402

# gfx950 peaks from /opt/skills/guides/MI355X_MICROARCH.md ("Chip-level parameters", dense, no sparsity)
PEAK = {"hbm_GBs": 8000.0, "mfma_f32_TFs": 157.3, "mfma_bf16_TFs": 2500.0}
FLOP_PER_FRAME = 29.0e6  # full forward, SURVEY 8(d) / BASELINE.md section 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--text-len", type=int, default=100)
    ap.add_argument("--mel-len", type=int, default=512)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="bf16 = BASELINE config 3 (throughput path); f32 = the 1e-4 parity path")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the short fp32 parity-path measurement (N=1 only)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (no roofline object)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU: consecutive steps alternate between this many HIP-graph instances on "
                         "their own streams, so one batch's small text-side launches overlap another's decoder (1 = serial)")
    ap.add_argument("--cpu-batch", type=int, default=16, help="utterances in the CPU-baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=5)
    return ap.parse_args()


def host_cores() -> int:
    """Cores this process may actually use: affinity mask, cgroup CPU quota, and the GPU box's per-GPU share (16)."""
    if os.environ.get("ISPK_CPU_THREADS"):
        return int(os.environ["ISPK_CPU_THREADS"])
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(args, sd):
    """Oracle forward (reference algorithm restated on PyTorch-CPU + C MAS) on a bounded sample: cpu_batch utterances
    of the same shape, 1 warm-up + cpu_iters timed passes, median."""
    from oracle import acoustic_oracle as orc
    torch.set_num_threads(host_cores())
    os.environ["OMP_NUM_THREADS"] = str(host_cores())
    inp = synth.make_inputs(args.cpu_batch, args.text_len, args.mel_len)
    a = (inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"], inp["flow_x0"],
         inp["flow_t"])
    orc.acoustic_forward(sd, *a)
    ts = []
    for _ in range(args.cpu_iters):
        t0 = time.perf_counter()
        orc.acoustic_forward(sd, *a)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return {"value": args.cpu_batch * args.mel_len / med, "unit": "mel-frames/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle acoustic_forward fp32, B={args.cpu_batch} x L={args.text_len} x M={args.mel_len}, "
                      f"1 warm-up + {args.cpu_iters} timed, median {med * 1e3:.0f} ms"}


def roofline(prof_summary: dict, steps: int, event_floor_us: float):
    """Per kernel label: average launch duration (HIP events on the launch stream), achieved algorithmic
    FLOP/s and B/s, and the fraction of the roofline that BINDS it: time floor = max(FLOPs / MFMA peak, bytes / HBM peak).
    The dominant kernel (largest summed time) is the headline object."""
    if not prof_summary:
        return None
    kernels = {}
    for label, d in sorted(prof_summary.items(), key=lambda kv: -kv[1]["total_ms"]):
        us = d["avg_us"]   # raw event-pair interval (the empty-pair interval is reported beside it, not subtracted:
        #                    inside a busy stream the real overhead is ~1-2 us and rocprofv3 agrees with the raw value)
        sec = us * 1e-6
        flops, nbytes = d["flops"] / d["launches"], d["bytes"] / d["launches"]
        peak_tf = PEAK["mfma_bf16_TFs"] if "bf16" in label else PEAK["mfma_f32_TFs"]
        t_mfma, t_hbm = flops / (peak_tf * 1e12), nbytes / (PEAK["hbm_GBs"] * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        kernels[label] = {"launches_per_step": round(d["launches"] / steps, 2), "avg_us": round(us, 2),
                          "ms_per_step": round(us * d["launches"] / steps / 1e3, 4),
                          "TFLOPs": round(flops / sec / 1e12, 2), "GBs": round(nbytes / sec / 1e9, 1), "bound": bound,
                          "frac": round(max(t_mfma, t_hbm) / sec, 4)}
    top = next(iter(kernels))
    k = kernels[top]
    if k["bound"] == "mfma":
        peak = PEAK["mfma_bf16_TFs"] if "bf16" in top else PEAK["mfma_f32_TFs"]
        rl = {"kernel": top, "bound": "mfma", "achieved": k["TFLOPs"], "peak": peak, "unit": "TFLOP/s"}
    else:
        rl = {"kernel": top, "bound": "hbm", "achieved": k["GBs"], "peak": PEAK["hbm_GBs"], "unit": "GB/s"}
    rl.update({"frac": k["frac"], "traffic": traffic_from_profiles(top), "avg_us": k["avg_us"],
               "event_floor_us": round(event_floor_us, 2), "kernels": kernels})
    return rl


def traffic_from_profiles(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/traffic.json, written by
    tools/pmc_traffic.py from separate rocprofv3 --pmc runs; FETCH_SIZE doubled per the gfx950 note of the guide)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(kernel)
            return entry["hbm_bytes_per_launch"] if entry else None
    except (OSError, ValueError):
        return None


def event_floor() -> float:
    """Median interval of an empty HIP-event pair on the launch stream, in us (subtracted from per-launch timings)."""
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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ISPK_BENCH_FORCE_DIST=1: take the distributed path (RCCL init, all-gather, barriers) even with one rank — lets the
    # multi-GPU code be rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get("ISPK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    sd = synth.make_state_dict()
    model = AcousticModel.init(AcousticDims().model_config()).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.set_compute_dtype(torch.float32 if args.dtype == "f32" else torch.bfloat16)

    B, L, M = args.batch, args.text_len, args.mel_len
    inp = synth.make_inputs(B, L, M, seed=synth.SEED + rank)           # each rank owns different utterances
    d = {k: v.to(dev) for k, v in inp.items()}

    def eager_step():
        return model(d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                     flow_noise=d["flow_x0"], flow_time=d["flow_t"])

    graphed = None
    lanes = None
    if not args.no_graph:   # the whole forward as one HIP graph: ~300 launches per step would otherwise be host-bound
        # `--in-flight` graph instances (own static buffers, own stream): step k replays instance k % n, so consecutive
        # batches overlap on the GPU - every step still runs the complete forward on its own batch of B utterances
        lanes = GraphedForwardLanes(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                    d["flow_x0"], d["flow_t"], lanes=args.in_flight)
        graphed = lanes.lanes[0][0]

    # the one exchange of the path: every rank's mel outputs gathered on rank 0 (ISPK_BENCH_ALLGATHER=1: on every rank)
    gather_root = None if os.environ.get("ISPK_BENCH_ALLGATHER") == "1" else 0
    gather = MelGatherPipeline(B, AcousticDims().mel_dim, M, dev, root=gather_root) if use_dist else None

    def step():
        if lanes is not None:
            g, stream = lanes.next_lane()
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                out = g.replay()
                if use_dist:   # the one exchange of the path: mel outputs over xGMI (RCCL gather), overlapped too
                    gather.submit(out.mel, out.adaptor_output.dec_lengths)
            return out
        out = eager_step()
        if use_dist:
            gather.submit(out.mel, out.adaptor_output.dec_lengths)
        return out

    def fence():
        if use_dist:
            gather.wait()          # every submitted gather has completed
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = None
    if not args.no_kernel_events and rank == 0 and graphed is None:
        prof = runtime.LaunchProfiler()          # eager mode: events around every launch of the timed region
        runtime.set_profiler(prof)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    runtime.set_profiler(None)
    assert torch.isfinite(out.mel).all()
    prof_steps = args.steps
    if not args.no_kernel_events and rank == 0 and graphed is not None:
        # graph replays cannot carry timing events: time the SAME kernels, launched eagerly, right after the timed region
        prof = runtime.LaunchProfiler()
        runtime.set_profiler(prof)
        prof_steps = min(args.steps, 5)
        for _ in range(prof_steps):
            eager_step()
        torch.cuda.synchronize()
        runtime.set_profiler(None)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        frames = world * B * M * args.steps
        value = frames / elapsed
        name, cus = runtime.device_info()
        line = {
            "metric": "mel-frames/s (whole node), batch=64 x 512-frame utterances per GPU, full forward incl. MAS",
            "value": round(value, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 3: AcousticModel.forward (TextEncoder + Aligner/MAS + TemporalAdaptor "
                                   "1 flow eval + MelDecoder + to_mel), fixed-length synthetic random-phoneme batch, "
                                   "random-init weights of the recipe architecture (23.2 M params)",
                       "batch_per_gpu": B, "global_batch": world * B, "text_len": L, "mel_len": M,
                       "parallelism": f"dp{world} (utterances sharded, RCCL {'all-gather' if gather_root is None else 'gather to rank 0'} of mel overlapped with the next step)" if world > 1 else "single GPU",
                       "device": name, "compute_units": cus,
                       "launch": "eager" if graphed is None else "HIP graph replay",
                       "batches_in_flight": len(lanes) if lanes is not None else 1,
                       "lane_overlap_gain": None if lanes is None or lanes.overlap is None else round(lanes.overlap, 3)},
            "model_TFLOPs": round(value * FLOP_PER_FRAME / 1e12, 2),
        }
        if prof is not None:
            line["roofline"] = roofline(prof.summary(), prof_steps, event_floor())
            line["roofline"]["timing"] = ("HIP events around every launch of the timed region" if graphed is None else
                                          f"HIP events around every launch of {prof_steps} eager passes of the same step "
                                          "(the timed region replays them as one HIP graph)")
        if world == 1 and lanes is not None and len(lanes) > 1:
            # for transparency: the same step with ONE batch in flight (lane 0 replayed back to back) - the latency-oriented
            # figure; `value` above is the throughput with len(lanes) batches in flight
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(10):
                graphed.replay()
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - ts) / 10 * 1e3
            line["one_batch_in_flight"] = {"value": round(B * M / ms1 * 1e3, 1), "unit": "mel-frames/s",
                                           "ms_per_step": round(ms1, 3), "steps": 10}
        if world == 1 and not args.no_f32_line and args.dtype == "bf16":
            # the same step on the fp32 parity path (exact-fp32 MFMA; the path that holds mel L-inf < 1e-4), 5 replays
            model.set_compute_dtype(torch.float32)
            g32 = GraphedForward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                 d["flow_x0"], d["flow_t"])
            g32.replay()
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            for _ in range(5):
                g32.replay()
            torch.cuda.synchronize()
            ms32 = (time.perf_counter() - t32) / 5 * 1e3
            line["f32_parity_path"] = {"value": round(B * M / ms32 * 1e3, 1), "unit": "mel-frames/s",
                                       "ms_per_step": round(ms32, 3), "steps": 5}
            model.set_compute_dtype(torch.bfloat16)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, sd)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
