"""CPU: host-side mirror of the reference interface (constructors, state_dict layout, masks), the rule that the product
never routes through the oracle or a CPU fallback, and the multi-GPU sharding/gather logic over gloo (world_size 2)."""
import os
import warnings

import pytest
import torch

from conftest import ROOT, golden_json

from isp_tts_amd import dist as idist
from isp_tts_amd import runtime, synth, utils
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.modules.constructor import Constructor
from isp_tts_amd.modules.transformer import Attention, FeedForward, Transformer, TransformerLayer
from oracle import acoustic_oracle as orc


def test_state_dict_layout_is_the_references():
    want = golden_json("state_dict_keys.json")        # dumped from the real reference model
    model = AcousticModel.init(AcousticDims().model_config())
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(got) == list(want) and got == want
    # non-persistent ALiBi slopes buffer (embeddings.py:35), AdaLN init (normalization.py:44-51)
    assert "encoder.layers.0.attention.rel_pos.slopes" not in got
    ada = model.temporal_adaptor.predictor.transformer.layers[0].attention_norm
    assert ada.weight.weight.abs().sum() == 0 and (ada.weight.bias == 1).all() and ada.bias.bias.abs().sum() == 0
    assert model.load_state_dict(synth.make_state_dict(), strict=True)
    assert model.encoder.layers[0].attention.rel_pos.slopes.flatten().tolist() == synth.alibi_default_slopes(6)


def test_constructor_init_semantics():
    class Toy(Constructor):
        def __init__(self, a: int = 1, b: int = 2):
            self.a, self.b = a, b

    t = Toy.init({"a": 5, "_name_": "ignored"}, b=7)
    assert (t.a, t.b) == (5, 7)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t = Toy.init({"a": 1, "zzz": 3})
        assert any("zzz" in str(x.message) for x in w) and t.a == 1
    with pytest.raises(RuntimeError, match="mandatory"):
        Toy.init({"a": "???"})
    ff = FeedForward.init({"inner_dim": 1024, "dropout": 0.3, "activation": "gelu"}, dim=256)
    assert ff.net[0].weight.shape == (1024, 256) and ff.net[3].weight.shape == (256, 1024) and ff.net[0].bias is None


def test_unsupported_configurations_are_refused_not_approximated():
    with pytest.raises(NotImplementedError):
        Attention(dim=384, heads=6, one_kv_head=False, alibi_pos_bias=True)
    with pytest.raises(NotImplementedError):
        Attention(dim=384, heads=6, one_kv_head=True, causal=True)
    with pytest.raises(NotImplementedError):
        FeedForward(dim=384, glu=True, activation="gelu")
    with pytest.raises(NotImplementedError):
        TransformerLayer(dim=384, attention={"heads": 6, "one_kv_head": True, "alibi_pos_bias": True},
                         feed_forward={"activation": "gelu"}, pre_norm=False)


def test_multi_speaker_model_builds_with_the_reference_key_and_forward_raises_like_the_reference():
    """model.py:93-97: `num_speakers > 0` adds `speaker_embedding` (an nn.Embedding(num_speakers, encoder.dim), xavier-uniform);
    :145-146: the reference's `forward` then reads a `speaker_encoder` no AcousticModel has - AttributeError, reproduced; only
    `infer` (:205-207) serves such a model (GPU test: tests/test_gpu_model.py)."""
    model = AcousticModel.init(dict(AcousticDims().model_config(), num_speakers=4)).eval()
    assert tuple(model.state_dict()["speaker_embedding.weight"].shape) == (4, 384)
    plain = AcousticModel.init(AcousticDims().model_config())
    assert set(model.state_dict()) - set(plain.state_dict()) == {"speaker_embedding.weight"}
    inp = synth.make_inputs(1, 16, 32)
    with pytest.raises(AttributeError, match="speaker_encoder"):
        model(inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"], speaker=torch.zeros(1, 1, dtype=torch.long))


def test_no_cpu_fallback_and_no_oracle_in_the_product():
    model = AcousticModel.init(AcousticDims().model_config()).eval()
    inp = synth.make_inputs(1, 16, 32)
    with pytest.raises(runtime.IspkError, match="GPU tensors"):
        model(inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"])
    tr = Transformer.init(AcousticDims().model_config()["encoder"], emb_dim=384)
    with pytest.raises(runtime.IspkError):
        tr(torch.zeros(1, 8, 384))
    # statically: no product source imports the oracle; dynamically: importing the whole product loads no oracle module
    import re
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "isp_tts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
    code = ("import sys, isp_tts_amd.acoustic, isp_tts_amd.dist, isp_tts_amd.modules.aligner; "
            "assert not [m for m in sys.modules if m.split('.')[0] == 'oracle']")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_mask_helpers_match_the_oracle():
    lens = torch.tensor([5, 1, 9])
    assert torch.equal(utils.get_mask_from_lengths(lens), orc.get_mask_from_lengths(lens))
    assert torch.equal(utils.get_mask_from_lengths(lens, 12), orc.get_mask_from_lengths(lens, 12))
    fl = torch.tensor([0.5, 3.25, 7.0])
    assert torch.equal(utils.get_float_mask_from_lengths(fl, 8), orc.get_float_mask_from_lengths(fl, 8))
    assert utils.min_dtype_value(torch.zeros(1)) // 2 == orc.F32_MIN // 2
    m3 = utils.get_mask_3d(torch.tensor([2, 3]), torch.tensor([4, 1]))
    assert m3.shape == (2, 3, 4) and m3[0].sum() == 8 and m3[1].sum() == 3


def test_shard_by_cost_balances_and_covers():
    _, mel_len = synth.make_lengths(256, 200, 1024, variable=True)
    for world in (1, 2, 4, 8):
        shards = idist.shard_by_cost(mel_len.tolist(), world)
        assert sorted(i for s in shards for i in s) == list(range(256))
        cost = [sum(int(mel_len[i]) * (1 + int(mel_len[i]) / 512) for i in s) for s in shards]
        assert max(cost) <= 1.05 * (sum(cost) / world), "greedy longest-first should be within 5% of perfect balance"
        for s in shards:
            assert [int(mel_len[i]) for i in s] == sorted((int(mel_len[i]) for i in s), reverse=True)


def _fake_forward(text_len, mel_len, max_m):
    """Stand-in for the per-rank model call: a deterministic mel per utterance that depends only on that utterance."""
    b = len(mel_len)
    t = torch.arange(max_m, dtype=torch.float32)[None, None, :]
    c = torch.arange(80, dtype=torch.float32)[None, :, None]
    mel = torch.sin(0.01 * t * text_len.view(b, 1, 1)) + 0.1 * c + mel_len.view(b, 1, 1) * 1e-3
    return mel * (t < mel_len.view(b, 1, 1))


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        text_len, mel_len = synth.make_lengths(13, 40, 96, variable=True)     # 13 utterances: uneven shards
        shards = idist.shard_by_cost(mel_len.tolist(), world)
        mine = torch.tensor(shards[rank], dtype=torch.int64)
        local_m = int(mel_len[mine].max())                                     # each rank pads to ITS max only
        mel = _fake_forward(text_len[mine], mel_len[mine], local_m)
        gathered, lens = idist.all_gather_mel(mel, mel_len[mine])
        full, dec = idist.unshard(gathered, lens, shards)
        want = _fake_forward(text_len, mel_len, int(mel_len.max()))
        ok = torch.equal(full, want) and torch.equal(dec, mel_len)
        # fixed-shape fast path (no MAX all-reduce), as bench.py uses it
        g2, l2 = idist.all_gather_mel(want[:2] + rank, mel_len[:2], max_frames=want.shape[2], max_batch=2)
        ok = ok and all(torch.equal(g2[r], want[:2] + r) for r in range(world))
        # the overlapped pipeline bench.py uses for N > 1: the source buffer is overwritten right after each submit (as a
        # HIP-graph replay would), three batches through two staging buffers, the last one is what wait() returns
        pipe = idist.MelGatherPipeline(2, 80, want.shape[2], "cpu")
        src, src_len = torch.empty_like(want[:2]), torch.empty(2, dtype=torch.int64)
        for step in range(3):
            src.copy_(want[:2] + rank + 10 * step)
            src_len.copy_(mel_len[:2] + step)
            pipe.submit(src, src_len)
            src.fill_(-1.0)
        g3, l3 = pipe.wait()
        ok = ok and all(torch.equal(g3[r], want[:2] + r + 20) for r in range(world))
        ok = ok and all(torch.equal(l3[r], mel_len[:2] + 2) for r in range(world))
        # ... and as a gather to one rank (bench.py's default: the north star's "RCCL gather of mel outputs")
        pipe = idist.MelGatherPipeline(2, 80, want.shape[2], "cpu", root=1)
        for step in range(3):
            src.copy_(want[:2] + rank + 10 * step)
            src_len.copy_(mel_len[:2] + step)
            pipe.submit(src, src_len)
            src.fill_(-1.0)
        res = pipe.wait()
        if rank == 1:
            ok = ok and all(torch.equal(res[0][r], want[:2] + r + 20) for r in range(world))
            ok = ok and all(torch.equal(res[1][r], mel_len[:2] + 2) for r in range(world))
        else:
            ok = ok and res is None
        # the same gather with a bf16 message (half the bytes over xGMI): values arrive rounded to bf16, lengths exact
        pipe = idist.MelGatherPipeline(2, 80, want.shape[2], "cpu", root=0, dtype=torch.bfloat16)
        pipe.submit(want[:2] + rank, mel_len[:2] + 7)
        res = pipe.wait()
        if rank == 0:
            ok = ok and res[0].dtype == torch.bfloat16
            ok = ok and all(torch.equal(res[0][r], (want[:2] + r).to(torch.bfloat16)) for r in range(world))
            ok = ok and all(torch.equal(res[1][r], mel_len[:2] + 7) for r in range(world))
        # BASELINE config 4 as bench.py runs it: cost-balanced shards cut into micro-batches that are padded to their OWN
        # maximum, outputs written into one fixed [slot, 80, M_max] block per rank, ONE gather of the blocks, unshard
        tl4, ml4 = synth.make_lengths(21, 40, 96, variable=True, seed=5)
        shards4, plans4 = idist.plan_micro_batches(ml4.tolist(), tl4.tolist(), world, frame_budget=4 * 96)
        slot = max(len(s) for s in shards4)
        block, blens = torch.zeros(slot, 80, 96), torch.full((slot,), -1, dtype=torch.int64)
        off = 0
        for idx, m_pad, l_pad in plans4[rank]:
            ii = torch.tensor(idx)
            assert m_pad >= int(ml4[ii].max()) and m_pad % 8 == 0 and l_pad >= int(tl4[ii].max())
            block[off:off + len(idx), :, :m_pad] = _fake_forward(tl4[ii], ml4[ii], m_pad)
            blens[off:off + len(idx)] = ml4[ii]
            off += len(idx)
        pipe = idist.MelGatherPipeline(slot, 80, 96, "cpu", root=0)
        pipe.submit(block, blens)
        res = pipe.wait()
        if rank == 0:
            full4, dec4 = idist.unshard(res[0], res[1], shards4)
            ok = ok and torch.equal(full4, _fake_forward(tl4, ml4, 96)) and torch.equal(dec4, ml4)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_shard_gather_unshard_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert res == [(0, True), (1, True)]


def test_plan_micro_batches_covers_every_utterance_once_and_balances_padded_cost():
    text_len, mel_len = synth.make_lengths(256, 200, 1024, variable=True)
    for world in (1, 2, 4, 8):
        shards, plans = idist.plan_micro_batches(mel_len.tolist(), text_len.tolist(), world, frame_budget=32768)
        assert sorted(i for p in plans for idx, _, _ in p for i in idx) == list(range(256))
        for r, p in enumerate(plans):
            assert [i for idx, _, _ in p for i in idx] == shards[r]
            lens = [int(mel_len[i]) for i in shards[r]]
            assert lens == sorted(lens, reverse=True)
            for idx, m_pad, l_pad in p:
                assert len(idx) * m_pad <= 32768 and m_pad <= 1024 and l_pad <= 200
                assert m_pad - max(int(mel_len[i]) for i in idx) < 8 and l_pad - max(int(text_len[i]) for i in idx) < 4
        # contiguous length ranges: a rank pads to a maximum close to its own members ...
        padded = sum(len(idx) * m for p in plans for idx, m, _ in p)
        assert padded <= 1.2 * int(mel_len.sum())
        # ... and the ranks' PADDED costs (what they really run) are balanced
        cost = [sum(len(idx) * m * (1 + m / 512) for idx, m, _ in p) for p in plans]
        assert max(cost) <= 1.05 * sum(cost) / world
    # fewer utterances than ranks: one each, the rest idle
    shards = idist.shard_by_length_range([96, 17, 14, 78, 74], 8)
    assert [len(s) for s in shards] == [1, 1, 1, 1, 1, 0, 0, 0] and sorted(i for s in shards for i in s) == [0, 1, 2, 3, 4]


def test_bench_starts_fresh_worker_processes_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent (which imports no torch and touches no
    GPU API) starts two workers and relays their exit status.  On this GPU-less box each worker stops at its own
    "needs a GPU" assertion - two of them prove both ranks were started with a rendezvous in their environment."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also holds on a GPU box: the workers must see no device
    env["CUDA_VISIBLE_DEVICES"] = ""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("AssertionError: bench.py needs a GPU") == 2 and r.stdout.strip() == ""


def test_from_pretrained_reads_the_reference_checkpoint_layout(tmp_path):
    """A file laid out as the reference's Trainer writes it (trainer.py:361-372: checkpoint["model"] = {"config",
    "state_dict"}) restores the same weights (base.py:39-56); keys the file lacks keep their initial value."""
    import torch
    cfg = AcousticDims().model_config()
    src = AcousticModel.init(cfg)
    with torch.no_grad():
        for i, p in enumerate(src.parameters()):
            p.add_(0.01 * ((i % 7) - 3))
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    dropped = "decoder.norm.bias"
    del sd[dropped]
    path = tmp_path / "ckpt.pt"
    torch.save({"epoch": 3, "iteration": 1000, "model": {"config": cfg, "state_dict": sd}, "optimizer": None}, path)
    model = AcousticModel.from_pretrained(str(path))
    got = model.state_dict()
    assert set(got) == set(src.state_dict())
    for k, v in src.state_dict().items():
        if k != dropped:
            assert torch.equal(got[k], v), k
    assert torch.equal(got[dropped], torch.zeros_like(got[dropped]))       # nn.LayerNorm's initial bias


def test_load_filters_keys_like_the_reference_and_freeze_sets_requires_grad():
    import torch
    model = AcousticModel.init(AcousticDims().model_config())
    before = {k: v.clone() for k, v in model.state_dict().items()}
    sd = {k: v + 1.0 if v.is_floating_point() else v for k, v in before.items()}
    sd["not.a.key"] = torch.zeros(3)
    sd["to_mel.weight"] = torch.zeros(81, 384)                             # wrong shape
    with pytest.warns(UserWarning, match="not.a.key"):
        model.load(sd, ignore_layers=["text_embedding"], ignore_mismatched_keys=True)
    after = model.state_dict()
    assert torch.equal(after["to_mel.weight"], before["to_mel.weight"])                      # mismatched: kept
    assert torch.equal(after["text_embedding.weight"], before["text_embedding.weight"])      # ignored layer: kept
    assert torch.equal(after["to_mel.bias"], before["to_mel.bias"] + 1.0)                    # everything else: loaded
    model.freeze(exception_list=["decoder.layers.5", "to_mel"])
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    assert trainable and all(n.startswith(("decoder.layers.5", "to_mel")) for n in trainable)
    assert any(n.startswith("to_mel") for n in trainable)


def test_bucket_by_length_and_batch_ingest_on_cpu():
    """Row f4 host logic: length buckets cover every sample once with little padding; BatchIngest hands back exactly what
    the collator-layout batch held, slot after slot (CPU tensors: pinned memory / streams need the GPU box)."""
    from isp_tts_amd import ingest
    text_len, mel_len = synth.make_lengths(300, 200, 1024, variable=True, seed=9)
    buckets = ingest.bucket_by_length(mel_len.tolist(), frame_budget=64 * 512)
    assert sorted(i for b in buckets for i in b) == list(range(300))
    padded = sum(len(b) * int(mel_len[b[0]]) for b in buckets)
    assert all(len(b) * int(mel_len[b[0]]) <= 64 * 512 for b in buckets) and padded <= 1.15 * int(mel_len.sum())
    assert all(int(mel_len[b[0]]) == max(int(mel_len[i]) for i in b) for b in buckets)
    assert max(len(b) for b in ingest.bucket_by_length(mel_len.tolist(), 64 * 512, max_batch=40)) <= 40

    ing = ingest.BatchIngest("cpu", max_batch=8, max_text=40, max_mel=96, slots=2)
    batches = []
    for k, (b, l, m) in enumerate([(8, 40, 96), (3, 17, 50), (5, 40, 33)]):
        inp = synth.make_inputs(b, l, m, variable=True, seed=k)
        batches.append({"text_vector": inp["text"], "text_vector_len": inp["text_len"], "mel": inp["mel"],
                        "mel_len": inp["mel_len"], "pitch": inp["pitch"], "energy": inp["energy"]})
    ing.submit(batches[0])
    ing.submit(batches[1])
    with pytest.raises(AssertionError):
        ing.submit(batches[2])                       # both slots still hold batches nobody has taken
    for k in range(3):
        got = ing.get()
        for name, t in batches[k].items():
            assert got[name].shape == t.shape and torch.equal(got[name], t), name
        kw = ingest.model_inputs(got)
        assert set(kw) == {"text", "text_len", "mel", "mel_len", "pitch", "energy"} and kw["text"] is got["text_vector"]
        ing.done()
        if k == 0:
            ing.submit(batches[2])


# ------------------------------------------------------------------------------------------------ training step (row f2)
def _train_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(48, 32), (5,), (64, 48), (32,), (1, 8, 1), (6, 1, 1), (13, 3)]
    return [torch.nn.Parameter(torch.randn(s, generator=g) * 0.3) for s in shapes]


def _train_grads(step, rank, params):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return [torch.randn(p.shape, generator=g) * (0.3 if step % 2 else 0.01) for p in params]


def test_flat_parameters_layout_and_flat_adamw_flow_on_cpu():
    """Host logic of the optimizer (no kernel: the oracle's flat AdamW is plugged into the hooks): the arena orders the
    weight-decay group first exactly as torch's param_groups do (optimizers.py:15-20, :34-40), parameters / gradients are
    views of it, and `step` = clip group 0 + AdamW + zero_grad reproduces torch.optim.AdamW step for step."""
    from isp_tts_amd import train
    from oracle import train_oracle as torc
    ref_p, my_p = _train_params(), _train_params()
    flat = train.FlatParameters(_train_params())
    wd, no_wd = torc.group_weight_decayable_params(ref_p)
    assert [tuple(p.shape) for p in flat.params] == [tuple(p.shape) for p in wd + no_wd]
    assert flat.n_decay_tensors == 3 and flat.n_decay % 64 == 0 and all(o % 64 == 0 for o in flat.offsets)
    assert flat.n_decay == sum((p.numel() + 63) // 64 * 64 for p in wd) and flat.total % 64 == 0
    for p, o in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.data.data_ptr() + 4 * o and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
    with pytest.raises(runtime.IspkError):          # the product's update is the HIP kernel: CPU arenas fail loudly
        train.FlatAdamW(_train_params(), lr=1e-3).step()

    ref = torc.reference_optimizer(ref_p, lr=2e-3, weight_decay=1e-2)
    opt = train.FlatAdamW(my_p, lr=2e-3, weight_decay=1e-2, grad_clip=1.0, update=torc.adamw_flat, sqnorm=torc.sqnorm_flat)
    for step in range(5):
        for a, b, g in zip(ref_p, my_p, _train_grads(step, 0, ref_p)):
            a.grad = g.clone()
            b.grad.copy_(g)
        versions = [p._version for p in my_p]
        n_ref, n_my = torc.reference_step(ref, 1.0), opt.step()
        assert abs(float(n_ref) - float(n_my)) < 1e-5 * float(n_ref)
        assert all(p._version > v for p, v in zip(my_p, versions)), "staged weight images key on the version counter"
        for a, b in zip(ref_p, my_p):
            assert (a - b).abs().max() < 1e-6 and float(b.grad.abs().max()) == 0.0
    # weight_decay == 0: ONE group (optimizers.py:34), so the clip covers every parameter
    one = train.FlatAdamW(_train_params(), lr=1e-3, weight_decay=0., update=torc.adamw_flat, sqnorm=torc.sqnorm_flat)
    assert one.flat.n_decay_tensors == 7 and len(one.get_last_lr()) == 1
    opt.anneal_on_epoch_end()
    assert abs(opt.lr - 2e-3 * 0.995) < 1e-12 and opt.state_dict()["lr_scheduler"]["last_epoch"] == 1


def _train_rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from isp_tts_amd import train
    from oracle import train_oracle as torc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        my_p, ref_p = _train_params(), _train_params()
        opt = train.FlatAdamW(my_p, lr=2e-3, weight_decay=1e-2, grad_clip=1.0, update=torc.adamw_flat, sqnorm=torc.sqnorm_flat)
        ref = torc.reference_optimizer(ref_p, lr=2e-3, weight_decay=1e-2)
        assert opt.world == world and opt.shard * world == opt.flat.total and opt.exp_avg.numel() == opt.shard
        ok = True
        for step in range(4):
            mine = _train_grads(step, rank, my_p)
            for p, g in zip(my_p, mine):
                p.grad.copy_(g)
            every = [_train_grads(step, r, ref_p) for r in range(world)]
            for i, p in enumerate(ref_p):                      # what DDP hands the reference: the mean over ranks
                p.grad = sum(e[i] for e in every) / world
            n_ref, n_my = torc.reference_step(ref, 1.0), opt.step()
            ok &= abs(float(n_ref) - float(n_my)) < 1e-5 * float(n_ref)
            ok &= all(float((a - b).abs().max()) < 1e-6 for a, b in zip(ref_p, my_p))
        sd = opt.state_dict()["optimizer"]["state"]               # sharded moments gathered into torch's layout
        rsd = ref.state_dict()["state"]
        ok &= all(float((sd[i]["exp_avg"] - rsd[i]["exp_avg"]).abs().max()) < 1e-6 for i in range(len(my_p)))
        # tensors frozen AFTER the optimizer was built (`model.freeze()`): torch's AdamW skips a tensor whose .grad is None - no
        # decay, no moment update; the flat update must leave value and (sharded) moments alone too.  Frozen here: a decay-group
        # matrix (its arena range straddles the two ranks' shards) and a no-decay vector.
        frozen = (2, 3)
        for i in frozen:
            my_p[i].requires_grad_(False)
        for step in range(4, 7):
            mine = _train_grads(step, rank, my_p)
            for i, (p, g) in enumerate(zip(my_p, mine)):
                if i not in frozen:                              # (what the backward delivers: nothing for a frozen tensor)
                    p.grad.copy_(g)
            every = [_train_grads(step, r, ref_p) for r in range(world)]
            for i, p in enumerate(ref_p):
                p.grad = None if i in frozen else sum(e[i] for e in every) / world
            n_ref, n_my = torc.reference_step(ref, 1.0), opt.step()
            ok &= abs(float(n_ref) - float(n_my)) < 1e-5 * float(n_ref)
            ok &= all(float((a - b).abs().max()) < 1e-6 for a, b in zip(ref_p, my_p))
        sd, rsd = opt.state_dict()["optimizer"]["state"], ref.state_dict()["state"]
        ok &= all(float((sd[i]["exp_avg"] - rsd[i]["exp_avg"]).abs().max()) < 1e-6 and
                  float((sd[i]["exp_avg_sq"] - rsd[i]["exp_avg_sq"]).abs().max()) < 1e-6 for i in range(len(my_p)))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_adamw_world_size_2_gloo_matches_ddp_plus_adamw():
    """N > 1 path of the training step: gradient arena summed across ranks, each rank updates its slice with its slice of
    the moments, parameters all-gathered - equal to averaged gradients + a replicated torch AdamW on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_train_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=10) for _ in range(2)) == [(0, True), (1, True)]


def test_bench_launcher_takes_siblings_down_when_a_rank_fails(tmp_path):
    """`launch_workers`: a rank that exits non-zero before the rendezvous must not leave its siblings waiting for the process
    group's timeout - the parent polls every rank and terminates the others.  Driven with a stand-in worker script (rank 1
    fails at once, rank 0 would sleep for minutes)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "bench.py"
    src = open(os.path.join(root, "bench.py")).read()
    # the same launcher code, with the worker replaced by a stub
    src = src.replace("def worker(args) -> int:", "def worker(args) -> int:\n    import time as _t\n    if os.environ['RANK'] == '1':\n"
                      "        return 7\n    _t.sleep(600)\n    return 0\n\n\ndef _unused_worker(args) -> int:")
    fake.write_text(src)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, str(fake), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and time.time() - t0 < 60
