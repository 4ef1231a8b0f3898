"""Data-parallel path on CPU (gloo, world_size 2): the bucketed all-reduce of DalleEngine over the flat
gradient buffer reproduces single-process gradients of the concatenated batch
(N-rank run on the concatenated batch == 1-rank run).  Gradients are produced by the
oracle here (no GPU in this container); the bucketing / reduction code under test is the engine's own."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dalle-mtf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

CFG = dict(n_embd=128, text_vocab_size=60, image_vocab_size=12, text_seq_len=8, image_seq_len=8, n_layers=2, n_heads=1)


def _flat_from_named(lay, named, d, V):
    """named reference-layout grads -> the engine's flat layout (q|k|v fused, vocab axis padded)."""
    flat = torch.zeros(lay.total)
    for name, shp in lay.entries:
        o = lay.offset[name]
        n = int(np.prod(shp))
        if name.endswith("attn/qkv"):
            b = name[:-3]
            a = np.concatenate([named[b + "q"], named[b + "k"], named[b + "v"]], axis=1)
        elif name == "to_logits/linear_out/kernel":
            a = np.zeros(shp, np.float32)
            a[:, :V] = named[name]
        elif name == "to_logits/linear_out/bias":
            a = np.zeros(shp, np.float32)
            a[:V] = named[name]
        else:
            a = named[name]
        flat[o:o + n] = torch.from_numpy(np.ascontiguousarray(a)).reshape(-1)
    return flat


def _worker(rank, world, init_file, out_file):
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import ParamLayout
    from src.dp import GradReducer
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    cfg = do.DalleConfig(**CFG)
    P = do.init_params(cfg, seed=3, perturb=0.05)
    B_global = 4
    text = do.synthetic_captions(B_global, cfg.text_seq_len, cfg.text_vocab_size, seed=1)
    img = do.synthetic_image_tokens(B_global, cfg.image_seq_len, cfg.image_vocab_size, seed=2)
    tokens = do.assemble_tokens(text, img, cfg.text_vocab_size)
    shard = tokens[rank * 2:(rank + 1) * 2]           # contiguous split along dim 0 (batch_dim:data)
    _, g_local = do.loss_and_grads(P, shard, cfg)     # gradient of the LOCAL mean loss
    scale = shard.shape[0] / B_global                 # engine scales dlogits by 1/(B_global*S)
    lay = ParamLayout(cfg.n_embd, cfg.n_layers, cfg.n_heads, cfg.total_tokens, cfg.total_seq_dim)
    g = _flat_from_named(lay, {k: v * np.float32(scale) for k, v in g_local.items()}, cfg.n_embd, cfg.total_tokens)
    # the engine's schedule: ready points in completion order, pieces of at most max_bucket_bytes (tiny here: several per range)
    red = GradReducer(g, world, comm=None, pg=dist.group.WORLD, max_bucket_bytes=16384)
    done = 0
    for upto in lay.ready_points:
        red.ready(done, upto)
        done = upto
    log = list(red.log)
    pending_before = len(red._pending)
    red.finish()                                        # what optimizer_step() calls first
    if rank == 0:
        _, g_full = do.loss_and_grads(P, tokens, cfg)
        ref = _flat_from_named(lay, g_full, cfg.n_embd, cfg.total_tokens)
        err = float((g - ref).abs().max())
        rel = float((g - ref).norm() / ref.norm())
        # schedule: contiguous, in order, covers [0, total), no piece above the cap, cuts at every ready point
        ok = log[0][0] == 0 and log[-1][1] == lay.total and all(a[1] == b[0] for a, b in zip(log, log[1:]))
        ok = ok and all(0 < b - a <= 4096 for a, b in log) and set(lay.ready_points) <= {b for _, b in log}
        ok = ok and lay.ready_points == sorted(lay.ready_points) and lay.ready_points[-1] == lay.total
        ok = ok and pending_before == len(log) and len(red._pending) == 0 and red.last_log == log and red.log == []
        with open(out_file, "w") as f:
            f.write(f"{err} {rel} {int(ok)}")
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out")
        mp.spawn(_worker, args=(2, init_file, out_file), nprocs=2, join=True)
        err, rel, ok = open(out_file).read().split()
        assert int(ok) == 1, "bucket schedule (order / sizes / coverage / join) is wrong"
        assert float(rel) < 1e-5 and float(err) < 1e-5, (err, rel)


def test_optimizer_never_starts_before_the_exchange_is_joined():
    """DalleEngine.optimizer_step and DiscreteVAE.optimizer_step begin with the reducer's finish(); checked on the source so
    that it holds without a GPU (the engines need one to be constructed)."""
    import inspect
    from src.dalle_mtf.engine import DalleEngine
    from src.vae_tf.models import DiscreteVAE
    src = inspect.getsource(DalleEngine.optimizer_step)
    body = src[src.index('"""', src.index('"""') + 3) + 3:]
    assert body.strip().startswith("self.wait_grads()"), body[:80]
    assert "self.reducer.finish()" in inspect.getsource(DalleEngine.wait_grads)
    src = inspect.getsource(DiscreteVAE.optimizer_step)
    body = src[src.index('"""', src.index('"""') + 3) + 3:]
    assert body.strip().startswith("self.reducer.finish()"), body[:80]


def _agree_worker(rank, world, init_file, out_file, scenario, strict):
    """scenario "init": rank 0's RCCL communicator comes up, rank 1's does not.  scenario "load": rank 1 cannot even bind
    librccl (before any blocking call).  Either way the DECISION is collective: with DALLE_DP_STRICT=0 both ranks end on the
    torch transport (rank 0 gives its communicator back); by default (strict) both ranks raise -- a split decision would
    hang the first collective, and a silent fallback would hide the failure on a real multi-GPU node."""
    import dalle_hip as dh
    from src import dp
    os.environ["DALLE_DP_STRICT"] = "1" if strict else "0"
    if strict == "default":
        del os.environ["DALLE_DP_STRICT"]
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    calls = {"destroyed": [], "init_entered": False}

    def fake_load():
        if scenario == "load" and rank == 1:
            raise dh.DalleHipError("comm_load: simulated dlopen failure on rank 1")

    def fake_init(world_, rank_, uid):
        calls["init_entered"] = True
        assert uid == b"uid-from-rank-0"
        if rank_ == 1:
            raise dh.DalleHipError("comm_init: simulated failure on rank 1")
        return 4242

    dh.comm_load = fake_load
    dh.comm_unique_id = lambda: b"uid-from-rank-0"
    dh.comm_init = fake_init
    dh.comm_destroy = lambda h: calls["destroyed"].append(h)
    handle, raised = None, None
    try:
        handle = dp.init_comm(world, rank, dist.group.WORLD)
    except RuntimeError as e:
        raised = str(e)
    torch.save({"handle": handle, "raised": raised, **calls}, f"{out_file}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,strict", [("init", False), ("init", "default"), ("load", False), ("load", True)])
def test_transport_choice_is_collective(scenario, strict):
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out")
        mp.spawn(_agree_worker, args=(2, init_file, out_file, scenario, strict), nprocs=2, join=True)
        r0, r1 = torch.load(f"{out_file}.0"), torch.load(f"{out_file}.1")
        assert r0["handle"] is None and r1["handle"] is None
        if strict:
            assert r0["raised"] and r1["raised"] and "DALLE_DP_STRICT=0" in r0["raised"]
        else:
            assert r0["raised"] is None and r1["raised"] is None
        if scenario == "init":
            assert r0["destroyed"] == [4242] and r1["destroyed"] == []
        else:      # nobody may enter the blocking communicator init when one rank could not load the library
            assert not r0["init_entered"] and not r1["init_entered"] and r0["destroyed"] == []


def _failing_rank_worker(rank, world, init_file, out_file, scenario):
    """[r06] rank 1 fails inside GradReducer.ready() on its SECOND piece; rank 0 issues its whole schedule.
    scenario "exit": rank 1's exception ends the process (what a training script does);
    scenario "stay": rank 1 catches it and stays alive without ever joining the remaining collectives (the hard case: no
    connection breaks, only the bounded wait in finish() can release rank 0)."""
    import time
    from src.dp import GradReducer
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    g = torch.full((4096,), float(rank + 1))
    red = GradReducer(g, world, comm=None, pg=dist.group.WORLD, max_bucket_bytes=4096, timeout_s=5.0)
    if rank == 1:
        real, calls = dist.all_reduce, [0]

        def flaky(*a, **k):
            calls[0] += 1
            if calls[0] == 2:
                raise RuntimeError("simulated collective failure on rank 1")
            return real(*a, **k)
        dist.all_reduce = flaky
    t0 = time.time()
    where, msg = None, ""
    try:
        red.ready(0, 2048)        # pieces 1, 2 (rank 1 dies in piece 2)
        where = "ready-2"
        red.ready(2048, 4096)     # pieces 3, 4: a failed rank must issue nothing more
        where = "finish"
        red.finish()
        where = "done"
    except RuntimeError as e:
        msg = str(e)
    issued = len(red.last_log) + len(red.log)
    # a failed reducer reports (again) from finish() and is clean afterwards
    again = None
    if rank == 1:
        again = []
        for _ in range(2):
            try:
                red.finish()
                again.append("no error")
            except RuntimeError as e:
                again.append(str(e))
    torch.save(dict(where=where, msg=msg, again=again, seconds=time.time() - t0, issued=issued,
                    pending=len(red._pending), failed=red.failed is not None), f"{out_file}.{rank}")
    if scenario == "stay" and rank == 1:
        time.sleep(12.0)          # alive, silent, never joins pieces 2..4
    os._exit(0)                   # (no destroy_process_group: it would try to drain the broken collectives)


@pytest.mark.parametrize("scenario", ["exit", "stay"])
def test_a_rank_failing_inside_ready_cannot_block_its_peers(scenario):
    """one rank's failure inside GradReducer.ready() must surface on EVERY rank inside a bounded time -- the failing
    rank raises at once, stops issuing and re-raises from finish(); its peer's finish() gives up after `timeout_s` per collective
    (or as soon as the connection breaks) with a message that names the exchange, instead of waiting in all_reduce for ever."""
    with tempfile.TemporaryDirectory() as d:
        init_file, out_file = os.path.join(d, "init"), os.path.join(d, "out")
        mp.spawn(_failing_rank_worker, args=(2, init_file, out_file, scenario), nprocs=2, join=True)
        r0, r1 = torch.load(f"{out_file}.0"), torch.load(f"{out_file}.1")
        # rank 1: raised inside its first ready(), issued exactly one collective, nothing pending afterwards
        assert r1["where"] is None and "simulated collective failure" in r1["msg"], r1
        assert r1["issued"] == 1 and not r1["failed"], r1
        assert "simulated collective failure" in r1["again"][0] and r1["again"][1] == "no error", r1     # reported once more by finish(), then clean
        # rank 0: issued all four, then finish() raised inside the bound (5 s per collective; far less when the peer's exit breaks the pair)
        assert r0["where"] == "finish" and "left the gradient exchange" in r0["msg"], r0
        assert r0["seconds"] < 30.0 and r0["pending"] == 0, r0


def test_bench_schedule_summary_and_strong_scaling_follow_the_layout():
    """bench.py's description of the exchange (`dp_piece_MB`, `dp_exposed_tail_MB`, `dp_largest_piece_MB`) against
    ParamLayout.ready_points at the BASELINE architecture, and `--scaling strong`'s batch rule -- so that the first multi-GPU line
    can be read against the layout without a GPU: pieces follow the ready points in order, none above 64 MB, the head's kernel +
    bias first (before the head's input gradient runs), the embeddings last and alone in the exposed tail."""
    import importlib.util
    from src.dalle_mtf.engine import ParamLayout
    from src.dp import GradReducer, MAX_BUCKET_BYTES
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = bench.MODELS["dalle_example"]
    d, L = cfg["n_embd"], cfg["n_layers"]
    V = cfg["text_vocab_size"] + cfg["image_vocab_size"] + 1
    S = cfg["text_seq_len"] + cfg["image_seq_len"]
    lay = ParamLayout(d, L, cfg["n_heads"], V, S)
    red = GradReducer.__new__(GradReducer)          # the piece rule only: no process group, no device
    red.max_elems = MAX_BUCKET_BYTES // 4
    sched, done = [], 0
    for upto in lay.ready_points:
        sched += red.pieces(done, upto)
        done = upto
    out = bench.dp_schedule_summary(sched, lay)
    mb = 4 / 2 ** 20
    assert sched[0][0] == 0 and sched[-1][1] == lay.total and all(a[1] == b[0] for a, b in zip(sched, sched[1:]))
    assert set(lay.ready_points) <= {b for _, b in sched} and len(lay.ready_points) == L + 2
    assert out["dp_pieces_per_step"] == len(sched) and len(out["dp_piece_MB"]) == len(sched)
    assert abs(sum(out["dp_piece_MB"]) - lay.total * mb) < 0.01 * len(sched)
    assert out["dp_largest_piece_MB"] <= 64.0 + 1e-9 and max(out["dp_piece_MB"]) <= 64.0
    # the head's kernel + bias (26.0 M + 50.8 k parameters, 99.4 MB) go first, in two pieces
    head = (lay.offset["to_logits/layer_norm/g"]) * mb
    assert abs(sum(out["dp_piece_MB"][:2]) - head) < 0.02 and 99.0 < head < 100.0
    # one piece per block (3.15 M parameters = 12.0 MB; the head LayerNorm's gain / bias ride with the last block)
    blocks = out["dp_piece_MB"][2:2 + L]
    assert all(11.9 < x < 12.2 for x in blocks), blocks
    # the exposed tail = wpe + wte = (1280 + 50771) x 512 fp32 = 101.7 MB, issued after the embedding backward
    tail = (lay.total - lay.offset["positional_embedding/wpe"]) * mb
    assert abs(out["dp_exposed_tail_MB"] - tail) < 1e-6 and abs(tail - (S + V) * d * mb) < 0.01
    assert abs(sum(out["dp_piece_MB"][2 + L:]) - tail) < 0.02
    # --scaling: weak keeps the per-GPU batch, strong keeps the global batch
    assert [bench.per_gpu_batch(0, 32, n, "weak") for n in (1, 2, 4, 8)] == [32, 32, 32, 32]
    assert [bench.per_gpu_batch(0, 32, n, "strong") for n in (1, 2, 4, 8)] == [32, 16, 8, 4]
    assert bench.per_gpu_batch(16, 32, 8, "strong") == 2
    with pytest.raises(AssertionError):
        bench.per_gpu_batch(0, 32, 3, "strong")
