#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json: train tokens/sec (text+image) per node on the
`dalle_example` transformer step (n_embd=512, 6 layers, 4 heads, seq 256+1024, bf16 compute),
synthetic captions + synthetic image-token ids, B=32 per GPU (weak scaling), one process per GPU.

A "step" = forward + backward + gradient all-reduce (RCCL behind the C ABI, N>1) + global-norm clip + Adam, nothing
skipped.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0 (contract in the task
statement) with two extra objects:
  roofline     -- the dominant single launch of the step: for the DALL-E models the vocabulary projection with the fused
                  softmax epilogue (gemm_nt4_kernel<65>: flags BIAS | SOFTMAX; dispatch `dmi_gemm_nt_softmax`), algorithmic
                  FLOPs 2*M*d*V per launch / mean launch duration measured live with HIP events on the launch stream inside
                  the timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak; plus the whole-step MFMA fraction
                  (train FLOPs of SURVEY.md §8(d)).  `traffic` comes from profiles/<round>_traffic_vocab_gemm.json (separate
                  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS kernel, tools/traffic_summary.py) and is null when
                  that file does not name the kernel that ran.
  cpu_baseline -- the CPU oracle (a restatement of the reference, NOT mesh-tensorflow, which cannot run
                  here) timed on this host's cores on a bounded sample of the same workload.
--model vae_example / vae_coco time the discrete-VAE train step (BASELINE.json configs 1 and 4; 32 / 16 images per GPU):
metric = image tokens/s (images/s x grid^2), roofline = the heaviest convolution launch (implicit-im2col MFMA GEMM).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dalle-mtf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

CFG = dict(n_embd=512, n_layers=6, n_heads=4, text_vocab_size=50258, image_vocab_size=512, text_seq_len=256,
           image_seq_len=1024)
MODELS = {  # --model: the default is BASELINE.json's metric config; "1.3B" is SURVEY.md §8(d) C5 (secondary datapoint)
    "dalle_example": CFG,
    "dalle_coco": dict(n_embd=1024, n_layers=12, n_heads=8, text_vocab_size=50258, image_vocab_size=2048, text_seq_len=256,
                       image_seq_len=1024),   # configs/dalle_coco.json (16 sequences per GPU on 8 GPUs: --batch 16)
    "1.3B": dict(n_embd=2048, n_layers=24, n_heads=16, text_vocab_size=50258, image_vocab_size=512, text_seq_len=256,
                 image_seq_len=1024),
}
VAE_MODELS = {"vae_example": 32, "vae_coco": 16}     # per-GPU batch (128 images on 8 GPUs)
HP = dict(lr=1e-3, train_steps=100000, warmup_steps=3000, gradient_clipping=1.0)
PER_GPU_BATCH = 32
PEAK_BF16_TFLOPS = 2500.0
ROUND_TAG = "r06"      # profiles/<ROUND_TAG>_traffic_*.json must come from this round's kernels


def fwd_flops_per_token(d, L, S, V):
    return L * (24 * d * d + 4 * S * d) + 2 * d * V


def synth_tokens(B, T, P, text_vocab, image_vocab, seed):
    rng = np.random.default_rng(seed)
    pad = text_vocab - 1
    out = np.full((B, T + P), pad, dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(1, T + 1))
        out[b, :n] = rng.integers(0, pad, size=n, dtype=np.int32)
    out[:, T:] = rng.integers(0, image_vocab, size=(B, P), dtype=np.int32) + text_vocab
    return out


def reference_baseline():
    """the unmodified mesh-tensorflow reference timed on this host, when it can run here at all (tools/ref_probe.py:
    `import tensorflow, mesh_tensorflow` + a reference checkout); None otherwise -- the expected case."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import ref_probe
        st = ref_probe.probe()
        if not st["available"]:
            return None, st["reason"]
        return ref_probe.time_reference(CFG), None
    except Exception as e:           # a half-working TF install must not take the bench line down
        return None, f"reference probe failed: {type(e).__name__}: {e}"


def cpu_baseline(budget_s=20.0):
    """The reference itself when it runs here (kind "reference"), else the oracle train step (fwd+bwd via autograd, clip,
    Adam; kind "port") on the host cores; B=1, S=1280 sample."""
    ref, why_not = reference_baseline()
    if ref is not None:
        return ref
    from oracle import dalle_oracle as do
    cores = min(os.cpu_count() or 1, 32)   # more threads than this only add contention on this op mix
    torch.set_num_threads(cores)
    cfg = do.DalleConfig(CFG["n_embd"], CFG["text_vocab_size"], CFG["image_vocab_size"], CFG["text_seq_len"],
                         CFG["image_seq_len"], CFG["n_layers"], CFG["n_heads"])
    P = do.init_params(cfg, seed=1234)
    m = {k: np.zeros_like(v) for k, v in P.items()}
    v = {k: np.zeros_like(v) for k, v in P.items()}
    tokens = synth_tokens(1, cfg.text_seq_len, cfg.image_seq_len, cfg.text_vocab_size, cfg.image_vocab_size, 7)
    do.train_step(P, m, v, tokens, cfg, 1, HP)  # warm-up (allocator, thread pool)
    t0 = time.time()
    n = 0
    while True:
        do.train_step(P, m, v, tokens, cfg, 2 + n, HP)
        n += 1
        if time.time() - t0 > budget_s or n >= 6:
            break
    dt = (time.time() - t0) / n
    S = cfg.total_seq_dim
    return {"value": S / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of B=1 x S={S} (dalle_example weights, fp32 PyTorch-CPU oracle, "
                      f"{torch.get_num_threads()} threads), {dt:.2f} s/step",
            "note": "CPU restatement of the reference; the mesh-tensorflow reference itself cannot run here "
                    f"(tools/ref_probe.py: {why_not})"}


def cpu_baseline_vae(p, grid, budget_s=15.0):
    """Oracle VAE train step (fwd + autograd bwd + TF-Adam) on the host cores; a 2-image sample of the same configuration."""
    from oracle import vae_oracle as vo
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = vo.VaeConfig(num_tokens=p["num_tokens"], dimensions=p["dataset"]["image_size"], convblocks=p["convblocks"])
    P = vo.init_params(cfg, seed=4321)
    m = {k: np.zeros_like(v) for k, v in P.items()}
    v = {k: np.zeros_like(v) for k, v in P.items()}
    B = 2
    img = vo.synthetic_images(B, cfg.H, seed=0)
    u = vo.synthetic_uniforms((B, cfg.grid, cfg.grid, cfg.num_tokens), seed=1)

    def step(t):
        _, g, _ = vo.loss_and_grads(P, img, u, cfg, hard=True, temp=1.0)
        vo.tf_adam_step(P, g, m, v, t, p["lr"])
    step(1)
    t0, n = time.time(), 0
    while True:
        step(2 + n)
        n += 1
        if time.time() - t0 > budget_s or n >= 8:
            break
    dt = (time.time() - t0) / n
    return {"value": B * grid * grid / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of {B} images (fp32 PyTorch-CPU oracle, {cores} threads), {dt:.2f} s/step",
            "note": "CPU restatement of the reference; tensorflow itself cannot run here"}


def per_gpu_batch(batch_arg, default, world, scaling):
    """weak (the primary line): per-GPU batch fixed; strong: the GLOBAL batch fixed at the model's BASELINE value"""
    B = batch_arg or default
    if scaling == "strong":
        assert B % world == 0, f"strong scaling: the global batch {B} must divide by the number of GPUs {world}"
        B = B // world
    return B


def dp_schedule_summary(schedule, lay):
    """what the JSON line says about the gradient exchange of one step: every piece in issue order (they follow
    ParamLayout.ready_points, src/dalle_mtf/engine.py, cut to <= 64 MB by src/dp.py) and the bytes issued after the last backward
    kernel -- the embedding gradients, which nothing is left to hide behind (tests/test_dp_gloo.py checks this against the layout)"""
    tail0 = lay.offset["positional_embedding/wpe"]
    return {"dp_pieces_per_step": len(schedule),
            "dp_largest_piece_MB": (max(b - a for a, b in schedule) * 4 / 2 ** 20) if schedule else None,
            "dp_piece_MB": [round((b - a) * 4 / 2 ** 20, 2) for a, b in schedule],
            "dp_exposed_tail_MB": sum(b - a for a, b in schedule if a >= tail0) * 4 / 2 ** 20}


def load_traffic(name, kernel_substr):
    """bytes/launch from this round's PMC passes, or None when absent / measured on another kernel"""
    path = os.path.join(ROOT, "profiles", f"{ROUND_TAG}_traffic_{name}.json")
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    if kernel_substr not in rec.get("kernel", ""):
        return None, None
    return rec.get("traffic_bytes"), os.path.relpath(path, ROOT)


def setup_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch ourselves under torch.distributed.run
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), __file__] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # DALLE_BENCH_SHARE_GPU=1 (tests only): all ranks share cuda:0 and exchange over gloo, to exercise the N>1 code path on a
    # 1-GPU box; the driver's multi-GPU runs use one GPU per rank and RCCL behind the C ABI (src/dp.py).
    share = os.environ.get("DALLE_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else local_rank)
    pg = comm = None
    if world > 1:
        from src import dp
        pg = dp.init_process_group(local_rank)
        _, _, pg, comm = dp.dist_setup()
        # one GPU per rank = the measured product path: the exchange must be RCCL behind the C ABI (dp.init_comm raises on
        # every rank when it cannot be -- DALLE_DP_STRICT defaults to 1); only the shared-GPU test mode runs on gloo
        assert share or comm, "bench.py --gpus N: the gradient exchange must run on RCCL (dmi_allreduce_bucket), not a fallback"
    return world, rank, pg, comm


def bench_vae(args, world, rank, pg, comm):
    from src.vae_tf import DiscreteVAE
    p = json.load(open(os.path.join(ROOT, "configs", args.model + ".json")))
    B = per_gpu_batch(args.batch, VAE_MODELS[args.model], world, args.scaling)   # (strong: global batch fixed, as on the DALL-E path)
    vae = DiscreteVAE(num_tokens=p["num_tokens"], dimensions=p["dataset"]["image_size"], convblocks=p["convblocks"],
                      dim=p.get("dim") or 512, hidden_dim=p.get("hidden_dim") or 64, input_channels=p.get("n_channels") or 3,
                      use_bf16=bool(p.get("use_bf16")), recompute_grad=bool(p.get("recompute_grad")),
                      stack_factor=p.get("stack_factor") or 1, batch_size=B, mode="train", process_group=pg, world_size=world,
                      comm=comm)
    vae.init_params()
    if world > 1:
        vae.reducer.broadcast(vae.p)
        vae.refresh_compute_copies(cast=True)
    g = torch.Generator(device="cuda").manual_seed(1000 + rank)
    imgs = [(torch.randint(0, 256, (B, vae.H, vae.W, vae.num_ch), device="cuda", generator=g).float() - 127.5) / 127.5 for _ in range(4)]
    hard = bool(p.get("train_gumbel_hard", True))
    # the heaviest convolution launch of the forward pass (most flops among the implicit-GEMM layers)
    heavy = max((c for c in vae.convs if c.kind in ("down", "res") and vae._implicit_ok(c)),
                key=lambda c: c.Ho * c.Wo * c.cout * c.kk * c.cin, default=None)
    evs = []
    # world 1: the step replays a captured HIP graph (DiscreteVAE.train_step) -- HIP events cannot be recorded inside a graph, so the
    # heaviest convolution is timed on eager steps run right AFTER the timed region (same kernels, same data); world > 1 runs eager
    # (exchange on its own stream) and times the launch inside the timed region as the DALL-E bench does.
    graphed = world == 1 and os.environ.get("DALLE_VAE_GRAPH", "1") != "0"
    vae.event_hook = None if graphed else ((heavy.name, evs) if heavy is not None else None)

    def step(i):
        vae.train_step(imgs[i % 4], p["lr"], hard_gumbel=hard, temperature=1.0)
    import torch.distributed as dist

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for i in range(max(args.warmup, 3)):      # >= 3: eager warm step, capture, first replay
        step(i)
    sync()
    evs.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    if graphed and heavy is not None:
        vae.event_hook = (heavy.name, evs)
        for i in range(22):
            step(i)
            if i == 1:
                sync()
                evs.clear()      # the first eager steps after the replays pay one-time costs (a 15 ms outlier was measured)
        sync()
        vae.event_hook = None
    loss = float(vae.loss.item())
    if world > 1:
        tmax = torch.tensor([dt])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    g2 = vae.grid ** 2
    fl_img = 0
    for c in vae.convs:   # 2*Hout*Wout*Cout*K*K*Cin per conv (transpose conv: input-sized), reference channel counts
        hw = c.Ho * c.Wo if c.kind != "up" else c.H * c.W
        fl_img += 2 * hw * c.cout_ref * c.kk * c.cin_ref
    fl_img += 2 * 2 * g2 * vae.n_hid * vae.num_tokens
    train_fl = 3 * fl_img * B
    k_ms = [a.elapsed_time(b) for a, b in evs]
    if os.environ.get("BENCH_DEBUG_EVENTS"):
        print("event ms:", [round(x, 3) for x in k_ms], file=sys.stderr)
    k_avg = sum(k_ms) / max(len(k_ms), 1) if k_ms else float("nan")
    roof = None
    if heavy is not None and k_ms:
        kfl = 2.0 * B * heavy.Ho * heavy.Wo * heavy.cout * heavy.kk * heavy.cin
        ach = kfl / (k_avg * 1e-3) / 1e12
        traffic, tsrc = load_traffic(f"{args.model}_conv", "conv_gemm_nt_kernel")
        roof = {"bound": "mfma", "kernel": f"conv_gemm_nt_kernel ({heavy.name}: implicit-im2col GEMM M={B * heavy.Ho * heavy.Wo}, "
                                           f"N={heavy.cout}, K={heavy.kk * heavy.cin})",
                "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                "traffic_source": tsrc, "launch_ms": k_avg, "launches_timed": len(k_ms),
                "launch_timing": ("eager steps right after the timed region (the timed steps replay a HIP graph)" if graphed
                                  else "HIP events inside the timed region"),
                "step_mfma_frac": train_fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                "step_tflops_per_gpu": train_fl / (ms * 1e-3) / 1e12}
    out = {"metric": f"train image tokens/sec per node, {args.model}", "value": B * world * g2 * args.steps / dt, "unit": "tokens/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"configs/{args.model}.json discrete-VAE train step ({vae.H}x{vae.W} images, convblocks "
                                  f"{p['convblocks']}, {vae.num_tokens} tokens, grid {vae.grid}x{vae.grid}), synthetic images",
                      "global_batch": B * world, "per_gpu_batch": B, "images_per_s": B * world * args.steps / dt,
                      "parallelism": f"dp{world}", "dp_transport": vae.reducer.transport if world > 1 else None,
                      "hip_graph": graphed, "final_loss": loss},
           "roofline": roof}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_vae(p, vae.grid)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling); default: the model's BASELINE value")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the primary line): per-GPU batch fixed, global batch = batch x N.  strong (BASELINE.md §2's "
                         "secondary number): GLOBAL batch fixed at the model's BASELINE value, per-GPU batch = global / N")
    ap.add_argument("--model", default="dalle_example", choices=sorted(MODELS) + sorted(VAE_MODELS))
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="N > 1: CUs the persistent kernels of the BACKWARD leave to the exchange's RCCL channels (engine option "
                         "dp_reserve_cus; default: DALLE_DP_RESERVE_CUS or 0).  For the first multi-GPU A/B: one call per value.")
    args = ap.parse_args()
    world, rank, pg, comm = setup_dist(args)
    if args.model in VAE_MODELS:
        bench_vae(args, world, rank, pg, comm)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    global CFG
    CFG = MODELS[args.model]
    import torch.distributed as dist
    from src.dalle_mtf.engine import DalleEngine
    B = per_gpu_batch(args.batch, PER_GPU_BATCH, world, args.scaling)
    eng = DalleEngine(CFG["n_embd"], CFG["n_layers"], CFG["n_heads"], CFG["text_vocab_size"], CFG["image_vocab_size"],
                      CFG["text_seq_len"], CFG["image_seq_len"], batch_size=B, global_batch_size=B * world,
                      hparams=dict(HP, **({"dp_reserve_cus": args.reserve_cus} if args.reserve_cus is not None else {})),
                      process_group=pg, world_size=world, comm=comm)
    eng.init_params(seed=1234)
    if world > 1:
        eng.reducer.broadcast(eng.p)
        eng.refresh_compute_copies(cast=True)
    S, T, P = eng.S, eng.T, eng.S - eng.T
    batches = [torch.from_numpy(synth_tokens(B, T, P, CFG["text_vocab_size"], CFG["image_vocab_size"], 1000 * rank + i)).cuda()
               for i in range(4)]
    eng.global_step = 3000  # past warm-up so the update is non-trivial
    evs = []
    eng.event_hook = lambda: evs  # the engine records (start, end) events around the vocab-projection GEMM

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        eng.train_step(batches[i % 4])
    sync()
    evs.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.train_step(batches[i % 4])
    sync()
    dt = time.perf_counter() - t0
    loss = float(eng.loss.item())
    schedule = list(eng.reducer.last_log)
    if world > 1:
        tmax = torch.tensor([dt])
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        tokens_per_s = B * world * S * args.steps / dt
        d, L, V = CFG["n_embd"], CFG["n_layers"], eng.V
        train_flops_step_gpu = 3 * fwd_flops_per_token(d, L, S, V) * B * S
        gemm_flops = 2.0 * B * S * d * V   # algorithmic (unpadded vocabulary)
        k_ms = [a.elapsed_time(b) for a, b in evs]
        k_avg = sum(k_ms) / max(len(k_ms), 1) if k_ms else float("nan")
        achieved = gemm_flops / (k_avg * 1e-3) / 1e12 if k_ms else float("nan")
        # 256x128 tiles when the grid covers >= 3 residencies and K <= 1024 (csrc/gemm.hip launch_nt), else 128x128
        tiles4 = ((B * S + 255) // 256) * ((eng.Vp + 127) // 128)
        tiles8 = ((B * S + 255) // 256) * ((eng.Vp + 255) // 256)
        if tiles8 >= 512 and d <= 1024 and d % 128 == 0:      # persistent 256x256 tiles (csrc/gemm.hip launch_nt, option nt8p = auto)
            kname = "gemm_nt8p_kernel<65>"
        else:
            kname = "gemm_nt4_kernel<65>" if (tiles4 >= 1536 and d <= 1024) else "gemm_nt2_kernel<65>"
        traffic, tsrc = (load_traffic("vocab_gemm", kname) if (B == PER_GPU_BATCH and args.model == "dalle_example") else (None, None))
        algo_bytes = (B * S * d + eng.Vp * d) * 2 + B * S * eng.Vp * 2 + (eng.Vp // 64) * B * S * 4
        out = {
            "metric": f"train tokens/sec (text+image) per node, {args.model}", "value": tokens_per_s, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs/{args.model if args.model != '1.3B' else 'dalle_example (1.3B dimensions, SURVEY §8(d) C5)'}.json transformer train step (n_embd={d}, {L} layers, "
                                   f"{CFG['n_heads']} heads, seq 256+1024, V={V}), synthetic captions + synthetic image-token ids",
                       "global_batch": B * world, "per_gpu_batch": B, "seq_len": S, "parallelism": f"dp{world}",
                       "dp_transport": eng.reducer.transport if world > 1 else None,
                       "dp_reserve_cus": eng.dp_reserve_cus if world > 1 else None,
                       # every exchange piece in issue order, and the bytes issued after the last backward kernel (the embedding
                       # gradients): nothing is left to hide those behind, they are the exposed tail of the exchange
                       **(dp_schedule_summary(schedule, eng.lay) if world > 1 else
                          {"dp_pieces_per_step": None, "dp_largest_piece_MB": None, "dp_piece_MB": None, "dp_exposed_tail_MB": None}),
                       "final_loss": loss},
            "roofline": {"bound": "mfma",
                         "kernel": f"{kname} (vocabulary projection with the softmax-numerator epilogue, dmi_gemm_nt_softmax: M=B*S={B * S}, N={eng.Vp}, K={d})",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS if k_ms else None, "traffic": traffic,
                         "traffic_source": tsrc, "algorithmic_bytes": algo_bytes,
                         "traffic_unit": "bytes/launch (PMC: FETCH_SIZE x2 + WRITE_SIZE, L2-side counters incl. Infinity-Cache hits)",
                         "launch_ms": k_avg, "launches_timed": len(k_ms),
                         # what else holds this launch below the matrix-pipe bound (DESIGN §4 "[r04] What bounds the short-K products"):
                         # K = 512 gives a 256x256 tile 8 k-steps per 128 KB of exponentiated output, the epilogue (v_exp_f32 at a quarter
                         # rate, ~13 B/clk/CU of output stores = 4.2 GB per launch) runs with the matrix pipe idle
                         "also_bound_by": "epilogue: exp at quarter VALU rate + 4.16 GB of output stores per launch (~28 % of a tile's life)",
                         "step_mfma_frac": train_flops_step_gpu / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "step_tflops_per_gpu": train_flops_step_gpu / (ms * 1e-3) / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline and args.model == "dalle_example":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
