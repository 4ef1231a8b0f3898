"""model_fn-level integration on the MI355X: the reference's entry contracts (src/model_fns.py:55, src/model_fns_tf.py:9)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(name, **over):
    from src.utils import fetch_model_params
    p = fetch_model_params(name)
    p.update(over)
    return p


def test_vae_model_fn_trains_and_tf_adam_semantics():
    from collections import OrderedDict
    from oracle import vae_oracle as vo
    from src.model_fns_tf import vae_model_fn
    from src.utils import ModeKeys
    p = _params("vae_example", train_batch_size=4, eval_batch_size=4, model_path=None, convblocks=[[2, 64], [2, 64]],
                num_tokens=128, dataset={"train_path": "synthetic", "eval_path": "synthetic", "image_size": 16})
    img = torch.from_numpy(vo.synthetic_images(4, 16, seed=1))
    spec = vae_model_fn(img, img, ModeKeys.TRAIN, p)
    model = p["_vae_state_train"]["model"]
    before = model.export_reference()
    step = spec.train_op()          # forward + backward + update run as one step inside train_op (spec.loss is filled by it)
    assert step == 1
    loss0 = float(spec.loss)
    assert np.isfinite(loss0) and loss0 > 0
    grads = model.export_reference(model.g)
    after = model.export_reference()
    # tf.train.AdamOptimizer step 1 on the HIP gradients must reproduce the HIP update (update-rule parity)
    m = OrderedDict((k, np.zeros_like(v)) for k, v in before.items())
    v = OrderedDict((k, np.zeros_like(v)) for k, v in before.items())
    ref = OrderedDict((k, a.copy()) for k, a in before.items())
    vo.tf_adam_step(ref, grads, m, v, 1, p["lr"])
    for k in ref:
        assert np.allclose(after[k], ref[k], rtol=1e-4, atol=2e-6), k
    # loss goes down over a few steps on a fixed batch
    for _ in range(30):
        spec = vae_model_fn(img, img, ModeKeys.TRAIN, p)
        spec.train_op()
    assert float(spec.loss) < loss0
    ev = vae_model_fn(img, img, ModeKeys.EVAL, p)
    assert np.isfinite(float(ev.loss))
    with pytest.raises(NotImplementedError):
        vae_model_fn(img, img, ModeKeys.PREDICT, p)


def test_vae_graph_replayed_step_is_bit_identical_to_the_eager_step():
    """DiscreteVAE.train_step captures forward + backward + TF-Adam as ONE HIP graph (launch-bound small configurations) with
    the scheduled scalars -- bias-corrected lr_t and the annealed Gumbel temperature -- in device memory.  Same kernels in the
    same order: after 6 steps with a CHANGING temperature and identical noise, weights, Adam state and losses must equal the
    eager run bit for bit."""
    from oracle import vae_oracle as vo
    from src.vae_tf import DiscreteVAE
    c = dict(num_tokens=128, dimensions=16, convblocks=[[2, 64], [2, 64]])
    P = vo.init_params(vo.VaeConfig(**c), seed=3, bias_perturb=0.02)
    runs = {}
    for mode in ("eager", "graph"):
        vae = DiscreteVAE(batch_size=4, use_bf16=True, **c)
        vae.load_reference_params(P)
        losses = []
        for t in range(6):
            img = torch.from_numpy(vo.synthetic_images(4, 16, seed=10 + t))
            u = torch.from_numpy(vo.synthetic_uniforms((4, vae.grid, vae.grid, 128), seed=20 + t))
            loss = vae.train_step(img, 1e-3, hard_gumbel=(t % 2 == 0), temperature=1.0 - 0.1 * t, noise=u, graph=(mode == "graph"))
            losses.append(float(loss))
        assert vae.global_step == 6
        if mode == "graph":
            assert len(vae._graphs) == 2          # one graph per Gumbel mode, each serving several temperatures / lr_t values
        runs[mode] = (losses, vae.p.clone(), vae.m.clone(), vae.v.clone())
        del vae
    assert runs["eager"][0] == runs["graph"][0], (runs["eager"][0], runs["graph"][0])
    for a, b in zip(runs["eager"][1:], runs["graph"][1:]):
        assert torch.equal(a, b)


def test_dalle_model_fn_with_vae_tokenisation():
    """images -> VAE encoder -> argmax tokens -> concat with text (src/model_fns.py:72-77,118-119) -> train step.
    S = 256 + 16 = 272 (the reference-faithful CIFAR grid, model_fns.py:68)."""
    from oracle import dalle_oracle as do
    from oracle import vae_oracle as vo
    from src.model_fns import dalle_model_fn
    from src.utils import ModeKeys
    p = _params("dalle_example", train_batch_size=2, eval_batch_size=2, model_path=None, n_layers=1, n_embd=256, n_heads=2,
                allow_random_vae=True)
    p["vae_params"] = _params("vae_example", model_path="/nonexistent")
    img = torch.from_numpy(vo.synthetic_images(2, 32, seed=2))
    text = torch.from_numpy(do.synthetic_captions(2, 256, p["text_vocab_size"], seed=3))
    spec = dalle_model_fn(img, text, ModeKeys.TRAIN, p)
    st = p["_dalle_state_train"]
    eng = st["model"].engine
    assert eng.S == 272 and st["image_seq_len"] == 16
    # integer path is bit-exact w.r.t. the oracle applied to the SAME logits
    logits = st["vae"].forward(img, return_logits=True).cpu().numpy()
    ref_tokens = do.assemble_tokens(text.numpy(), do.image_tokens_from_logits(logits), p["text_vocab_size"])
    assert np.array_equal(eng.tokens.cpu().numpy(), ref_tokens)
    assert ref_tokens[:, 256:].min() >= 50258 and ref_tokens[:, 256:].max() < 50770
    l0 = float(spec.loss)
    assert abs(l0 - np.log(eng.V)) < 1.0
    assert spec.train_op() == 1
    with pytest.raises(NotImplementedError):
        dalle_model_fn(img, text, ModeKeys.PREDICT, p)


def test_dalle_model_fn_synthetic_tokens_loss_decreases():
    from oracle import dalle_oracle as do
    from src.model_fns import dalle_model_fn
    from src.utils import ModeKeys
    p = _params("dalle_example", train_batch_size=2, eval_batch_size=2, model_path=None, n_layers=2, n_embd=256, n_heads=2,
                synthetic_image_tokens=112, text_seq_len=16, warmup_steps=1, lr=3e-3)
    text = torch.from_numpy(do.synthetic_captions(2, 16, p["text_vocab_size"], seed=3))
    imgtok = torch.from_numpy(do.synthetic_image_tokens(2, 112, 512, seed=4))
    losses = []
    for _ in range(25):
        spec = dalle_model_fn(imgtok, text, ModeKeys.TRAIN, p)
        losses.append(float(spec.loss))
        spec.train_op()
    assert losses[-1] < losses[0] - 1.0, losses


def _write_shards(tmp_path, n=12, size=32):
    """Paired JPEG/caption TFRecord shards in the reference's format (src/data/create_tfrecords.py:47-56)."""
    import io
    from PIL import Image
    from src.data.create_tfrecords import TFRecordWriter, serialize_example
    rng = np.random.default_rng(0)
    paths = []
    for k in range(2):
        paths.append(str(tmp_path / f"pairs_{k}.tfrecords"))
        w = TFRecordWriter(paths[-1])
        for i in range(n // 2):
            arr = rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)
            buf = io.BytesIO()
            Image.fromarray(arr).save(buf, format="JPEG", quality=90)
            w.write(serialize_example(buf.getvalue(), rng.integers(0, 50257, size=int(rng.integers(1, 300))).tolist()))
        w.close()
    return str(tmp_path / "pairs_*.tfrecords")


def test_estimator_trains_from_tfrecords(tmp_path):
    """TFRecord shards -> input_fn -> model_fn -> train_op for both entry points (train_vae_tf.py:63-95,
    train_dalle.py:87-112): the real-data path feeds the same HIP step as the synthetic one."""
    from functools import partial
    from src.estimator import Estimator
    from src.input_fns import dalle_input_fn, vae_input_fn
    from src.model_fns import dalle_model_fn
    from src.model_fns_tf import vae_model_fn
    glob = _write_shards(tmp_path)
    ds = {"train_path": glob, "eval_path": glob, "image_size": 32, "tfrecords": True}
    vp = _params("vae_example", train_batch_size=4, eval_batch_size=4, batch_size=4, model_path=None, dataset=ds)
    est = Estimator(vae_model_fn, None, vp, log_every=1000)
    assert est.train(partial(vae_input_fn, eval=False), max_steps=3) == 3
    assert np.isfinite(est.evaluate(partial(vae_input_fn, eval=True), steps=2)["loss"])

    dp = _params("dalle_example", train_batch_size=2, eval_batch_size=2, batch_size=2, model_path=None, n_layers=1,
                 n_embd=256, n_heads=2, allow_random_vae=True, dataset=ds)
    dp["padding_id"] = dp["text_vocab_size"] - 1
    dp["vae_params"] = _params("vae_example", model_path="/nonexistent")
    it = dalle_input_fn(dp, eval=True)
    img, cap = next(it)
    it.close()
    assert tuple(img.shape) == (2, 32, 32, 3) and tuple(cap.shape) == (2, dp["text_seq_len"])
    assert int(cap.max()) <= dp["padding_id"] and int(cap.min()) >= 0
    est = Estimator(dalle_model_fn, None, dp, log_every=1000)
    assert est.train(partial(dalle_input_fn, eval=False), max_steps=2) == 2
    st = dp["_dalle_state_train"]
    toks = st["model"].engine.tokens.cpu().numpy()
    assert toks[:, :dp["text_seq_len"]].max() < dp["text_vocab_size"] and toks[:, dp["text_seq_len"]:].min() >= dp["text_vocab_size"]


def test_dalle_model_fn_microbatching():
    """tokens_per_mb_per_replica -> num_microbatches (src/model_fns.py:141-166): the engine is built for one
    micro-batch, train_op runs the serialized step, spec.loss reports the full-batch mean."""
    from oracle import dalle_oracle as do
    from src.model_fns import dalle_model_fn
    from src.utils import ModeKeys
    p = _params("dalle_example", train_batch_size=4, eval_batch_size=4, model_path=None, n_layers=1, n_embd=256, n_heads=2,
                synthetic_image_tokens=112, text_seq_len=16, warmup_steps=1, lr=3e-3, tokens_per_mb_per_replica=256)
    text = torch.from_numpy(do.synthetic_captions(4, 16, p["text_vocab_size"], seed=3))
    imgtok = torch.from_numpy(do.synthetic_image_tokens(4, 112, 512, seed=4))
    losses = []
    for i in range(12):
        spec = dalle_model_fn(imgtok, text, ModeKeys.TRAIN, p)
        assert spec.train_op() == i + 1
        losses.append(float(spec.loss))
    eng = p["_dalle_state_train"]["model"].engine
    assert p["num_microbatches"] == 2 and eng.B == 2 and eng.hp["num_microbatches"] == 2
    assert abs(losses[0] - np.log(eng.V)) < 1.0 and losses[-1] < losses[0] - 0.5, losses


def test_greedy_sampling_matches_oracle_and_decodes():
    """DALLE.sample (generation path, unfinished upstream): greedy image tokens equal the fp32 oracle's greedy continuation
    wherever the oracle's top-2 gap exceeds the bf16 logit error (checked position by position under teacher forcing with
    the HIP tokens), and the VAE decodes token ids to the same image as its hard one-hot forward."""
    import numpy as np
    from collections import OrderedDict
    from oracle import dalle_oracle as do
    from oracle import vae_oracle as vo
    from src.dalle_mtf import DALLE
    from src.vae_tf import DiscreteVAE
    T, P, tv, iv = 8, 16, 60, 64
    cfg = do.DalleConfig(128, tv, iv, T, P, 2, 1)
    P0 = do.init_params(cfg, seed=3, perturb=0.05)
    model = DALLE(n_embd=128, text_vocab_size=tv, image_vocab_size=iv, text_seq_len=T, image_seq_len=P, n_layers=2, n_heads=1,
                  batch_size=2, mode="eval", params=dict(lr=1e-3, train_steps=10))
    model.engine.load_reference_params(P0)
    text = torch.from_numpy(do.synthetic_captions(2, T, tv, seed=1)).cuda()
    toks = model.sample(text, temperature=0.0)
    assert toks.shape == (2, P) and int(toks.min()) >= 0 and int(toks.max()) < iv
    full = np.concatenate([text.cpu().numpy(), toks.cpu().numpy() + tv], 1).astype(np.int32)
    Pt = OrderedDict((k, torch.tensor(v)) for k, v in P0.items())
    _, _, ref = do.forward(Pt, full, cfg, bf16=False, return_logits=True)
    ref = ref.numpy()[:, T - 1:T + P - 1, tv:tv + iv]                      # the logits that chose each image token
    srt = np.sort(ref, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 5e-2
    assert safe.mean() > 0.3      # (the test has power: a good share of the positions is decided by a clear margin)
    assert np.array_equal(ref.argmax(-1)[safe], toks.cpu().numpy()[safe])
    # sampled (temperature 1, top-k 8) tokens are valid, seeded and reproducible
    a = model.sample(text, temperature=1.0, top_k=8, seed=7)
    b = model.sample(text, temperature=1.0, top_k=8, seed=7)
    assert torch.equal(a, b) and int(a.max()) < iv
    # decode: token ids -> image == decoder applied to the one-hot of those ids (oracle decoder)
    vc = dict(num_tokens=64, dimensions=16, convblocks=[[2, 64], [2, 64]])
    vcfg = vo.VaeConfig(**vc)
    VP = vo.init_params(vcfg, seed=5, bias_perturb=0.02)
    vae = DiscreteVAE(batch_size=2, mode="eval", **vc)
    vae.load_reference_params(VP)
    img = vae.decode_tokens(toks)
    assert img.shape == (2, 16, 16, 3)
    onehot = torch.nn.functional.one_hot(toks.cpu().long().view(2, 4, 4), 64).float()
    want = vo.decoder({k: torch.tensor(v) for k, v in VP.items()}, onehot, vcfg).numpy()
    assert float(np.abs(img.cpu().numpy() - want).max()) <= 5e-2 * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("B,H,S", [(2, 1, 72), (3, 2, 320)])
def test_attention_decode_kernel_vs_fp32_math(B, H, S):
    """dmi_attention_decode: one query position against the cache, for positions at the start, at 64-key chunk boundaries and at
    the end; vs fp32 softmax(q K^T) V on the same bf16 inputs (unscaled logits, keys <= pos)."""
    import dalle_hip as dh
    d = H * 128
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B * S, 3 * d, generator=g) * 0.3).to(torch.bfloat16)
    qd = qkv.cuda()
    o = torch.empty(B, d, dtype=torch.bfloat16, device="cuda")
    f = qkv.float().view(B, S, 3, H, 128)
    for pos in sorted({0, 1, 63, 64, 65, 127, 128, S // 2, S - 2, S - 1} & set(range(S))):
        dh.attention_decode(qd, o, B, H, S, pos)
        q, k, v = f[:, pos, 0], f[:, :pos + 1, 1], f[:, :pos + 1, 2]           # [B,H,128], [B,p+1,H,128]
        w = torch.softmax(torch.einsum("bhd,bkhd->bhk", q, k), -1)
        ref = torch.einsum("bhk,bkhd->bhd", w, v).reshape(B, d)
        err = float((o.float().cpu() - ref).abs().max())
        assert err <= 1.6e-2 * float(ref.abs().max()) + 2e-3, (pos, err)
        # graph-replayable form: position from device memory, q | k | v of the step handed over in a staging buffer ->
        # the same bits, and the staging row lands in cache row pos
        cache = qd.clone()
        cache.view(B, S, 3 * d)[:, pos] = 0
        fresh = qd.view(B, S, 3 * d)[:, pos].contiguous()
        o2 = torch.full_like(o, 7.0)
        dh.attention_decode(cache, o2, B, H, S, 0, fresh=fresh, pos_dev=torch.tensor([pos], dtype=torch.int32, device="cuda"))
        assert torch.equal(o2, o), pos
        assert torch.equal(cache, qd), pos
    o2.fill_(7.0)            # a position outside [0, S) in device memory: the launch is a no-op
    dh.attention_decode(cache, o2, B, H, S, 0, fresh=fresh, pos_dev=torch.tensor([S], dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert bool((o2 == 7.0).all()) and torch.equal(cache, qd)


def test_kv_cached_decode_equals_full_forward():
    """Incremental inference (reference hooks src/dalle_mtf/models.py:246-254,281-285): with the key/value cache, the logits of
    position p computed from ONE new row equal the full forward's logits at p (teacher forcing over a 320-position sequence:
    every 64-key chunk count, batch 3), and the cached greedy sampler emits the same tokens as the one-forward-per-token
    sampler wherever the top-2 logit gap exceeds twice the measured logit difference."""
    from oracle import dalle_oracle as do
    from src.dalle_mtf.engine import DalleEngine
    T, P, tv, iv, B = 16, 304, 60, 64, 3
    cfg = do.DalleConfig(128, tv, iv, T, P, 2, 1)
    eng = DalleEngine(128, 2, 1, tv, iv, T, P, batch_size=B, hparams=dict(lr=1e-3, train_steps=10))
    eng.load_reference_params(do.init_params(cfg, seed=9, perturb=0.05))
    S = T + P
    toks = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(B, T, tv, seed=1),
                                               do.synthetic_image_tokens(B, P, iv, seed=2), tv)).cuda()
    eng.forward(toks, need_grad=False)                      # also the prefill: k, v of every position are in the cache
    full = eng.z.view(B, S, eng.Vp)[:, :, tv:tv + iv].float().clone()
    worst = 0.0
    for i, p in enumerate(list(range(T - 1, T + 70)) + [127, 128, 129, 255, 256, S - 2, S - 1]):
        z = eng.decode_step(toks[:, p].contiguous(), p)     # rewrites row p of the cache with the same values
        worst = max(worst, float((z - full[:, p]).abs().max()))
        if i % 7 == 0 or p > 100:   # the replayed HIP graph (steps >= 2) and the eager launches give the same bits
            zg = z.clone()
            assert torch.equal(eng.decode_step(toks[:, p].contiguous(), p, graph=False), zg), p
    assert eng._dec["graphs"].get(False) is not None
    scale = float(full.abs().max())
    print("decode vs full forward: max |dlogit|", worst, "of", scale)
    assert worst <= 2.5e-2 * scale, (worst, scale)
    text = toks[:, :T].contiguous()
    a = eng.sample_image_tokens(text, temperature=0.0, kv_cache=True)      # decode + draw replayed as one graph per position
    a2 = eng.sample_image_tokens(text, temperature=0.0, kv_cache=True, fused_sampling=False)   # draw kernel launched from the host
    a3 = eng.sample_image_tokens(text, temperature=0.0, kv_cache=True, decode_graph=False)     # host-launched
    assert torch.equal(a, a2) and torch.equal(a, a3) and eng._dec["graphs"].get(True) is not None
    b = eng.sample_image_tokens(text, temperature=0.0, kv_cache=False)
    agree = (a == b)
    # positions after a first disagreement see different prefixes; compare up to and including it
    first_bad = [int((~agree[i]).nonzero()[0]) if not bool(agree[i].all()) else P for i in range(B)]
    for i in range(B):
        if first_bad[i] < P:      # a disagreement must be a near-tie of the full-forward logits at that position
            seq = torch.cat([text[i], b[i, :first_bad[i]].to(torch.int32) + tv, torch.full((S - T - first_bad[i],), tv, dtype=torch.int32, device="cuda")])
            eng.forward(seq.repeat(B, 1), need_grad=False)
            zz = eng.z.view(B, S, eng.Vp)[0, T + first_bad[i] - 1, tv:tv + iv].float()
            top2 = zz.topk(2).values
            assert float(top2[0] - top2[1]) <= 2 * worst + 1e-3, (i, first_bad[i], float(top2[0] - top2[1]), worst)
    # (a random-initialised model has nearly flat logits, so near-ties -- and with them a first divergence -- are common; on
    # MI355X: decode-vs-full logit difference 0.0043 of a 0.85 logit range, first divergences at positions 304 (none), 95, 50)
    # seeded stochastic sampling runs on the cached path and is reproducible
    c1 = eng.sample_image_tokens(text, temperature=1.0, top_k=8, seed=3)
    c2 = eng.sample_image_tokens(text, temperature=1.0, top_k=8, seed=3)
    c3 = eng.sample_image_tokens(text, temperature=1.0, top_k=8, seed=4)
    assert torch.equal(c1, c2) and int(c1.max()) < iv and int(c1.min()) >= 0 and not torch.equal(c1, c3)
    for kw in (dict(fused_sampling=False), dict(decode_graph=False)):      # host-launched draw: the same kernel, the same noise
        d1 = eng.sample_image_tokens(text, temperature=1.0, top_k=8, seed=3, **kw)
        d2 = eng.sample_image_tokens(text, temperature=1.0, top_k=8, seed=3, **kw)
        assert torch.equal(d1, d2) and int(d1.max()) < iv and int(d1.min()) >= 0
        assert torch.equal(d1, c1), "the draw is a pure function of (seed, position, row): graph-replayed and host-launched draws agree"


@pytest.mark.parametrize("nv", [64, 512, 2048, 8192])
def test_sample_tokens_kernel(nv):
    """dmi_sample_tokens: greedy = first maximum; top_k = 1 = the maximum at any temperature; draws stay inside the top-k set
    (ties of the k-th value kept); a draw is a pure function of (seed, position, row); and the empirical distribution of 40 000
    draws matches softmax((z + bias) / T) over the kept entries (Gumbel-max is an exact categorical draw)."""
    import dalle_hip as dh
    B = 5
    g = torch.Generator().manual_seed(nv)
    z = (torch.randn(B, nv + 8, generator=g) * 2).to(torch.bfloat16)
    z[1, 7] = z[1, 3] = z[1].float().max() + 1          # a tie of the maximum: the first one wins
    bias = (torch.randn(nv, generator=g) * 0.5).to(torch.bfloat16)
    bias[7] = bias[3]
    zd, bd = z.cuda(), bias.cuda()
    v = z[:, :nv].float() + bias.float()
    nxt = torch.zeros(B, dtype=torch.int32, device="cuda")
    out = torch.full((B, 4), -1, dtype=torch.int32, device="cuda")
    dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=0.0, pos=11, token_offset=100, next_tok=nxt, out=out, out_col0=9)
    want = torch.stack([(v[b] == v[b].max()).nonzero()[0, 0] for b in range(B)]).to(torch.int32)
    assert torch.equal(out[:, 2].cpu(), want) and torch.equal(nxt.cpu(), want + 100)
    assert bool((out[:, [0, 1, 3]] == -1).all())
    dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=0.7, top_k=1, seed=5, pos=50, out=out, out_col0=50)
    assert torch.equal(out[:, 0].cpu(), want)
    dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=0.7, top_k=1, seed=5, pos=99, out=out, out_col0=50)   # column out of range
    assert torch.equal(out[:, 0].cpu(), want)
    # device-side settings and position give the same draw as the by-value ones
    T, k = 0.8, 6
    o1 = torch.zeros(B, 1, dtype=torch.int32, device="cuda")
    o2 = torch.zeros(B, 1, dtype=torch.int32, device="cuda")
    seed = (123 << 32) | 77
    dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=T, top_k=k, seed=seed, pos=4, out=o1, out_col0=4)
    prm = np.array([np.array([1 / T], np.float32).view(np.uint32)[0], k, seed & 0xffffffff, seed >> 32], dtype=np.uint32).view(np.int32)
    dh.sample_tokens(zd, nv + 8, bd, B, nv, params_dev=torch.from_numpy(prm).cuda(), pos_dev=torch.tensor([4], dtype=torch.int32, device="cuda"),
                     out=o2, out_col0=4)
    assert torch.equal(o1, o2)
    # distribution: many positions = many independent draws of the same rows
    N = 40000
    draws = torch.zeros(B, N, dtype=torch.int32, device="cuda")
    for p in range(N):
        dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=T, top_k=k, seed=9, pos=p, out=draws, out_col0=0)
    draws = draws.cpu().long()
    for b in range(B):
        vb = v[b] / T
        kth = vb.topk(k).values[-1]
        keep = vb >= kth
        prob = torch.softmax(vb.masked_fill(~keep, float("-inf")), -1)
        freq = torch.bincount(draws[b], minlength=nv).float() / N
        assert bool((freq[~keep] == 0).all()), b
        sigma = torch.sqrt(prob * (1 - prob) / N)
        assert bool(((freq - prob).abs() <= 5 * sigma + 1e-4).all()), (b, float((freq - prob).abs().max()))
    dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=T, top_k=k, seed=9, pos=3, out=o1, out_col0=3)
    assert torch.equal(o1[:, 0].cpu().long(), draws[:, 3])
    # advance: the kernel itself moves the device-side position on (one graph replay per generated token, no host arithmetic)
    pd = torch.tensor([0, 0], dtype=torch.int32, device="cuda")
    seq = torch.zeros(B, 6, dtype=torch.int32, device="cuda")
    for _ in range(6):
        dh.sample_tokens(zd, nv + 8, bd, B, nv, temperature=T, top_k=k, seed=9, pos_dev=pd, advance=True, out=seq, out_col0=0)
    assert pd.cpu().tolist() == [6, 0] and torch.equal(seq.cpu().long(), draws[:, :6])


def test_logits_f32_kernel():
    """dmi_logits_f32: fp32(z) + fp32(bias) over a column slice of a wider bf16 matrix (exact: both terms are bf16 values)"""
    import dalle_hip as dh
    g = torch.Generator().manual_seed(0)
    z = torch.randn(5, 700, generator=g).to(torch.bfloat16).cuda()
    bias = torch.randn(300, generator=g).to(torch.bfloat16).cuda()
    out = torch.full((5, 300), -7.0, dtype=torch.float32, device="cuda")
    dh.logits_f32(z[:, 100:], 700, bias, out, 5, 300)
    assert torch.equal(out, z[:, 100:400].float() + bias.float())
    dh.logits_f32(z, 700, None, out, 5, 300)
    assert torch.equal(out, z[:, :300].float())


def test_reference_tf_checkpoints_load_by_variable_name(tmp_path):
    """(f)2: a checkpoint in the reference's own format (tf.train.Saver bundle, variable names of SURVEY Appendix B, optimizer
    slots and global_step alongside) warm-starts the DALL-E engine (`tf_checkpoint`) and restores the VAE's variables from scope
    `vae/` exactly as reference src/model_fns.py:11-32 does -- read without TensorFlow."""
    from oracle import dalle_oracle as do
    from oracle import vae_oracle as vo
    from src.data import tf_checkpoint as tfc
    from src.dalle_mtf.engine import DalleEngine
    from src.model_fns import initialize_vae_weights
    from src.vae_tf import DiscreteVAE
    cfg = do.DalleConfig(128, 60, 64, 8, 16, 2, 1)
    P = do.init_params(cfg, seed=4, perturb=0.05)
    bundle = {k: v for k, v in P.items()}
    bundle.update({k + "/adam_m": np.zeros_like(v) for k, v in P.items()})
    bundle["global_step"] = np.asarray(321, np.int64)
    prefix = str(tmp_path / "model.ckpt-321")
    tfc.save_checkpoint(prefix, bundle)
    eng = DalleEngine(128, 2, 1, 60, 64, 8, 16, batch_size=2, hparams=dict(lr=1e-3, train_steps=10))
    eng.load_reference_params(tfc.load_model_variables(prefix))
    back = eng.export_reference()
    assert sorted(back) == sorted(P) and all(np.array_equal(back[k], P[k]) for k in P)
    assert tfc.global_step_of(prefix) == 321
    vc = dict(num_tokens=64, dimensions=16, convblocks=[[2, 64], [2, 64]])
    VP = vo.init_params(vo.VaeConfig(**vc), seed=5, bias_perturb=0.02)
    vprefix = str(tmp_path / "vae" / "model.ckpt-9")
    tfc.save_checkpoint(vprefix, {"vae/" + k: v for k, v in VP.items()})
    vae = DiscreteVAE(batch_size=2, mode="eval", **vc)
    initialize_vae_weights(vae, vprefix)
    vb = vae.export_reference()
    assert all(np.allclose(vb[k], VP[k]) for k in VP)
