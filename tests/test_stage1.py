"""Stage-1 distillation forward pieces: loss + teacher-embedding payload.

CPU: the oracle reproduces the values the reference's own functions gave (tests/golden/stage1, written by
oracle/gen_golden_stage1.py); the product's closed-form valid mask equals the oracle's interpolate-based one; the
payload codec round-trips.  GPU: esam3_distill_loss vs the fixtures."""
import json
import os

import numpy as np
import pytest
import torch

from efficientsam3_amd import stage1, synth


@pytest.fixture(scope="module")
def stage1_gold(golden_dir):
    with open(os.path.join(golden_dir, "stage1", "manifest.json")) as f:
        return json.load(f)["cases"]


def test_oracle_loss_matches_reference_values(stage1_gold):
    from oracle import ref_stage1
    for name, (b, c, hw, img, sizes) in synth.stage1_cases().items():
        preds, teacher = synth.stage1_embeddings(name)
        m = ref_stage1.build_valid_mask(img, sizes, (hw, hw))
        assert [int(v) for v in m.sum(dim=(1, 2, 3))] == stage1_gold[name]["valid_pixels"]
        p, t = torch.from_numpy(preds), torch.from_numpy(teacher)
        assert float(ref_stage1.masked_mse(p, t, m)) == pytest.approx(stage1_gold[name]["mse"], rel=1e-6)
        assert float(ref_stage1.masked_cosine_loss(p, t, m)) == pytest.approx(stage1_gold[name]["cosine"], rel=1e-6)


def test_valid_mask_closed_form_equals_interpolate():
    from oracle import ref_stage1
    rng = np.random.default_rng(0)
    for img, hw in ((1008, 72), (126, 9), (1024, 64), (100, 7), (64, 64), (50, 72)):
        sizes = [(img, img), (1, 1), (img // 2, img), (img, img // 3)] + [tuple(int(v) for v in rng.integers(1, img + 1, 2)) for _ in range(8)]
        want = ref_stage1.build_valid_mask(img, sizes, (hw, hw)).numpy().reshape(len(sizes), -1).astype(np.uint8)
        got = stage1.valid_mask(img, sizes, (hw, hw))
        assert np.array_equal(got, want), (img, hw)


def test_embedding_payload_round_trip():
    from oracle import ref_stage1
    emb = np.random.default_rng(1).standard_normal((8, 3, 5)).astype(np.float32)
    blob = stage1.pack_embedding(123456789, emb)
    assert blob == ref_stage1.pack_embedding(123456789, emb) and len(blob) == 4 + 2 * emb.size
    seed, back = stage1.unpack_embedding(blob, emb.shape)
    seed_o, back_o = ref_stage1.unpack_embedding(blob, emb.shape)
    assert seed == seed_o == 123456789 and back.dtype == np.float16
    assert np.array_equal(back, back_o) and np.array_equal(back, emb.astype(np.float16))
    with pytest.raises(ValueError):
        stage1.unpack_embedding(blob[:-2], emb.shape)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small", "full"])
def test_distill_loss_kernel_vs_reference_values(stage1_gold, name):
    b, c, hw, img, sizes = synth.stage1_cases()[name]
    preds, teacher = synth.stage1_embeddings(name)
    valid = torch.from_numpy(stage1.valid_mask(img, sizes, (hw, hw))).cuda()
    p = torch.from_numpy(preds).permute(0, 2, 3, 1).reshape(b, hw * hw, c).contiguous().cuda()
    t = torch.from_numpy(teacher).permute(0, 2, 3, 1).reshape(b, hw * hw, c).contiguous().cuda()
    g = stage1_gold[name]
    for tdt in (torch.float32, torch.float16):          # the teacher values are fp16-representable: same result
        mse, cos, per = stage1.distill_loss(p, t.to(tdt), valid)
        assert float(mse) == pytest.approx(g["mse"], rel=2e-5) and float(cos) == pytest.approx(g["cosine"], rel=2e-5)
        assert per.shape == (b, 2)
        if name == "small":
            assert float(per[2].abs().max()) == 0.0      # an image without valid pixels contributes zero
    mse, cos, _ = stage1.distill_loss(p.to(torch.bfloat16), t.to(torch.bfloat16), valid)   # bf16 inputs: rounding only
    assert float(mse) == pytest.approx(g["mse"], rel=2e-2) and float(cos) == pytest.approx(g["cosine"], rel=5e-2)
    a = stage1.distill_loss(p, t, valid)[2]
    assert torch.equal(a, stage1.distill_loss(p, t, valid)[2])   # fixed-order reductions: bit-identical repeats


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small", "full"])
@pytest.mark.parametrize("w_cos", [0.0, 0.5])
def test_distill_loss_backward_vs_autograd(name, w_cos):
    """dL/dpreds of masked_mse + w * masked_cosine_loss / ACCUMULATION_STEPS (train_image_encoder_stage1.py:186-210) vs
    torch.autograd through the oracle's restatement of the two loss functions (itself pinned to the reference's values by
    test_oracle_loss_matches_reference_values)."""
    from oracle import ref_stage1
    b, c, hw, img, sizes = synth.stage1_cases()[name]
    preds, teacher = synth.stage1_embeddings(name)
    accum = 2.0
    pt = torch.from_numpy(preds).clone().requires_grad_(True)          # [B, C, H, W] fp32 on the CPU
    tt = torch.from_numpy(teacher)
    mask = ref_stage1.build_valid_mask(img, sizes, (hw, hw))
    loss = ref_stage1.masked_mse(pt, tt, mask)
    if w_cos:
        loss = loss + w_cos * ref_stage1.masked_cosine_loss(pt, tt, mask)
    (loss / accum).backward()
    want = pt.grad.permute(0, 2, 3, 1).reshape(b, hw * hw, c)
    valid = torch.from_numpy(stage1.valid_mask(img, sizes, (hw, hw))).cuda()
    p = torch.from_numpy(preds).permute(0, 2, 3, 1).reshape(b, hw * hw, c).contiguous().cuda()
    t = torch.from_numpy(teacher).permute(0, 2, 3, 1).reshape(b, hw * hw, c).contiguous().cuda()
    scale = float(want.abs().max())
    for tdt in (torch.float32, torch.float16):          # the teacher values are fp16-representable: same result
        g = stage1.distill_loss_backward(p, t.to(tdt), valid, cosine_weight=w_cos, grad_scale=1.0 / accum)
        assert g.dtype == torch.float32 and g.shape == p.shape
        assert float((g.cpu() - want).abs().max()) <= 2e-6 * max(scale, 1e-30) + 1e-12
        assert float(g[valid == 0].abs().max() if (valid == 0).any() else 0.0) == 0.0   # masked pixels get no gradient
    gb = stage1.distill_loss_backward(p.to(torch.bfloat16), t.to(torch.bfloat16), valid, cosine_weight=w_cos, grad_scale=1.0 / accum)
    assert gb.dtype == torch.bfloat16
    assert float((gb.float().cpu() - want).abs().max()) <= 3e-2 * scale    # bf16 inputs and output: rounding only
    assert torch.equal(stage1.distill_loss_backward(p, t, valid, w_cos, 0.5), stage1.distill_loss_backward(p, t, valid, w_cos, 0.5))
    # the loss scale read on the device (esam3_distill_loss_backward_ds) = the same scale folded into grad_scale on the host, bit for bit
    sd = torch.tensor([65536.0, 3.0, 7.0], dtype=torch.float32, device="cuda")
    assert torch.equal(stage1.distill_loss_backward(p, t, valid, w_cos, 0.5, scale_dev=sd), stage1.distill_loss_backward(p, t, valid, w_cos, 0.5 * 65536.0))


@pytest.mark.gpu
def test_paired_forward_composes_trunks_and_loss():
    """BASELINE config 5 (forward): ViT-H teacher trunk + EV-M student trunk on the same images, then the loss.  The
    trunks have their own parity tests (test_students_gpu.py, test_e2e_gpu.py); here the composition is checked:
    the device loss equals the oracle's loss evaluated on the very tensors the two engines produced."""
    from efficientsam3_amd import build_efficientsam3_image_model, build_sam3_image_model, schema
    from oracle import ref_stage1
    teacher = build_sam3_image_model(device="cuda", enable_inst_interactivity=False, dtype="bf16",
                                     state_dict=schema.synthetic_state_dict("sam3", "vit_h", seed=0, enable_inst_interactivity=False))
    student = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                              model_name="b1", dtype="bf16",
                                              state_dict=schema.synthetic_state_dict("efficientvit", "b1", seed=0,
                                                                                     enable_inst_interactivity=False))
    x = torch.from_numpy(np.stack([synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s)) for s in (1, 3)]))
    sizes = [(1008, 1008), (700, 900)]
    out = stage1.paired_forward(teacher, student, x, sizes)
    assert tuple(out["teacher"].shape) == (2, 5184, 1024) == tuple(out["student"].shape)
    nchw = lambda t: t.float().cpu().reshape(2, 72, 72, 1024).permute(0, 3, 1, 2)
    m = ref_stage1.build_valid_mask(1008, sizes, (72, 72))
    assert np.array_equal(out["valid"].cpu().numpy().reshape(2, 1, 72, 72), m.numpy().astype(np.uint8))
    mse = float(ref_stage1.masked_mse(nchw(out["student"]), nchw(out["teacher"]), m))
    cos = float(ref_stage1.masked_cosine_loss(nchw(out["student"]), nchw(out["teacher"]), m))
    assert float(out["mse"]) == pytest.approx(mse, rel=1e-4) and float(out["cosine"]) == pytest.approx(cos, rel=1e-4)
    assert 0.0 < float(out["cosine"]) < 2.0 and float(out["mse"]) > 0.0


# ---- stage-1 input pipeline (BASELINE config 5): ResizeLongestSide + ImageNet mean / std + bottom-right padding -----------
@pytest.fixture(scope="module")
def preproc_gold(golden_dir):
    with open(os.path.join(golden_dir, "stage1", "preproc_manifest.json")) as f:
        return json.load(f), np.load(os.path.join(golden_dir, "stage1", "preproc.npz"))


def test_oracle_preprocess_matches_reference_fixture(preproc_gold):
    """oracle/ref_stage1.preprocess_sa1b against what the REAL ResizeLongestSide / SA1BDataset.norm / .pad produced
    (oracle/gen_golden_stage1_preproc.py): every 61st element bit-exact, row / column sums of channel 0 (every pixel)."""
    from oracle import ref_stage1
    man, arr = preproc_gold
    assert set(man["cases"]) == set(synth.stage1_preproc_cases())
    for name, c in man["cases"].items():
        img = synth.stage1_preproc_image(name)
        assert list(img.shape[:2]) == c["hw"]
        x, hw = ref_stage1.preprocess_sa1b(torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0))), man["img_size"])
        assert list(hw) == c["new_hw"] and tuple(x.shape) == (3, man["img_size"], man["img_size"])
        xn = x.numpy()
        assert np.array_equal(xn.reshape(-1)[::man["stride"]], arr[f"{name}/sample"]), name
        np.testing.assert_allclose(xn[0].astype(np.float64).sum(axis=1), arr[f"{name}/rowsum0"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(xn[0].astype(np.float64).sum(axis=0), arr[f"{name}/colsum0"], rtol=0, atol=1e-9)
        assert (xn[:, hw[0]:, :] == 0).all() and (xn[:, :, hw[1]:] == 0).all()       # the padding is 0, not -mean / std


def test_preprocess_shape_and_prompt_transforms_match_the_oracle():
    """get_preprocess_shape goes through the C ABI (host arithmetic in the library); apply_coords / apply_boxes follow
    transforms.py:35-46."""
    from oracle import ref_stage1
    rng = np.random.default_rng(3)
    for _ in range(200):
        h, w = (int(v) for v in rng.integers(1, 4000, 2))
        for side in (1008, 1024, 64):
            assert stage1.get_preprocess_shape(h, w, side) == ref_stage1.get_preprocess_shape(h, w, side), (h, w, side)
    assert stage1.get_preprocess_shape(1500, 2250, 1008) == (672, 1008)
    pts = np.array([[[10.0, 20.0], [300.0, 150.0]]])
    got = stage1.apply_coords(pts, (600, 800), 1008)
    np.testing.assert_allclose(got, pts * np.array([1008 / 800, 756 / 600]), rtol=1e-12)
    assert pts[0, 0, 0] == 10.0                                                         # the input is not modified
    box = stage1.apply_boxes(np.array([[10.0, 20.0, 300.0, 150.0]]), (600, 800), 1008)
    np.testing.assert_allclose(box, [[12.6, 25.2, 378.0, 189.0]], rtol=1e-12)


def test_preprocess_refuses_the_cpu():
    with pytest.raises(ValueError):
        stage1.preprocess_sa1b([synth.stage1_preproc_image("upscale_300x420")], device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(synth.stage1_preproc_cases()))
def test_preprocess_kernel_vs_reference_fixture_and_oracle(preproc_gold, name):
    """esam3_stage1_preprocess_u8 through stage1.preprocess_sa1b: against the reference's fixture (strided sample, row /
    column sums) and against the oracle run live on every element.  fp32 arithmetic in the reference's summation order;
    2e-5 covers the ulp-level differences of the tap weights (values are O(1) after the normalisation)."""
    from oracle import ref_stage1
    man, arr = preproc_gold
    img = synth.stage1_preproc_image(name)
    x, sizes = stage1.preprocess_sa1b([img], man["img_size"])
    assert sizes == [tuple(man["cases"][name]["new_hw"])]
    got = x[0].cpu().numpy()
    want, _ = ref_stage1.preprocess_sa1b(torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0))), man["img_size"])
    err = float(np.abs(got - want.numpy()).max())
    e_s = float(np.abs(got.reshape(-1)[::man["stride"]] - arr[f"{name}/sample"]).max())
    print(f"[stage1 preprocess] {name}: max-abs-err vs oracle (all elements) {err:.3g}, vs reference fixture sample {e_s:.3g}")
    assert err <= 2e-5 and e_s <= 2e-5
    nh, nw = sizes[0]
    assert (got[:, nh:, :] == 0).all() and (got[:, :, nw:] == 0).all()
    np.testing.assert_allclose(got[0].astype(np.float64).sum(axis=1), arr[f"{name}/rowsum0"], atol=2e-5 * got.shape[2])


@pytest.mark.gpu
def test_preprocess_batch_feeds_paired_forward_shapes(preproc_gold):
    """A list of differently sized images -> one [B, 3, 1008, 1008] batch + sizes_before_pad, the two inputs of
    stage1.paired_forward / valid_mask."""
    man, _ = preproc_gold
    names = ["landscape_600x800", "portrait_900x700", "sa1b_1500x2250"]
    x, sizes = stage1.preprocess_sa1b([synth.stage1_preproc_image(n) for n in names], man["img_size"])
    assert tuple(x.shape) == (3, 3, 1008, 1008) and sizes == [tuple(man["cases"][n]["new_hw"]) for n in names]
    m = stage1.valid_mask(1008, sizes, (72, 72))
    assert m.shape == (3, 72 * 72) and [int(v) for v in m.sum(axis=1)] == [54 * 72, 72 * 56, 48 * 72]


# ---- offline teacher embeddings: the reference's file format ------------------------------------------------------------------
def _reference_manager():
    """the reference's TxtManager, imported from where it lies (pure-Python module); None where /root/reference is absent"""
    import importlib.util
    path = "/root/reference/stage1/data/augmentation/manager.py"
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("ref_stage1_manager", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_embedding_store_round_trip_and_layout(tmp_path):
    shape = (4, 3, 3)
    size = stage1.embedding_item_size(shape)
    rng = np.random.default_rng(5)
    embs = {f"sa_{i}": rng.standard_normal(shape).astype(np.float32) for i in range(5)}
    for rank in (0, 1):
        with stage1.EmbeddingStore(str(tmp_path / "emb"), size, rank=rank) as st:
            for j, (k, e) in enumerate(embs.items()):
                if j % 2 == rank:
                    st.write(k, stage1.pack_embedding(100 + j, e))
                    st.write(k, stage1.pack_embedding(999, e))          # duplicates are skipped, like the reference's writer
    assert sorted(os.listdir(tmp_path / "emb")) == ["rank0-keys.txt", "rank0-values.bin", "rank1-keys.txt", "rank1-values.bin"]
    assert (tmp_path / "emb" / "rank0-keys.txt").read_text() == "sa_0\nsa_2\nsa_4\n"
    assert os.path.getsize(tmp_path / "emb" / "rank0-values.bin") == 3 * size
    rd = stage1.EmbeddingStore(str(tmp_path / "emb"), size, rank=1)
    for j, (k, e) in enumerate(embs.items()):
        seed, back = stage1.unpack_embedding(rd.read(k), shape)
        assert seed == 100 + j and np.array_equal(back, e.astype(np.float16))
    rd.close()
    with pytest.raises(ValueError):
        stage1.EmbeddingStore(str(tmp_path / "x"), size).write("k", b"short")


def test_embedding_store_is_the_reference_format(tmp_path):
    """files written here are read by the reference's TxtManager and vice versa (build container only)"""
    ref = _reference_manager()
    if ref is None:
        pytest.skip("/root/reference is not present")
    shape = (2, 3, 3)
    size = stage1.embedding_item_size(shape)
    rng = np.random.default_rng(6)
    embs = {f"img{i}": rng.standard_normal(shape).astype(np.float32) for i in range(4)}
    with stage1.EmbeddingStore(str(tmp_path / "ours"), size, rank=0) as st:
        for j, (k, e) in enumerate(embs.items()):
            st.write(k, stage1.pack_embedding(j, e))
    mgr = ref.TxtManager(str(tmp_path / "ours"), size, 0)
    for j, (k, e) in enumerate(embs.items()):
        assert mgr.read(k) == stage1.pack_embedding(j, e)
    w = ref.TxtManager(str(tmp_path / "theirs"), size, 0)
    for j, (k, e) in enumerate(embs.items()):
        w.write(k, stage1.pack_embedding(j, e))
    w.writer.__del__()                       # the worker process flushes and moves the files on KILL
    w.writer.worker = None
    rd = stage1.EmbeddingStore(str(tmp_path / "theirs"), size, rank=0)
    for j, (k, e) in enumerate(embs.items()):
        assert rd.read(k) == stage1.pack_embedding(j, e)
    rd.close()


@pytest.mark.gpu
def test_save_teacher_embeddings_writes_the_trunk_output(tmp_path):
    """save_embeddings_one_epoch on the engine with a small student standing in for the teacher (same code path, seconds
    instead of a ViT-H build): the stored fp16 [C, H, W] equals the trunk output of the same preprocessed batch."""
    from efficientsam3_amd import build_efficientsam3_image_model, schema
    model = build_efficientsam3_image_model(device="cuda", enable_inst_interactivity=False, backbone_type="efficientvit",
                                            model_name="b0", dtype="bf16",
                                            state_dict=schema.synthetic_state_dict("efficientvit", "b0", seed=0, enable_inst_interactivity=False))
    names = ["landscape_600x800", "upscale_300x420"]
    imgs = [synth.stage1_preproc_image(n) for n in names]
    shape = (1024, 72, 72)
    with stage1.EmbeddingStore(str(tmp_path / "emb"), stage1.embedding_item_size(shape)) as st:
        n = stage1.save_teacher_embeddings(model, [(imgs, names, [11, 12])], st)
    assert n == 2
    x, _ = stage1.preprocess_sa1b(imgs, 1008)
    want = model.engine.encode(x, want_sam3=False, want_sam2=False, want_trunk=True)["trunk"].permute(0, 3, 1, 2).to(torch.float16).cpu().numpy()
    rd = stage1.EmbeddingStore(str(tmp_path / "emb"), stage1.embedding_item_size(shape))
    for i, k in enumerate(names):
        seed, emb = stage1.unpack_embedding(rd.read(k), shape)
        assert seed == 11 + i and np.array_equal(emb, want[i])
    rd.close()


# ---- the update half of the training step: AMP loss scaler + clip_grad_norm_ + AdamW (esam3_stage1_update) -----------------
@pytest.fixture(scope="module")
def update_gold(golden_dir):
    with open(os.path.join(golden_dir, "stage1", "update_manifest.json")) as f:
        man = json.load(f)
    return man, np.load(os.path.join(golden_dir, "stage1", "update.npz"))


def _names(man):
    return [n for n, _ in man["shapes"]]


def test_update_oracle_reproduces_the_torch_run(update_gold):
    """oracle/ref_stage1.update_step replays the fixture written by the REAL stack (the reference's build_optimizer ->
    torch.optim.AdamW, torch.amp.GradScaler in the call order of NativeScalerWithGradNormCount): parameters after every step
    within 2e-7, loss scale and growth tracker exact, the skipped step skipped, the norm within 1e-5 relative."""
    from oracle import ref_stage1
    man, g = update_gold
    hy, names = man["hyper"], _names(man)
    shapes = [(n, tuple(s)) for n, s in man["shapes"]]
    decay = ref_stage1.weight_decay_groups(shapes, skip_keywords=man["skip_keywords"])
    p = {n: g[f"init/{n}"].copy() for n in names}
    m = {n: np.zeros_like(v) for n, v in p.items()}
    v = {n: np.zeros_like(x) for n, x in p.items()}
    st = {"scale": hy["init_scale"], "tracker": 0, "step": 0}
    for t, s in enumerate(man["steps"]):
        grads = {n: g[f"step{t}/grad/{n}"] for n in names}
        norm, found = ref_stage1.update_step(p, grads, m, v, st, decay, man["lr_scale"], s["lr"], hy["weight_decay"], tuple(hy["betas"]),
                                             hy["eps"], hy["clip_grad"], True, hy["growth_factor"], hy["backoff_factor"], hy["growth_interval"])
        assert found == s["skipped"] and st["scale"] == s["scale_after"] and st["tracker"] == s["growth_tracker"]
        if not s["skipped"]:
            assert norm == pytest.approx(s["grad_norm"], rel=1e-5)
        for n in names:
            assert float(np.abs(p[n] - g[f"step{t}/param/{n}"]).max()) <= 2e-7, (t, n)
    assert st["step"] == man["optimizer_steps_taken"]
    for n in names:
        assert float(np.abs(m[n] - g[f"final/exp_avg/{n}"]).max()) <= 1e-7
        assert np.allclose(v[n], g[f"final/exp_avg_sq/{n}"], rtol=1e-5, atol=1e-12)


def test_arena_layout_groups_are_the_reference_param_groups(update_gold):
    """ArenaLayout's per-chunk tables = the param groups stage1/optimizer.py:build_optimizer produced for the same names (recorded
    in the manifest): weight decay only on the has_decay group, lr_scale per parameter; every tensor starts on a chunk."""
    man, _ = update_gold
    lay = stage1.ArenaLayout([(n, tuple(s)) for n, s in man["shapes"]], skip_keywords=man["skip_keywords"], lr_scales=man["lr_scale"])
    assert lay.n % stage1.CHUNK == 0 and lay.chunk_decay.size == lay.n // stage1.CHUNK == lay.chunk_lr_scale.size
    end = 0
    for (name, shape) in man["shapes"]:
        start, numel = lay.offsets[name]
        assert start % stage1.CHUNK == 0 and start >= end and numel == int(np.prod(shape)) if shape else numel == 1
        end = start + numel
        grp = [gp for gp in man["groups_from_build_optimizer"] if name in gp["names"]]
        assert len(grp) == 1
        c0, c1 = start // stage1.CHUNK, (start + numel + stage1.CHUNK - 1) // stage1.CHUNK
        assert (lay.chunk_decay[c0:c1] == int(grp[0]["decay"])).all(), name
        assert (lay.chunk_lr_scale[c0:c1] == np.float32(grp[0]["lr_scale"])).all(), name


@pytest.mark.gpu
def test_stage1_update_matches_the_torch_run(update_gold):
    """esam3_stage1_update through Stage1Updater vs the fixture of the real AdamW + GradScaler run: parameters within 1e-6 after
    every step (fp32 arithmetic in torch's operation order), loss scale / growth tracker / skipped step exact, gradient norm
    within 1e-5 relative (inf on the overflow step), gradients zeroed, padding untouched, the bf16 copy = the rounded weights."""
    man, g = update_gold
    hy, names = man["hyper"], _names(man)
    lay = stage1.ArenaLayout([(n, tuple(s)) for n, s in man["shapes"]], skip_keywords=man["skip_keywords"], lr_scales=man["lr_scale"])
    up = stage1.Stage1Updater(lay, "cuda", lr=man["steps"][0]["lr"], weight_decay=hy["weight_decay"], betas=tuple(hy["betas"]), eps=hy["eps"],
                              clip_grad=hy["clip_grad"], amp=True, init_scale=hy["init_scale"], growth_factor=hy["growth_factor"],
                              backoff_factor=hy["backoff_factor"], growth_interval=hy["growth_interval"], keep_bf16_copy=True)
    up.load_params({n: g[f"init/{n}"] for n in names})
    used = torch.zeros(lay.n, dtype=torch.bool)
    for n in names:
        s0, ne = lay.offsets[n]
        used[s0:s0 + ne] = True
    for t, s in enumerate(man["steps"]):
        assert float(up.loss_scale) == s["scale_before"]
        for n in names:
            up.grad(n).copy_(torch.from_numpy(g[f"step{t}/grad/{n}"]))
        norm = float(up.step(lr=s["lr"]))
        st = up.state.cpu().numpy()
        assert st[0] == s["scale_after"] and int(st[1]) == s["growth_tracker"] and bool(st[2]) == s["skipped"], (t, st[:5])
        if s["skipped"]:
            assert not np.isfinite(norm)
        else:
            assert norm == pytest.approx(s["grad_norm"], rel=1e-5)
        for n in names:
            assert float(np.abs(up.param(n).cpu().numpy() - g[f"step{t}/param/{n}"]).max()) <= 1e-6, (t, n)
        assert float(up.grads.abs().max()) == 0.0                       # optimizer.zero_grad()
        assert float(up.params.cpu()[~used].abs().max()) == 0.0         # padding stays zero
        assert torch.equal(up.bf16.cpu()[used], up.params.cpu()[used].to(torch.bfloat16)) or s["skipped"]
    assert int(up.state[4]) == man["optimizer_steps_taken"]
    for n in names:
        assert float(np.abs(up.view(up.exp_avg, n).cpu().numpy() - g[f"final/exp_avg/{n}"]).max()) <= 1e-7
        assert np.allclose(up.view(up.exp_avg_sq, n).cpu().numpy(), g[f"final/exp_avg_sq/{n}"], rtol=1e-5, atol=1e-12)
    # checkpoint round trip (utils.py:364-368 state_dict / load_state_dict of the scaler, plus the optimizer's moments)
    sd = up.state_dict()
    assert sd["amp_scaler"]["scale"] == man["steps"][-1]["scale_after"] and sd["optimizer"]["step"] == man["optimizer_steps_taken"]
    up2 = stage1.Stage1Updater(lay, "cuda", amp=True)
    up2.load_state_dict(sd)
    assert torch.equal(up2.exp_avg, up.exp_avg) and float(up2.state[0]) == sd["amp_scaler"]["scale"]


@pytest.mark.gpu
@pytest.mark.parametrize("amp,clip", [(True, 5.0), (False, 5.0), (True, 0.0)])
def test_stage1_update_large_arena_vs_oracle(amp, clip):
    """An EV-M-sized arena (14 M parameters in a handful of tensors, several steps) against the oracle: amp on / off
    (GradScaler(enabled=False): scale 1, never skipped) and clip_grad <= 0 (ampscaler_get_grad_norm: norm only)."""
    from oracle import ref_stage1
    shapes = [("a.weight", (1024, 4096)), ("a.bias", (4096,)), ("b.weight", (2304, 2048)), ("b.norm.weight", (2048,)),
              ("c.weight", (256, 256, 3, 3)), ("c.pos_embed", (1, 5184, 64)), ("d.weight", (4736, 1024))]
    rng = np.random.default_rng(5)
    lay = stage1.ArenaLayout(shapes, skip_keywords=("pos_embed",), lr_scales={"a.weight": 0.25, "a.bias": 0.25})
    up = stage1.Stage1Updater(lay, "cuda", weight_decay=0.05, clip_grad=clip, amp=amp, init_scale=1024.0, growth_interval=2)
    decay = ref_stage1.weight_decay_groups(shapes, skip_keywords=("pos_embed",))
    p = {n: (rng.standard_normal(s) * 0.1).astype(np.float32) for n, s in shapes}
    m = {n: np.zeros_like(v) for n, v in p.items()}
    v = {n: np.zeros_like(x) for n, x in p.items()}
    st = {"scale": 1024.0 if amp else 1.0, "tracker": 0, "step": 0}
    up.load_params(p)
    for t, lr in enumerate((5e-4, 3e-4, 1e-4)):
        gain = (1e-5, 3e-3, 1e-4)[t] * st["scale"]
        grads = {n: (rng.standard_normal(s) * gain).astype(np.float32) for n, s in shapes}
        if t == 1 and amp:
            grads["c.weight"].reshape(-1)[12345] = np.nan
        for n, _ in shapes:
            up.grad(n).copy_(torch.from_numpy(grads[n]))
        norm = float(up.step(lr=lr))
        o_norm, found = ref_stage1.update_step(p, grads, m, v, st, decay, {"a.weight": 0.25, "a.bias": 0.25}, lr, 0.05, (0.9, 0.999), 1e-8,
                                               clip if clip > 0 else None, amp, 2.0, 0.5, 2)
        s_dev = up.state.cpu().numpy()
        assert bool(s_dev[2]) == found and s_dev[0] == np.float32(st["scale"]) and int(s_dev[1]) == st["tracker"] and int(s_dev[4]) == st["step"]
        if np.isfinite(o_norm):
            assert norm == pytest.approx(o_norm, rel=2e-5)
        for n, _ in shapes:
            assert float(np.abs(up.param(n).cpu().numpy() - p[n]).max()) <= 2e-6, (t, n)


# ---- BatchNorm2d in training mode (building blocks of the trunk backward) --------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 9, 7, 24), (3, 16, 16, 64), (1, 5, 5, 8), (2, 20, 12, 256), (1, 33, 31, 1024), (2, 3, 2, 2048),
                                   (4, 63, 63, 128)])
def test_bn_train_forward_backward_vs_torch(mode, shape):
    """esam3_bn_train_forward / _backward (NHWC) against torch's batch_norm in training mode + autograd on the same (bf16-quantised
    where the mode is bf16) inputs: output, saved statistics, running statistics (unbiased variance, momentum 0.1), dx, dgamma, dbeta."""
    import torch.nn.functional as F
    tdt = torch.float32 if mode == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    x = (torch.randn(shape, generator=g) * (0.5 + torch.rand(c, generator=g) * 2.0) + torch.randn(c, generator=g)).to(tdt)
    dy = torch.randn(shape, generator=g).to(tdt)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    # reference: fp32 arithmetic on the quantised inputs, NCHW as the module sees it
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    yr.backward(dy.float().permute(0, 3, 1, 2).contiguous())
    to_nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()  # noqa: E731
    dev = "cuda"
    rm_d, rv_d = rm0.to(dev), rv0.to(dev)
    y, mean, rstd = stage1.bn_train_forward(x.to(dev).contiguous(), gamma.to(dev), beta.to(dev), rm_d, rv_d, momentum=0.1, eps=1e-5)
    dx, dgamma, dbeta = stage1.bn_train_backward(x.to(dev).contiguous(), dy.to(dev).contiguous(), gamma.to(dev), mean, rstd)
    n = x.numel() // c
    xf = x.float().reshape(n, c)
    assert torch.allclose(mean.cpu(), xf.mean(0), rtol=1e-5, atol=1e-5)
    assert torch.allclose(rstd.cpu(), 1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5), rtol=2e-5, atol=1e-6)
    assert torch.allclose(rm_d.cpu(), rm, rtol=1e-5, atol=1e-5) and torch.allclose(rv_d.cpu(), rv, rtol=2e-5, atol=1e-5)
    rel = 2.0 ** -8 if mode == "bf16" else 2e-5     # bf16: the outputs are rounded once
    for got, ref, what in ((y, to_nhwc(yr.detach()), "y"), (dx, to_nhwc(xr.grad), "dx")):
        d = (got.float().cpu() - ref).abs()
        assert bool((d <= rel * ref.abs() + (2e-3 if mode == "bf16" else 2e-5) * float(ref.abs().max())).all()), (what, float(d.max()))
    assert torch.allclose(dbeta.cpu(), br.grad, rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))
    assert torch.allclose(dgamma.cpu(), gr.grad, rtol=1e-4, atol=1e-4 * float(gr.grad.abs().max()))
    with pytest.raises(RuntimeError):   # C not a multiple of 8
        stage1.bn_train_forward(torch.zeros((4, 12), device=dev), torch.ones(12, device=dev), torch.zeros(12, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,offset", [((2, 20, 12, 64), 1.0e3), ((4, 63, 63, 128), 3.0e3), ((1, 5, 5, 8), 1.0e4)])
def test_bn_train_statistics_with_a_large_offset(shape, offset):
    """|mean| >> std (a large conv bias in front of the norm): E[x^2] - E[x]^2 on raw fp32 sums loses the variance entirely at
    offset / std = 1e3 (relative precision 2^-24 x 1e6 of the square); the kernels accumulate sums shifted by a per-channel pivot,
    so mean, rstd and the normalised output still match a float64 reference."""
    g = torch.Generator().manual_seed(7)
    c = shape[-1]
    std = 0.5 + torch.rand(c, generator=g)
    x = torch.randn(shape, generator=g) * std + offset * (1.0 + torch.rand(c, generator=g))
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    dev = "cuda"
    y, mean, rstd = stage1.bn_train_forward(x.to(dev).contiguous(), gamma.to(dev), beta.to(dev), None, None, momentum=0.1, eps=1e-5)
    xd = x.double().reshape(-1, c)
    m64, v64 = xd.mean(0), xd.var(0, unbiased=False)
    r64 = 1.0 / torch.sqrt(v64 + 1e-5)
    assert torch.allclose(mean.cpu().double(), m64, rtol=1e-6, atol=0)
    assert torch.allclose(rstd.cpu().double(), r64, rtol=1e-3, atol=0), float(((rstd.cpu().double() - r64) / r64).abs().max())
    y64 = ((xd - m64) * r64 * gamma.double() + beta.double()).reshape(shape)
    # the input itself is fp32: x - mean carries an absolute error of 2^-24 |x| ~ 6e-5 x (offset / 1e3), times rstd
    tol = 4.0 * offset * 2.0 ** -24 * float(r64.max()) * float(gamma.max()) + 1e-5
    assert float((y.cpu().double() - y64).abs().max()) <= tol


def test_adamw_state_round_trips_through_torch_optim():
    """Resume in both directions (host logic, no device work): the state of a REAL torch.optim.AdamW with the two parameter groups of
    stage1/optimizer.py:32-46 (has_decay first, then no_decay: 1-D tensors and biases) after three steps -> `load_adamw_state_dict` into the
    arena -> `adamw_state_dict` back out: moments, step count and groups identical, and torch's own `load_state_dict` accepts the export."""
    from efficientsam3_amd.stage1 import ArenaLayout, Stage1Updater
    from efficientsam3_amd.stage1_train import adamw_state_dict, load_adamw_state_dict
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 5, 1, bias=True), torch.nn.Linear(5, 4))
    named = list(net.named_parameters())
    no_decay = lambda n, p: p.ndim == 1 or n.endswith(".bias")  # noqa: E731   (set_weight_decay, stage1/optimizer.py:36-44)
    groups = [{"params": [p for n, p in named if not no_decay(n, p)]}, {"params": [p for n, p in named if no_decay(n, p)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=3e-4, betas=(0.9, 0.95), eps=1e-7, weight_decay=0.05)
    for step in range(3):
        for _, p in named:
            p.grad = torch.randn_like(p) * (step + 1)
        opt.step()
    sd = opt.state_dict()
    names = [n for n, _ in named]
    up = Stage1Updater(ArenaLayout([(n, p.shape) for n, p in named]), "cpu", amp=False)
    load_adamw_state_dict(up, names, sd)
    assert up.lr == 3e-4 and tuple(up.betas) == (0.9, 0.95) and up.eps == 1e-7 and up.weight_decay == 0.05 and int(up.state[4]) == 3
    out = adamw_state_dict(up, names)
    order = [n for n, p in named if not no_decay(n, p)] + [n for n, p in named if no_decay(n, p)]
    assert out["param_names"] == order
    assert [g_["params"] for g_ in out["param_groups"]] == [g_["params"] for g_ in sd["param_groups"]]
    assert out["param_groups"][0]["weight_decay"] == 0.05 and out["param_groups"][1]["weight_decay"] == 0.0
    for i in sd["state"]:
        assert torch.equal(out["state"][i]["exp_avg"], sd["state"][i]["exp_avg"]) and torch.equal(out["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
        assert float(out["state"][i]["step"]) == float(sd["state"][i]["step"]) == 3.0
    # the padding between parameters stays zero (AdamW leaves zero-gradient elements with zero moments at zero)
    used = torch.zeros(up.layout.n, dtype=torch.bool)
    for n in names:
        s0, num = up.layout.offsets[n]
        used[s0:s0 + num] = True
    assert float(up.exp_avg[~used].abs().max()) == 0.0 and float(up.exp_avg_sq[~used].abs().max()) == 0.0
    opt2 = torch.optim.AdamW(groups, lr=1.0)
    opt2.load_state_dict({"state": out["state"], "param_groups": [{k: v for k, v in g_.items()} for g_ in out["param_groups"]]})
    assert opt2.state_dict()["param_groups"][0]["lr"] == 3e-4
    with pytest.raises(ValueError):
        load_adamw_state_dict(up, names[:-1], sd)
