"""Stage-1 TRAINING STEP on the GPU: the device-resident-weight operators (csrc/kernels_train_dev.hip), the student
head's forward / backward, and two whole iterations of `stage1_train.Stage1Trainer` against the fixture the REAL reference stack wrote
(oracle/gen_golden_stage1_step.py: stage1/model.py + stage1/optimizer.py + the loss functions of train_image_encoder_stage1.py, two
steps on the CPU in fp32, plus a bf16-autocast run as the yardstick)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from efficientsam3_amd import schema, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "stage1")
PREFIX = "backbone.vision_backbone.trunk.model."
DTS = {"f32": torch.float32, "bf16": torch.bfloat16}

# step-1 gradient limits of the fp32 run: max-abs-err <= GRAD_REL x the tensor's largest gradient + GRAD_ABS x the network's largest.
# Achieved on the MI355X (round 5, profiles/r05/parity_margins.txt): the worst tensor WITH a gradient is the stem's depthwise weight --
# 1.66e-2 of its own maximum (nine sums of 2 M products each at 504 x 504, heavy cancellation, another summation order than torch's);
# the five next-worst tensors are printed by the test.  The noise-only tensors (analytically zero gradient) reach 7.1e-6 of the network's
# largest gradient: GRAD_ABS = 3.5 x that (it was 1e-4).  GRAD_REL stays at 1.5 x the achieved worst.
GRAD_REL, GRAD_ABS = 2.5e-2, 2.5e-5
RV_GRAD_REL, RV_GRAD_ABS = 2.5e-3, 5e-6    # the RepViT students (round 5): first run used 0.015 - 0.018 of (5e-2, 1e-4), worst tensor 4.6e-4 of its
                                           # maximum (profiles/r05/parity_margins_repvit_steps.txt); this allowance is 1 / 20 of that one
TV_GRAD_REL, TV_GRAD_ABS = 2.5e-3, 2.5e-6    # the TinyViT students (round 5): first run used 0.011 of (2.5e-2, 2.5e-5), worst tensor 3.0e-4 of its maximum
                                             # (profiles/r05/parity_margins_tinyvit_steps.txt); this allowance is 1 / 10 of that one
B2_GRAD_REL, B2_GRAD_ABS = 5e-2, 1e-4      # EfficientViT-B2: 35 x larger gradients through a deeper chain of training-mode BatchNorms



def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _q(x, mode):
    return x.to(torch.bfloat16).float() if mode == "bf16" else x


def _nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)


def _nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def _close(got, ref, mode, what, f32=2e-4, bf16=2e-2):
    tol = f32 if mode == "f32" else bf16
    err = float((got.double() - ref.double()).abs().max())
    peak = float(ref.abs().max())
    assert err <= tol * max(peak, 1e-6), (what, mode, err, peak)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_train_ops_on_device_weights(mode):
    """esam3_train_linear (+ transposed), esam3_train_conv3x3 (+ dgrad), esam3_train_dwconv, esam3_train_stem,
    esam3_resize_bilinear_backward against torch on the same (quantised) inputs; weights are DEVICE fp32 tensors."""
    from efficientsam3_amd import stage1_train as st
    from efficientsam3_amd import train_blocks as tb
    dt = DTS[mode]
    # Linear + its data gradient
    x, w, b = _rand(300, 48, seed=1), _rand(72, 48, seed=2) / 7.0, _rand(72, seed=3) * 0.1
    y = tb.linear_forward(x.to("cuda", dt), w.cuda(), b.cuda())
    _close(y.float().cpu(), F.linear(_q(x, mode), _q(w, mode), b), mode, "linear")
    dy = _rand(300, 72, seed=4)
    dx = tb.linear_dgrad(dy.to("cuda", dt), w.cuda())
    _close(dx.float().cpu(), _q(dy, mode) @ _q(w, mode), mode, "linear dgrad")
    # dense 3x3 + data gradient (Cin = 128 exercises the channel-chunk-major K order in bf16, Cin = 40 the tap-major one)
    for cin, cout in ((40, 24), (128, 64)):
        xi, wc, bc = _rand(2, cin, 9, 11, seed=5), _rand(cout, cin, 3, 3, seed=6) / (3.0 * cin ** 0.5), _rand(cout, seed=7) * 0.1
        yc = st.conv3x3_forward(_nhwc(xi, dt), wc.cuda(), bc.cuda())
        _close(_nchw(yc), F.conv2d(_q(xi, mode), _q(wc, mode), bc, padding=1), mode, f"conv3x3 {cin}->{cout}")
        dyc = _rand(2, cout, 9, 11, seed=8)
        dxc = st.conv3x3_dgrad(_nhwc(dyc, dt), wc.cuda())
        _close(_nchw(dxc), F.conv_transpose2d(_q(dyc, mode), _q(wc, mode), padding=1), mode, f"conv3x3 dgrad {cin}->{cout}")
        xr = _q(xi, mode).clone()
        wr = wc.clone().requires_grad_(True)
        F.conv2d(xr, wr, None, padding=1).backward(_q(dyc, mode))
        dwc = st.conv3x3_wgrad(_nhwc(dyc, dt), _nhwc(xi, dt))
        _close(dwc.cpu(), wr.grad, mode, f"conv3x3 wgrad {cin}->{cout}", f32=1e-4, bf16=1e-2)
    # depthwise, stem
    xd, wd_ = _rand(2, 64, 13, 10, seed=9), _rand(64, 1, 3, 3, seed=10) * 0.3
    yd = tb.dwconv_forward(_nhwc(xd, dt), wd_.cuda(), 2)
    _close(_nchw(yd), F.conv2d(_q(xd, mode), wd_ if mode == "f32" else _q(wd_, mode), None, stride=2, padding=1, groups=64), mode, "dwconv s2", bf16=3e-2)
    img, ws_ = _rand(2, 3, 38, 42, seed=11), _rand(16, 3, 3, 3, seed=12) / 27 ** 0.5
    ys = tb.stem_forward(img.cuda(), ws_.cuda(), dt)
    _close(_nchw(ys), F.conv2d(img, ws_, None, stride=2, padding=1), mode, "stem", bf16=3e-2)
    # bilinear resize: forward's adjoint
    for (ih, oh) in ((32, 72), (9, 20), (16, 16), (20, 9)):
        xr = _rand(2, 16, ih, ih, seed=13).requires_grad_(True)
        dyr = _rand(2, 16, oh, oh, seed=14)
        F.interpolate(xr, size=(oh, oh), mode="bilinear", align_corners=False).backward(_q(dyr, mode))
        dxr = st.resize_backward(_nhwc(dyr, dt), (ih, ih))
        _close(_nchw(dxr), xr.grad, mode, f"resize backward {ih}->{oh}")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_head_forward_backward_vs_autograd(mode):
    """HeadTrain (stage1/model.py:193-211: Conv1x1 no bias + BatchNorm (training) + GELU + Conv3x3 + bilinear resize) against autograd"""
    from efficientsam3_amd import stage1_train as st
    from efficientsam3_amd import train_blocks as tb
    tb.DEVICE = "cuda"
    dt = DTS[mode]
    b, cin, e, h, s = 2, 32, 64, 8, 18
    sd = {"head.0.weight": _rand(e, cin, 1, 1, seed=1) / cin ** 0.5, "head.1.weight": torch.rand(e, generator=torch.Generator().manual_seed(2)) + 0.5,
          "head.1.bias": _rand(e, seed=3) * 0.2, "head.3.weight": _rand(e, e, 3, 3, seed=4) / (3.0 * e ** 0.5), "head.3.bias": _rand(e, seed=5) * 0.1,
          "head.1.running_mean": _rand(e, seed=6) * 0.1, "head.1.running_var": torch.rand(e, generator=torch.Generator().manual_seed(7)) + 0.5}
    x, dy = _rand(b, cin, h, h, seed=8), _rand(b, e, s, s, seed=9)
    p = {k: v.clone().requires_grad_(not k.endswith(("running_mean", "running_var"))) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    a = F.conv2d(xr, p["head.0.weight"])
    rm, rv = p["head.1.running_mean"].clone(), p["head.1.running_var"].clone()
    a = F.gelu(F.batch_norm(a, rm, rv, p["head.1.weight"], p["head.1.bias"], training=True, momentum=0.1, eps=1e-5))
    yr = F.interpolate(F.conv2d(a, p["head.3.weight"], p["head.3.bias"], padding=1), size=(s, s), mode="bilinear", align_corners=False)
    yr.backward(dy)
    head = st.HeadTrain({k: v.cuda() for k, v in sd.items()}, embed_size=s)
    y = head.forward(_nhwc(x, dt))
    dx, grads = head.backward(_nhwc(dy, dt))
    tol = dict(f32=5e-4, bf16=4e-2)
    _close(_nchw(y), yr.detach(), mode, "head y", **tol)
    _close(_nchw(dx), xr.grad, mode, "head dx", **tol)
    for k in ("head.0.weight", "head.1.weight", "head.1.bias", "head.3.weight", "head.3.bias"):
        assert tuple(grads[k].shape) == tuple(sd[k].shape), k          # state-dict shapes
        _close(grads[k].cpu(), p[k].grad, mode, k, **tol)
    assert torch.allclose(head.l0.running_mean.cpu(), rm, atol=1e-4 if mode == "f32" else 2e-2)
    assert torch.allclose(head.l0.running_var.cpu(), rv, atol=1e-4 if mode == "f32" else 2e-2)


def _student_sd():
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0)
    return {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}


def _inputs(man):
    imgs = torch.stack([torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s))) for s in man["image_seeds"]])
    for i, (h, w) in enumerate(man["sizes_before_pad"]):
        imgs[i, :, h:, :] = 0
        imgs[i, :, :, w:] = 0
    g = torch.Generator().manual_seed(man["teacher_seed"])
    teacher = torch.randn((len(man["sizes_before_pad"]), man["embed_dim"], man["embed_size"], man["embed_size"]), generator=g) * 0.5
    return imgs, teacher.permute(0, 2, 3, 1).contiguous()


def _sample(t, n):
    flat = t.detach().float().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].numpy()


@pytest.fixture(scope="module")
def step_gold():
    with open(os.path.join(GOLD, "step_manifest.json")) as f:
        return json.load(f), np.load(os.path.join(GOLD, "step.npz"))


def test_two_training_steps_match_the_reference_run(step_gold):
    """Two iterations of train_one_epoch (stage1/train_image_encoder_stage1.py:165-226) in fp32 on the HIP kernels -- forward of the
    EfficientViT-B1 student at 1008^2 (batch 2, BatchNorm in training mode), masked MSE + 0.5 x cosine, backward through head and trunk
    into the gradient arena, clip at 5, AdamW, zero_grad -- against the REAL reference stack's run: loss and total gradient norm of both
    steps, every parameter's gradient (samples), every parameter after step 1 and step 2 (samples), the BatchNorm running statistics."""
    from efficientsam3_amd.stage1_train import Stage1Trainer
    man, g = step_gold
    hy = man["hyper"]
    sd = _student_sd()
    assert set(man["fp32"]["names"]) == {k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    tr = Stage1Trainer(sd, "b1", embed_size=man["embed_size"], dtype="f32", lr=hy["lr"], weight_decay=hy["weight_decay"], betas=tuple(hy["betas"]),
                       eps=hy["eps"], clip_grad=hy["clip_grad"], amp=False, cosine_weight=hy["cosine"], accumulation_steps=hy["accumulation_steps"])
    imgs, teacher = _inputs(man)
    imgs, teacher = imgs.cuda(), teacher.cuda()
    ns = man["samples_per_tensor"]
    ref = man["fp32"]
    gmax = max(float(g[f"gradmax1/{n}"]) for n in ref["names"])
    lr = hy["lr"]
    own = []                 # our own clipped gradients and parameters per step: the recurrence check below
    prev = {n: _sample(v, ns) for n, v in tr.state_dict().items() if n in ref["shapes"]}
    for step in range(2):
        # the gradients the update consumes are captured before the step zeroes them
        out = tr.step(imgs, teacher, [tuple(s) for s in man["sizes_before_pad"]], update_grad=False)
        grads = {n: _sample(v, ns) for n, v in tr.gradients().items()}
        tr._allreduce()
        norm = float(tr.updater.step())
        tr._micro = 0
        loss = float(out["loss"])
        clip = min(1.0, hy["clip_grad"] / (norm + 1e-6))          # torch.nn.utils.clip_grad_norm_'s coefficient
        print(f"[stage-1 step {step + 1}] loss {loss:.6f} (reference {ref['losses'][step]:.6f})  grad norm {norm:.6f} ({ref['grad_norms'][step]:.6f})")
        assert abs(loss - ref["losses"][step]) <= (1e-4 if step == 0 else 1e-3) * abs(ref["losses"][step])
        # the total gradient norm of step 1 is tight. Step 2 is ill-conditioned BY CONSTRUCTION: AdamW's first update is -lr sign(g), so
        # every element whose step-1 gradient is rounding noise (analytically zero: shifts feeding a training-mode BatchNorm, dead
        # channels) moves by +-lr with a noise sign, in the reference as well; about sixty training-mode BatchNorms in a row amplify
        # that (the reference's OWN bf16-autocast run reports 2.9 x its fp32 norm there, step_manifest.json). The step-2 state is
        # pinned by the loss (1e-3), by the recurrence on our own gradients (tight) and statistically against the reference's parameters.
        # (round 5) the step-2 bound is no longer a hand-set 15 % -- two equally valid fp32 summation orders of the HIP kernels gave 7 921 and
        # 11 184 against the reference's 8 957 -- but the reference's OWN sensitivity: its bf16-autocast run reports 25 748 there (2.9 x its
        # fp32 norm); the step-2 norm has to stay within HALF of that distance.  What pins step 2 tightly is the loss and the recurrence below.
        if step == 0:
            assert abs(norm - ref["grad_norms"][0]) <= 2e-3 * ref["grad_norms"][0]
        else:
            ref_dist = abs(man["bf16_autocast"]["grad_norms"][1] - ref["grad_norms"][1])
            print(f"  step-2 gradient norm: |ours - reference| = {abs(norm - ref['grad_norms'][1]):.1f}; the reference's own bf16-vs-fp32 distance there = {ref_dist:.1f} "
                  f"(allowed: half of it)")
            assert abs(norm - ref["grad_norms"][1]) <= 0.5 * ref_dist
        if step == 0:
            # Achieved margins (profiles/r05/parity_margins.txt keeps the printed line): tensors whose own gradient is above 1e-3 of the
            # network's largest are judged relative to THEIR maximum, the rest (shifts feeding a training-mode BatchNorm: analytically
            # zero gradient, rounding noise on both sides) relative to the network's largest gradient
            worst_rel, worst_rel_n, worst_abs, worst_abs_n, used = 0.0, "", 0.0, "", 0.0
            ranked = []
            for n in ref["names"]:
                got = grads[n] * clip                 # the fixture's gradients are the clipped ones (what AdamW saw)
                want = g[f"grad1/{n}"]
                tmax = float(g[f"gradmax1/{n}"])
                err = float(np.abs(got - want).max())
                if tmax >= 1e-3 * gmax:
                    if err / tmax > worst_rel:
                        worst_rel, worst_rel_n = err / tmax, n
                elif err / gmax > worst_abs:
                    worst_abs, worst_abs_n = err / gmax, n
                used = max(used, err / (GRAD_REL * tmax + GRAD_ABS * gmax))
                if tmax >= 1e-3 * gmax:
                    ranked.append((err / tmax, n))
                assert err <= GRAD_REL * tmax + GRAD_ABS * gmax, (n, err, tmax, gmax)
            print(f"  gradients: worst max-abs-err / tensor max = {worst_rel:.3e} ({worst_rel_n}) over the tensors with a gradient; "
                  f"worst max-abs-err / network max = {worst_abs:.3e} ({worst_abs_n}) over the noise-only tensors; "
                  f"largest fraction of the allowance ({GRAD_REL:g} x tensor max + {GRAD_ABS:g} x network max) used = {used:.3f}")
            ranked.sort(reverse=True)
            print("  gradients, the six worst tensors (max-abs-err / tensor max): " + "; ".join(f"{r:.2e} {n}" for r, n in ranked[:6])
                  + f"; median over {len(ranked)} tensors {ranked[len(ranked) // 2][0]:.2e}")
        params = {n: _sample(v, ns) for n, v in tr.state_dict().items() if n in ref["shapes"]}
        own.append({n: grads[n].astype(np.float64) * clip for n in ref["names"]})
        # (1) the update itself: torch.optim.AdamW's recurrence (decoupled decay, bias-corrected moments) on OUR clipped gradients of
        # step 1 and step 2 reproduces our parameters -- the first / second moments carried between the steps, the step counter, the
        # two decay groups of stage1/optimizer.py:32-46
        b1, b2 = hy["betas"]
        worst_rec = 0.0
        for n in ref["names"]:
            decay = 0.0 if (len(ref["shapes"][n]) == 1 or n.endswith(".bias")) else hy["weight_decay"]
            m = v = 0.0
            for t, gk in enumerate(own, start=1):
                m = b1 * m + (1 - b1) * gk[n]
                v = b2 * v + (1 - b2) * gk[n] ** 2
            t = len(own)
            want = prev[n].astype(np.float64) * (1 - lr * decay) - lr * (m / (1 - b1 ** t)) / (np.sqrt(v / (1 - b2 ** t)) + hy["eps"])
            worst_rec = max(worst_rec, float(np.abs(params[n] - want).max()))
        print(f"  AdamW recurrence on our own gradients: worst |param - expected| = {worst_rec:.3e} ({worst_rec / lr:.2e} lr)")
        assert worst_rec <= 2e-3 * lr + 2e-7            # fp32 parameter rounding (|p| up to ~1) + the kernel's fp32 moments
        prev = params
        if step == 0:
            # running statistics after ONE training-mode forward: momentum 0.1, unbiased batch variance (torch.nn.BatchNorm2d)
            sd1 = tr.state_dict()
            for k in [k for k in sd1 if k.endswith(("running_mean", "running_var"))]:
                want = g[f"buffer1/{k}"]
                assert np.abs(_sample(sd1[k], ns) - want).max() <= 1e-3 * max(1.0, float(np.abs(want).max())), k
        # (2) against the reference's parameters
        nconf = nbad = 0
        diffs = []
        for n in ref["names"]:
            got, want = params[n], g[f"param{step + 1}/{n}"]
            # elements whose step-1 gradient is far above the rounding noise: AdamW's first update is -lr sign(g) (+ decay), reproducible;
            # its second, lr (0.9 g1 + g2) / 1.9 / sqrt((0.999 g1^2 + g2^2) / 1.999), follows the ill-conditioned step-2 gradient
            conf = np.abs(g[f"grad1/{n}"]) > 1e-3 * float(g[f"gradmax1/{n}"]) + 1e-6 * gmax
            tol = (2e-2 if step == 0 else 0.2) * lr
            bad = np.abs(got - want)[conf] > tol
            diffs.append(np.abs(got - want)[conf])
            nconf += int(conf.sum()); nbad += int(bad.sum())
            assert np.abs(got - want).max() <= 2.5 * (step + 1) * lr + 1e-6, n        # nothing moves further than the steps allow
        diffs = np.concatenate(diffs)
        q = np.quantile(diffs, [0.5, 0.9, 0.99]) / lr
        print(f"  parameters after step {step + 1}: {nbad} of {nconf} confident samples outside {tol / lr:.2f} lr; |diff| quantiles 50/90/99 % = {q[0]:.3f} / {q[1]:.3f} / {q[2]:.3f} lr")
        if step == 0:
            assert nbad <= 2e-3 * nconf, (nbad, nconf)
        else:
            assert nbad <= 0.3 * nconf and q[0] <= 0.1, (nbad, nconf, q)
    # BatchNorm buffers after two training-mode forwards (step 2's batch statistics see the ill-conditioned parameters: looser)
    sd2 = tr.state_dict()
    for k in [k for k in sd2 if k.endswith(("running_mean", "running_var"))]:
        want = g[f"buffer2/{k}"]
        got = _sample(sd2[k], ns)
        assert np.abs(got - want).max() <= 2e-2 * max(1.0, float(np.abs(want).max())), k
    assert int(sd2["head.1.num_batches_tracked"]) == int(sd["head.1.num_batches_tracked"]) + 2


def test_bf16_training_step_inside_the_reference_autocast_yardstick(step_gold):
    """the same two iterations with bf16 activations (fp32 master weights and gradients): loss and gradient norm against the
    reference's fp32 run, allowed 1.5 x the distance of the reference's own bf16-autocast run (+ 1 % of the value)"""
    from efficientsam3_amd.stage1_train import Stage1Trainer
    man, g = step_gold
    hy = man["hyper"]
    tr = Stage1Trainer(_student_sd(), "b1", embed_size=man["embed_size"], dtype="bf16", lr=hy["lr"], weight_decay=hy["weight_decay"],
                       betas=tuple(hy["betas"]), eps=hy["eps"], clip_grad=hy["clip_grad"], amp=False, cosine_weight=hy["cosine"])
    imgs, teacher = _inputs(man)
    imgs, teacher = imgs.cuda(), teacher.cuda().to(torch.bfloat16)
    for step in range(2):
        out = tr.step(imgs, teacher, [tuple(s) for s in man["sizes_before_pad"]])
        loss, norm = float(out["loss"]), float(out["grad_norm"])
        r32, r16 = man["fp32"], man["bf16_autocast"]
        lim_l = 1.5 * abs(r16["losses"][step] - r32["losses"][step]) + 1e-2 * abs(r32["losses"][step])
        lim_n = 1.5 * abs(r16["grad_norms"][step] - r32["grad_norms"][step]) + 5e-2 * r32["grad_norms"][step]
        print(f"[stage-1 bf16 step {step + 1}] loss {loss:.5f} (fp32 ref {r32['losses'][step]:.5f}, ref bf16 {r16['losses'][step]:.5f}, allowed +-{lim_l:.4f}) "
              f"grad norm {norm:.4f} ({r32['grad_norms'][step]:.4f} / {r16['grad_norms'][step]:.4f}, +-{lim_n:.4f})")
        assert np.isfinite(loss) and np.isfinite(norm)
        assert abs(loss - r32["losses"][step]) <= lim_l and abs(norm - r32["grad_norms"][step]) <= lim_n


# Note on the gradient-norm column of the RepViT / TinyViT fixtures: this trainer's norm is 2.8e-4 - 3.4e-4 ABOVE the fixture's in every model and
# step, on the GPU and in the CPU emulation alike.  The fixture holds what the reference's loop logs, torch.nn.utils.clip_grad_norm_'s fp32
# value; the float64 norm of the reference's own gradients (computed when the offset was chased: 673.4554 for repvit_m0_9 against the logged
# 673.2244) equals this trainer's (673.4554): clip_grad_norm_ sums the 9.4 M squares of head.3.weight in fp32.  Hence norm limits of 2e-3, not 1e-5.


def _fixture_drop_path(g, step_no):
    """the DropPath factors the reference's run drew (oracle/gen_golden_stage1_step.py records them per module and call), as the trainer's
    ``drop_path_sampler``; None for fixtures of students without stochastic depth"""
    if not any(k.startswith("droppath") for k in g.files):
        return None
    return lambda mod, call, batch, keep: torch.from_numpy(g[f"droppath{step_no[0]}/backbone.model.{mod}/{call}"])


def _first_step_vs_reference(model, suffix, sd, grad_rel, grad_abs, second_step=None, check_buffers=False, loss_rel=1e-4, norm_rel=1e-2):
    """the first iteration of the trainer (fp32) against the reference stack's own run (tests/golden/stage1/step_<suffix>.*): loss, total
    gradient norm, every parameter's clipped gradient (samples), every parameter after the update; ``second_step`` = (loss rel, norm rel):
    also the loss and gradient norm of a second iteration"""
    from efficientsam3_amd.stage1_train import Stage1Trainer
    with open(os.path.join(GOLD, f"step_{suffix}_manifest.json")) as f:
        man = json.load(f)
    g = np.load(os.path.join(GOLD, f"step_{suffix}.npz"))
    hy, ref, ns = man["hyper"], man["fp32"], man["samples_per_tensor"]
    assert set(ref["names"]) == {k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    step_no = [1]
    tr = Stage1Trainer(sd, model, embed_size=man["embed_size"], dtype="f32", lr=hy["lr"], weight_decay=hy["weight_decay"], betas=tuple(hy["betas"]),
                       eps=hy["eps"], clip_grad=hy["clip_grad"], amp=False, cosine_weight=hy["cosine"], accumulation_steps=hy["accumulation_steps"],
                       drop_path_sampler=_fixture_drop_path(g, step_no))
    imgs, teacher = _inputs(man)
    imgs, teacher = imgs.cuda(), teacher.cuda()
    sizes = [tuple(s_) for s_ in man["sizes_before_pad"]]
    out = tr.step(imgs, teacher, sizes, update_grad=False)
    grads = {n: _sample(v, ns) for n, v in tr.gradients().items()}
    tr._allreduce()
    norm = float(tr.updater.step())
    tr._micro = 0
    loss = float(out["loss"])
    clip = min(1.0, hy["clip_grad"] / (norm + 1e-6))
    print(f"[stage-1 {model} step 1] loss {loss:.6f} (reference {ref['losses'][0]:.6f})  grad norm {norm:.3f} ({ref['grad_norms'][0]:.3f})")
    assert abs(loss - ref["losses"][0]) <= loss_rel * abs(ref["losses"][0])
    assert abs(norm - ref["grad_norms"][0]) <= norm_rel * ref["grad_norms"][0]
    gmax = max(float(g[f"gradmax1/{n}"]) for n in ref["names"])
    ranked, used = [], 0.0
    for n in ref["names"]:
        err = float(np.abs(grads[n] * clip - g[f"grad1/{n}"]).max())
        tmax = float(g[f"gradmax1/{n}"])
        used = max(used, err / (grad_rel * tmax + grad_abs * gmax))
        if tmax >= 1e-3 * gmax:
            ranked.append((err / tmax, n))
    ranked.sort(reverse=True)
    print(f"  gradients: largest fraction of the allowance ({grad_rel:g} x tensor max + {grad_abs:g} x network max) used = {used:.3f}; six worst "
          "tensors (max-abs-err / tensor max): " + "; ".join(f"{r:.2e} {n}" for r, n in ranked[:6]))
    assert used <= 1.0
    lr, nbad, nconf = hy["lr"], 0, 0
    params = {n: _sample(v, ns) for n, v in tr.state_dict().items() if n in ref["shapes"]}
    for n in ref["names"]:
        conf = np.abs(g[f"grad1/{n}"]) > 1e-3 * float(g[f"gradmax1/{n}"]) + 1e-6 * gmax
        d = np.abs(params[n] - g[f"param1/{n}"])
        assert d.max() <= 2.5 * lr + 1e-6, n
        nconf += int(conf.sum()); nbad += int((d[conf] > 2e-2 * lr).sum())
    print(f"  parameters after step 1: {nbad} of {nconf} confident samples outside 0.02 lr")
    assert nbad <= 5e-3 * nconf, (nbad, nconf)
    bufs = {k: v for k, v in tr.state_dict().items() if k.endswith(("running_mean", "running_var"))} if check_buffers else {}
    worst_buf = 0.0
    for k, v in bufs.items():
        want = g[f"buffer1/{k}"]
        worst_buf = max(worst_buf, float(np.abs(_sample(v, ns) - want).max()) / max(1.0, float(np.abs(want).max())))
    print(f"  BatchNorm running statistics after step 1 ({len(bufs)} buffers): worst |diff| / max(1, |ref|) = {worst_buf:.2e}")
    assert worst_buf <= 1e-5          # measured 1.5e-7
    if second_step is not None:
        step_no[0] = 2
        out = tr.step(imgs, teacher, sizes)
        loss2, norm2 = float(out["loss"]), float(out["grad_norm"])
        print(f"[stage-1 {model} step 2] loss {loss2:.6f} (reference {ref['losses'][1]:.6f}, allowed {second_step[0]:g} rel)  "
              f"grad norm {norm2:.3f} ({ref['grad_norms'][1]:.3f}, allowed {second_step[1]:g} rel)")
        assert abs(loss2 - ref["losses"][1]) <= second_step[0] * abs(ref["losses"][1])
        assert abs(norm2 - ref["grad_norms"][1]) <= second_step[1] * ref["grad_norms"][1]
    return man


def test_b2_training_step_matches_the_reference_run():
    """EfficientViT-B2 (EV-L: widths 24 .. 384, LiteMLA heads of dim 32, 1 + 3 + 4 + 4 + 6 blocks) through the same trainer: the first
    iteration of the REAL reference stack (oracle/gen_golden_stage1_step.py --model b2 -> tests/golden/stage1/step_b2.*) -- loss,
    total gradient norm, every parameter's clipped gradient (samples) and every parameter after the update.  (A randomly initialised
    B2 at 1008^2 has a gradient norm of 2.3e5 that its own bf16-autocast run moves to 1.1e6: the second step of that run is not a
    fixture worth holding anything to, see the B1 test's note on step 2.)

    History (round 5): with the one-workgroup LiteMLA backward -- every element of S and dS one fp32 chain over all 1024 - 3969 tokens --
    this test FAILED (loss 601.9367 vs 602.0063, gradient norm 346 695 vs 232 970) and tools/trunk_train_layer_diff.py showed the forward
    drifting from a torch stand-in run by 2 - 3 x per stage-4 block (profiles/r05/layer_diff_b2.txt): B2 at random initialisation
    amplifies summation error.  The token-split kernels (256-token partial sums combined in a fixed order, a shorter and more accurate
    summation) brought it inside the limits; the margins are printed."""
    sd = schema.synthetic_state_dict("efficientvit", "b2", seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    _first_step_vs_reference("b2", "b2", sd, B2_GRAD_REL, B2_GRAD_ABS)


def test_b0_training_step_matches_the_reference_run():
    """EfficientViT-B0 (EV-S: widths 8 .. 128, 1 + 2 + 2 + 2 + 2 blocks, LiteMLA heads of dim 16) through the same trainer, against the REAL
    reference stack's run (oracle/gen_golden_stage1_step.py --model b0 -> tests/golden/stage1/step_b0.*): round 5 benchmarked this student
    (675 img/s) but only held its step to the fixture through the CPU stand-in (tests/test_stage1_trainer_host.py); first iteration as for
    B1 / B2 -- loss, total gradient norm, every parameter's clipped gradient, every updated parameter, every BatchNorm running statistic."""
    sd = schema.synthetic_state_dict("efficientvit", "b0", seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    _first_step_vs_reference("b0", "b0", sd, GRAD_REL, GRAD_ABS, check_buffers=True)


def _repvit_sd(name):
    sd = schema.synthetic_state_dict("repvit", name, seed=0)
    return {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}


@pytest.mark.parametrize("model,name", [("repvit_m0_9", "m0.9"), ("repvit_m1_1", "m1.1"), ("repvit_m2_3", "m2.3")])
def test_repvit_training_steps_match_the_reference_run(model, name):
    """The RepViT students (RV-S / RV-M: stage1/model.py:386-395 -> RepViTAdapter over sam3/backbones/repvit.py) through the same trainer --
    patch embedding with its dense stride-2 3x3, RepVGGDW token mixers, SqueezeExcite in every other block, stride-2 blocks, GELU channel
    mixers, every BatchNorm in training mode -- against the REAL reference stack's run (oracle/gen_golden_stage1_step.py --model repvit_m0_9 |
    repvit_m1_1): first iteration as for B2 (loss, norm, every clipped gradient, every updated parameter, every running statistic), and the
    loss and gradient norm of the second iteration (these nets are well conditioned: the reference's own bf16-autocast run moves its
    loss by 4e-4 and its norm by 3e-4)."""
    # measured (first run): loss 2e-7 / 1e-7 rel, norm 3.4e-4 / 3.9e-4 rel in step 1; 1.3e-7 and 3.0e-4 in step 2
    _first_step_vs_reference(model, model, _repvit_sd(name), RV_GRAD_REL, RV_GRAD_ABS, second_step=(5e-6, 2e-3), check_buffers=True, loss_rel=5e-6, norm_rel=2e-3)


@pytest.mark.parametrize("model,name", [("repvit_m0_9", "m0.9"), ("repvit_m1_1", "m1.1")])
def test_repvit_bf16_training_step_inside_the_reference_autocast_yardstick(model, name):
    """the same two iterations with bf16 activations: loss and gradient norm against the reference's fp32 run, allowed 1.5 x the distance of
    the reference's own bf16-autocast run + 1 % (loss) / 5 % (norm) of the value -- the rule of the EfficientViT-B1 test"""
    from efficientsam3_amd.stage1_train import Stage1Trainer
    with open(os.path.join(GOLD, f"step_{model}_manifest.json")) as f:
        man = json.load(f)
    hy = man["hyper"]
    tr = Stage1Trainer(_repvit_sd(name), model, embed_size=man["embed_size"], dtype="bf16", lr=hy["lr"], weight_decay=hy["weight_decay"],
                       betas=tuple(hy["betas"]), eps=hy["eps"], clip_grad=hy["clip_grad"], amp=False, cosine_weight=hy["cosine"])
    imgs, teacher = _inputs(man)
    imgs, teacher = imgs.cuda(), teacher.cuda().to(torch.bfloat16)
    for step in range(2):
        out = tr.step(imgs, teacher, [tuple(s_) for s_ in man["sizes_before_pad"]])
        loss, norm = float(out["loss"]), float(out["grad_norm"])
        r32, r16 = man["fp32"], man["bf16_autocast"]
        lim_l = 1.5 * abs(r16["losses"][step] - r32["losses"][step]) + 1e-2 * abs(r32["losses"][step])
        lim_n = 1.5 * abs(r16["grad_norms"][step] - r32["grad_norms"][step]) + 5e-2 * r32["grad_norms"][step]
        print(f"[stage-1 {model} bf16 step {step + 1}] loss {loss:.5f} (fp32 ref {r32['losses'][step]:.5f}, ref bf16 {r16['losses'][step]:.5f}, "
              f"allowed +-{lim_l:.4f}) grad norm {norm:.4f} ({r32['grad_norms'][step]:.4f} / {r16['grad_norms'][step]:.4f}, +-{lim_n:.4f})")
        assert np.isfinite(loss) and np.isfinite(norm)
        assert abs(loss - r32["losses"][step]) <= lim_l and abs(norm - r32["grad_norms"][step]) <= lim_n


def _tinyvit_sd(name):
    sd = schema.synthetic_state_dict("tinyvit", name, seed=0)
    return {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}


@pytest.mark.parametrize("model,name", [("tiny_vit_5m", "5m"), ("tiny_vit_11m", "11m"), ("tiny_vit_21m", "21m")])
def test_tinyvit_training_steps_match_the_reference_run(model, name):
    """The TinyViT students (TV-S / TV-M: stage1/model.py:397-406 -> TinyViTAdapter over sam3/backbones/tiny_vit.py) through the same trainer --
    MBConv stage, PatchMerging, window attention over zero-padded windows (63 -> 70 for the 14-token windows, 32 -> 35 for the last stage)
    with its learned bias table, LayerNorm, MLP, and for 11m stochastic depth with the per-sample factors of the reference's own run --
    against the REAL reference stack's run (oracle/gen_golden_stage1_step.py --model tiny_vit_5m | tiny_vit_11m): first iteration (loss,
    norm, every clipped gradient, every updated parameter, every running statistic), loss and gradient norm of the second."""
    # measured (first run): loss equal to all printed digits in step 1, 1e-7 rel in step 2; norm + 2.8e-4 rel in both (see the note below)
    _first_step_vs_reference(model, model, _tinyvit_sd(name), TV_GRAD_REL, TV_GRAD_ABS, second_step=(5e-6, 2e-3), check_buffers=True, loss_rel=5e-6, norm_rel=2e-3)


@pytest.mark.parametrize("model,name", [("tiny_vit_5m", "5m"), ("tiny_vit_11m", "11m")])
def test_tinyvit_bf16_training_step_inside_the_reference_autocast_yardstick(model, name):
    """the same two iterations with bf16 activations: loss and gradient norm against the reference's fp32 run, allowed 1.5 x the distance of
    the reference's own bf16-autocast run + 1 % (loss) / 5 % (norm) of the value -- the rule of the EfficientViT-B1 test"""
    from efficientsam3_amd.stage1_train import Stage1Trainer
    with open(os.path.join(GOLD, f"step_{model}_manifest.json")) as f:
        man = json.load(f)
    g = np.load(os.path.join(GOLD, f"step_{model}.npz"))
    hy = man["hyper"]
    step_no = [1]
    tr = Stage1Trainer(_tinyvit_sd(name), model, embed_size=man["embed_size"], dtype="bf16", lr=hy["lr"], weight_decay=hy["weight_decay"],
                       betas=tuple(hy["betas"]), eps=hy["eps"], clip_grad=hy["clip_grad"], amp=False, cosine_weight=hy["cosine"],
                       drop_path_sampler=_fixture_drop_path(g, step_no))
    imgs, teacher = _inputs(man)
    imgs, teacher = imgs.cuda(), teacher.cuda().to(torch.bfloat16)
    for step in range(2):
        step_no[0] = step + 1
        out = tr.step(imgs, teacher, [tuple(s_) for s_ in man["sizes_before_pad"]])
        loss, norm = float(out["loss"]), float(out["grad_norm"])
        r32, r16 = man["fp32"], man["bf16_autocast"]
        lim_l = 1.5 * abs(r16["losses"][step] - r32["losses"][step]) + 1e-2 * abs(r32["losses"][step])
        lim_n = 1.5 * abs(r16["grad_norms"][step] - r32["grad_norms"][step]) + 5e-2 * r32["grad_norms"][step]
        print(f"[stage-1 {model} bf16 step {step + 1}] loss {loss:.5f} (fp32 ref {r32['losses'][step]:.5f}, ref bf16 {r16['losses'][step]:.5f}, "
              f"allowed +-{lim_l:.4f}) grad norm {norm:.4f} ({r32['grad_norms'][step]:.4f} / {r16['grad_norms'][step]:.4f}, +-{lim_n:.4f})")
        assert np.isfinite(loss) and np.isfinite(norm)
        assert abs(loss - r32["losses"][step]) <= lim_l and abs(norm - r32["grad_norms"][step]) <= lim_n
