"""CPU: the HOST LOGIC of efficientsam3_amd/stage1_train.py (``Stage1Trainer``: arena views handed to the layers, head and trunk sequencing,
gradient sink and arrival order, state-dict names, BatchNorm buffers, the update) with a RepViT or TinyViT student, every kernel wrapper replaced by a
plain torch / numpy stand-in of the same contract (tests/test_train_blocks_host.py, tests/test_train_repvit_host.py, the oracle's loss and
update), against the REAL reference stack's run of the same two iterations (tests/golden/stage1/step_repvit_m0_9.*, step_tiny_vit_5m.*,
step_tiny_vit_11m.*, made by oracle/gen_golden_stage1_step.py --model ...).  The GPU twins, with the HIP kernels in place of the stand-ins, are
tests/test_stage1_step.py::test_repvit_training_steps_match_the_reference_run and ::test_tinyvit_training_steps_match_the_reference_run."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from efficientsam3_amd import schema, stage1, stage1_train, synth
from oracle import ref_stage1
from tests.test_train_blocks_host import _to_nchw, _to_nhwc, cpu_kernels  # noqa: F401
from tests.test_train_repvit_host import repvit_kernels  # noqa: F401
from tests.test_train_tinyvit_host import tinyvit_kernels  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage1")
PREFIX = "backbone.vision_backbone.trunk.model."


class HostUpdater(stage1.Stage1Updater):
    """the arena of Stage1Updater on the CPU; ``step`` = the oracle's update (oracle/ref_stage1.py: update_step) on views of it"""

    def __init__(self, layout, device, lr=5e-4, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, clip_grad=5.0, amp=False, **_):
        self.layout, self.device = layout, torch.device("cpu")
        self.lr, self.weight_decay, self.betas, self.eps, self.clip_grad, self.amp = float(lr), float(weight_decay), betas, float(eps), clip_grad, False
        z = lambda: torch.zeros(layout.n, dtype=torch.float32)  # noqa: E731
        self.params, self.grads, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
        self.state = torch.zeros(16, dtype=torch.float32)
        self.state[0] = 1.0
        self._st = {"scale": 1.0, "tracker": 0, "step": 0}

    def step(self, lr=None, zero_grads=True):
        names = [n for n, _ in self.layout.named_shapes]
        as_np = lambda buf: {n: self.view(buf, n).numpy() for n in names}  # noqa: E731  (views: updated in place)
        decay = stage1.weight_decay_groups(self.layout.named_shapes)
        norm, _ = ref_stage1.update_step(as_np(self.params), as_np(self.grads), as_np(self.exp_avg), as_np(self.exp_avg_sq), self._st, decay, {},
                                         self.lr if lr is None else lr, self.weight_decay, self.betas, self.eps, self.clip_grad, amp=False)
        if zero_grads:
            self.grads.zero_()
        self.state[3], self.state[4] = float(norm), self._st["step"]
        return self.state[3].clone()


def install_host_trainer(monkeypatch):  # noqa: F811
    def loss(p2, t2, valid):          # [B, HW, C] rows -> the reference's NCHW functions on [B, C, HW, 1]
        p, t, m = p2.permute(0, 2, 1)[..., None].float(), t2.permute(0, 2, 1)[..., None].float(), valid[:, None, :, None].float()
        return ref_stage1.masked_mse(p, t, m), ref_stage1.masked_cosine_loss(p, t, m), None

    def loss_backward(p2, t2, valid, cosine_weight=0.0, grad_scale=1.0):
        pr = p2.clone().float().requires_grad_(True)
        mse, cos, _ = loss(pr, t2, valid)
        ((mse + cosine_weight * cos) * grad_scale).backward()
        return pr.grad

    def conv3x3_wgrad(dy, x):
        wr = torch.zeros((dy.shape[-1], x.shape[-1], 3, 3), requires_grad=True)
        F.conv2d(_to_nchw(x), wr, None, padding=1).backward(_to_nchw(dy))
        return wr.grad

    def resize_backward(dy, in_hw):
        xr = torch.zeros((dy.shape[0], dy.shape[-1]) + tuple(in_hw), requires_grad=True)
        F.interpolate(xr, size=tuple(dy.shape[1:3]), mode="bilinear", align_corners=False).backward(_to_nchw(dy))
        return _to_nhwc(xr.grad)

    st = stage1_train
    monkeypatch.setattr(st, "Stage1Updater", HostUpdater)
    monkeypatch.setattr(st, "distill_loss", loss)
    monkeypatch.setattr(st, "distill_loss_backward", loss_backward)
    monkeypatch.setattr(st, "conv3x3_forward", lambda x, w, bias: _to_nhwc(F.conv2d(_to_nchw(x), w, bias, padding=1)))
    monkeypatch.setattr(st, "conv3x3_dgrad", lambda dy, w: _to_nhwc(F.conv_transpose2d(_to_nchw(dy), w, None, padding=1)))
    monkeypatch.setattr(st, "conv3x3_wgrad", conv3x3_wgrad)
    monkeypatch.setattr(st, "resize_forward", lambda x, size: _to_nhwc(F.interpolate(_to_nchw(x), size=(size, size), mode="bilinear", align_corners=False)))
    monkeypatch.setattr(st, "resize_backward", resize_backward)


@pytest.fixture
def host_trainer(tinyvit_kernels, monkeypatch):  # noqa: F811
    install_host_trainer(monkeypatch)


def _sample(t, n):
    flat = t.detach().float().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step][:n].numpy()


@pytest.mark.parametrize("model,family,name", [("repvit_m0_9", "repvit", "m0.9"), ("tiny_vit_5m", "tinyvit", "5m"), ("tiny_vit_11m", "tinyvit", "11m"),
                                               ("tiny_vit_21m", "tinyvit", "21m"), ("b0", "efficientvit", "b0")])
def test_trainer_host_logic_vs_the_reference_run(host_trainer, model, family, name):
    """RepViT-M0.9; TinyViT-5M (no stochastic depth); TinyViT-11M and 21M with the DropPath factors of the reference's run (part of the fixture)
    fed through ``drop_path_sampler``; EfficientViT-B0 (the GPU tests hold B1 and B2 to their fixtures) -- with the GPU fixtures that is eight
    of the nine students of stage1/model.py:386-420 against the reference's own run (RepViT-M2.3, 58 blocks, is checked against the reference
    MODULE in tests/test_train_repvit_host.py)"""
    with open(os.path.join(GOLD, f"step_{model}_manifest.json")) as f:
        man = json.load(f)
    g = np.load(os.path.join(GOLD, f"step_{model}.npz"))
    hy, ref, ns = man["hyper"], man["fp32"], man["samples_per_tensor"]
    sd = schema.synthetic_state_dict(family, name, seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    step_no = [1]
    used = []

    def sampler(mod, call, batch, keep):
        key = f"droppath{step_no[0]}/backbone.model.{mod}/{call}"
        used.append(key)
        f = g[key]
        assert f.shape == (batch,) and set(np.round(f * keep, 5).tolist()) <= {0.0, 1.0}, (key, f, keep)
        return torch.from_numpy(f)

    tr = stage1_train.Stage1Trainer(sd, model, embed_size=man["embed_size"], dtype="f32", device="cpu", lr=hy["lr"], weight_decay=hy["weight_decay"],
                                    betas=tuple(hy["betas"]), eps=hy["eps"], clip_grad=hy["clip_grad"], amp=False, cosine_weight=hy["cosine"],
                                    accumulation_steps=hy["accumulation_steps"], drop_path_sampler=sampler)
    imgs = torch.stack([torch.from_numpy(synth.normalise_to_chw_f32(synth.smooth_image_u8(seed=s))) for s in man["image_seeds"]])
    for i, (h, w) in enumerate(man["sizes_before_pad"]):
        imgs[i, :, h:, :] = 0
        imgs[i, :, :, w:] = 0
    gen = torch.Generator().manual_seed(man["teacher_seed"])
    teacher = (torch.randn((len(man["sizes_before_pad"]), man["embed_dim"], man["embed_size"], man["embed_size"]), generator=gen) * 0.5).permute(0, 2, 3, 1).contiguous()
    sizes = [tuple(s) for s in man["sizes_before_pad"]]

    out = tr.step(imgs, teacher, sizes, update_grad=False)
    assert sorted(used) == sorted(k for k in g.files if k.startswith("droppath1/"))          # every factor the reference drew was asked for
    grads = {n: _sample(v, ns) for n, v in tr.gradients().items()}
    norm = float(tr.updater.step())
    tr._micro = 0
    loss = float(out["loss"])
    assert abs(loss - ref["losses"][0]) <= 1e-5 * abs(ref["losses"][0]), (loss, ref["losses"][0])
    print(f"[host {model}] loss {loss:.6f} ({ref['losses'][0]:.6f}) grad norm {norm:.4f} ({ref['grad_norms'][0]:.4f})")
    assert abs(norm - ref["grad_norms"][0]) <= 1e-3 * ref["grad_norms"][0], (norm, ref["grad_norms"][0])
    clip = min(1.0, hy["clip_grad"] / (norm + 1e-6))
    gmax = max(float(g[f"gradmax1/{n}"]) for n in ref["names"])
    assert sorted(grads) == sorted(ref["names"])
    # EfficientViT at random initialisation amplifies rounding differences between two fp32 implementations (its GPU test allows 2.5e-2 of a
    # tensor's maximum for the same reason: tests/test_stage1_step.py); the other two families are well conditioned
    rel, absl = (2.5e-2, 2.5e-5) if family == "efficientvit" else (5e-3, 1e-5)
    worst = 0.0
    for n in ref["names"]:
        err = float(np.abs(grads[n] * clip - g[f"grad1/{n}"]).max())
        worst = max(worst, err / (rel * float(g[f"gradmax1/{n}"]) + absl * gmax))
        assert err <= rel * float(g[f"gradmax1/{n}"]) + absl * gmax, (n, err, float(g[f"gradmax1/{n}"]))
    print(f"[host {model}] largest fraction of the gradient allowance ({rel:g} x tensor max + {absl:g} x network max) used: {worst:.3f}")
    state = tr.state_dict()
    for n in ref["names"]:
        assert tuple(state[n].shape) == tuple(ref["shapes"][n]), n
        assert float(np.abs(_sample(state[n], ns) - g[f"param1/{n}"]).max()) <= 2.5 * hy["lr"] + 1e-6, n
    for k in [k for k in state if k.endswith(("running_mean", "running_var"))]:
        want = g[f"buffer1/{k}"]
        assert float(np.abs(_sample(state[k], ns) - want).max()) <= 1e-4 * max(1.0, float(np.abs(want).max())), k
    assert all(int(v) == int(sd[k]) + 1 for k, v in state.items() if k.endswith("num_batches_tracked"))
    # the arrival order of the gradients (= the bucket order of the all-reduce): head first, then the trunk from its last block to the stem
    first = {"repvit": "backbone.model.features.0.0.bn.bias", "tinyvit": "backbone.model.patch_embed.seq.0.bn.bias",
             "efficientvit": "backbone.model.input_stem.op_list.0.norm.bias"}[family]
    assert tr._arrival[0].startswith("head.") and tr._arrival[-1] == first

    step_no[0] = 2
    out = tr.step(imgs, teacher, sizes)
    loss2, norm2 = float(out["loss"]), float(out["grad_norm"])
    assert abs(loss2 - ref["losses"][1]) <= 1e-3 * abs(ref["losses"][1]), (loss2, ref["losses"][1])
    assert abs(norm2 - ref["grad_norms"][1]) <= 1e-2 * ref["grad_norms"][1], (norm2, ref["grad_norms"][1])


def test_drop_path_factors_one_draw_per_step_same_sequence_and_resumable(host_trainer):
    """TinyViT-11M's default stochastic depth: the factors of a step are drawn in ONE piece at the start of forward() -- the
    same generator sequence as per-residual draws in forward order -- so a step costs one upload instead of one synchronous copy per block;
    ``rng_state`` / ``set_rng_state`` let a resumed run continue the mask sequence; `seed` is the BASE seed (the rank is added: the
    reference seeds config.SEED + rank, train_image_encoder_stage1.py:340)."""
    from efficientsam3_amd import train_tinyvit as tv
    sd = schema.synthetic_state_dict("tinyvit", "11m", seed=0)
    sd = {k[len(PREFIX):]: v.clone() for k, v in sd.items() if k.startswith(PREFIX)}
    tr = stage1_train.Stage1Trainer(sd, "tiny_vit_11m", embed_size=8, dtype="f32", device="cpu", seed=5)
    trunk = tr.trunk
    gen = torch.Generator().manual_seed(5)          # one rank, no process group: base seed + 0
    expect = {}
    for blk in trunk.blocks:
        res = getattr(blk, "res", None)
        if res is None or res.rate == 0.0:
            continue
        for call in range(2 if isinstance(blk, tv.TinyViTBlockTrain) else 1):
            keep = 1.0 - res.rate
            expect[(res.name, call)] = torch.empty(3, dtype=torch.float32).bernoulli_(keep, generator=gen) / keep
    assert len(expect) >= 10
    state0 = tr.rng_state()
    trunk._predraw(3, "cpu")
    assert set(trunk._pre) == set(expect)
    for k, v in expect.items():
        assert torch.equal(trunk._pre[k], v), k
    first = {k: v.clone() for k, v in trunk._pre.items()}
    trunk._predraw(3, "cpu")                                               # the next step continues the sequence ...
    assert any(not torch.equal(trunk._pre[k], first[k]) for k in first)
    tr.set_rng_state(state0)                                               # ... and a restored state replays it exactly
    trunk._predraw(3, "cpu")
    assert all(torch.equal(trunk._pre[k], first[k]) for k in first)
    # a factor asked for through the sampler interface is the pre-drawn row (consumed once), then fresh draws
    k0 = next(iter(first))
    assert torch.equal(trunk._draw(k0[0], k0[1], 3, 0.9), first[k0]) and k0 not in trunk._pre


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE + "/sam3"), reason="the reference tree is only present in the build container")
def test_trainer_gradient_accumulation_vs_the_reference_loop(host_trainer):
    """ACCUMULATION_STEPS = 2 (train_image_encoder_stage1.py:186-219): two micro-batches, each loss / 2, gradients accumulated, ONE clip + AdamW
    step + zero_grad after the second -- ``step(update_grad=False)`` then ``step()`` -- against that loop written out with the real RepViT module,
    the head of ``ImageStudentEncoder`` (stage1/model.py:193-211), the oracle's loss functions and torch.optim.AdamW in the two groups of
    ``set_weight_decay`` (stage1/optimizer.py:32-46); then a second accumulation cycle (the arena was zeroed, the BatchNorm buffers moved on)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pth in (os.path.join(root, "oracle", "shims"), REFERENCE + "/sam3"):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    from sam3.backbones import repvit as ref_repvit
    E, S, ACC, LR, WD, CLIP, COS = 1024, 8, 2, 1e-3, 0.05, 5.0, 0.5
    sd = schema.synthetic_state_dict("repvit", "m0.9", seed=2)
    sd = {k[len(PREFIX):]: v.clone().float() for k, v in sd.items() if k.startswith(PREFIX)}

    class Student(torch.nn.Module):            # ImageStudentEncoder (stage1/model.py:188-211) over RepViTAdapter (:287-296)
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Module()
            self.backbone.model = ref_repvit.repvit_m0_9(pretrained=False, num_classes=0, distillation=False)
            self.head = torch.nn.Sequential(torch.nn.Conv2d(384, E, 1, bias=False), torch.nn.BatchNorm2d(E), torch.nn.GELU(), torch.nn.Conv2d(E, E, 3, padding=1))

        def forward(self, x):
            for layer in self.backbone.model.features:
                x = layer(x)
            return F.interpolate(self.head(x), size=(S, S), mode="bilinear", align_corners=False)

    net = Student()
    net.load_state_dict(sd, strict=True)
    net.train()
    decay, no_decay = [], []
    for n, p in net.named_parameters():
        (no_decay if p.dim() == 1 or n.endswith(".bias") else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay}, {"params": no_decay, "weight_decay": 0.0}], lr=LR, weight_decay=WD, betas=(0.9, 0.999), eps=1e-8)
    tr = stage1_train.Stage1Trainer({k: v.clone() for k, v in sd.items()}, "repvit_m0_9", embed_size=S, dtype="f32", device="cpu", lr=LR, weight_decay=WD,
                                    clip_grad=CLIP, amp=False, cosine_weight=COS, accumulation_steps=ACC)
    g = torch.Generator().manual_seed(31)
    sizes = [(128, 96), (100, 96)]
    for cycle in range(2):
        norms = []
        for micro in range(ACC):
            imgs = torch.randn(2, 3, 128, 128, generator=g)
            for i, (h, w) in enumerate(sizes):                 # the dataset pads below / right of the resized image with zeros
                imgs[i, :, h:, :] = 0
                imgs[i, :, :, w:] = 0
            teacher = torch.randn(2, E, S, S, generator=g) * 0.5
            preds = net(imgs)
            valid = ref_stage1.build_valid_mask(128, sizes, (S, S))
            loss = (ref_stage1.masked_mse(preds, teacher, valid) + COS * ref_stage1.masked_cosine_loss(preds, teacher, valid)) / ACC
            loss.backward()
            out = tr.step(imgs, teacher.permute(0, 2, 3, 1).contiguous(), sizes, update_grad=(micro == ACC - 1))
            assert abs(float(out["loss"]) - float(loss)) <= 2e-5 * abs(float(loss)), (cycle, micro, float(out["loss"]), float(loss))
            norms.append(out["grad_norm"])
        ref_norm = float(torch.nn.utils.clip_grad_norm_(net.parameters(), CLIP))
        ref_grads = {n: p.grad.clone() for n, p in net.named_parameters()}
        opt.step()
        opt.zero_grad()
        assert norms[0] is None and abs(float(norms[1]) - ref_norm) <= 2e-3 * ref_norm, (cycle, norms, ref_norm)
        state = tr.state_dict()
        named = dict(net.named_parameters())
        worst = max(float((state[n] - p.detach()).abs().max()) for n, p in named.items())
        assert worst <= (2.1 if cycle == 0 else 4.2) * LR, (cycle, worst)       # AdamW's first steps move by about lr: a sign flip of a noise-level gradient is 2 lr
        # elements whose gradient is rounding noise (a shift in front of a BatchNorm has none at all) get a noise sign from AdamW in the reference
        # too: compare where the gradient is confidently non-zero, as the GPU test does
        gmax = max(float(v.abs().max()) for v in ref_grads.values())
        bad = conf = 0
        for n, p in named.items():
            mask = ref_grads[n].abs() > 1e-3 * float(ref_grads[n].abs().max()) + 1e-6 * gmax
            conf += int(mask.sum())
            bad += int((((state[n] - p.detach()).abs() > 2e-2 * LR) & mask).sum())
        assert conf > 1e6 and bad <= 5e-3 * conf, (cycle, bad, conf)
        for k, v in net.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                # cycle 2 runs on parameters that already differ by AdamW's noise-sign elements
                assert float((state[k] - v).abs().max()) <= (1e-4 if cycle == 0 else 5e-3) * max(1.0, float(v.abs().max())), (cycle, k)
