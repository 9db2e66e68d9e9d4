"""The drop-in claim at the API level: the reference's OWN caller, eval/eval_coco.py::evaluate_model,
runs UNCHANGED with ``compat/`` ahead of it on sys.path.

* CPU (build container: the reference is present, no GPU): evaluate_model is imported from the reference checkout and
  executed as is -- facade imports, builder signature, checkpoint ingestion, ``model.to / eval``, Sam3Processor(PIL),
  ``predict_inst(box=...)`` shapes -- against a synthetic COCO directory, with the HIP engine replaced by the
  oracle-backed TEST DOUBLE tests/fake_engine.py (the real engine refuses to exist without a HIP device).  Its mIoU
  must equal the oracle's own evaluation of the same boxes.
* GPU: the same directory and checkpoint through the facade on the real engine (f32 and bf16); evaluate_model itself
  when a reference checkout is available (ESAM3_REFERENCE_ROOT), else the same loop restated here (the GPU box has no
  reference); mIoU vs the oracle.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ROOT = os.environ.get("ESAM3_REFERENCE_ROOT", "/root/reference")
REF_EVAL = os.path.join(REF_ROOT, "eval", "eval_coco.py")


def _rle_counts(mask):  # uncompressed COCO RLE (column-major runs, starting with zeros)
    flat = np.asarray(mask, dtype=np.uint8).T.reshape(-1)
    change = np.flatnonzero(np.diff(flat)) + 1
    runs = np.diff(np.concatenate([[0], change, [flat.size]])).tolist()
    return ([0] + runs) if flat[0] else runs


def make_coco_dir(root):
    """2 synthetic images (600x800 and 480x640) with 2 elliptical ground-truth objects each; a third image with a
    crowd annotation only (skipped by the evaluator)."""
    from PIL import Image

    from efficientsam3_amd import synth
    os.makedirs(os.path.join(root, "annotations"), exist_ok=True)
    os.makedirs(os.path.join(root, "images", "val2017"), exist_ok=True)
    images, anns, boxes = [], [], {}
    aid = 1
    for iid, (h, w, seed) in enumerate([(600, 800, 21), (480, 640, 22), (256, 256, 23)], start=1):
        img = np.ascontiguousarray(synth.smooth_image_u8(seed=seed, size=max(h, w))[:h, :w])
        name = f"{iid:012d}.png"
        Image.fromarray(img).save(os.path.join(root, "images", "val2017", name))
        images.append({"id": iid, "file_name": name, "height": h, "width": w})
        rng = np.random.default_rng(100 + iid)
        for k in range(2):
            cx, cy = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h
            rx, ry = rng.uniform(0.1, 0.25) * w, rng.uniform(0.1, 0.25) * h
            yy, xx = np.mgrid[0:h, 0:w]
            m = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0
            ys, xs = np.nonzero(m)
            bbox = [float(xs.min()), float(ys.min()), float(xs.max() - xs.min() + 1), float(ys.max() - ys.min() + 1)]
            crowd = 1 if iid == 3 else 0
            anns.append({"id": aid, "image_id": iid, "category_id": 1, "iscrowd": crowd, "bbox": bbox, "area": float(m.sum()),
                         "segmentation": {"counts": _rle_counts(m), "size": [h, w]}})
            if not crowd:
                boxes.setdefault(iid, []).append((bbox, m))
            aid += 1
    with open(os.path.join(root, "annotations", "instances_val2017.json"), "w") as f:
        json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "thing"}]}, f)
    return images, boxes


def make_checkpoint(path):
    """a converter-style checkpoint: {"model": {"detector.<key>": tensor}} (sam3/sam3/model_builder.py:584-630)"""
    from efficientsam3_amd import schema
    sd = schema.synthetic_state_dict("efficientvit", "b1", seed=0, enable_inst_interactivity=True)
    ck = {}
    for k, v in sd.items():
        if k.startswith("inst_interactive_predictor.model."):
            ck["tracker." + k[len("inst_interactive_predictor.model."):]] = v
        else:
            ck["detector." + k] = v
    torch.save({"model": ck}, path)
    return sd


def oracle_miou(sd, root, images, boxes):
    from PIL import Image

    from oracle import ref_model
    ious = []
    for info in images:
        if info["id"] not in boxes:
            continue
        img = np.asarray(Image.open(os.path.join(root, "images", "val2017", info["file_name"])).convert("RGB"))
        chw = torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, -1, 0)))
        with torch.inference_mode():
            x = ref_model.processor_transform(chw)[None]
            st = ref_model.set_image(sd, x, (info["height"], info["width"]), "b1")
            for bbox, gt in boxes[info["id"]]:
                box = np.array([bbox[0], bbox[1], bbox[0] + bbox[2], bbox[1] + bbox[3]])
                masks, _, _ = ref_model.predict_inst(sd, st, box=box[None, :], multimask_output=False)
                pm = masks[0] > 0
                ious.append(np.logical_and(pm, gt).sum() / max(np.logical_or(pm, gt).sum(), 1))
    return float(np.mean(ious)), len(ious)


def _import_reference_evaluator():
    """eval/eval_coco.py from the reference checkout, unmodified, with the facade and the pycocotools shim ahead of it"""
    for p in (os.path.join(ROOT, "oracle", "shims"), os.path.join(ROOT, "compat")):
        if p not in sys.path:
            sys.path.insert(0, p)
    for name in [n for n in sys.modules if n == "sam3" or n.startswith("sam3.")]:
        del sys.modules[name]  # a reference `sam3` imported by another test must not shadow the facade
    spec = importlib.util.spec_from_file_location("reference_eval_coco", REF_EVAL)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sam3
    assert os.path.join("compat", "sam3") in sam3.__file__, sam3.__file__
    return mod


def _restated_evaluate(model_path, backbone, model_name, coco_root, dtype):
    """eval/eval_coco.py:29-137 restated for the GPU box (no reference checkout there): same calls, same arithmetic."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "shims"))
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    from PIL import Image
    from pycocotools.coco import COCO
    from sam3 import build_efficientsam3_image_model
    from sam3.device import get_device
    from sam3.model.sam3_image_processor import Sam3Processor
    model = build_efficientsam3_image_model(bpe_path="sam3/assets/bpe_simple_vocab_16e6.txt.gz", enable_inst_interactivity=True,
                                            checkpoint_path=model_path, load_from_HF=False, backbone_type=backbone,
                                            model_name=model_name, dtype=dtype)
    model.to(get_device())
    model.eval()
    processor = Sam3Processor(model)
    coco = COCO(os.path.join(coco_root, "annotations/instances_val2017.json"))
    ious = []
    for img_id in coco.getImgIds():
        info = coco.loadImgs(img_id)[0]
        image = Image.open(os.path.join(coco_root, "images", "val2017", info["file_name"])).convert("RGB")
        state = processor.set_image(image)
        for ann in coco.loadAnns(coco.getAnnIds(imgIds=img_id)):
            if ann["iscrowd"]:
                continue
            b = ann["bbox"]
            box = np.array([b[0], b[1], b[0] + b[2], b[1] + b[3]])
            with torch.no_grad():
                masks, scores, _ = model.predict_inst(state, point_coords=None, point_labels=None, box=box[None, :],
                                                      multimask_output=False)
            assert isinstance(masks, np.ndarray) and masks.shape == (1, info["height"], info["width"]) and scores.shape == (1,)
            pm, gt = masks[0] > 0, coco.annToMask(ann)
            ious.append(np.logical_and(pm, gt).sum() / max(np.logical_or(pm, gt).sum(), 1))
    return float(np.mean(ious)), len(ious)


@pytest.mark.skipif(not os.path.exists(REF_EVAL), reason="needs the reference checkout (build container)")
def test_reference_eval_coco_runs_unchanged_through_the_facade(tmp_path, monkeypatch):
    from tests.fake_engine import OracleEngine
    coco_root = str(tmp_path / "coco")
    images, boxes = make_coco_dir(coco_root)
    ckpt = str(tmp_path / "efficient_sam3_efficientvit_m.pt")
    sd = make_checkpoint(ckpt)
    import efficientsam3_amd.sam3_image as si
    monkeypatch.setattr(si, "HipEngine", OracleEngine)  # TEST DOUBLE: no HIP device in this container
    mod = _import_reference_evaluator()
    result = mod.evaluate_model(ckpt, "efficientvit", "b1", coco_root, split="val2017", num_samples=-1, device="cuda")
    assert result is not None, "evaluate_model swallowed an exception (it prints it)"
    miou, _elapsed = result
    want, n = oracle_miou(sd, coco_root, images, boxes)
    assert n == 4
    assert abs(miou - want) <= 1e-6, (miou, want)
    # the facade also answers the other spellings callers use
    import sam3.sam3.model_builder as mb2
    from sam3.model.box_ops import box_xyxy_to_cxcywh  # noqa: F401
    from sam3.model.tokenizer_ve import SimpleTokenizer  # noqa: F401
    from sam3.model_builder import build_sam3_image_model  # noqa: F401
    assert mb2.build_efficientsam3_image_model is mod.build_efficientsam3_image_model


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_ROOT, "sam3", "sam3")), reason="needs the reference checkout")
def test_facade_forwards_off_path_modules_to_a_reference_checkout(monkeypatch):
    """``sam3.visualization_utils`` (efficientsam3_image_predictor_example.py:28) is not part of the hot path: with
    ESAM3_REFERENCE_SAM3 set it resolves to the reference's module, while the hot-path names stay the facade's."""
    monkeypatch.setenv("ESAM3_REFERENCE_SAM3", os.path.join(REF_ROOT, "sam3", "sam3"))
    for p in (os.path.join(ROOT, "oracle", "shims"), os.path.join(ROOT, "compat")):
        monkeypatch.syspath_prepend(p)
    for name in [n for n in sys.modules if n == "sam3" or n.startswith("sam3.")]:
        monkeypatch.delitem(sys.modules, name)
    import sam3
    spec = importlib.util.find_spec("sam3.visualization_utils")
    assert spec is not None and spec.origin.startswith(REF_ROOT)
    spec2 = importlib.util.find_spec("sam3.model.sam3_image_processor")
    assert os.path.join("compat", "sam3") in spec2.origin
    assert os.path.join("compat", "sam3") in sam3.__file__


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_eval_coco_miou_on_the_engine(tmp_path, dtype):
    coco_root = str(tmp_path / "coco")
    images, boxes = make_coco_dir(coco_root)
    ckpt = str(tmp_path / "efficient_sam3_efficientvit_m.pt")
    sd = make_checkpoint(ckpt)
    want, n = oracle_miou(sd, coco_root, images, boxes)
    if os.path.exists(REF_EVAL) and dtype == "bf16":  # the unmodified caller (it builds the default, bf16, engine)
        mod = _import_reference_evaluator()
        got, _ = mod.evaluate_model(ckpt, "efficientvit", "b1", coco_root, split="val2017", num_samples=-1, device="cuda")
        m = 4
    else:
        got, m = _restated_evaluate(ckpt, "efficientvit", "b1", coco_root, dtype)
    print(f"[eval_coco {dtype}] mIoU engine {got:.6f} oracle {want:.6f} over {m} boxes")
    assert m == n == 4
    assert abs(got - want) <= (1e-4 if dtype == "f32" else 2e-2), (got, want)
