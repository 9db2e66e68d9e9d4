"""Shim of pycocotools.coco (absent third-party dependency; test infrastructure only): the subset of the COCO API that
the reference's eval/eval_coco.py:53-108 calls -- COCO(annotation_file), getImgIds, loadImgs, getAnnIds, loadAnns,
annToMask -- restated from cocoapi's published behaviour (PythonAPI/pycocotools/coco.py).  annToMask handles the
uncompressed / compressed RLE and bounding-box-only annotations the tests generate (no polygons)."""
import json

import numpy as np


class COCO:
    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.imgs, self.imgToAnns = {}, {}, {}, {}
        if annotation_file is not None:
            with open(annotation_file) as f:
                self.dataset = json.load(f)
            for img in self.dataset.get("images", []):
                self.imgs[img["id"]] = img
                self.imgToAnns.setdefault(img["id"], [])
            for ann in self.dataset.get("annotations", []):
                self.anns[ann["id"]] = ann
                self.imgToAnns.setdefault(ann["image_id"], []).append(ann)

    def getImgIds(self, imgIds=(), catIds=()):
        return sorted(self.imgs) if not imgIds else sorted(set(imgIds) & set(self.imgs))

    def loadImgs(self, ids=()):
        ids = [ids] if isinstance(ids, int) else ids
        return [self.imgs[i] for i in ids]

    def getAnnIds(self, imgIds=(), catIds=(), areaRng=(), iscrowd=None):
        imgIds = [imgIds] if isinstance(imgIds, int) else list(imgIds)
        anns = [a for i in imgIds for a in self.imgToAnns.get(i, [])] if imgIds else list(self.anns.values())
        if iscrowd is not None:
            anns = [a for a in anns if a.get("iscrowd", 0) == iscrowd]
        return [a["id"] for a in anns]

    def loadAnns(self, ids=()):
        ids = [ids] if isinstance(ids, int) else ids
        return [self.anns[i] for i in ids]

    def annToMask(self, ann):
        img = self.imgs[ann["image_id"]]
        h, w = img["height"], img["width"]
        seg = ann.get("segmentation")
        if isinstance(seg, dict) and isinstance(seg.get("counts"), list):  # uncompressed RLE, column-major runs
            flat = np.zeros(h * w, dtype=np.uint8)
            pos, val = 0, 0
            for c in seg["counts"]:
                if val:
                    flat[pos:pos + c] = 1
                pos += c
                val ^= 1
            return flat.reshape((w, h)).T.copy()
        if isinstance(seg, dict):
            from pycocotools import mask as maskUtils
            return maskUtils.decode(seg)
        x, y, bw, bh = [int(round(v)) for v in ann["bbox"]]
        m = np.zeros((h, w), dtype=np.uint8)
        m[y:y + bh, x:x + bw] = 1
        return m
