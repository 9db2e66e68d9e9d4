"""Design prototype (CPU, numpy + scipy): hole filling by TILED connected-component labelling, the planned replacement of the grid-wide
union-find of csrc/kernels_decoder.hip (cc_init / cc_merge / cc_count / cc_apply; reference: sam3/model/utils/sam1_utils.py:77-119 +
perflib/connected_components.py).  Only components of area <= max_area matter, and the one huge background component is what makes the
grid-wide version slow (every walk and every count converges on its root).  Plan:

  1. a workgroup labels ONE tile (strip of TH rows) in LDS: local union-find, local areas;
  2. components that touch the strip's first / last row are "open": their (tile, local id) pairs are unioned ACROSS the seam with the
     8-connected neighbours in the adjacent strip -- a union-find over a few hundred seam components per mask, not over pixels;
  3. area of an open component = sum of the local areas of its merged parts (saturating at max_area + 1); closed components keep theirs;
  4. a background pixel is filled when the area of its (merged) component is <= max_area.

This file checks the PLAN, sequentially, against scipy's 8-connected labelling on the mask kinds of tests/test_ops_gpu.py::test_fill_holes;
it is not product code.      python tools/cc_tiled_prototype.py
"""
import numpy as np
from scipy import ndimage

ST = np.ones((3, 3), dtype=np.int32)


def fill_holes_reference(m, thr, max_area):
    out = m.copy()
    for i in range(m.shape[0]):
        lab, n = ndimage.label(m[i] <= thr, structure=ST)
        if n:
            areas = np.bincount(lab.ravel(), minlength=n + 1)
            out[i][(lab > 0) & (areas[lab] <= max_area)] = thr + 10.0
    return out


def fill_holes_tiled(m, thr, max_area, th=16):
    out = m.copy()
    for i in range(m.shape[0]):
        bg = m[i] <= thr
        H, W = bg.shape
        strips = [(y0, min(H, y0 + th)) for y0 in range(0, H, th)]
        labs, areas, base = [], [], [0]
        for y0, y1 in strips:                                   # step 1: per-strip labelling (a workgroup each)
            lab, n = ndimage.label(bg[y0:y1], structure=ST)
            labs.append(lab)
            areas.append(np.bincount(lab.ravel(), minlength=n + 1))
            base.append(base[-1] + n + 1)
        parent = np.arange(base[-1])                            # step 2: union-find over strip-local components

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for s in range(len(strips) - 1):
            up, dn = labs[s][-1], labs[s + 1][0]                # last row of strip s, first row of strip s + 1
            for x in range(W):
                if not up[x]:
                    continue
                for xx in (x - 1, x, x + 1):
                    if 0 <= xx < W and dn[xx]:
                        a, b = find(base[s] + up[x]), find(base[s + 1] + dn[xx])
                        if a != b:
                            parent[max(a, b)] = min(a, b)
        total = np.zeros(base[-1], dtype=np.int64)              # step 3: merged areas, saturating
        for s in range(len(strips)):
            for c in range(1, len(areas[s])):
                r = find(base[s] + c)
                total[r] = min(total[r] + areas[s][c], max_area + 1)
        for s, (y0, y1) in enumerate(strips):                   # step 4
            lab = labs[s]
            roots = np.array([find(base[s] + c) if c else 0 for c in range(len(areas[s]))])
            small = (lab > 0) & (total[roots[lab]] <= max_area)
            out[i, y0:y1][small] = thr + 10.0
    return out


def masks(kind, rng, n=3, H=288, W=288):
    if kind == "noise":
        return rng.normal(0.3, 1.0, (n, H, W)).astype(np.float32)
    if kind == "blobs":
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.ones((n, H, W), np.float32)
        for i in range(n):
            for _ in range(60):
                cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, 14)
                m[i][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = -1.0
        return m
    if kind == "all_bg":
        return -np.ones((n, H, W), np.float32)
    if kind == "all_fg":
        return np.ones((n, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    return np.where(((yy // 3) + (xx // 5)) % 2 == 0, 1.0, -1.0).astype(np.float32)[None].repeat(n, 0)


if __name__ == "__main__":
    rng = np.random.default_rng(7)
    for kind in ("noise", "blobs", "all_bg", "all_fg", "checker"):
        for th in (16, 32, 48):
            m = masks(kind, rng)
            assert np.array_equal(fill_holes_tiled(m, 0.0, 256.0, th), fill_holes_reference(m, 0.0, 256.0)), (kind, th)
        print(kind, "ok")
