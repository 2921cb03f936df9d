"""Pins the four third-party seams of the hot path against the REAL libraries, wherever they import.

Test infrastructure (see oracle/__init__.py): nothing in point_slam_amd/ imports this.

The build image has none of faiss / cv2 / skimage / pytorch_msssim, so oracle/ restates their published algorithms
(pointslam_oracle.knn_exact, pointslam_oracle.remap_linear_cv2, frame_oracle.rgb2gray / sobel_axis, eval_oracle.ms_ssim) and
DESIGN.md lists them as "parity unpinned".  Run this script in the reference's own environment (env.yaml of the reference:
faiss-gpu 1.7.2, opencv-python, scikit-image, pytorch-msssim):

    python oracle/pin_third_party.py            # writes tests/golden/third_party_<lib>.npz for every library that imports

Each file holds the seeded INPUTS and the library's OUTPUTS; tests/test_third_party_pins.py compares the oracle restatements
(CPU) and the HIP kernels (GPU) with every file that is present and skips the ones that are not.  The call sites pinned:
  faiss      IndexIVFFlat / GpuIndexIVFFlat search, src/neural_point.py:37-41,169-197 (nprobe = nlist: exact)
  cv2        cv2.remap(depth, u, v, INTER_LINEAR), src/Mapper.py:149-155
  skimage    color.rgb2gray + filters.sobel_h / sobel_v, src/Tracker.py:235-250, src/Mapper.py:686-701, src/common.py:92-159
  msssim     pytorch_msssim.ms_ssim(data_range=1.0, size_average=True), src/Mapper.py:861-867
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def inputs(seed=1219):
    """The seeded inputs every pin shares (small enough for the repository, ragged enough to hit the border rules)."""
    g = np.random.default_rng(seed)
    cloud = g.uniform(-1.0, 1.0, size=(20000, 3)).astype(np.float32)
    queries = np.concatenate([g.uniform(-1.1, 1.1, size=(3000, 3)), cloud[:96] + 1e-4]).astype(np.float32)
    H, W = 120, 160
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    depth = (2.0 + 0.5 * np.sin(xx / 17.0) + 0.3 * np.cos(yy / 11.0)).astype(np.float32)
    depth[g.uniform(size=depth.shape) < 0.02] = 0.0                                     # sensor holes
    u = g.uniform(-3.0, W + 2.0, size=4000).astype(np.float32)                          # in, on and beyond the border
    v = g.uniform(-3.0, H + 2.0, size=4000).astype(np.float32)
    u[:64] = np.round(u[:64]); v[:64] = np.round(v[:64])                                # exact pixel centres
    u[64:128] = (np.floor(u[64:128]) + 1.0 / 64.0).astype(np.float32)                   # half a quantisation step of cv2's 1/32 grid
    color = g.uniform(0.0, 1.0, size=(H, W, 3)).astype(np.float32)
    color[20:60, 30:90] = (0.2, 0.7, 0.4)                                               # a flat region: zero gradient
    a = g.uniform(0.0, 1.0, size=(1, 3, 176, 192)).astype(np.float32)                   # MS-SSIM needs > 160 px per side
    b = np.clip(a + g.normal(0.0, 0.05, size=a.shape), 0.0, 1.0).astype(np.float32)
    return dict(cloud=cloud, queries=queries, depth=depth, u=u, v=v, color=color, img_a=a, img_b=b)


def pin_faiss(x):
    import faiss
    nlist = 64
    quant = faiss.IndexFlatL2(3)
    index = faiss.IndexIVFFlat(quant, 3, nlist, faiss.METRIC_L2)
    index.train(x["cloud"]); index.add(x["cloud"])
    index.nprobe = nlist                                         # every list probed: the exact answer, in FAISS's tie order
    D, I = index.search(x["queries"], 8)
    return dict(cloud=x["cloud"], queries=x["queries"], D=D.astype(np.float32), I=I.astype(np.int64), version=faiss.__version__)


def pin_cv2(x):
    import cv2
    out = cv2.remap(x["depth"], x["u"].reshape(-1, 1), x["v"].reshape(-1, 1), interpolation=cv2.INTER_LINEAR)[:, 0]
    return dict(depth=x["depth"], u=x["u"], v=x["v"], out=out.astype(np.float32), version=cv2.__version__)


def pin_skimage(x):
    import skimage
    from skimage import color, filters
    gray = color.rgb2gray(x["color"].astype(np.float64))
    return dict(color=x["color"], gray=gray, sobel_h=filters.sobel_h(gray), sobel_v=filters.sobel_v(gray), version=skimage.__version__)


def pin_msssim(x):
    import torch
    import pytorch_msssim
    v = pytorch_msssim.ms_ssim(torch.from_numpy(x["img_a"]), torch.from_numpy(x["img_b"]), data_range=1.0, size_average=True)
    return dict(img_a=x["img_a"], img_b=x["img_b"], ms_ssim=np.float64(v.item()), version=getattr(pytorch_msssim, "__version__", "?"))


PINS = dict(faiss=pin_faiss, cv2=pin_cv2, skimage=pin_skimage, msssim=pin_msssim)


def main():
    x = inputs()
    os.makedirs(OUT, exist_ok=True)
    for name, fn in PINS.items():
        try:
            rec = fn(x)
        except ImportError as e:
            print(f"{name}: not importable here ({e}); no fixture written")
            continue
        path = os.path.join(OUT, f"third_party_{name}.npz")
        np.savez_compressed(path, **rec)
        print(f"{name}: {path} ({rec['version']})")


if __name__ == "__main__":
    sys.exit(main())
