"""CPU restatement of the per-frame image operators (TEST INFRASTRUCTURE: only tests/ may import this).

PARITY UNPINNED for the skimage pieces: scikit-image 0.19.3 (env.yaml:135) is not installed here and the reference
holds no golden vectors for them; this file restates what skimage 0.19 does -- rgb2gray = rgb @ [0.2125, 0.7154,
0.0721]; filters.sobel_h / sobel_v = scipy.ndimage.convolve with the separable kernel [1,0,-1] (x) [1,2,1]/4,
mode='reflect' -- with the very scipy call skimage makes.  The numpy / scipy parts (clip, interp1d, argpartition,
the projection arithmetic) are the reference's own lines: src/Tracker.py:235-250, src/common.py:116-159,
src/Mapper.py:197-229.
"""
import numpy as np
from scipy import ndimage as ndi
from scipy.interpolate import interp1d


def rgb2gray(image):
    return np.asarray(image, dtype=np.float64) @ np.array([0.2125, 0.7154, 0.0721])


def sobel_axis(gray, axis):
    edge = np.array([1.0, 0.0, -1.0])
    smooth = np.array([1.0, 2.0, 1.0]) / 4.0
    kernel = edge.reshape(-1, 1) * smooth.reshape(1, -1) if axis == 0 else smooth.reshape(-1, 1) * edge.reshape(1, -1)
    return ndi.convolve(gray, kernel, mode="reflect")


def grad_magnitude(image):
    g = rgb2gray(image)
    gy, gx = sobel_axis(g, 0), sobel_axis(g, 1)          # sobel_h, sobel_v
    return np.sqrt(gx ** 2 + gy ** 2)


def dynamic_radius_maps(image, cfg):
    """src/Tracker.py:235-250."""
    pc = cfg["pointcloud"]
    thr, rmax, rmin, ratio = pc["color_grad_threshold"], pc["radius_add_max"], pc["radius_add_min"], \
        pc["radius_query_ratio"]
    gm = grad_magnitude(image)
    c = np.clip(gm, 0.0, thr)
    r_add = interp1d([0, 0.01, thr], [rmax, rmax, rmin])(c)
    r_query = interp1d([0, 0.01, thr], [ratio * rmax, ratio * rmax, ratio * rmin])(c)
    return r_add, r_query, gm


def selected_index_with_grad(H0, H1, W0, W1, n, grad_mag, ratio=15, gt_depth=None, depth_limit=False):
    """src/common.py:139-157 on a given gradient-magnitude image; returns the SET as a sorted array."""
    k = min(ratio * n, grad_mag.size)
    sel = np.argpartition(grad_mag, -k, axis=None)[-k:]
    ih, iw = np.unravel_index(sel, grad_mag.shape)
    mask = (ih >= H0) & (ih < H1) & (iw >= W0) & (iw < W1)
    if gt_depth is not None:
        d = gt_depth[ih, iw]
        mask &= ((d <= 5.0) & (d > 0.0)) if depth_limit else (d > 0.0)
    return np.sort(np.ravel_multi_index((ih[mask], iw[mask]), grad_mag.shape))


def keyframe_overlap(rays_o, rays_d, gt_depth, c2w_list, H, W, fx, fy, cx, cy, n_samples=8, edge=20):
    """src/Mapper.py:190-229 (numpy part in the reference's dtypes: float32 points and poses, float64 intrinsics)."""
    d = np.asarray(gt_depth, dtype=np.float32).reshape(-1, 1).repeat(n_samples, 1)
    t = np.linspace(0.0, 1.0, n_samples, dtype=np.float32)
    z = (d * np.float32(0.8)) * (np.float32(1.0) - t) + (d + np.float32(0.5)) * t
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    vertices = pts.reshape(-1, 3).astype(np.float32)
    out = []
    for c2w in c2w_list:
        c2w = np.asarray(c2w, dtype=np.float32)
        w2c = np.linalg.inv(c2w)
        homo = np.concatenate([vertices, np.ones_like(vertices[:, :1])], axis=1).reshape(-1, 4, 1)
        cam = (w2c @ homo)[:, :3]
        K = np.array([[fx, .0, cx], [.0, fy, cy], [.0, .0, 1.0]]).reshape(3, 3)
        cam[:, 0] *= -1
        uv = K @ cam
        zz = uv[:, -1:] + 1e-5
        uv = (uv[:, :2] / zz).astype(np.float32)
        m = (uv[:, 0] < W - edge) * (uv[:, 0] > edge) * (uv[:, 1] < H - edge) * (uv[:, 1] > edge)
        m = m & (zz[:, :, 0] < 0)
        out.append(m.reshape(-1).sum() / uv.shape[0])
    return np.array(out)


def keyframe_selection(percent_inside, k, rng=np.random):
    """src/Mapper.py:229-235: keyframes sorted by percent_inside (descending, stable), those with any overlap, a random
    permutation of them (np.random -- the caller seeds it), the first k."""
    order = sorted(range(len(percent_inside)), key=lambda i: percent_inside[i], reverse=True)
    sel = [i for i in order if percent_inside[i] > 0.0]
    return [int(i) for i in rng.permutation(np.array(sel))[:k]]
