"""CPU restatement of the end-of-run image metrics (TEST INFRASTRUCTURE: only tests/ may import this).

Reference call sites: src/Mapper.py:861-879 -- PSNR over the pixels with sensor depth, MS-SSIM through
`pytorch_msssim.ms_ssim(gt.transpose(0,2)[None].float(), img.transpose(0,2)[None].float(), data_range=1.0,
size_average=True)`, mean depth L1 over the pixels with sensor depth.

PARITY UNPINNED for MS-SSIM: pytorch-msssim 0.2.1 (env.yaml:132) is not installed and not vendored; this restates
its published algorithm (Wang et al. 2003 as implemented by pytorch_msssim 0.2.x): 11-tap Gaussian (sigma 1.5)
separable 'valid' filtering, K = (0.01, 0.03), 5 scales with weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333),
2x2 average pooling with padding = size % 2 (zeros counted), relu on cs / ssim, product of powers, mean over
channels.
"""
import torch
import torch.nn.functional as F

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gauss_1d(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_filter(x, win1d):
    C = x.shape[1]
    w = win1d.reshape(1, 1, -1).repeat(C, 1, 1)
    out = F.conv2d(x, w.unsqueeze(-1), groups=C)        # along dim 2
    out = F.conv2d(out, w.unsqueeze(-2), groups=C)      # along dim 3
    return out


def ssim_and_cs(X, Y, win1d, data_range=1.0, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = gaussian_filter(X, win1d), gaussian_filter(Y, win1d)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = gaussian_filter(X * X, win1d) - mu1_sq
    s2 = gaussian_filter(Y * Y, win1d) - mu2_sq
    s12 = gaussian_filter(X * Y, win1d) - mu12
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(X, Y, data_range=1.0):
    """X, Y: [1, C, A, B] float32."""
    win = gauss_1d()
    levels = len(MS_WEIGHTS)
    assert min(X.shape[-2:]) > (11 - 1) * 2 ** 4
    mcs = []
    for i in range(levels):
        ssim_c, cs = ssim_and_cs(X, Y, win, data_range)
        if i < levels - 1:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    ssim_c = torch.relu(ssim_c)
    stack = torch.stack(mcs + [ssim_c], dim=0)
    w = torch.tensor(MS_WEIGHTS).view(-1, 1, 1)
    return torch.prod(stack ** w, dim=0).mean()


def image_metrics(gt_color, gt_depth, color, depth):
    """(psnr, ms_ssim, depth_l1) as src/Mapper.py:861-879; images [H,W,3] / [H,W]."""
    m = gt_depth > 0
    mse = torch.nn.functional.mse_loss(gt_color[m].double(), color[m].double())
    psnr = -10.0 * torch.log10(mse)
    ms = ms_ssim(gt_color.transpose(0, 2).unsqueeze(0).float(), color.transpose(0, 2).unsqueeze(0).float())
    l1 = torch.abs(gt_depth[m].double() - depth[m].double()).mean()
    return float(psnr), float(ms), float(l1)
