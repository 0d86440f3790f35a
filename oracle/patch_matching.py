"""Oracle restatement of the numeric functions of /root/reference/models/Patch_Matching.py (TEST INFRASTRUCTURE).

CPU, plain PyTorch / numpy; same names and argument meaning as the reference functions
(L2_or_pearson_corr :854-910, create_gaussian_masks :779-807, SI_Wraper :218-240, SI_Finder_at_Image_Domain :87-122,
rgb_transform :926-934, reduce_mean_and_std_normalize_images :913-924) without their .cuda()/sleep/empty_cache calls.
Pinned against the genuine functions (AST-extracted in the build container): tests/golden/patch_matching.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

_MEANS = torch.tensor([93.70454143384742, 98.28243432206516, 94.84678088809876]).float().view(1, 3, 1, 1)
_VARS = torch.tensor([73.56493292844912, 75.88547006820752, 76.74838442810665]).float().view(1, 3, 1, 1)


def reduce_mean_and_std_normalize_images(x):
    return (x - _MEANS) / _VARS


def rgb_transform(x):
    R, G, B = torch.chunk(x, 3, dim=1)
    return torch.cat([R + G, R - G, 0.5 * (R + B)], dim=1)


def create_gaussian_masks(img_h, img_w, patch_h, patch_w):
    n = np.arange(0, (img_h * img_w) // (patch_h * patch_w))
    patch_img_w = img_w / patch_w
    w = np.arange(1, img_w + 1, 1, float) - (patch_w % 2) / 2
    h = (np.arange(1, img_h + 1, 1, float) - (patch_h % 2) / 2)[:, np.newaxis]
    center_h = (n // patch_img_w + 0.5) * patch_h
    center_w = ((n % patch_img_w) + 0.5) * patch_w
    cols = (w - center_w[:, np.newaxis])[:, np.newaxis, :] ** 2 / (0.5 * img_w) ** 2
    rows = np.transpose(h - center_h)[:, :, np.newaxis] ** 2 / (0.5 * img_h) ** 2
    g = np.exp(-4 * np.log(2) * (rows + cols))
    g = g[:, (patch_h + 1) // 2 - 1: img_h - patch_h // 2, (patch_w + 1) // 2 - 1: img_w - patch_w // 2]
    return torch.from_numpy(g.astype(np.float32)[np.newaxis])


def L2_or_pearson_corr(x, y, patch_h, patch_w):
    N, C, H, W = x.shape
    patch_size = int(H * W * C)
    xy = F.conv2d(y, x)
    y_mean = F.conv2d(y, torch.ones(1, C, H, W) / patch_size)
    x_sum = torch.sum(x, dim=[1, 2, 3])
    numerator = xy - y_mean * x_sum.view(1, -1, 1, 1)
    den_x = torch.sum(torch.square(x), dim=[1, 2, 3]) - torch.mean(x, dim=[1, 2, 3]) * x_sum
    den_y = F.conv2d(torch.square(y), torch.ones(1, C, H, W)) - y_mean * y_mean * patch_size
    return numerator / torch.sqrt(den_y * den_x.view(1, -1, 1, 1))


def _gather_patches(y, index, patch_h, patch_w, corr_w):
    """y [1,C,H,W]; index [...] flat positions in the corr map -> patches [..., C, ph, pw]."""
    ih, iw = torch.div(index, corr_w, rounding_mode="floor"), index % corr_w
    dy, dx = torch.meshgrid(torch.arange(patch_h), torch.arange(patch_w), indexing="ij")
    rows = ih[..., None, None] + dy
    cols = iw[..., None, None] + dx
    return y[0][:, rows, cols].movedim(0, -3)


def SI_Wraper(cross_corr, patch_h, patch_w, patchs_num, y, k=1, temperature=15, is_stack=False):
    _, _, corr_h, corr_w = cross_corr.shape
    _, C, fh, fw = y.shape
    value, index = torch.topk(cross_corr.reshape(-1, corr_h * corr_w), k, dim=1)
    weight = F.softmax(value * temperature, dim=1)
    patches = _gather_patches(y, index, patch_h, patch_w, corr_w)              # [P,k,C,ph,pw]
    if is_stack:   # Patch_Matching.py:235-236: candidate-major channel stack, no weights
        p = patches.reshape(fh // patch_h, fw // patch_w, k, C, patch_h, patch_w).permute(2, 3, 0, 4, 1, 5)
        return p.reshape(1, k * C, fh, fw)
    p = (patches * weight[:, :, None, None, None]).sum(1)                       # [P,C,ph,pw]
    return p.reshape(fh // patch_h, fw // patch_w, C, patch_h, patch_w).permute(2, 0, 3, 1, 4).reshape(1, C, fh, fw)


def SI_Finder_at_Image_Domain(x_dec, y_imgs, patch_h, patch_w, y_dec, mask=None):
    N, C, H, W = x_dec.shape
    outs = []
    for n in range(N):
        xp = x_dec[n:n + 1].reshape(1, C, H // patch_h, patch_h, W // patch_w, patch_w).permute(0, 2, 4, 1, 3, 5).reshape(-1, C, patch_h, patch_w)
        q = rgb_transform(reduce_mean_and_std_normalize_images(xp * 255))
        r = rgb_transform(reduce_mean_and_std_normalize_images(y_dec[n:n + 1] * 255))
        corr = L2_or_pearson_corr(q, r, patch_h, patch_w)
        if mask is not None:
            corr = corr * mask
        _, _, ch, cw = corr.shape
        index = torch.argmax(corr.reshape(-1, ch * cw), dim=1)
        patches = _gather_patches(y_imgs[n:n + 1], index, patch_h, patch_w, cw)  # [P,C,ph,pw]
        ih, iw = y_imgs.shape[2], y_imgs.shape[3]
        outs.append(patches.reshape(ih // patch_h, iw // patch_w, C, patch_h, patch_w).permute(2, 0, 3, 1, 4).reshape(1, C, ih, iw))
    return torch.cat(outs, 0)
