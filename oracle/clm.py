"""Oracle restatement of /root/reference/models/CLM.py (TEST INFRASTRUCTURE).

Vectorised plain-PyTorch form of the reference's Python loops, same parameters / names:
  * DeformableAlignment.forward (CLM.py:11-33): the double loop adds sim[:, i, j] * x over all (i, j), i.e. x times the
    column sums of the similarity matrix;
  * deform_conv (CLM.py:35-60): 9-tap modulated bilinear sampling, taps whose centre falls outside [0,H-1]x[0,W-1] skipped.
Pinned against the genuine module (loaded by path in the build container): tests/golden/clm.npz.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DeformableAlignment(nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.offset_conv = nn.Conv2d(input_dim * 2, 2 * 3 * 3, kernel_size=3, padding=1)
        self.modulation_conv = nn.Conv2d(input_dim * 2, 3 * 3, kernel_size=3, padding=1)

    def forward(self, x, similarity_map):
        B, C, H, W = x.shape
        colsum = similarity_map.sum(dim=1).view(B, 1, H, W)
        concat = torch.cat([x, colsum * x], dim=1)
        offset = self.offset_conv(concat).view(B, 9, 2, H, W)
        modulation = torch.sigmoid(self.modulation_conv(concat)).view(B, 9, 1, H, W)
        hh = torch.arange(H, dtype=x.dtype).view(1, 1, H, 1)
        ww = torch.arange(W, dtype=x.dtype).view(1, 1, 1, W)
        off_h, off_w = hh + offset[:, :, 0], ww + offset[:, :, 1]          # [B,9,H,W]
        valid = (off_h >= 0) & (off_h <= H - 1) & (off_w >= 0) & (off_w <= W - 1)
        h0 = off_h.clamp(0, H - 1).long()
        w0 = off_w.clamp(0, W - 1).long()
        h1, w1 = (h0 + 1).clamp(max=H - 1), (w0 + 1).clamp(max=W - 1)
        lh, lw = off_h - h0.to(x.dtype), off_w - w0.to(x.dtype)
        flat = x.reshape(B, C, H * W)

        def gather(hi, wi):
            idx = (hi * W + wi).reshape(B, 1, -1).expand(B, C, -1)
            return torch.gather(flat, 2, idx).reshape(B, C, 9, H, W)

        val = ((1 - lh) * (1 - lw)).unsqueeze(1) * gather(h0, w0) + (lh * (1 - lw)).unsqueeze(1) * gather(h1, w0) + \
              ((1 - lh) * lw).unsqueeze(1) * gather(h0, w1) + (lh * lw).unsqueeze(1) * gather(h1, w1)
        val = val * (valid.to(x.dtype) * modulation[:, :, 0]).unsqueeze(1)
        return val.sum(dim=2)


class CLM(nn.Module):
    def __init__(self, input_dim, temperature=0.5):
        super().__init__()
        self.temperature = temperature
        self.feature_transform = nn.Sequential(nn.Conv2d(input_dim, input_dim, 1), nn.ReLU(inplace=True), nn.Conv2d(input_dim, input_dim, 1))
        self.alignment = DeformableAlignment(input_dim)
        self.attention_conv = nn.Conv2d(input_dim, 1, 1)
        self.fusion_conv = nn.Sequential(nn.Conv2d(input_dim, input_dim, 3, padding=1), nn.ReLU(inplace=True),
                                         nn.Conv2d(input_dim, input_dim, 3, padding=1))

    def forward(self, y, y_refs):
        B, C, H, W = y.shape
        y_t = self.feature_transform(y)
        aligned, att = [], []
        for y_ref in y_refs:
            y_ref_t = self.feature_transform(y_ref)
            sim = torch.bmm(y_t.view(B, C, -1).transpose(1, 2), y_ref_t.view(B, C, -1)) / self.temperature
            a = self.alignment(y_ref, F.softmax(sim, dim=-1))
            aligned.append(a)
            att.append(self.attention_conv(a))
        wts = F.softmax(torch.stack(att, dim=1), dim=1)
        return self.fusion_conv((torch.stack(aligned, dim=1) * wts).sum(dim=1) + y)


class SimpleCLM(nn.Module):
    def __init__(self, input_dim, temperature=0.5):
        super().__init__()
        self.temperature = temperature
        self.feature_transform = nn.Conv2d(input_dim, input_dim, 1)
        self.attention_conv = nn.Conv2d(input_dim, 1, 1)
        self.fusion_conv = nn.Sequential(nn.Conv2d(input_dim, input_dim, 3, padding=1), nn.ReLU(inplace=True))

    def forward(self, y, y_refs):
        feats, att = [], []
        for y_ref in y_refs:
            r = self.feature_transform(y_ref)
            a = self.attention_conv(r)
            att.append(a)
            feats.append(r * torch.sigmoid(a))
        wts = F.softmax(torch.stack(att, dim=1), dim=1)
        return self.fusion_conv((torch.stack(feats, dim=1) * wts).sum(dim=1) + y)
