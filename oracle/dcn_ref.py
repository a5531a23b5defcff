"""ORACLE (test infrastructure only): CPU restatement of the reference's deformable convolution forward.

Follows ``visualDet3D/networks/lib/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu``: bilinear sampling :84-115 / :467-497
(corner used iff inside the image), im2col :190-243 (v1) / :570-633 (v2: value * mask), sample taken iff
``-1 < h_im < H and -1 < w_im < W``; offset channel layout ``[y0, x0, y1, x1, ...]`` per deformable group; host side
``src/cuda/deform_conv_cuda.cpp:531-569``: output = W . columns per group (+ bias).
Vectorised torch fp32 on the host.  Pinned against the reference's own im2col device code compiled for the host
(``oracle/build_ref.sh`` -> ``oracle/_ref/libdcn_ref.so``) through ``tests/golden/dcn_cases.npz``
(``oracle/make_golden_native.py``)."""
import ctypes
import os

import numpy as np
import torch


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def im2col(x, offset, mask, kernel, stride, padding, dilation, deformable_groups):
    """x [B,C,H,W], offset [B,dg*2*K,Ho,Wo], mask [B,dg*K,Ho,Wo] or None -> columns [B, C, K, Ho, Wo] (fp32)."""
    kh, kw = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    B, C, H, W = x.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    K = kh * kw
    cpd = C // deformable_groups
    cols = torch.zeros(B, C, K, Ho, Wo, dtype=torch.float32)
    hs = (torch.arange(Ho) * sh - ph).view(1, Ho, 1).float()
    ws = (torch.arange(Wo) * sw - pw).view(1, 1, Wo).float()
    xf = x.float().reshape(B, C, H * W)
    for g in range(deformable_groups):
        xg = xf[:, g * cpd:(g + 1) * cpd]                      # [B, cpd, H*W]
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                oh = offset[:, g * 2 * K + 2 * t].float()      # [B, Ho, Wo]
                ow = offset[:, g * 2 * K + 2 * t + 1].float()
                h_im = (hs + float(i * dh)) + oh
                w_im = (ws + float(j * dw)) + ow
                ok = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
                h_low = torch.floor(h_im)
                w_low = torch.floor(w_im)
                lh, lw = h_im - h_low, w_im - w_low
                hh, hw = 1 - lh, 1 - lw
                h_low, w_low = h_low.long(), w_low.long()
                h_high, w_high = h_low + 1, w_low + 1

                def corner(hi, wi, cond):
                    cond = cond & ok
                    idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, cpd, Ho * Wo)
                    v = torch.gather(xg, 2, idx).view(B, cpd, Ho, Wo)
                    return v * cond.view(B, 1, Ho, Wo).float()

                v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
                v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
                v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
                v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
                w1, w2, w3, w4 = (hh * hw).unsqueeze(1), (hh * lw).unsqueeze(1), (lh * hw).unsqueeze(1), (lh * lw).unsqueeze(1)
                val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
                if mask is not None:
                    val = val * mask[:, g * K + t].float().unsqueeze(1)
                cols[:, g * cpd:(g + 1) * cpd, t] = val
    return cols


def deform_conv_forward(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                        rnd=None):
    """modulated (mask given) or v1 (mask None) deformable conv forward, NCHW fp32.  ``rnd``: optional rounding hook
    applied to the sampled columns and the weights (the bf16 path rounds both before the MFMA)."""
    kh, kw = weight.shape[2:]
    cols = im2col(x, offset, mask, (kh, kw), stride, padding, dilation, deformable_groups)
    w = weight.float()
    if rnd is not None:
        cols, w = rnd(cols), rnd(w)
    B, C, K, Ho, Wo = cols.shape
    O = w.shape[0]
    Cg, Og = C // groups, O // groups
    out = torch.zeros(B, O, Ho, Wo)
    for g in range(groups):
        cg = cols[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * K, Ho * Wo)
        wg = w[g * Og:(g + 1) * Og].reshape(Og, Cg * K)
        out[:, g * Og:(g + 1) * Og] = torch.matmul(wg, cg).view(B, Og, Ho, Wo)
    if bias is not None:
        out = out + bias.float().view(1, -1, 1, 1)
    return out


# ---- the reference's own im2col device code compiled for the host ------------------------------------------------
_REF_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libdcn_ref.so')


def native_im2col(x, offset, mask, kernel, stride, padding, dilation, deformable_groups):
    """columns [B, C, K, Ho, Wo] from oracle/_ref/libdcn_ref.so (per image, like the reference's host loop)."""
    lib = ctypes.CDLL(_REF_SO)
    kh, kw = _pair(kernel); sh, sw = _pair(stride); ph, pw = _pair(padding); dh, dw = _pair(dilation)
    B, C, H, W = x.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = np.zeros((B, C * kh * kw, Ho * Wo), dtype=np.float32)
    vp = ctypes.c_void_p
    for b in range(B):
        xb = np.ascontiguousarray(x[b].numpy(), dtype=np.float32)
        ob = np.ascontiguousarray(offset[b].numpy(), dtype=np.float32)
        col = np.zeros((C * kh * kw, Ho * Wo), dtype=np.float32)
        if mask is not None:
            mb = np.ascontiguousarray(mask[b].numpy(), dtype=np.float32)
            lib.ref_dcn_v2_im2col(xb.ctypes.data_as(vp), ob.ctypes.data_as(vp), mb.ctypes.data_as(vp), 1, C, H, W, kh, kw,
                                  ph, pw, sh, sw, dh, dw, deformable_groups, Ho, Wo, col.ctypes.data_as(vp))
        else:
            lib.ref_dcn_v1_im2col(xb.ctypes.data_as(vp), ob.ctypes.data_as(vp), 1, C, H, W, kh, kw,
                                  ph, pw, sh, sw, dh, dw, deformable_groups, Ho, Wo, col.ctypes.data_as(vp))
        out[b] = col
    return torch.from_numpy(out).view(B, C, kh * kw, Ho, Wo)
