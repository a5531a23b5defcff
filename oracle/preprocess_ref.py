"""ORACLE (test infrastructure only): numpy float32 restatement of the test-time image pipeline
``data/pipeline/stereo_augmentator.py`` ConvertToFloat (:30-36) -> CropTop (:214-249) -> Resize (:62-134) -> Normalize (:39-59)
and the collate transpose (``data/kitti/dataset/stereo_dataset.py:141-147``).

PARITY UNPINNED AGAINST cv2 ITSELF for the resize step: it is ``cv2.resize`` (third party, ``requirement.txt``: opencv-python, no
version pin; not installed in this container and absent from /root/reference), restated from OpenCV's published algorithm
(modules/imgproc/src/resize.cpp: resizeGeneric_ coordinate table with ``fx = (float)((dx+0.5)*scale_x - 0.5)``, clamping of the
first / last source column, HResizeLinear then VResizeLinear in fp32).  The restatement IS pinned against a second, independent
implementation of that algorithm (torch ``F.interpolate(bilinear, align_corners=False, antialias=False)``,
tests/test_preprocess_oracle.py): identical to 1e-10 with float64 coordinates; with the fp32 coordinate tables the two differ
by the table's own rounding (<= 2^-13 of a weight at x ~ 1200 -> 0.03 on the 0..255 scale, 1.2e-4 relative).  Everything else (crop, pad, normalise, calibration update) is the
reference's own numpy code and is followed line by line."""
import numpy as np


def _coords(n_dst, n_src, dtype=np.float32):
    """``dtype``: float32 = OpenCV's table (the source coordinate is cast to float before the weight is taken, so at x ~ 1200
    the weight carries 2^-13 of rounding); float64 = the exact algorithm, used only to pin the restatement against an
    independent implementation (tests/test_preprocess_oracle.py)."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    s0 = np.zeros(n_dst, dtype=np.int64)
    w = np.zeros(n_dst, dtype=dtype)
    for d in range(n_dst):
        f = dtype((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = dtype(f - dtype(s))
        if s < 0:
            f, s = dtype(0), 0
        if s >= n_src - 1:
            f, s = dtype(0), n_src - 1
        s0[d], w[d] = s, f
    return s0, w


def resize_linear(img, w_dst, h_dst, dtype=np.float32):
    """cv2.resize(img float32 HWC, (w_dst, h_dst)) with INTER_LINEAR."""
    img = img.astype(dtype)
    hs, ws = img.shape[:2]
    sx, fx = _coords(w_dst, ws, dtype)
    sy, fy = _coords(h_dst, hs, dtype)
    sx1 = np.minimum(sx + 1, ws - 1)
    sy1 = np.minimum(sy + 1, hs - 1)
    a0, a1 = (dtype(1) - fx)[None, :, None], fx[None, :, None]
    hor = img[:, sx] * a0 + img[:, sx1] * a1                       # horizontal pass
    b0, b1 = (dtype(1) - fy)[:, None, None], fy[:, None, None]
    return (hor[sy] * b0 + hor[sy1] * b1).astype(dtype)


def preprocess(frame_u8, crop_top, size, mean, std):
    """uint8 HWC -> float32 CHW network input, as the reference pipeline + collate_fn."""
    img = frame_u8.astype(np.float32)                              # ConvertToFloat
    img = img[crop_top:]                                           # CropTop
    scale = size[0] / img.shape[0]                                 # Resize, preserve_aspect_ratio
    h = np.round(img.shape[0] * scale).astype(int)
    w = np.round(img.shape[1] * scale).astype(int)
    img = resize_linear(img, int(w), int(h))
    if img.shape[1] > size[1]:
        img = img[:, 0:size[1], :]
    elif img.shape[1] < size[1]:
        img = np.pad(img, [(0, 0), (0, size[1] - img.shape[1]), (0, 0)], 'constant')
    img = img[:size[0]]
    img = img.astype(np.float32)                                   # Normalize
    img /= 255.0
    img -= np.asarray(mean, dtype=np.float32)
    img /= np.asarray(std, dtype=np.float32)
    return img.transpose(2, 0, 1).copy()
