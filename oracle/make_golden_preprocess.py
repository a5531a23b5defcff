"""Generates tests/golden/preprocess_cases.npz with the REFERENCE's own augmentation classes (ConvertToFloat, CropTop, Resize,
Normalize of data/pipeline/stereo_augmentator.py, imported from /root/reference through oracle/ref_shim.py) on seeded uint8
frames.  cv2 is not installed: the reference's ``cv2.resize`` call is served by oracle/preprocess_ref.resize_linear (the
restatement of OpenCV's INTER_LINEAR), so the fixture pins everything around the resize -- crop, crop/pad to the network width,
normalisation order, calibration update -- but NOT the interpolation itself (parity unpinned, see oracle/preprocess_ref.py).
Run here: python -m oracle.make_golden_preprocess"""
import os

import numpy as np

from oracle import preprocess_ref, ref_shim

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
MEAN, STD = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
CASES = [dict(Hs=40, Ws=120, crop_top=8, size=(36, 128)),       # wider than the target after resize -> cropped
         dict(Hs=40, Ws=100, crop_top=10, size=(32, 128)),      # narrower -> zero padded (before Normalize)
         dict(Hs=36, Ws=128, crop_top=4, size=(32, 128))]       # scale 1: resize is the identity


def main():
    ref_shim.load()
    import visualDet3D.data.pipeline.stereo_augmentator as sa
    sa.cv2.resize = lambda img, wh, *a, **k: preprocess_ref.resize_linear(img, int(wh[0]), int(wh[1]))
    rng = np.random.default_rng(9)
    out = {}
    for i, c in enumerate(CASES):
        left = rng.integers(0, 256, (c['Hs'], c['Ws'], 3), dtype=np.uint8)
        right = rng.integers(0, 256, (c['Hs'], c['Ws'], 3), dtype=np.uint8)
        P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
        P3 = P2.copy()
        P3[0, 3] = -339.5242
        args = (left, right, P2.copy(), P3.copy(), None, None, None)
        for aug in (sa.ConvertToFloat(), sa.CropTop(crop_top_index=c['crop_top']), sa.Resize(size=c['size']), sa.Normalize(mean=MEAN, stds=STD)):
            args = aug(*args)
        l, r, p2, p3 = args[:4]
        out['c%d_left_u8' % i], out['c%d_right_u8' % i] = left, right
        out['c%d_left' % i], out['c%d_right' % i] = l.transpose(2, 0, 1).astype(np.float32), r.transpose(2, 0, 1).astype(np.float32)
        out['c%d_P2_in' % i], out['c%d_P2' % i], out['c%d_P3' % i] = P2, p2, p3
        out['c%d_cfg' % i] = np.array([c['Hs'], c['Ws'], c['crop_top'], c['size'][0], c['size'][1]])
        print('case', i, l.shape, float(l.mean()))
    np.savez_compressed(os.path.join(GOLDEN_DIR, 'preprocess_cases.npz'), **out)


if __name__ == '__main__':
    main()
