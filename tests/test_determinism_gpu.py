"""GPU: race check -- the same batch through the Stereo3D / Yolo3D (DCN) / KM3D paths repeatedly, eager and as a hipGraph replay
(with the side-stream fork/joins captured), must give bit-identical raw maps and detections every time (tools/stress.py)."""
import subprocess
import sys
import os

import pytest

pytestmark = pytest.mark.gpu


def test_repeated_forwards_are_bit_identical():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'stress.py'), '6'], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'TOTAL mismatches 0' in r.stdout
