"""torchvision.ops stand-in: nms is served by the oracle's restatement (oracle/nms_ref.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle.nms_ref import nms_torch as nms  # noqa: E402,F401
