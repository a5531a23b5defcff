"""A/B of BASELINE config 3 alone (bench.py's own `time_other_config`), for same-box comparisons of two builds of the library:
    python tools/ab_c3.py                                   # the product library
    VD3D_TUNING_LIB=libvd3d_hip_<name>.so python tools/ab_c3.py   # an A/B build next to it"""
import json
import sys
import torch
sys.path.insert(0, '.')
import bench

c = [c for c in bench.OTHER_CONFIGS if c['key'] == (sys.argv[1] if len(sys.argv) > 1 else 'C3')][0]
r = bench.time_other_config(c, torch.device('cuda', 0), 20, 5)
print(json.dumps({k: r.get(k) for k in ('config', 'ms_per_step', 'value', 'spread', 'whole_path_frac', 'error')}))
