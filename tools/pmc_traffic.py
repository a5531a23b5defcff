"""HBM/fabric traffic of the conv launches from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs of the
same bench command): writes profiles/<round>_pmc_traffic.json (read by bench.py for roofline.traffic).
    python tools/pmc_traffic.py <fetch-dir-or-db> <write-dir-or-db> <out.json>"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualdet3d_amd import build as _build  # noqa: E402

# the HBM-bound stages north_star names: FETCH / WRITE per launch next to the launch duration of the same pass
HBM_KERNELS = ('stem_pool_kernel', 'psm_cosine', 'cost_volume_fused_kernel', 'dwconv3x3_kernel', 'head_select_kernel', 'head_nms_kernel',
               'dcn_nhwc_kernel', 'dwconvT_phase_kernel', 'image_conv7_kernel', 'conv_pair_kernel', 'look_ground_kernel')
CONV = ('conv_igemm', 'conv_halo', 'conv_resident', 'conv_regw', 'conv_ksplit', 'conv_small', 'conv_pw', 'splitk_reduce')      # (the reduction launch of a split-K convolution is part of that convolution)


def per_kernel(path, counter):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[-1]
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    ip = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    kcols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
    scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
    name_col = 'display_name' if 'display_name' in scol else 'kernel_name'
    evcol = 'event_id' if 'event_id' in kcols else 'id'
    q = ('select s.%s, count(distinct d.%s), sum(e.value) from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.%s '
         'join %s s on d.kernel_id = s.id where i.name = ? group by s.%s' % (name_col, evcol, pe, ip, kd, evcol, ks, name_col))
    vals = {r[0]: (r[1], r[2]) for r in cur.execute(q, (counter,))}
    # mean launch duration per kernel name in the same pass (the counter pass serialises dispatches: durations are additive)
    cols = set(kcols)
    dur = {}
    if {'start', 'end'} <= cols:
        for name, n, ns in cur.execute('select s.%s, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s'
                                       % (name_col, kd, ks, name_col)):
            dur[name] = ns / max(n, 1) / 1e3
    per_kernel.durations_us = dur
    return vals


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
dur_us = dict(per_kernel.durations_us)
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
forwards = max(n for k, (n, v) in fetch.items() if 'head_nms_kernel' in k)
fk = sum(v for k, (n, v) in fetch.items() if any(c in k for c in CONV))
wk = sum(v for k, (n, v) in write.items() if any(c in k for c in CONV))
rows = {k[:110]: {'dispatches': n, 'fetch_kb': v, 'write_kb': write.get(k, (0, 0.0))[1]} for k, (n, v) in fetch.items() if any(c in k for c in CONV)}
hbm_rows = {}
for k, (n, v) in fetch.items():
    if any(c in k for c in HBM_KERNELS):
        w = write.get(k, (0, 0.0))[1]
        nb = (2.0 * v + w) * 1024.0 / max(n, 1)
        us = dur_us.get(k)
        hbm_rows[k[:110]] = {'dispatches': n, 'fetch_kb_per_launch': v / max(n, 1), 'write_kb_per_launch': w / max(n, 1),
                             'hbm_bytes_per_launch': nb, 'avg_launch_us_under_the_counter_pass': us,
                             'counter_gbps': (nb / (us * 1e-6) / 1e9) if us else None}
out = {
    'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 4 --warmup 1 --no-graph --no-cpu-baseline`',
    # which kernels were measured: hash of the convolution family's sources (bench.py marks the record stale when it moves)
    'conv_sources': list(_build.CONV_SOURCES), 'conv_sources_sha16': _build.source_hash(_build.CONV_SOURCES), 'library_sources_sha16': _build.source_hash(),
    'forwards': forwards, 'conv_fetch_kb_sum': fk, 'conv_write_kb_sum': wk, 'fetch_correction': 2.0,
    'note': 'gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section); '
            'WRITE_SIZE is 1:1. Counters are in KB (x1024).',
    'conv_hbm_bytes_per_forward': (fk * 2.0 + wk) * 1024.0 / forwards,
    'per_kernel': rows,
    # HBM-bound kernels of the step: counter bytes per launch (FETCH x2 + WRITE) and the GB/s they amount to over the launch duration
    'hbm_kernels': hbm_rows,
}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print('forwards', forwards, 'conv bytes/forward %.3f GB' % (out['conv_hbm_bytes_per_forward'] / 1e9))
