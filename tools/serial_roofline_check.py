"""Cross-check of bench.py's `roofline.achieved` against the rocprofv3 kernel trace of the SAME command (`bench.py --no-overlap`,
one stream, so the per-kernel durations are additive): sum of the conv-family kernel time per forward vs flops / achieved.
    python tools/serial_roofline_check.py <serial_kernel_stats.csv> <serial_bench.json>"""
import csv
import json
import sys

CONV = ('conv_igemm', 'conv_halo', 'conv_resident', 'conv_regw', 'conv_ksplit', 'conv_small', 'conv_pw', 'splitk_reduce')      # (the reduction launch of a split-K convolution is part of that convolution)
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads(open(sys.argv[2]).read())
fwd = max(int(r['calls']) for r in rows if 'head_nms_kernel' in r['name'])
conv_ns = sum(int(r['total_ns']) for r in rows if any(c in r['name'] for c in CONV))
all_ns = sum(int(r['total_ns']) for r in rows)
n_conv = sum(int(r['calls']) for r in rows if any(c in r['name'] for c in CONV))
B = line['config']['global_batch']
flops = 473.82e9 * B                     # conv + conv3d flops of one step; the conv2d launches carry 3.79 TF of it
ach = line['roofline']['achieved']
t_bench = 3.7903e12 / (ach * 1e12)       # seconds per step the bench's HIP-event timing saw over the conv launches
print('forwards in the trace: %d; conv launches per forward: %.1f' % (fwd, n_conv / fwd))
print('rocprofv3 : sum of conv-family kernel durations per forward = %.1f us  -> %.1f TF/s = %.4f of 2.5 PF' % (
    conv_ns / fwd / 1e3, 3.7903e12 / (conv_ns / fwd * 1e-9) / 1e12, 3.7903e12 / (conv_ns / fwd * 1e-9) / 2.5e15))
print('bench.py  : roofline.achieved = %.1f TF/s (frac %.4f) = %.1f us per forward over the conv launches (HIP events)' % (ach, line['roofline']['frac'], t_bench * 1e6))
print('agreement : %.2f %%' % (100.0 * (conv_ns / fwd * 1e-9 / t_bench - 1.0)))
print('all kernels per forward (serial): %.1f us; step time of this run %.3f ms; whole-path %.4f of 2.5 PF' % (
    all_ns / fwd / 1e3, line['ms_per_step'], flops / (line['ms_per_step'] * 1e-3) / 2.5e15))
