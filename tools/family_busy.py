"""Per-kernel-family MFMA busy / wait shares from the SQ PMC table of tools/profile_round.sh (<tag>_sq_pmc.txt):
    python tools/family_busy.py gpurun_out/r02/r02_sq_pmc.txt
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x GRBM_GUI_ACTIVE) (the counter sums to 32 per fully busy chip cycle: it reproduces
flops / (duration x 2.5 PF x clock / 2.4 GHz) of every family); parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave cycles spent in
s_waitcnt / barriers); issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import sys

lines = [l.rstrip('\n') for l in open(sys.argv[1]) if ' | ' in l]
hdr = [h.strip() for h in lines[0].split('|')]
col = {h: i for i, h in enumerate(hdr)}
print('%-78s %5s %8s %9s %8s %8s %9s' % ('kernel (launches per 3 profiled forwards)', 'n', 'avg_us', 'mfma_busy', 'parked', 'stalled', 'lds_confl'))
for l in lines[1:]:
    f = [x.strip() for x in l.split('|')]
    g = lambda k: float(f[col[k]])
    busy = g('SQ_VALU_MFMA_BUSY_CYCLES') / (32.0 * g('GRBM_GUI_ACTIVE'))
    wc = g('SQ_WAVE_CYCLES')
    idx = g('SQ_LDS_IDX_ACTIVE')
    print('%-78s %5d %8.1f %8.1f%% %7.1f%% %7.1f%% %8.1f%%' % (f[0].replace('void (anonymous namespace)::', '')[:78], int(f[col['n']]), g('avg_us'), 100 * busy,
                                                    100 * g('SQ_WAIT_ANY') / wc, 100 * g('SQ_WAIT_INST_ANY') / wc,
                                                    100 * g('SQ_LDS_BANK_CONFLICT') / idx if idx else 0.0))
