#!/usr/bin/env python
"""Joins a `rocprofv3 --pmc ... --kernel-trace` run of tools/infer_trace.py: per launch of the LAST forward, duration,
core clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs))."""
import csv, glob, os, sys, collections
root = sys.argv[1]
trace = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)[0]
cc = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)[0]
ks = {}
for r in csv.DictReader(open(trace)):
    ks[r['Dispatch_Id']] = (int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'])
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[r['Dispatch_Id']][r['Counter_Name']] = cnt[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
order = sorted(ks, key=lambda d: ks[d][0])
starts = [i for i, d in enumerate(order) if ks[d][2].startswith('nchw_to_nhwc_kernel')]
ends = [i for i, d in enumerate(order) if ks[d][2].startswith('nhwc_to_nchw_kernel')]
for d in order[starts[-1]:ends[-1] + 1]:
    s, e, n = ks[d]
    c = cnt.get(d, {})
    us = (e - s) / 1e3
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    ghz = cyc / (us * 1e3) if us > 0 else 0.0
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024) if cyc else 0.0
    print('%7.1f us  %5.2f GHz  mfma %5.1f %%  busy_cycles %9.0f  %s' % (us, ghz, 100 * mf, c.get('SQ_BUSY_CYCLES', 0.0), n.replace('void ', '')[:60]))
