"""Summarise rocprofv3 --pmc CSVs: mean counter value per (kernel name, grid size) for conv kernels."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if not any(t in k for t in ('conv_', 'wino_', 'wino2_', 'reduce_kernel')):
            continue
        key = '%s grid=%s' % (k.replace('void ', '')[:48], r.get('Grid_Size', '?'))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k)
    wc = sum(d.get('SQ_WAVE_CYCLES', [0])) / max(len(d.get('SQ_WAVE_CYCLES', [1])), 1)
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        extra = ''
        if wc and c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA'):
            extra = '  %.1f%% of wave cycles' % (100.0 * m / wc)
        print('   %-28s %16.0f (n=%d)%s' % (c, m, len(v), extra))
    # derived (profiles/README.md): counters are summed over the 8 XCDs; MFMA pipe utilisation = busy cycles / (cycles x 1024 SIMDs)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d:
        busy = sum(d['SQ_VALU_MFMA_BUSY_CYCLES']) / len(d['SQ_VALU_MFMA_BUSY_CYCLES'])
        act = sum(d['GRBM_GUI_ACTIVE']) / len(d['GRBM_GUI_ACTIVE'])
        print('   %-28s %15.1f%%' % ('-> MFMA pipe busy', 100.0 * busy / (act / 8.0 * 1024.0)))

