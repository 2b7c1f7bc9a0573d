"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> JSON (profiles/rNN_traffic.json).

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide coalesced reads, i.e. HALF the bytes - the fetch figure is doubled here;
WRITE_SIZE is uncalibrated and left as reported.
"""
import csv, glob, json, os, sys, collections
root, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # training steps each PMC pass ran (bench.py --steps 1 --warmup 1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in agg.items():
    if not any(t in k for t in ('conv_', 'bn_', 'wino_', 'wino2_', 'reduce_kernel', 'first_block')):
        continue
    fetch = d.get('FETCH_SIZE', [])
    write = d.get('WRITE_SIZE', [])
    res[k.replace('void ', '')[:80]] = {
        'launches': max(len(fetch), len(write)),
        'fetch_bytes_per_launch_corrected': 2.0 * 1024.0 * sum(fetch) / max(len(fetch), 1),
        'write_bytes_per_launch_reported': 1024.0 * sum(write) / max(len(write), 1),
    }
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from singleshotpose_amd._lib import csrc_digest
res['_meta'] = {'csrc_sha1': csrc_digest(), 'note': 'digest of singleshotpose_amd/csrc + include/ssp_hip.h at profile time',
                'steps_profiled': steps}
# per-step totals: all kernels, and the Winograd transform / finishing family
tot = sum(v['launches'] * (v['fetch_bytes_per_launch_corrected'] + v['write_bytes_per_launch_reported']) for k, v in res.items() if k != '_meta')
wino = sum(v['launches'] * (v['fetch_bytes_per_launch_corrected'] + v['write_bytes_per_launch_reported']) for k, v in res.items()
           if k != '_meta' and k.startswith('wino_'))
res['_meta']['total_bytes_per_step'] = tot / steps
res['_meta']['onchip_winograd_bytes_per_step'] = sum(
    v['launches'] * (v['fetch_bytes_per_launch_corrected'] + v['write_bytes_per_launch_reported']) for k, v in res.items()
    if k != '_meta' and k.startswith('wino2_')) / steps
res['_meta']['wino_transform_bytes_per_step'] = wino / steps
json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in list(res.items())[:40]}, indent=1)[:3000])
