"""Write a copy of a tune cache with some launch shapes' codes replaced: python tools/cache_variant.py in.json out.json KEYPREFIX=code ...
(same-box A/B of ONE plan choice with everything else pinned: AB_CFGS="-;SSP_TUNE_CACHE=out.json" tools/gpu_round.sh TAG ab)"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
for spec in sys.argv[3:]:
    pref, code = spec.rsplit('=', 1)
    hits = [k for k in d if k.startswith(pref)]
    for k in hits:
        print('%s: %s -> %s' % (k, d[k], code))
        d[k] = int(code)
    if not hits:
        print('no key starts with', pref)
json.dump(d, open(dst, 'w'))
