#!/bin/bash
# A/B of environment settings on the headline bench: tools/ab_env.sh "GSX_X=0" "GSX_X=1" ...  (one bench.py run per argument)
for setting in "$@"; do echo "== $setting"; env $setting timeout 300 python bench.py --steps 300 --warmup 60 --no-cpu-baseline --no-pmc --no-robustness ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'one lane', d.get('value_one_frame_in_flight'), 'unspeculated', d.get('value_unspeculated'), 'synchronised', d.get('value_synchronised'))
for sch in (('speculated','unspeculated') if 'passes' in d else ()):
    print(sch, {k:(v['us_per_frame'] if isinstance(v,dict) else v) for k,v in d['passes'][sch].items()})
"; done
