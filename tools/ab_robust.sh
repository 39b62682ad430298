#!/bin/bash
# A/B of environment settings on the robustness legs (random pose order, open sky, translucent, surfaces): tools/ab_robust.sh "GSX_X=0" "GSX_X=1" ...
for setting in "$@"; do echo "== $setting"; env $setting timeout 600 python bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-pmc --no-extra-legs --no-cfg5 ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'one lane', d.get('value_one_frame_in_flight'), 'unspeculated', d.get('value_unspeculated'))
print({k:[v['speculated']['fps'], v['unspeculated']['fps'], v['unspeculated']['tile_entries']] for k,v in d['robustness'].items() if isinstance(v,dict)})
"; done
