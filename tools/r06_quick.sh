#!/bin/bash
# quick A/B on one box: bench value for a list of "ENVVAR=value" settings (K=48, no extras)
cd ${GRAFT_REPO_ROOT:-.}
for setting in "$@"; do
  env $setting python bench.py --steps 48 --warmup 6 --repeats 5 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$setting', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
done
