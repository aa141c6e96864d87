#!/bin/bash
# First GPU pass of round 2: the test-suite, then bench.py in the variants that the design
# decisions of this round hang on (tree layout, builder, exchange chain).
set -u
O=gpurun_out/${1:-r02a}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="timeout 400 python bench.py"
$B > $O/bench_default.json 2> $O/bench_default.err
$B --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
$B --no-secondary --no-cpu-baseline --binary-traversal > $O/bench_c3_binary.json 2> $O/bench_c3_binary.err
$B --no-secondary --no-cpu-baseline --bvh sah_host > $O/bench_c3_sah_host.json 2> $O/bench_c3_sah_host.err
$B --no-secondary --no-cpu-baseline --bvh lbvh_device > $O/bench_c3_lbvh.json 2> $O/bench_c3_lbvh.err
$B --config 2 --no-secondary --no-cpu-baseline > $O/bench_c2_wide.json 2> $O/bench_c2_wide.err
$B --config 2 --no-secondary --no-cpu-baseline --binary-traversal > $O/bench_c2_binary.json 2> $O/bench_c2_binary.err
$B --config target --no-secondary --no-cpu-baseline > $O/bench_target.json 2> $O/bench_target.err
$B --force-distributed --no-secondary --no-cpu-baseline > $O/bench_c3_exchange_1rank.json 2> $O/bench_c3_exchange_1rank.err
$B --force-distributed --exchange rgb8 --config 2 --no-secondary --no-cpu-baseline > $O/bench_c2_exchange_rgb8_1rank.json 2> $O/bench_c2_exchange_rgb8_1rank.err
tail -3 $O/pytest.log
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json,sys
try:
    d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('traversal'), d.get('secondary',{}).get('value'))
except Exception as e: print('FAILED', e)
")"; done
