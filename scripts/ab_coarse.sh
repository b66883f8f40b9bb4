#!/bin/bash
# Run ON THE GPU BOX: coarse-phase time at the headline point under the coarse select's ablation knobs
# (VI_SELECT_XMODE_COARSE bits — wrong results: 1 no exact evaluation, 16 no row listing either, 32 probe order = probe rank),
# then its sampled stage clocks and row counts (VI_FILTER_STATS=2).
for x in 0 1 16 32; do
  VI_SELECT_XMODE_COARSE=$x timeout -k 10 120 python scripts/profile_headline.py --steps 10 --warmup 3 2>&1 | grep pipeline_ms |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('coarse xmode $x', d['pipeline_ms']['coarse'])"
done
timeout -k 10 200 python scripts/stats_probe.py 2 2>&1 | grep -E "STATS|ticks"
