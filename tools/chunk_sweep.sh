#!/bin/bash
# bench.py at several spectral chunk counts (10 000 columns): value, ms/step, roofline fraction
for c in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --chunks $c 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks $c', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
done
