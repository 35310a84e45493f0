#!/bin/bash
# usage: ab.sh B libA.so libB.so ...   -> alternates the builds 3 times on this box, prints value per run
B=$1; shift
for r in 1 2 3; do
  for L in "$@"; do
    FSGM_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-per-gpu $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', '%.4g'%d['value'], '%.3f ms'%d['ms_per_step'])"
  done
done
