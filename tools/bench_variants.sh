#!/bin/bash
# usage (on the GPU box): tools/bench_variants.sh v1 v2 ...   -> first-shape ms of tools/bench_fused.py per variant
for v in "$@"; do
  printf "%s " $v; timeout -k 10 100 python tools/bench_fused.py --lib tools/variants/libirm_$v.so 2>/dev/null | grep '"ms"' | tr -d ' \n'; echo
done
