#!/bin/bash
# development aid (on the GPU box): alternate two builds of the library under tools/quick_bench.py
#   tools/ab.sh <libA.so> <libB.so> [quick_bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    printf "%-28s " "$(basename $L)"; CS_LIB_PATH=$PWD/$L timeout 200 python tools/quick_bench.py "$@" 2>&1 | tail -1 | sed 's/.*: //'
  done
done
