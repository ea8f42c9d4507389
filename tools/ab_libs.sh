#!/bin/bash
# A/B of library builds on one GPU box: tools/ab_libs.sh "<bench args>" lib1.so lib2.so ...  (2 rounds, interleaved)
ARGS=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    INFERA_LIB_PATH=$PWD/$lib timeout 300 python bench.py $ARGS --no-cpu-baseline 2>&1 | grep -o 'ms_per_step.: [0-9.]*' | sed "s|^|$lib |"
  done
done
