#!/bin/bash
# A/B timing of several builds of the library in ONE call (boxes differ by 5-10 %): tools/ab_libs.sh "cmd" lib1.so lib2.so ...
cmd="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    LBM_MI355X_LIB=$PWD/$lib bash -c "$cmd"
  done
done
