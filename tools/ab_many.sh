#!/bin/bash
# A/B several builds on one box, interleaved: tools/ab_many.sh "<kbench args>" this tools/ab/a.so tools/ab/b.so ...
args=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $rep)"
    if [ "$lib" = this ]; then python tools/kbench.py $args; else DEFF_AMD_LIB=$PWD/$lib python tools/kbench.py $args; fi
  done
done
