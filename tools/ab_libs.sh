#!/bin/bash
# Process-level A/B of two builds of the library on one box: alternates `python tools/ab_encoder.py X=0` under KIRAG_AMD_LIB=<lib> (bash tools/ab_libs.sh libA libB [rounds]).
A=$1; B=$2; N=${3:-2}
for r in $(seq 1 $N); do
  for lib in $A $B; do
    echo "== round $r: $lib"
    KIRAG_AMD_LIB=$lib python tools/ab_encoder.py KIRAG_AMD_UNUSED=0 2>&1 | grep -v amdgpu.ids | sed 's/KIRAG_AMD_UNUSED=0: //; s/  outputs identical: True//'
  done
done
