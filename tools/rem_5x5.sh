#!/bin/bash
# `reveal rem` on five related genomes of 5 Mbp: wall clock of the command, its stages (REVEAL_AMD_TIMES=1)
d=${1:-/tmp/rem5x5}; mkdir -p $d
python - <<PY
import sys; sys.path.insert(0, ".")
from reveal_amd import synth
for k, s in enumerate(synth.genomes(5000000, 5, seed=42)):
    open("$d/g%d.fa" % k, "w").write(">genome%d\n%s\n" % (k, s.decode()))
PY
for i in 1 2 3; do
  t0=$(date +%s%N)
  REVEAL_AMD_TIMES=1 python -m reveal_amd.rem $d/g0.fa $d/g1.fa $d/g2.fa $d/g3.fa $d/g4.fa -o $d/out.gfa 2>&1 | grep -v amdgpu.ids
  echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
done
