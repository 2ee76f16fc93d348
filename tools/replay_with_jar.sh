#!/bin/bash
# Replays tests/golden/stage1_small against a real RNA-Bloom (needs a JRE and RNA-Bloom.jar v2.0.1 — neither exists in
# the build image, which is why parity below the hash layer is "unpinned"; this is the one command that pins it).
#   tools/replay_with_jar.sh /path/to/RNA-Bloom.jar [workdir]
# Stage 1 only (-stage 1), one thread (-t 1: the reference's filter updates are not atomic, so only -t 1 is reproducible),
# -savebf keeps the graph files (R/RNABloom.java:7181-7185), which are then compared byte for byte with the fixture.
set -e
JAR=${1:?usage: replay_with_jar.sh RNA-Bloom.jar [workdir]}
HERE=$(cd "$(dirname "$0")/.." && pwd)
FIX=$HERE/tests/golden/stage1_small
OUT=${2:-$(mktemp -d)}/O
mkdir -p "$OUT"
java -jar "$JAR" -left "$FIX/L.fq" -right "$FIX/R.fq" -revcomp-right -k 25 -t 1 -fpr 0.01 -nk 4000 -stage 1 -savebf -outdir "$OUT"
rc=0
for ext in "" .dbgbf .dbgbf.desc .cbf .cbf.desc .rpkbf .rpkbf.desc; do
  if cmp -s "$OUT/rnabloom.graph$ext" "$FIX/rnabloom.graph$ext"; then echo "identical: rnabloom.graph$ext"
  else echo "DIFFERENT: rnabloom.graph$ext"; rc=1; fi
done
[ $rc -eq 0 ] && echo "the oracle's stage-1 semantics match the reference on this input" || echo "mismatch: see the files under $OUT"
exit $rc
