#!/bin/bash
# Replays the committed stage-1 fixtures against a real RNA-Bloom (needs a JRE and RNA-Bloom.jar v2.0.1 — neither exists in
# the build image, which is why parity below the hash layer is "unpinned"; this is the one command that pins it).
#   tools/replay_with_jar.sh /path/to/RNA-Bloom.jar [workdir]
#   tools/replay_with_jar.sh --check        no java needed: every fixture directory, read file, argument list and cmp target the
#                                           replay names must exist (the CPU suite runs this, tests/test_golden_stage1.py, so the
#                                           one command that would pin the oracle cannot rot while no JRE is around)
# Stage 1 only (-stage 1), one thread (-t 1: the reference's filter updates are not atomic, so only -t 1 is reproducible),
# -savebf keeps the graph files (R/RNABloom.java:7181-7185), which are then compared byte for byte with the fixtures:
#   tests/golden/stage1_small          uniform 100 bp pairs (the plain case)
#   tests/golden/stage1_rich/pe        300 bp ragged pairs with lower case, U, N, '#' and '$' qualities, -revcomp-right
#   tests/golden/stage1_rich/stranded  the same files with -stranded
#   tests/golden/stage1_rich/sef       one single-end forward file
#   tests/golden/stage1_rich/long      long reads at k = 35 (-long: no pair filter)
# (generators: tests/golden/gen_stage1_small.py, gen_stage1_rich.py; every counter stays below 16, so no run draws a random number)
# A -long run makes main() probe for minimap2 and racon (R/RNABloom.java:6778-6784) although stage 1 never calls them: if they
# are not installed, two do-nothing stand-ins are put on the PATH of that one run.
JAR=${1:?usage: replay_with_jar.sh RNA-Bloom.jar [workdir] | --check}
HERE=$(cd "$(dirname "$0")/.." && pwd)
CHECK=0
if [ "$JAR" = "--check" ]; then CHECK=1; WORK=/nonexistent; else WORK=${2:-$(mktemp -d)}; fi
rc=0
check() {    # what a replay of this run would touch: the read files its arguments name, and one expected file per graph file
  local name=$1 want=$2 reads=$3; shift 3
  local n=0 prev=
  [ $# -ge 6 ] || { echo "$name: no argument list"; rc=1; return; }
  for a in "$@"; do
    case "$prev" in -left|-right|-sef|-ser|-long) [ -s "$reads/$a" ] || { echo "$name: read file $reads/$a missing"; rc=1; }; n=$((n+1));; esac
    prev=$a
  done
  [ $n -ge 1 ] || { echo "$name: the arguments name no read file: $*"; rc=1; }
  case " $* " in *" -stage 1 "*) ;; *) echo "$name: not a stage-1 run: $*"; rc=1;; esac
  case " $* " in *" -savebf "*) ;; *) echo "$name: -savebf missing (no graph files would be kept): $*"; rc=1;; esac
  case " $* " in *" -t 1 "*) ;; *) echo "$name: -t 1 missing (only one thread is reproducible): $*"; rc=1;; esac
  [ -d "$want" ] || { echo "$name: fixture directory $want missing"; rc=1; return; }
  local need="rnabloom.graph rnabloom.graph.dbgbf rnabloom.graph.dbgbf.desc rnabloom.graph.cbf rnabloom.graph.cbf.desc"
  case " $* " in *" -long "*) ;; *) need="$need rnabloom.graph.rpkbf rnabloom.graph.rpkbf.desc";; esac
  for f in $need; do [ -s "$want/$f" ] || { echo "$name: cmp target $want/$f missing"; rc=1; }; done
  echo "ok: $name ($n read file(s), $(echo $need | wc -w) cmp targets): $*"
}
replay() {   # name, fixture dir with the expected files, directory the read files are in, arguments
  if [ $CHECK -eq 1 ]; then check "$@"; return; fi
  local name=$1 want=$2 reads=$3; shift 3
  local out=$WORK/$name
  mkdir -p "$out"
  echo "== $name: java -jar RNA-Bloom.jar $* -outdir $out"
  ( cd "$reads" && java -jar "$JAR" "$@" -outdir "$out" ) > "$out/log.txt" 2>&1 || { echo "RNA-Bloom failed, see $out/log.txt"; rc=1; return; }
  local exts="$(cd "$want" && ls rnabloom.graph*)"
  for f in $exts; do
    if cmp -s "$out/$f" "$want/$f"; then echo "identical: $f"; else echo "DIFFERENT: $f"; rc=1; fi
  done
  for f in $(cd "$out" && ls rnabloom.graph* 2>/dev/null); do
    [ -e "$want/$f" ] || { echo "UNEXPECTED file written by the reference: $f"; rc=1; }
  done
}
S=$HERE/tests/golden/stage1_small
R=$HERE/tests/golden/stage1_rich
args() { python3 -c "import json,sys; print(json.load(open('$R/MANIFEST.json'))['runs'][sys.argv[1]]['args'])" "$1"; }
replay small "$S" "$S" -left L.fq -right R.fq -revcomp-right -k 25 -t 1 -fpr 0.01 -nk 4000 -stage 1 -savebf
for run in pe stranded sef; do replay $run "$R/$run" "$R" $(args $run); done
if [ $CHECK -eq 0 ] && { ! command -v minimap2 >/dev/null || ! command -v racon >/dev/null; }; then
  mkdir -p "$WORK/bin"
  for t in minimap2 racon; do command -v $t >/dev/null || { printf '#!/bin/sh\nexit 0\n' > "$WORK/bin/$t"; chmod +x "$WORK/bin/$t"; }; done
  PATH=$WORK/bin:$PATH
fi
replay long "$R/long" "$R" $(args long)
if [ $CHECK -eq 1 ]; then [ $rc -eq 0 ] && echo "replay recipe complete: 5 runs" || echo "replay recipe BROKEN"; exit $rc; fi
[ $rc -eq 0 ] && echo "the oracle's stage-1 semantics match the reference on all five inputs" || echo "mismatch: see the files under $WORK"
exit $rc
