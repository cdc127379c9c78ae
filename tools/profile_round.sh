#!/bin/bash
# Round profile of the headline command, run ON THE GPU BOX from the repo root:
#   tools/profile_round.sh NAME GIT_SHA     (e.g. r02/final 6102449)
# writes profiles/NAME_kernel_stats.csv, _pmc_traffic.json, _bench.json into gpurun_out/profiles/ (copy them to profiles/).
# Three separate runs of the same command (kernel trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE), as the guide prescribes.
set -e -o pipefail
NAME=$1; SHA=$2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles/$(dirname $NAME); mkdir -p $OUT
BASE=$(basename $NAME)
export TMPDIR=/tmp IRM_GIT_SHA=$SHA
CMD="python3 $ROOT/bench.py --steps 4 --warmup 2 --no-legs"
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $CMD > $OUT/${BASE}_bench_profiled.json 2> /tmp/prof_kt.err
KS=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] || { tail -20 /tmp/prof_kt.err; find /tmp/prof_kt | head; exit 1; }
cp $KS $OUT/${BASE}_kernel_stats.csv
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- $CMD > /dev/null 2> /tmp/prof_f.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- $CMD > /dev/null 2> /tmp/prof_w.err
echo "write pass done"
cd $ROOT
python3 tools/pmc_traffic.py /tmp/prof_f /tmp/prof_w $OUT/${BASE}_pmc_traffic.json
# the bench line itself (un-profiled), reading the traffic file just written
mkdir -p profiles/$(dirname $NAME); cp $OUT/${BASE}_pmc_traffic.json profiles/${NAME}_pmc_traffic.json
python3 bench.py > $OUT/${BASE}_bench.json 2> $OUT/${BASE}_bench.err
tail -c 400 $OUT/${BASE}_bench.json
