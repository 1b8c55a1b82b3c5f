#!/bin/bash
# rocprofv3 kernel-trace summary of the device front end: tools/prof_parse.sh <tag> [frames]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-parse}; N=${2:-65536}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/parse_rate.py --frames $N --steps 20 --warmup 5 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/prof_${TAG}_kernel_stats.csv && head -5 $f
