#!/bin/bash
# rocprofv3 kernel-trace summary of the resident pipeline (parse -> refresh -> decode): tools/prof_pipeline.sh <tag> [streams]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pipeline}; S=${2:-4096}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/pipeline_rate.py --streams $S --frames 16 --resident --steps 20 > $OUT/log.txt 2>&1 < /dev/null
tail -1 $OUT/log.txt | cut -c1-300
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/prof_${TAG}_kernel_stats.csv; head -8 "$f" | cut -c1-160; fi
