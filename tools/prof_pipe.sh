#!/bin/bash
# Runs on the GPU box (via gpurun): the headline route from tools/micro/pipe_drive (a tight C loop, no Python) — untraced, then
# under `rocprofv3 --kernel-trace --stats` with the program directly behind `--`, then bench.py's own line on the same box.
# usage: tools/prof_pipe.sh <tag> [pipe_drive args...]
set -u
TAG=${1:-pipe}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/micro/pipe_drive "$@" > $OUT/pipe_drive.json 2> $OUT/pipe_drive.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $R/tools/micro/pipe_drive --launches 20000 --repeats 3 "$@" > $OUT/pipe_drive_traced.json 2> $OUT/pipe_drive_traced.err
python3 $R/tools/kernel_intervals.py $OUT/trace --regions > $OUT/intervals.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
cd $R
python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err
cat $OUT/pipe_drive.json $OUT/pipe_drive_traced.json $OUT/intervals.txt
python3 - <<PY
import json
for n in ("default", "20steps"):
    try:
        d = json.loads(open("$OUT/bench_%s.json" % n).read().strip().splitlines()[-1])
        print("bench.py", n, "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
    except Exception as e:
        print("bench.py", n, "failed:", e)
PY
