#!/bin/bash
# Round-end record of a build on one box: full GPU suite, the default bench line, rocprofv3 kernel stats of the overlapped and the
# single-stream step.  Usage (on the GPU box, from the repo root): bash tools/final_profile.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
python -m pytest $R/tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
python $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ov -- python3 $R/bench.py --steps 19 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o se -- python3 $R/bench.py --steps 19 --warmup 5 --no-cpu-baseline --no-extras --serial > $O/prof_serial_bench.json 2> $O/prof_serial.err || exit 3
find $O -name "*kernel_stats.csv" -exec ls -la {} \;
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -size +20M -delete
python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['hbm_classes']['bn']['ms'])"
