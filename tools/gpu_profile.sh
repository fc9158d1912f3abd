#!/bin/bash
# rocprofv3 kernel trace + stats of one bench run; summaries are copied to gpurun_out/prof_* for profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
SHA=$(python tools/lib_id.py)
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- $CMD > gpurun_out/prof_${TAG}_bench.log 2> gpurun_out/prof_${TAG}_bench.err
find /tmp/prof -name "*stats*" | head > gpurun_out/prof_${TAG}_files.log
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp "$f" gpurun_out/prof_${TAG}_kernel_stats.csv; done
for f in $(find /tmp/prof -name "*kernel_trace.csv"); do python tools/trace_summary.py "$f" $SHA $CMD > gpurun_out/prof_${TAG}_trace_summary.txt; done
ls -la /tmp/prof/* >> gpurun_out/prof_${TAG}_files.log 2>&1
