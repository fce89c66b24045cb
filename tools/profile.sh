#!/bin/bash
# rocprofv3 evidence for bench.py, run ON the GPU box (via gpurun):  bash tools/profile.sh <tag> [bench args]
# Kernel trace + stats in one run; every PMC group in its own run (never combined with traces).
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-variants $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 3 --warmup 1 > $O/trace.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- $B --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- $B --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mem -- $B --steps 1 --warmup 0 > $O/pmc_mem.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq2 -- $B --steps 1 --warmup 0 > $O/pmc_sq2.log 2>&1
timeout 900 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $O/pmc_ta -- $B --steps 1 --warmup 0 > $O/pmc_ta.log 2>&1
grep "^{\"metric\"" $O/trace.log | tail -1 > $O/bench_under_trace.json
python $R/tools/summarize_profile.py $O > $O/summary.md 2>&1
find $O -type f -size +4M -delete
cat $O/summary.md
