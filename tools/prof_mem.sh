# tools/prof_mem.sh <tag> <cic_one args...> -- memory-side PMC passes for one CIC shape (read / write request latencies, EA stalls, write request sizes).
# Few counters of one block per pass (a request the hardware cannot schedule makes rocprofv3 abort and then hang: every pass under `timeout`).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/cic_one.py $*"
T="timeout -k 5 120"
$T rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
$T rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $OUT -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
$T rocprofv3 --pmc TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
$T rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
$T rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum -d $OUT -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
$T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum -d $OUT -o pmc5 -- $CMD > $OUT/pmc5.log 2>&1
python $R/tools/pmc_summary.py $OUT $OUT.txt > /dev/null
rm -rf $OUT/*.db $OUT/*/
