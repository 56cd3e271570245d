#!/bin/bash
# PMC passes on the vector-memory / texture-addresser side of the kernels whose staging cost
# DESIGN 3.2a / 3.4 discuss: python tools/kbench.py --what conv,f16
#   tools/staging_pmc.sh r02   ->  gpurun_out/profiles_r02/pmc_staging_{a,b}.md
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
pass() { # name, title, counters...
  local name=$1; shift; local title=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/prof_$name -o t -- python tools/kbench.py --what conv,f16 > $O/$name.log 2>&1
  local db=$(ls /tmp/prof_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db $O/$name.md --json $O/$name.json --title "$title" > /dev/null; fi
  tail -2 $O/$name.log | cut -c1-200 > $O/$name.tail; rm -f $O/$name.log
}
pass pmc_staging_a "PMC staging pass A (issue side): kbench conv,f16" \
  SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE
# (a second pass on TA_BUSY_avr / TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_BUFFER_* / SQ_VMEM_TA_*_FIFO_FULL did not
#  finish within 25 minutes on this pool and was dropped: the TA-side derived counters need many replays)
ls $O | grep staging
