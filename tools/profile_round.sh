#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box; only small summaries
# land in gpurun_out/ (the .db captures stay in /tmp).
#   tools/profile_round.sh r01
set -u
TAG=${1:-rXX}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
SUMMARY_ARGS=""
run() { # name, title, rocprof args..., -- cmd
  local name=$1; shift; local title=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(ls /tmp/prof_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    python tools/rocpd_summary.py $db $O/$name.md --json $O/$name.json --title "$title" $SUMMARY_ARGS > /dev/null
  else
    echo "no db for $name" >> $O/$name.log
  fi
  tail -2 $O/$name.log | cut -c1-300 > $O/$name.tail; rm -f $O/$name.log
}
# bench traces: only the 5 timed step periods (the once-per-step fused loss kernel
# is the period marker), so MIOpen's find-phase candidates of step 1 do not show
SUMMARY_ARGS="--marker cls_losses_fused_kernel --last 5"
run bench_full_trace "rocprofv3 --kernel-trace --stats: python bench.py --steps 5 --warmup 2 (default = full workload), the 5 timed steps" \
    --kernel-trace --stats -d /tmp/prof_bench_full_trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0
run bench_heads_trace "rocprofv3 --kernel-trace --stats: python bench.py --workload heads --steps 5 --warmup 2, the 5 timed steps" \
    --kernel-trace --stats -d /tmp/prof_bench_heads_trace -o t -- python bench.py --workload heads --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0
run bench_heads_f16_trace "rocprofv3 --kernel-trace --stats: python bench.py --workload heads --precision f16 --steps 5 --warmup 2 (fp16 storage subnets; not the headline precision), the 5 timed steps" \
    --kernel-trace --stats -d /tmp/prof_bench_heads_f16_trace -o t -- python bench.py --workload heads --precision f16 --steps 5 --warmup 2 --no-cpu-baseline --profile-steps 0
SUMMARY_ARGS=""
run pmc_fetch "PMC pass 1 (FETCH_SIZE, KB): python tools/kbench.py" \
    --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_pmc_fetch -o t -- python tools/kbench.py
run pmc_write "PMC pass 2 (WRITE_SIZE, KB): python tools/kbench.py" \
    --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_pmc_write -o t -- python tools/kbench.py
run pmc_mfma "PMC pass 3 (MFMA / LDS): python tools/kbench.py --what conv" \
    --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/prof_pmc_mfma -o t -- python tools/kbench.py --what conv
run pmc_mfma_f16 "PMC pass 4 (MFMA / LDS, fp16 kernels): python tools/kbench.py --what f16" \
    --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/prof_pmc_mfma_f16 -o t -- python tools/kbench.py --what f16
run pmc_mfma_gemm "PMC pass 5 (MFMA / LDS / stalls, pointwise-convolution GEMM kernels): python tools/gemm_pmc.py" \
    --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/prof_pmc_mfma_gemm -o t -- python tools/gemm_pmc.py
# FETCH_SIZE calibration on known byte counts in this repo's access patterns (tools/fetch_calib.hip)
hipcc --offload-arch=gfx950 -O3 -w tools/fetch_calib.hip -o /tmp/fetch_calib > /dev/null 2>&1
run pmc_fetch_calib "FETCH_SIZE calibration (KB per dispatch for a 1 GiB = 1048576 KB single pass): /tmp/fetch_calib" \
    --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_pmc_fetch_calib -o t -- /tmp/fetch_calib
ls -la $O
