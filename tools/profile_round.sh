#!/bin/bash
# rocprofv3 evidence of a round (r03 onwards; rounds 1-2 used an earlier script of this name) on the GPU box.  Kernel traces of the three bench configurations and
# SEPARATE counter passes (MI355X_MICROARCH.md: one --pmc group per run, with --kernel-trace only)
# over the launches bench.py itself times (`--workload heads`, bs 16), attributed to timing classes by
# tools/pmc_by_class.py.  Only summaries land in gpurun_out/; copy them to profiles/ afterwards.
#   tools/profile_round.sh r04 [traces|pmc|all]
set -u
TAG=${1:-r03}
WHAT=${2:-all}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
trace() { # name, marker, title, bench args...
  local name=$1; shift; local marker=$1; shift; local title=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t -- python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-also --profile-steps 0 > $O/$name.log 2>&1
  local db=$(ls /tmp/prof_$name/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    local cls=""
    # the subnets-only capture: rows of kernels that serve several timing classes are also split by class, so that
    # roofline.avg_launch_ms (class 23) can be read from this file
    [ "$name" = "bench_heads_trace" ] && cls="--classes heads"
    python tools/rocpd_summary.py $db $O/$name.md --json $O/$name.json --title "$title" --marker $marker --last 5 --busy $cls > /dev/null
  fi
  tail -2 $O/$name.log | cut -c1-400 > $O/$name.tail; rm -f $O/$name.log
}
pmc() { # name, counters..., -- bench args
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d /tmp/prof_$name -o t -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-also --profile-steps 0 > $O/$name.log 2>&1
  tail -2 $O/$name.log | cut -c1-400 > $O/$name.tail; rm -f $O/$name.log
}
if [ "$WHAT" = "traces" ] || [ "$WHAT" = "all" ]; then
  trace bench_full_trace cls_losses_fused_kernel "rocprofv3 --kernel-trace --stats: python bench.py --steps 5 --warmup 2 (default: config 3, fp32, native), the 5 timed steps"
  trace bench_heads_trace cls_losses_fused_kernel "rocprofv3 --kernel-trace --stats: python bench.py --workload heads --steps 5 --warmup 2, the 5 timed steps" --workload heads
  trace bench_cfg5_f16_trace cls_losses_fused_kernel "rocprofv3 --kernel-trace --stats: python bench.py --student r101 --teacher x101-64x4d --px 500 --precision f16 --steps 5 --warmup 2 (BASELINE config 5 on native fp16 kernels), the 5 timed steps" --student r101 --teacher x101-64x4d --px 500 --precision f16
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  pmc pmc_fetch FETCH_SIZE -- --workload heads
  pmc pmc_write WRITE_SIZE -- --workload heads
  pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT -- --workload heads
  pmc pmc_wait SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES -- --workload heads
  pmc pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- --workload heads
  python tools/pmc_by_class.py --out $O/pmc_classes.json --md $O/pmc_classes.md \
      fetch=$(ls /tmp/prof_pmc_fetch/*.db | head -1) write=$(ls /tmp/prof_pmc_write/*.db | head -1) \
      mfma=$(ls /tmp/prof_pmc_mfma/*.db | head -1) wait=$(ls /tmp/prof_pmc_wait/*.db | head -1) \
      tcc=$(ls /tmp/prof_pmc_tcc/*.db | head -1) > /dev/null 2> $O/pmc_classes.err
fi
if [ "$WHAT" = "pmc_rr" ]; then
  # fetch / L2 passes with round 2's round-robin work order (A/B of the XCD-aware order; not part of "all")
  SSAD_WINO_XCD_GROUP=1 SSAD_WGRAD_XCD_GROUP=1 pmc pmc_fetch_rr FETCH_SIZE -- --workload heads
  SSAD_WINO_XCD_GROUP=1 SSAD_WGRAD_XCD_GROUP=1 pmc pmc_tcc_rr TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- --workload heads
  python tools/pmc_by_class.py --out $O/pmc_classes_roundrobin.json --md $O/pmc_classes_roundrobin.md \
      fetch=$(ls /tmp/prof_pmc_fetch_rr/*.db | head -1) tcc=$(ls /tmp/prof_pmc_tcc_rr/*.db | head -1) > /dev/null 2>> $O/pmc_classes.err
fi
if [ "$WHAT" = "pmc16" ] || [ "$WHAT" = "all" ]; then
  # the fp16-storage subnets (bench.py --workload heads --precision f16), same attribution
  pmc pmc16_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT -- --workload heads --precision f16
  pmc pmc16_fetch FETCH_SIZE -- --workload heads --precision f16
  pmc pmc16_write WRITE_SIZE -- --workload heads --precision f16
  pmc pmc16_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- --workload heads --precision f16
  python tools/pmc_by_class.py --f16 --out $O/pmc_classes_f16.json --md $O/pmc_classes_f16.md \
      mfma=$(ls /tmp/prof_pmc16_mfma/*.db | head -1) fetch=$(ls /tmp/prof_pmc16_fetch/*.db | head -1) \
      write=$(ls /tmp/prof_pmc16_write/*.db | head -1) tcc=$(ls /tmp/prof_pmc16_tcc/*.db | head -1) > /dev/null 2>> $O/pmc_classes.err
fi
ls -la $O
