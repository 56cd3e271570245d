export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $grp -d /tmp/pp -o t -- python tools/pw_f16_probe.py "x101 res4" > /dev/null 2>&1
  python - <<P
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/pp/*.db")[0])
rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%pw_f16%' group by 1,2").fetchall()
for r in rows: print(r[0][:40], r[1], "%.4g" % r[2], r[3])
P
done
