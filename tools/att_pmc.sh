# Dev tool (GPU box): hardware counters of the attention kernels under tools/att_probe.py, one rocprofv3 --pmc pass per counter group.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4/attpmc; mkdir -p $O; cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/att_probe.py > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
done
cd $R; python3 tools/pmc_by_kernel.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 > $O/summary.txt; rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
python3 - <<'PY'
import os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4/attpmc/summary.txt")
L=[l.split() for l in open(p)]
names=L[0][4:-2]
for row in L[1:]:
    if "attention" not in row[0]: continue
    # kernel name may contain no spaces; row = name blocks calls avg_us counters...
    print(row[0], "blocks", row[1], "calls", row[2], "avg_us", row[3])
    for n, v in zip(names, row[4:4+len(names)]): print(f"    {n:32s} {v}")
PY
