cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
export BIGINT_ROUNDS=1000 PROBE_T=1
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_VMEM_RD"
B="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
Cc="SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_I8"
rocprofv3 -L > $O/pmc_avail.txt 2>&1
for m in 2 3; do
  if [ $m = 3 ]; then export CWC_FORCE_MODE3=1; fi
  rocprofv3 --pmc $A -d $O/pmc_lay_m${m}_a -o runc -- python tools/gpu_bigint.py > $O/pmc_lay_m${m}_a.log 2>&1
  rocprofv3 --pmc $B -d $O/pmc_lay_m${m}_b -o runc -- python tools/gpu_bigint.py > $O/pmc_lay_m${m}_b.log 2>&1
  rocprofv3 --pmc $Cc -d $O/pmc_lay_m${m}_c -o runc -- python tools/gpu_bigint.py > $O/pmc_lay_m${m}_c.log 2>&1
  python tools/gpu_bigint.py > $O/lay_m${m}_plain.log 2>&1
done
tail -2 $O/lay_m2_plain.log $O/lay_m3_plain.log
tail -3 $O/pmc_lay_m2_a.log $O/pmc_lay_m3_c.log
