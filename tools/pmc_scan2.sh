#!/bin/bash
# On the GPU box: SQ issue/stall/LDS counters of the scan kernels, forward and backward apart.  usage: tools/pmc_scan2.sh OUTDIR [LIB] [BATCH]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; LIB=$2; B=${3:-8}
R=$GRAFT_REPO_ROOT
mkdir -p $O
if [ -n "$LIB" ]; then export GFE_HIP_LIB=$R/$LIB; fi
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/scan_exp/time_scan.py $B 4"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --kernel-trace --output-format csv -d $O/pmc_sq3 -o p -- $P > /dev/null 2>&1
cd $R
python3 $R/tools/pmc_summ.py $O
