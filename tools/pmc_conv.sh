#!/bin/bash
# PMC passes over tools/convbench.py for one shape: SQ wait / busy split and LDS activity per kernel.  Usage (on the GPU box): tools/pmc_conv.sh "8,120,160,800,320,3" outdir
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SHAPES="$1"
O=$R/gpurun_out/$2
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools/convbench.py > $O/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $R/tools/convbench.py > $O/p2.log 2>&1 || exit 1
cd $R
python tools/pmc_by_kernel.py $O/p1 igemm_dma > $O/pmc1.txt
python tools/pmc_by_kernel.py $O/p2 igemm_dma > $O/pmc2.txt
rm -rf $O/p1 $O/p2
cat $O/pmc1.txt $O/pmc2.txt
