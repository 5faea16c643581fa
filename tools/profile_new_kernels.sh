#!/bin/bash
# rocprofv3 kernel stats + HBM-traffic counters (separate --pmc passes) for round 6's new kernels (tools/new_kernels_prof.py).
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06newk
mkdir -p $OUT
RP="timeout -k 5 600 rocprofv3"
cd /tmp && export TMPDIR=/tmp
$RP --kernel-trace --stats --output-format csv -d $OUT/stats -o newk -- python3 $ROOT/tools/new_kernels_prof.py > $OUT/stats.log 2>&1
$RP --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o fetch -- python3 $ROOT/tools/new_kernels_prof.py > $OUT/fetch.log 2>&1
$RP --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -o write -- python3 $ROOT/tools/new_kernels_prof.py > $OUT/write.log 2>&1
$RP --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $OUT/pmc -o sq -- python3 $ROOT/tools/new_kernels_prof.py > $OUT/sq.log 2>&1
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc $OUT/r06_new_kernels_hbm_traffic.json > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/pmc > $OUT/r06_new_kernels_pmc_summary.txt 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r06_new_kernels_kernel_stats.csv
ls -la $OUT; grep -E 'k_mr_reg|k_interp_frac' $OUT/r06_new_kernels_kernel_stats.csv | cut -c1-200
