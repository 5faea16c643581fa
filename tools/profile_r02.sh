#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC counters for the bench step
# and for every BASELINE config (tools/bench_configs.py).  Outputs under gpurun_out/r02prof/, copied to profiles/.
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --- the bench step: kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_stats.err
# --- the bench step: PMC (separate passes, no trace domains besides the kernel trace)
pmc() { # outdir name script args -- counters
  dir=$1; name=$2; shift 2
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir -o $name -- python3 $PMC_SCRIPT $PMC_ARGS > $OUT/$dir.$name.log 2>&1
}
PMC_SCRIPT=$ROOT/bench.py; PMC_ARGS="--steps 3 --warmup 1 --prewarm 0.02 --no-cpu-baseline"
pmc bench_pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc bench_pmc sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
pmc bench_pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc bench_pmc write WRITE_SIZE
pmc bench_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
# --- every BASELINE config: timings, kernel stats, PMC
python3 $ROOT/tools/bench_configs.py > $OUT/config_table.jsonl 2> $OUT/config_table.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs_stats -o configs -- python3 $ROOT/tools/bench_configs.py > $OUT/configs_under_rocprof.jsonl 2> $OUT/configs_stats.err
PMC_SCRIPT=$ROOT/tools/bench_configs.py; PMC_ARGS="--quick"
pmc configs_pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc configs_pmc sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
pmc configs_pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc configs_pmc write WRITE_SIZE
cd $ROOT
python3 tools/pmc_summary.py $OUT/bench_pmc > $OUT/bench_pmc_summary.txt 2>&1
python3 tools/pmc_traffic.py $OUT/bench_pmc $OUT/hbm_traffic.json > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/configs_pmc > $OUT/configs_pmc_summary.txt 2>&1
python3 tools/pmc_traffic.py $OUT/configs_pmc $OUT/configs_hbm_traffic.json > /dev/null 2>&1
find $OUT -name "*kernel_stats.csv" | head; ls $OUT
