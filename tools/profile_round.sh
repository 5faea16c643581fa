#!/bin/bash
# Per-round profiles (TAG=r04 ...) (run on the GPU box through gpurun): rocprofv3 kernel stats and PMC counters for the bench step
# and for every BASELINE config (tools/bench_configs.py).  Outputs under gpurun_out/${TAG}prof/, copied to profiles/.
set -u
TAG=${TAG:-r06}
RP="timeout -k 5 600 rocprofv3"   # a counter set the hardware cannot collect makes the tool abort and then hang: bound every run
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${TAG}prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --- the bench step: kernel trace + stats
$RP --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-first-call --no-self-check > $OUT/bench_line_under_rocprof.json 2> $OUT/bench_stats.err
# --- the bench step: PMC (separate passes, no trace domains besides the kernel trace)
pmc() { # outdir name script args -- counters
  dir=$1; name=$2; shift 2
  $RP --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir -o $name -- python3 $PMC_SCRIPT $PMC_ARGS > $OUT/$dir.$name.log 2>&1
}
PMC_SCRIPT=$ROOT/bench.py; PMC_ARGS="--steps 3 --warmup 1 --prewarm 0.02 --no-cpu-baseline --no-first-call --no-self-check"
pmc bench_pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc bench_pmc sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS
pmc bench_pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc bench_pmc write WRITE_SIZE
pmc bench_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
# --- every BASELINE config: timings, kernel stats, PMC
python3 $ROOT/tools/bench_configs.py > $OUT/config_table.jsonl 2> $OUT/config_table.err
$RP --kernel-trace --stats --output-format csv -d $OUT/configs_stats -o configs -- python3 $ROOT/tools/bench_configs.py --no-cpu > $OUT/configs_under_rocprof.jsonl 2> $OUT/configs_stats.err
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
# --- what the figures were collected on: bench.py quotes them only while the kernel sources still hash to this value
python3 - <<PY
import json, sys, time
sys.path.insert(0, "$ROOT")
import bench
json.dump({"source_sha16": bench.kernel_source_sha16(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
           "command": "tools/profile_round.sh (rocprofv3 --kernel-trace --stats, then separate --pmc passes)"},
          open("$OUT/profile_meta.json", "w"), indent=1)
PY
# --- the per-round names under profiles/ (copy these into the repository)
mkdir -p $OUT/for_profiles
cp $OUT/profile_meta.json $OUT/for_profiles/${TAG}_profile_meta.json
cp $OUT/hbm_traffic.json $OUT/for_profiles/${TAG}_hbm_traffic.json
cp $OUT/configs_hbm_traffic.json $OUT/for_profiles/${TAG}_configs_hbm_traffic.json
cp $OUT/bench_pmc_summary.txt $OUT/for_profiles/${TAG}_bench_pmc_summary.txt
cp $OUT/configs_pmc_summary.txt $OUT/for_profiles/${TAG}_configs_pmc_summary.txt
cp $OUT/bench_line_under_rocprof.json $OUT/for_profiles/${TAG}_bench_line_under_rocprof.json
cp $OUT/config_table.jsonl $OUT/for_profiles/${TAG}_config_table.jsonl
cp $(find $OUT/bench_stats -name "*kernel_stats.csv" | head -1) $OUT/for_profiles/${TAG}_bench_kernel_stats.csv
cp $(find $OUT/configs_stats -name "*kernel_stats.csv" | head -1) $OUT/for_profiles/${TAG}_configs_kernel_stats.csv
ls -la $OUT/for_profiles
