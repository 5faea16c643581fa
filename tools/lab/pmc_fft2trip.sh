#!/bin/bash
# GPU box: HBM-side traffic and L2 hit counters of the two-trip lab kernels (one counter set per pass, kernel-trace only;
# every rocprofv3 run under its own timeout: a counter set the hardware cannot collect makes the tool abort and then hang)
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/lab/${1:-fft2trip_lab_w3}
shift
ARGS="${@:-3 2 6 0}"
O=$GRAFT_REPO_ROOT/gpurun_out/lab/pmc_$(date +%H%M%S)
mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $C | tr ' ' '_')
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $C -d $O/$n -o p --output-format csv -- $B $ARGS > $O/$n.log 2>&1 || echo "rocprofv3 $n: rc $?"
done
python3 - $O <<'PY'
import sys,glob,csv,collections
o=sys.argv[1]
for f in sorted(glob.glob(o+'/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:48],r['Counter_Name'])
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    for (k,c),(n,v) in sorted(agg.items()):
        if 'k_trip' in k or 'k_fft_pass' in k: print(f"{k:48s} {c:20s} launches {n:4d} avg {v/n:16.1f}")
PY
