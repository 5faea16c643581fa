#!/bin/bash
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/ubench/l2_writeback
O=$GRAFT_REPO_ROOT/gpurun_out/l2wb
mkdir -p $O
$B 50 | tee $O/plain.txt
for C in WRITE_SIZE TCC_EA0_WRREQ_sum FETCH_SIZE; do
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $C -d $O/$C -o p --output-format csv -- $B 50 > $O/$C.log 2>&1 || echo "rocprofv3 $C rc $?"
done
python3 - $O <<'PY'
import sys,glob,csv
for f in sorted(glob.glob(sys.argv[1]+'/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'k_rewrite' in r['Kernel_Name']: print(r['Kernel_Name'][:40], r['Counter_Name'], r['Dispatch_Id'], r['Counter_Value'])
PY
