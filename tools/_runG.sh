SECTIONS="L" bash tools/plan_matrix.sh > gpurun_out/plan_matrix.log 2>&1
cp gpurun_out/plan_matrix.txt gpurun_out/plan_matrix_L.txt; tail -14 gpurun_out/plan_matrix_L.txt
