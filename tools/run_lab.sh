cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lab
export LAB_DUMP=$GRAFT_REPO_ROOT/gpurun_out/lab
BIN=${LAB_BIN:-conv_lab}
(cd tools/lab && timeout 300 ./$BIN "$@") 2>&1 | tee gpurun_out/lab/${BIN}_$(date +%H%M%S).txt
