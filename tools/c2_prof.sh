cd /tmp && export TMPDIR=/tmp
for b in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2prof$b -o c2 -- python3 $GRAFT_REPO_ROOT/tools/c2_prof.py $b > /dev/null 2>&1
echo "== batch $b"; grep k_fft_pass $(find $GRAFT_REPO_ROOT/gpurun_out/c2prof$b -name "*kernel_stats.csv") | sed 's/(bdsp::FftIo.*)"//' | cut -c1-200
done
