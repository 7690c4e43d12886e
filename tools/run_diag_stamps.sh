# usage (GPU box): bash tools/run_diag_stamps.sh [voices] — builds the -DPG_DIAG variant in place (takes ~2 min on the box), prints the shader-clock
# stamps of workgroup 0 through the three stages (single-block launches), restores the build
cd $GRAFT_REPO_ROOT/phonic_amd/csrc
cp libphonic_gpu.so /tmp/keep.so
rm -f *.o; make -s FAST_WAVES="2 -DPG_DIAG" 2>&1 | grep -i " error"
cd ../..; python tools/diag_stamps.py ${1:-1024} --staged; python tools/diag_stamps.py 256 --staged
cp /tmp/keep.so phonic_amd/csrc/libphonic_gpu.so
