# usage (GPU box): bash tools/run_diag_bus_chain.sh — builds the -DPG_DIAG variant in place, prints the stamps of a one-Reverb bus chain, restores the build
cd $GRAFT_REPO_ROOT/phonic_amd/csrc
cp libphonic_gpu.so /tmp/keep.so
rm -f *.o; make -s FAST_WAVES="2 -DPG_DIAG" 2>&1 | grep -i " error"
cd ../..; python tools/diag_bus_chain.py
cp /tmp/keep.so phonic_amd/csrc/libphonic_gpu.so
