# usage (GPU box): bash tools/run_diag_stamps_c5.sh — the stamps of workgroup 0 through the wide staged kernel on BASELINE config 5 (Filter -> Eq5 -> Delay -> Reverb)
cd $GRAFT_REPO_ROOT/phonic_amd/csrc
cp libphonic_gpu.so /tmp/keep.so
rm -f *.o; make -s FAST_WAVES="2 -DPG_DIAG" 2>&1 | grep -i " error"
cd ../..; python tools/diag_stamps.py 1024 --staged c5; python tools/diag_stamps.py 256 --staged c5
cp /tmp/keep.so phonic_amd/csrc/libphonic_gpu.so
