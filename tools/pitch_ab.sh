cd $GRAFT_REPO_ROOT
cp phonic_amd/csrc/libphonic_gpu.so /tmp/keep.so
for f in old new; do cp tools/ab_libs/$f.so phonic_amd/csrc/libphonic_gpu.so; echo "[$f]"; python tools/exp_generic_paths.py 1024 headline,pitch05,pitch07,pitch12,pitch15,pitch20,pitch30 2>&1 | grep -v '"blocks_per_call": 1,'; done
cp /tmp/keep.so phonic_amd/csrc/libphonic_gpu.so
