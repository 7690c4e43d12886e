cd $GRAFT_REPO_ROOT
# (the variant is selected with PHONIC_LIB: the in-tree library is never overwritten)
for f in old new; do echo "[$f]"; PHONIC_LIB=$PWD/tools/ab_libs/$f.so python tools/exp_generic_paths.py 1024 headline,pitch05,pitch07,pitch12,pitch15,pitch20,pitch30 2>&1 | grep -v '"blocks_per_call": 1,'; done
