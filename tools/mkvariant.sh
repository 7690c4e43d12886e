# usage: bash tools/mkvariant.sh <name> ["extra flags"]   — builds the CURRENT tree into tools/ab_libs/<name>.so (travels to the GPU box).
# Built in a scratch copy of the sources: the in-tree objects and libphonic_gpu.so are left alone.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=/tmp/pg_variant_$1
rm -rf $W; mkdir -p $W/phonic_amd $W/include
cp -r $ROOT/phonic_amd/csrc $W/phonic_amd/csrc
cp $ROOT/include/phonic_gpu.h $W/include/
cd $W/phonic_amd/csrc
rm -f *.o libphonic_gpu.so
make -s -j8 FAST_WAVES="2 $2" 2>&1 | grep -iE " error|spill" | head -5 || true
mkdir -p $ROOT/tools/ab_libs
cp libphonic_gpu.so $ROOT/tools/ab_libs/$1.so
echo "built tools/ab_libs/$1.so"
