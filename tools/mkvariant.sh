# usage: bash tools/mkvariant.sh <name> ["extra flags"]   — builds the CURRENT tree into tools/ab_libs/<name>.so (travels to the GPU box)
set -e
cd "$(dirname "$0")/../phonic_amd/csrc"
rm -f *.o libphonic_gpu.so
make -s FAST_WAVES="2 $2" 2>&1 | grep -iE " error|spill" | head -5 || true
cp libphonic_gpu.so ../../tools/ab_libs/$1.so
echo "built tools/ab_libs/$1.so"
