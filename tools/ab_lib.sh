# A/B two builds of the library on one box: tools/ab_lib.sh <other.so> <kbench args...>
set -e
other=$1; shift
for rep in 1 2; do
echo "== this build"; python tools/kbench.py "$@"
echo "== $other"; DEFF_AMD_LIB=$PWD/$other python tools/kbench.py "$@"
done
