# A/B of two builds of the library in one box: experiments/ab_lib.sh build_ab/lib_a.so build_ab/lib_b.so [env...]
A=$1; B=$2; shift 2
for rep in 1 2; do for L in $A $B; do cp $L fql_amd/libfql_amd.so; echo "== $L $@"; env "$@" FQL_CHECK_SKIP=1 timeout -k 10 120 python experiments/aql_check.py 256 512 only 2>&1 | grep -v "^\[fql\]" | tail -3 | head -2; done; done
