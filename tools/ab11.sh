bash tools/ab7.sh
cd $GRAFT_REPO_ROOT
FA_VERBOSE=1 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 2>&1 >/dev/null | grep "sketch tuples"
FA_LIB_VARIANT=prev FA_VERBOSE=1 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 2>&1 >/dev/null | grep "sketch tuples"
