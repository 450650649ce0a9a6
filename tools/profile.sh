#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun).  Separate passes: kernel trace +
# stats, then one PMC pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950;
# PMC passes never carry sys/hip/hsa trace options).  Output: gpurun_out/prof/<tag>/...
# PROF_CMD="python tools/config3_run.py --records 200000000 --timing-only" profiles another command (relative to the repo root)
# instead of bench.py; PROF_PASSES="trace" (default: trace fetch write) picks the passes.
TAG=${1:-r03}
shift
ARGS=${@:---steps 12 --warmup 3 --cpu-sample 0 --no-verify --no-host-fed}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
CMD=${PROF_CMD:-python $ROOT/bench.py $ARGS}
PASSES=${PROF_PASSES:-trace fetch write}
case " $PASSES " in *" trace "*) rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1;; esac
case " $PASSES " in *" fetch "*) rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1;; esac
case " $PASSES " in *" write "*) rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1;; esac
if [ -n "$PROF_TCC" ]; then  # memory-side atomics / L2 requests (counter names as this rocprofv3 lists them; a pass with an unknown name just fails)
  rocprofv3 -L 2>/dev/null | grep -i -o "TCC_[A-Z0-9_]*ATOMIC[A-Za-z0-9_]*\|TCC_REQ_sum\|TCC_HIT_sum\|TCC_MISS_sum" | sort -u > $OUT/tcc_counter_names.txt
  rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o tcc -- $CMD > $OUT/pmc_tcc.log 2>&1
fi
if [ -n "$PROF_SQ" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq1 -o sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $OUT/pmc_sq2 -o sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM -d $OUT/pmc_sq3 -o sq3 -- $CMD > $OUT/pmc_sq3.log 2>&1
fi
cd $ROOT
PROF_BENCH_ARGS="${PROF_CMD:-$ARGS}" python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# (gpurun copies at most 64 MiB back: the databases stay on the box, the summary and traffic.json travel)
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -type f -size +2M -delete 2>/dev/null
