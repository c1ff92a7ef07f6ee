# Counters for the kernels of the configs[2] training step AT THE STEP'S OWN SHAPES (B = 16 x 2 s, predictor heads, every chain
# serial so that kernel times add up): one un-profiled kernel trace for the durations, then separate --pmc passes (SQ issue /
# wait / matrix-pipe counters, LDS counters, FETCH_SIZE, WRITE_SIZE -- never combined with other trace domains).
# usage: bash tools/tune/pmc_train.sh <tag> [extra env assignments ...]      -> gpurun_out/<tag>/pmc_train.json
TAG=${1:-pmc_train}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1 FAC_QUANT_STREAMS=1 FAC_MEL_STREAMS=1 "$@"
CMD="$R/tools/train_bench.py --batch 16 --steps 2 --warmup 1 --predictors"
rocprofv3 -L > $O/counters_available.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $O/counters_available.txt && printf "%s " $c; done; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python $CMD > $O/train_unprofiled.json 2>$O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
A=$(have SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE)
B=$(have SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM)
echo "pass A: $A"; echo "pass B: $B"
rocprofv3 --pmc $A --kernel-trace --output-format csv -d $O/pa -o a -- python $CMD > $O/pa.log 2>&1
rocprofv3 --pmc $B --kernel-trace --output-format csv -d $O/pb -o b -- python $CMD > $O/pb.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o f -- python $CMD > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o w -- python $CMD > $O/pw.log 2>&1
CC=""
for d in pa pb pf pw; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && CC="$CC $f" || { echo "pass $d produced no counters"; tail -5 $O/$d.log; }; done
python $R/tools/pmc_kernels.py $O/pmc_train.json $O/train_kernel_stats.csv $CC
rm -rf $O/kt $O/pa $O/pb $O/pf $O/pw
