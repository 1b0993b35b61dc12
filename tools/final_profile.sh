#!/bin/bash
# Everything profiles/ cites for a round, in phases that each fit one gpurun call (20 minutes):
#     bash tools/final_profile.sh <tag> tests counts   then   COUNT_CONFIGS="c5 n882 ..." bash tools/final_profile.sh <tag> counts   then   ... <tag> lines traces
# (`counts` measures COUNT_CONFIGS, default "c3 c4 c1 c1phi c3r": four rocprofv3 --pmc passes of 20-40 s per configuration — all thirteen do not fit one call)
# (GPU tests; the PMC passes of the BASELINE configurations -> traffic.json; the five bench lines; a rocprofv3 kernel trace of the
# same bench commands; the batch-size table).  Copy what is to be judged from gpurun_out/<tag>/ into profiles/ afterwards.
set -e
TAG=${1:-r00}
shift || true
PHASES=${@:-tests counts lines traces}
has() { [[ " $PHASES " == *" $1 "* ]]; }
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if has tests; then
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
fi
if has counts; then
# counts first: the bench lines below then quote THIS library's PMC passes (library_is_the_profiled_binary: true)
bash tools/refresh_traffic.sh $TAG ${COUNT_CONFIGS:-c3 c4 c1 c1phi c3r} >> $O/refresh.log 2>&1
cp $O/traffic.json profiles/traffic.json
cp profiles/traffic.json $O/traffic_after_counts.json
fi
if has lines; then
python bench.py --no-build --steps 20 --warmup 5 --require-roofline > $O/bench_c3.json 2> $O/bench.err   # the driver's own command line
python bench.py --no-build --config c4 --require-roofline > $O/bench_c4.json 2>> $O/bench.err
python bench.py --no-build --config c5 --require-roofline > $O/bench_c5.json 2>> $O/bench.err
# configs[0] as the reference constructs it (cn_type='boxplus', normalization_factor=0.625) and the QLDPC.ipynb cell 11 variant ('boxplus-phi')
python bench.py --no-build --config c1 --steps 50 --warmup 5 --require-roofline > $O/bench_c1_boxplus.json 2>> $O/bench.err
python bench.py --no-build --config c1 --cn-type boxplus-phi --steps 50 --warmup 5 --require-roofline > $O/bench_c1_boxplus_phi.json 2>> $O/bench.err
python bench.py --no-build --config c2 --require-roofline > $O/bench_c2.json 2>> $O/bench.err
# the workloads of the reference's published timings: batch_size 5 000, one and two streams in the same run
python bench.py --no-build --config n882_3r --steps 100 --warmup 5 --require-roofline > $O/bench_n882_3r.json 2>> $O/bench.err
python bench.py --no-build --config n882_5r --steps 100 --warmup 5 --require-roofline > $O/bench_n882_5r.json 2>> $O/bench.err
python bench.py --no-build --config n1270_3r --steps 100 --warmup 5 --require-roofline > $O/bench_n1270_3r.json 2>> $O/bench.err
python bench.py --no-build --config n1270_5r --steps 100 --warmup 5 --require-roofline > $O/bench_n1270_5r.json 2>> $O/bench.err
python bench.py --no-build --config n1270_coarse --steps 100 --warmup 5 --require-roofline > $O/bench_n1270_coarse.json 2>> $O/bench.err
python bench.py --no-build --config osd_bp4_minsum --steps 20 --warmup 3 --require-roofline > $O/bench_osd_bp4_minsum.json 2>> $O/bench.err
python bench.py --no-build --config qldpc_882 --steps 100 --warmup 5 --require-roofline > $O/bench_qldpc_882.json 2>> $O/bench.err
python bench.py --no-build --config qldpc_1270 --steps 100 --warmup 5 --require-roofline > $O/bench_qldpc_1270.json 2>> $O/bench.err
cat $O/bench_c3.json
echo "bench stderr:"; cat $O/bench.err || true
fi
if has traces; then
# kernel traces of the same commands (no CPU legs, no extras, --no-literal: only the launches of the warm-up and of the timed region —
# the literal-forms launches are the same BP4 symbol with a runtime flag and would be averaged into its statistics)
for cfg in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- python3 bench.py --config $cfg --steps 8 --warmup 2 --cpu-sample 0 --no-extras --no-literal --no-build > $O/trace_bench_$cfg.json 2>&1
  python tools/dispatch_summary.py $O/trace_$cfg/*/*_kernel_trace.csv > $O/dispatches_$cfg.txt
  cp $O/trace_$cfg/*/*_kernel_stats.csv $O/kernel_stats_$cfg.csv
done
cat $O/dispatches_c3.txt
python tools/batch_size_table.py $O/batch_sizes.json > $O/batch_sizes.txt 2>&1 || true
python tools/harness_rate.py 0.05 3 40 2>&1 | grep "^(" > $O/harness_rate.txt || true
fi
grep -E "(bp4_kernel|gnn_stream_kernel|gnn_bp4).* (FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU |SQ_INSTS_MFMA|GRBM_GUI_ACTIVE)" $O/c*_pmc_summary.txt || true
