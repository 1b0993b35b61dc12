#!/bin/bash
# Re-measure what bench.py's rooflines quote from profiles/traffic.json (HBM bytes, VALU / MFMA instruction counts per launch of the
# dominant kernels of BASELINE configs[2], [3] and [4] at their per-GPU shard shapes) after a kernel source changed:
#     bash tools/refresh_traffic.sh <tag> [c1 c1phi c3 c4 c5 n882 n1270 q882 q1270 osdms c3r c4r c5r]   (c2 = the first launch of c3: no pass of its own)
# A configuration name ending in `r` runs the same shape under the two OPT-IN re-associations (FGNN_OPT_BP4_SHARED_LSE / FGNN_OPT_GNN_FACTORED = 1):
# its counts go to the `_shared` / `_factored` keys; every other name runs the library default, the reference's formulas term by term.
# Four rocprofv3 --pmc passes per configuration (~20-40 s each) on the GPU box; writes the summaries and the regenerated traffic.json into
# gpurun_out/<tag>/ (then: cp gpurun_out/<tag>/traffic.json profiles/traffic.json; cp gpurun_out/<tag>/*pmc_summary.txt profiles/).
set -e
TAG=${1:-r00}
shift || true
CONFIGS=${@:-c3 c4 c5}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SETS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY")
SPECS=""
for cfg in $CONFIGS; do
  FORMS=literal
  base=$cfg
  case $cfg in c3r|c4r|c5r) FORMS=reassociated; base=${cfg%r};; esac
  case $base in
    c1) PROG="tools/prof_kernels.py ghp882 256 fixed 32 0.05 boxplus 0.625"; SPEC="sandwich:ghp882:256:32:boxplus";;   # configs[0] as the reference constructs it
    c1phi) PROG="tools/prof_kernels.py ghp882 256 fixed 32 0.05 boxplus-phi 0.625"; SPEC="sandwich:ghp882:256:32:boxplus-phi";;     # the QLDPC.ipynb cell 11 variant
    n882) PROG="tools/prof_kernels.py ghp882 5000 fixed 64,16 0.05"; SPEC="sandwich:ghp882:5000:64:boxplus-phi";;                  # the published workloads' batch (n882.py:39)
    n1270) PROG="tools/prof_kernels.py ghp1270 5000 fixed 64,16 0.07"; SPEC="sandwich:ghp1270:5000:64:boxplus-phi";;
    osdms) PROG="tools/prof_kernels.py ghp882 50000 fixed 120 0.09 minsum 0.8"; SPEC="sandwich:ghp882:50000:120:minsum";;   # OSD.ipynb cell 6
    q882) PROG="tools/prof_kernels.py ghp882 10000 fixed 64 0.01 boxplus-phi 0.8"; SPEC="sandwich:ghp882:10000:64:boxplus-phi";;     # QLDPC.ipynb cell 12: batch 10 000
    q1270) PROG="tools/prof_kernels.py ghp1270 10000 fixed 64 0.01 boxplus-phi 0.8"; SPEC="sandwich:ghp1270:10000:64:boxplus-phi";;
    c3) PROG="tools/prof_kernels.py ghp882 65536 fixed 64,16"; SPEC="sandwich:ghp882:65536:64:boxplus-phi";;
    c4) PROG="tools/prof_kernels.py ghp1270 32768 fixed 64,64"; SPEC="sandwich:ghp1270:32768:64:boxplus-phi";;
    c5) PROG="tools/prof_gnnbp4.py 16384"; SPEC="gnnbp4:ghp1270:16384:10:";;
    *) echo "unknown config $cfg"; exit 1;;
  esac
  SPEC="$SPEC:$FORMS"
  # the forms reach the profiled program through the environment (exported here, inherited by rocprofv3's child: no exec hop in between)
  if [ $FORMS = reassociated ]; then export FGNN_BENCH_BP4_LSE=shared FGNN_BENCH_GNN_ORDER=factored; else export FGNN_BENCH_BP4_LSE=literal FGNN_BENCH_GNN_ORDER=literal; fi
  for set in "${SETS[@]}"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/${cfg}_pmc_$tag -- python3 $PROG > $O/${cfg}_pmc_$tag.log 2>&1
    echo "$cfg $tag done"
  done
  python tools/pmc_summary.py $O/${cfg}_pmc_*/*/*_counter_collection.csv > $O/${cfg}_pmc_summary.txt
  SPECS="$SPECS $SPEC=$O/${cfg}_pmc_summary.txt"
done
unset FGNN_BENCH_BP4_LSE FGNN_BENCH_GNN_ORDER
python tools/make_traffic_json.py "$TAG" merge=profiles/traffic.json $SPECS > $O/traffic.json  # entries of configurations not re-measured are kept
grep -E "(bp4_kernel|gnn_stream_kernel|gnn_bp4).* (FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU |SQ_INSTS_MFMA|GRBM_GUI_ACTIVE)" $O/c*_pmc_summary.txt || true
