set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r01n
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_abi.py -x -q -m gpu > $O/pytest_abi.log 2>&1 || { tail -20 $O/pytest_abi.log; exit 1; }
tail -2 $O/pytest_abi.log
export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python3 tools/prof_kernels.py ghp882 65536 fixed > $O/pmc_$tag.log 2>&1
  echo "pass $tag done"
done
python tools/pmc_summary.py $O/pmc_*/*/*_counter_collection.csv > $O/pmc_summary.txt
grep -E "bp4_kernel|gnn_mfma" $O/pmc_summary.txt | head -60
