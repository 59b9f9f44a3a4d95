# SQ counter passes of k_decode through the torch-free probe (through gpurun):  bash tools/scripts/dec_pmc.sh <tag> [lib-dir]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-dp}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export LD_LIBRARY_PATH=$R/$2:$LD_LIBRARY_PATH
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p -- $R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0 > $O/p$i.log 2>&1
  ( cd $R; python tools/rocpd_stats.py $O/p$i/p_results.db --min-grid 2000000 --per 16777216 | grep "k_decode" | grep "SQ_" )
done
