# VMEM / TA / TCP counter passes of k_decode through the torch-free probe, with and without its stores
# (through gpurun):  bash tools/scripts/dec_store_pmc.sh <tag> [lib-dir]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-dsp}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export LD_LIBRARY_PATH=$R/$2:$LD_LIBRARY_PATH
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
P2="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU"
# (TA_* / TCP_* sets were tried as third and fourth passes: this rocprofv3 build wrote no database for them)
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p -- $R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0 > $O/p$i.log 2>&1
  ( cd $R; python tools/rocpd_stats.py $O/p$i/p_results.db --min-grid 2000000 --per 16777216 | grep "k_decode" )
done
