# Diagnostic PMC passes over the torch-free A/B probe (run on the GPU box through gpurun):
#   bash tools/scripts/profile_diag.sh <outdir-under-gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > $O/counters.txt
AB="$R/tools/probes/encode_ab 32 8 128 16384 256 0 3 0"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/sq -o sq -- $AB > $O/sq.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $O/sq2 -o sq2 -- $AB > $O/sq2.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_LEVEL_WAVES -d $O/sq3 -o sq3 -- $AB > $O/sq3.log 2>&1
cd $R; find gpurun_out/$1 -name "*.db" | xargs ls -la; for d in sq sq2 sq3; do python tools/rocpd_stats.py $O/$d/${d}_results.db --min-grid 2000000 --per 16777216 | sed -n '/## PMC/,$p' > $O/$d.md; done; tail -3 $O/sq2.log $O/sq3.log
