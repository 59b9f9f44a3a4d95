cd $GRAFT_REPO_ROOT; R=$PWD; O=$R/gpurun_out/r3S; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P="$R/tools/probes/encode_modes 3 2"
timeout 80 rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL -d $O/a -o a -- $P > $O/a.log 2>&1
echo "rc $?"; grep -c fused $O/a.log
cd $R; python tools/pmc_by_dispatch.py $O/a/a_results.db k_encode_fused > $O/a.md 2>&1; wc -l $O/a.md; grep "fused" $O/a.log | head -20
