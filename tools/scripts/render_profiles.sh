#!/bin/bash
# Turn the databases tools/scripts/profile_round.sh left under gpurun_out/<dir> into the committed summaries:
#   bash tools/scripts/render_profiles.sh r2c r02_c
set -e
O=gpurun_out/$1; TAG=$2
python tools/rocpd_stats.py $O/stats/bench_results.db --min-grid 4000000 > profiles/${TAG}_bench_kernel_stats.md
{
  echo "# Round 2 (${TAG}): rocprofv3 PMC passes of \`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras\`"
  echo
  echo "Three separate \`rocprofv3 --kernel-trace --pmc ...\` runs (SQ set; FETCH_SIZE + GRBM_GUI_ACTIVE; WRITE_SIZE), as MI355X_MICROARCH.md prescribes (no trace domain besides --kernel-trace; the commands are in \`tools/scripts/profile_round.sh\`). Only full-context dispatches (64 chunks, >= 4 M work-items) are averaged. \"per unit\" = per wave token-step (64 chunks x 64 planes x 16 groups x 256 tokens = 16.78 M per dispatch). \`tools/make_latest_profile.py\` turns the same three databases into \`profiles/latest.json\`, which is where \`bench.py\` takes \`roofline.traffic\` and the VALU roof from. The kernel-stats summary of the default command (\`python bench.py --no-cpu-baseline --no-extras\`: 3 warm-up + 20 timed + 10 profiled steps) is \`${TAG}_bench_kernel_stats.md\`."
  echo
  for d in sq fe wr; do python tools/rocpd_stats.py $O/$d/${d}_results.db --min-grid 4000000 --per 16777216 | sed -n '/## PMC/,$p'; echo; done
  echo "# k_decode: the same three passes over \`python tools/probes/decode_rate.py --reps 3\` (blobs of the bench workload resident in HBM; full-context launches only)"
  echo
  echo "FETCH_SIZE x 2 (the guide's gfx950 correction for this kernel's loads) = the 510 MB of blobs + headers / counts / scales re-read by the 16 groups of a plane; WRITE_SIZE = 2.147 GB = the raw KV, written once: traffic 1.01 x algorithmic."
  echo
  for d in dsq dfe dwr; do python tools/rocpd_stats.py $O/$d/${d}_results.db --min-grid 4000000 --per 16777216 | sed -n '/## PMC/,$p' | grep -i "k_decode\|^|---\|^| kernel\|^##"; echo; done
} > profiles/${TAG}_pmc.md
python tools/make_latest_profile.py $TAG $O/sq/sq_results.db $O/fe/fe_results.db $O/wr/wr_results.db > profiles/latest.json
