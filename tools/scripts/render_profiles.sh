#!/bin/bash
# Turn the databases tools/scripts/profile_round.sh left under gpurun_out/<dir> into the committed summaries:
#   bash tools/scripts/render_profiles.sh rprof r04_h
set -e
O=gpurun_out/$1; TAG=$2
python tools/rocpd_stats.py $O/stats/bench_results.db --min-grid 2000000 > profiles/${TAG}_bench_kernel_stats.md
{
  echo "# ${TAG}: rocprofv3 PMC passes of \`python bench.py --no-cpu-baseline --no-extras --profile-legs --ramp-ms 0 --steps 5 --warmup 2\`"
  echo
  echo "The bench command itself (BASELINE configs[1]: Llama-3-8B bf16, 16 384 tokens, 64 chunks); \`--profile-legs\` adds five jobs of the two-kernel path and five HBM-resident decodes of the context behind the timed region, so each database holds \`k_encode_fused\` (the default at this size) next to \`k_quantize\` + \`k_cdf_encode\` and \`k_decode\`. Three separate \`rocprofv3 --kernel-trace --pmc ...\` runs (SQ set; FETCH_SIZE + GRBM_GUI_ACTIVE; WRITE_SIZE), as MI355X_MICROARCH.md prescribes (no trace domain besides --kernel-trace; the commands are in \`tools/scripts/profile_round.sh\`). Only full-context dispatches (>= 2 M work-items) are averaged. \"per unit\" = per wave token-step (64 chunks x 64 planes x 16 groups x 256 tokens = 16.78 M per dispatch). \`tools/make_latest_profile.py\` turns the same three databases into \`profiles/latest.json\`, which is where \`bench.py\` takes \`roofline.traffic\` and the VALU roof from. The kernel-stats summary of the same command with the default ramp / warm-up / steps is \`${TAG}_bench_kernel_stats.md\` (it has the \`k_decode\` row too)."
  echo
  for d in sq fe wr; do python tools/rocpd_stats.py $O/$d/${d}_results.db --min-grid 2000000 --per 16777216 | sed -n '/## PMC/,$p'; echo; done
} > profiles/${TAG}_pmc.md
python tools/make_latest_profile.py $TAG $O/sq/sq_results.db $O/fe/fe_results.db $O/wr/wr_results.db > profiles/latest.json
