# A/B of two builds of liblmc_hip.so through the torch-free probe, alternating, on one box (through gpurun):
#   bash tools/scripts/ab_libs.sh <alt-name> [rounds]     (alt = build_alt/<alt-name>/liblmc_hip.so; the other is the in-tree build)
# prints per run: fused ms, two-kernel ms, decode ms, parity line
R=$GRAFT_REPO_ROOT; ALT=${1:-base}; N=${2:-3}
run() {  # $1 = label, $2 = LD_LIBRARY_PATH prefix
  LD_LIBRARY_PATH=$2:$LD_LIBRARY_PATH timeout 100 $R/tools/probes/encode_ab 32 8 128 16384 256 0 40 0 3 2>&1 | awk -v l="$1" '
    /^fused/ {f=$2} /^two-kernel/ {t=$2} /^decode/ {d=$2} /PARITY|MISMATCH|differ/ {p=p" "$0} END {printf "%-8s fused %s  two-kernel %s  decode %s |%s\n", l, f, t, d, p}'
}
for i in $(seq $N); do run "$ALT" $R/build_alt/$ALT; run new ""; done
