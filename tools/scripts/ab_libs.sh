# A/B of builds of liblmc_hip.so through the torch-free probe, alternating, on one box (through gpurun):
#   bash tools/scripts/ab_libs.sh "<alt-name> [<alt-name> ...]" [rounds]    (alt = build_alt/<alt-name>/liblmc_hip.so; "new" = the in-tree build)
# prints per run: fused ms, two-kernel ms (k_quantize + k_cdf_encode), decode ms, parity line
R=$GRAFT_REPO_ROOT; ALTS=${1:-base}; N=${2:-3}
run() {  # $1 = label, $2 = LD_LIBRARY_PATH prefix
  LD_LIBRARY_PATH=$2:$LD_LIBRARY_PATH timeout 100 $R/tools/probes/encode_ab 32 8 128 16384 256 0 40 0 3 2>&1 | awk -v l="$1" '
    /^fused/ {f=$2} /^two-kernel/ {t=$2; q=$10; c=$11} /^decode/ {d=$2} /PARITY|MISMATCH/ {p=$1" "$2} END {printf "%-8s fused %s  two-kernel %s (%s + %s)  decode %s | %s\n", l, f, t, q, c, d, p}'
}
# the order rotates from round to round (a fixed order would hand the same build the coolest / warmest slot every time)
LIST="$ALTS new"; K=$(echo $LIST | wc -w)
for i in $(seq $N); do
  for j in $(seq $K); do
    a=$(echo $LIST | cut -d" " -f$(( (i + j - 2) % K + 1 )))
    if [ "$a" = new ]; then run new ""; else run "$a" $R/build_alt/$a; fi
  done
done
