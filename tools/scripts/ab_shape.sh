# A/B of library builds on ANOTHER geometry through the torch-free probe (through gpurun):
#   bash tools/scripts/ab_shape.sh "<alt> [<alt> ...]" rounds L H D ctx chunk     (alt = build_alt/<alt>/liblmc_hip.so; "new" = in-tree)
R=$GRAFT_REPO_ROOT; ALTS=$1; N=$2; shift 2
LIST="$ALTS new"; K=$(echo $LIST | wc -w)
for i in $(seq $N); do
  for j in $(seq $K); do
    a=$(echo $LIST | cut -d" " -f$(( (i + j - 2) % K + 1 )))
    if [ "$a" = new ]; then P=""; else P=$R/build_alt/$a; fi
    LD_LIBRARY_PATH=$P:$LD_LIBRARY_PATH timeout 100 $R/tools/probes/encode_ab $1 $2 $3 $4 $5 0 20 0 3 2>&1 | awk -v l="$a" '
      /^fused/ {f=$2} /^two-kernel/ {t=$2} /^decode/ {d=$2} /PARITY|MISMATCH/ {p=$1" "$2} END {printf "%-6s fused %s  two-kernel %s  decode %s | %s\n", l, f, t, d, p}'
  done
done
