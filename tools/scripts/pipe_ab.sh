# LMC_ENCODE_PIPE (the two-kernel path's part pipelining) on / off, over the shapes that take that path (through gpurun)
R=$GRAFT_REPO_ROOT
for shape in "32 8 128 16384 256 0" "32 32 128 4096 256 1" "32 8 128 16384 236 0" "32 8 128 16384 128 0" "80 1 128 32768 256 0"; do
  for p in 0 1 0 1; do
    echo -n "shape [$shape] pipe=$p: "
    LMC_ENCODE_PIPE=$p timeout 100 $R/tools/probes/encode_ab $shape 20 0 2 2>&1 | awk '/^two-kernel/ {t=$2; g=$6} /^fused/ {f=$2} /PARITY|MISMATCH/ {p=$1" "$2} END {printf "two-kernel %s ms %s GB/s | fused-setting %s | %s\n", t, g, f, p}'
  done
done
