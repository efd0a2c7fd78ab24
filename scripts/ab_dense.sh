# Same-box A/B of Dense-1 builds (diagnostic: make -C mmwave_msc_amd/csrc DIAG=<name> DIAGFLAGS=...): scripts/bench_dense1.py per library,
# alternating, two rounds.  usage: bash scripts/ab_dense.sh "<lib> <lib> ..." [rows]   -> gpurun_out/ab_dense.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LIBS="$1"; ROWS=${2:-31744}
: > gpurun_out/ab_dense.txt
for rep in 1 2; do
  for L in $LIBS; do
    [ -f mmwave_msc_amd/$L ] || continue
    echo "== $L" >> gpurun_out/ab_dense.txt
    MMW_LIB_NAME=$L timeout 300 python3 scripts/bench_dense1.py $ROWS 2>&1 | grep -E "k_mars_dense1|conv pair|whole|max" >> gpurun_out/ab_dense.txt
  done
done
cat gpurun_out/ab_dense.txt
