# Same-box timing of conv16 builds (diagnostic: make -C mmwave_msc_amd/csrc DIAG=<name> DIAGFLAGS=-DMMW_DIAG_CONV_...): scripts/ubench_conv16.py
# per library, alternating, two rounds.  usage: bash scripts/ab_conv16.sh "<lib> <lib> ..." [samples]   -> gpurun_out/ab_conv16.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LIBS="$1"; B=${2:-31744}
: > gpurun_out/ab_conv16.txt
for rep in 1 2; do
  for L in $LIBS; do
    [ -f mmwave_msc_amd/$L ] || continue
    echo "$L $(MMW_LIB_NAME=$L timeout 300 python3 scripts/ubench_conv16.py $B 3 2>&1 | grep -E '^conv16')" >> gpurun_out/ab_conv16.txt
  done
done
cat gpurun_out/ab_conv16.txt
