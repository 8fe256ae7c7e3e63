# A/B of the fused MFMA likelihood kernel: product library vs variant libraries
# (LB_LIB) at the config-3 / config-5 shapes.  Usage: bash tools/lb_ab.sh lib1.so ...
for lib in "" "$@"; do
  echo "== ${lib:-product}"
  LB_LIB=$lib python tools/lb_bench.py 32768 100000 1 256 | tail -1
  LB_LIB=$lib python tools/lb_bench.py 32768 100000 1 128 | tail -1
  LB_LIB=$lib python tools/lb_bench.py 8192 12419 4 128 | tail -1
done
