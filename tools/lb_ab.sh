for lib in "" build/variants/lib_lbb2.so; do
  echo "== ${lib:-product}"
  LB_LIB=$lib python tools/lb_bench.py 32768 100000 1 128 | tail -1
  LB_LIB=$lib python tools/lb_bench.py 32768 100000 1 256 | tail -1
  LB_LIB=$lib python tools/lb_bench.py 8192 12419 4 128 | tail -1
done
python tools/lntm_bench.py | tail -2 | head -1
python -m pytest tests/test_gpu_linear_bernoulli.py tests/test_gpu_mixture_multinomial.py -x -q -m gpu 2>&1 | tail -3
