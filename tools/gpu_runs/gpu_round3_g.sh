#!/bin/bash
# round 3, seventh trip: document-major tiles of the multinomial kernel
# (timing at the full configs[4] shape, both tilings), GPU suite, lntm line
mkdir -p gpurun_out/r03g
{
ZSHMC_LB_DOC_MAJOR=0 timeout 300 python tools/lntm_docmajor_bench.py 8192
ZSHMC_LB_DOC_MAJOR=1 timeout 300 python tools/lntm_docmajor_bench.py 8192
ZSHMC_LB_DOC_MAJOR=0 timeout 300 python tools/lntm_docmajor_bench.py 1024
ZSHMC_LB_DOC_MAJOR=1 timeout 300 python tools/lntm_docmajor_bench.py 1024
} > gpurun_out/r03g/docmajor.txt 2>&1
grep ZSHMC gpurun_out/r03g/docmajor.txt | cut -c1-250
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r03g/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 30 gpurun_out/r03g/pytest.log | cut -c1-300
timeout 900 python bench.py --workload lntm --steps 2 --warmup 1 > gpurun_out/r03g/bench_lntm.json 2> gpurun_out/r03g/bench_lntm.err
echo "lntm bench rc=$?"; tail -c 400 gpurun_out/r03g/bench_lntm.err; cut -c1-1500 gpurun_out/r03g/bench_lntm.json
