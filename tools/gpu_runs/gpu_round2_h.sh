#!/bin/bash
mkdir -p gpurun_out/r02h
O=gpurun_out/r02h
timeout 300 python tools/kbench.py build/variants/libzshmc_r01.so build/variants/libzshmc_nopair.so build/variants/libzshmc_base.so build/variants/libzshmc_fence.so > $O/kbench.txt 2>&1
KB_REPS=2 timeout 300 python tools/kbench.py build/variants/libzshmc_nopair.so build/variants/libzshmc_base.so build/variants/libzshmc_fence.so --mean --mass > $O/kbench_mm.txt 2>&1
grep -v amdgpu $O/kbench.txt $O/kbench_mm.txt
timeout 900 python -m pytest tests/test_gpu_mixture_multinomial.py tests/test_gpu_fused.py tests/test_gpu_hmc_reference.py -m gpu -q > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
