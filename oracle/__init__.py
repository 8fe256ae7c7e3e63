"""CPU oracle for the zhusuan.HMC hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm
(/root/reference/zhusuan/hmc.py, distributions/univariate.py, diagnostics.py).
It exists to CHECK the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``zhusuan_amd/`` does (tests/test_no_oracle_in_product.py enforces that).

A second, independent restatement of the diag-Normal transition in C + OpenMP
(oracle/c/hmc_diag_normal_port.c, wrapper oracle/hmc_c.py) serves as the
all-cores CPU baseline of bench.py; it is held to this NumPy oracle by
tests/test_oracle_c_port.py.

Pinning status (see DESIGN.md section "Oracle"):
  * log_prob closed forms (Normal/Bernoulli/Categorical/UnnormalizedMultinomial)
    -- PINNED against the reference's own test vectors (scipy oracle), see
    tests/golden/logprob_vectors.json + tests/test_oracle_distributions.py.
  * ESS -- PINNED against outputs of the reference's own
    zhusuan/diagnostics.py imported in the build container
    (oracle/make_golden.py -> tests/golden/ess_fixture.npz).
  * Philox4x32-7 -- PINNED against the Random123 known-answer vectors.
  * HMC transition numerics (leapfrog / MH / dual averaging / mass / step-size
    search) -- PINNED against traces of the reference's own zhusuan/hmc.py,
    run unmodified over the eager TensorFlow-API shim oracle/tf_shim.py
    (oracle/make_golden_hmc.py -> tests/golden/hmc_reference_traces.npz,
    tests/test_oracle_hmc_reference.py; the device path is held to the same
    traces in tests/test_gpu_hmc_reference.py).  Not pinnable without a real
    TensorFlow: its Eigen kernels' last-bit rounding and its random stream.
  * SGMCMC update numerics (oracle/sgmcmc_ref.py) -- PINNED the same way
    against the reference's own zhusuan/sgmcmc.py (oracle/make_golden_sgmcmc.py
    -> tests/golden/sgmcmc_reference_traces.npz).
"""
