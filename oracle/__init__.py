"""CPU oracle for the zhusuan.HMC hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm
(/root/reference/zhusuan/hmc.py, distributions/univariate.py, diagnostics.py).
It exists to CHECK the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``zhusuan_amd/`` does (tests/test_no_oracle_in_product.py enforces that).

Pinning status (see DESIGN.md section "Oracle"):
  * log_prob closed forms (Normal/Bernoulli/Categorical/UnnormalizedMultinomial)
    -- PINNED against the reference's own test vectors (scipy oracle), see
    tests/golden/logprob_vectors.json + tests/test_oracle_distributions.py.
  * ESS -- PINNED against outputs of the reference's own
    zhusuan/diagnostics.py imported in the build container
    (oracle/make_golden.py -> tests/golden/ess_fixture.npz).
  * Philox4x32-10 -- PINNED against the Random123 known-answer vectors.
  * HMC transition numerics (leapfrog / MH / dual averaging / mass / step-size
    search) -- **parity unpinned**: the reference runs on TensorFlow, which is
    not installable here, and its own tests hold no trajectory-level vectors
    (only an unseeded KDE bound, tests/test_mcmc.py:55-62).  The restatement
    follows hmc.py line by line and is validated statistically.
"""
