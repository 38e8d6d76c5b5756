# round 6, GPU session R: the suite with the arch tables switched off ($PETIT_AMD_NO_TUNED=1: every default pick comes from the formula heuristic) -- parity must hold for THOSE kernels too;
# tests that assert table facts are expected to fail and are listed
PETIT_AMD_NO_TUNED=1 python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_notuned.log 2>&1; tail -25 gpurun_out/r06_gputest_notuned.log
