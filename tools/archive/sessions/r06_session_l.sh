# round 6, GPU session L: the suite with the native row-split capture test added (durations kept)
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r06_gputest_l.log 2>&1; tail -40 gpurun_out/r06_gputest_l.log
