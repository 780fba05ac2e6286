mkdir -p gpurun_out/r05j gpurun_out/profiles_out
(time python -m pytest tests -m gpu -q) > gpurun_out/r05j/gputests.log 2>&1; tail -4 gpurun_out/r05j/gputests.log
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r05j/bench_default.json 2> gpurun_out/r05j/bench_default.err; tail -3 gpurun_out/r05j/bench_default.err
PROFILE_PASSES="trace fetch write dram ea sq1 sq2" bash tools/profile2.sh r05_sift1b_shape_host sift1b_shape_host > gpurun_out/r05j/profile2.log 2>&1; tail -3 gpurun_out/r05j/profile2.log
ls gpurun_out/profiles_out
