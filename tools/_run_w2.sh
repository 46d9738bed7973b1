set -o pipefail
mkdir -p gpurun_out/r02_w10
timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_harness.py -x -q -m gpu > gpurun_out/r02_w10/t.log 2>&1; tail -3 gpurun_out/r02_w10/t.log
echo "== default"; for i in 1 2 3; do timeout -k 10 120 bash tools/bench_stages.sh; done
timeout -k 10 200 bash tools/prof_bench.sh r02_w10/prof > /dev/null 2>&1; python tools/frame_gaps.py "gpurun_out/r02_w10/prof/*kernel_trace.csv"
