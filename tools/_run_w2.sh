set -o pipefail
mkdir -p gpurun_out/r02_w12
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "sort or bin or depth or order or tile or bucket or forward_backward or list" > gpurun_out/r02_w12/t.log 2>&1; tail -3 gpurun_out/r02_w12/t.log
echo "== default"; for i in 1 2 3; do timeout -k 10 120 bash tools/bench_stages.sh; done
timeout -k 10 200 bash tools/prof_bench.sh r02_w12/prof > /dev/null 2>&1; sed -n 3,20p gpurun_out/r02_w12/prof/summary.md | cut -c1-100
