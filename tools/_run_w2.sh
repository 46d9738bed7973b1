set -o pipefail
mkdir -p gpurun_out/r02_w11
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02_w11/t.log 2>&1; tail -3 gpurun_out/r02_w11/t.log
echo "== default"; for i in 1 2 3; do timeout -k 10 120 bash tools/bench_stages.sh; done
