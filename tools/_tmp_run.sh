set -o pipefail
mkdir -p gpurun_out/r02_w17
echo "== tests, bin 32 default"
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02_w17/t32.log 2>&1; tail -3 gpurun_out/r02_w17/t32.log
echo "== bench bin 32"; for i in 1 2 3; do timeout -k 10 120 bash tools/bench_stages.sh --steps 100 --warmup 20; done
echo "== bench bin 16"; for i in 1 2; do GSPLAT_AMD_BIN_TILE=16 timeout -k 10 120 bash tools/bench_stages.sh --steps 100 --warmup 20; done
