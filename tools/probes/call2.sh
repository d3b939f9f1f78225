set -u
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extract.py -x -q -m gpu > gpurun_out/c2/pytest_extract.txt 2>&1
tail -5 gpurun_out/c2/pytest_extract.txt
timeout 300 python bench.py --no-extras --cpu-frames 0 > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err
timeout 300 python bench.py --no-extras --cpu-frames 0 --steps 8 > gpurun_out/c2/bench_b.json 2> gpurun_out/c2/bench_b.err
