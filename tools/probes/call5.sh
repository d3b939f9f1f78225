set -u
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_match.py tests/test_gpu_table.py -x -q -m gpu > gpurun_out/c5/pytest.txt 2>&1
tail -5 gpurun_out/c5/pytest.txt
timeout 300 python bench.py --no-extras --cpu-frames 0 > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
timeout 300 python bench.py --workload pairs10k --cpu-frames 0 --steps 5 > gpurun_out/c5/pairs.json 2> gpurun_out/c5/pairs.err
