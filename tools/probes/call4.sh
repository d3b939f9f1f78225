set -u
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extract.py -x -q -m gpu > gpurun_out/c4/pytest_extract.txt 2>&1
tail -5 gpurun_out/c4/pytest_extract.txt
timeout 300 python bench.py --no-extras --cpu-frames 0 > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err
for K in k_describe k_blur_strips k_apron_copy; do
AFV_EXP_KERNEL=$K python tools/experiments.py run base > gpurun_out/c4/pmc_$K.json 2> gpurun_out/c4/pmc_$K.err
done
cat gpurun_out/c4/pmc_*.json
