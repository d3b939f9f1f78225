set -u
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
for K in k_describe k_blur_apron k_fast_nms; do
AFV_EXP_KERNEL=$K python tools/experiments.py run base > gpurun_out/c3/pmc_$K.json 2> gpurun_out/c3/pmc_$K.err
done
cat gpurun_out/c3/pmc_*.json
