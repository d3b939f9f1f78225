set -u
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/probes/probe_cvt_pk_u8.hip -o /tmp/p 2>/dev/null && /tmp/p > gpurun_out/c1/probe_cvt.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_match.py tests/test_gpu_table.py -x -q -m gpu > gpurun_out/c1/pytest_match.txt 2>&1
for E in 0 1; do
  timeout 300 python bench.py --no-extras --cpu-frames 0 --match-engine $E > gpurun_out/c1/bench_e$E.json 2> gpurun_out/c1/bench_e$E.err
  timeout 300 python bench.py --workload pairs10k --cpu-frames 0 --steps 5 --match-engine $E > gpurun_out/c1/pairs_e$E.json 2> gpurun_out/c1/pairs_e$E.err
done
tail -3 gpurun_out/c1/pytest_match.txt; cat gpurun_out/c1/probe_cvt.txt
