timeout 800 python -m pytest tests/test_gpu_akaze.py -x -q 2>&1 | tail -3
python bench.py --workload akaze61 --batch 64 --steps 10 --cpu-frames 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']), round(d['ms_per_step'],3), round(d['scale_space_ms_per_step'],3), d.get('single_frame'))"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/akz_prof4 -o akz -- python /root/repo/bench.py --workload akaze61 --batch 64 --cpu-frames 0 > /dev/null 2>&1
