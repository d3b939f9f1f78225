cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${AB_LIST:-base dh4 dh5 base dh4 dh5}; do
  if [ $v = base ]; then L=$R/anyfeature-vslam_amd/libafv_hip.so; else L=$R/anyfeature-vslam_amd/build_exp/libafv_$v.so; fi
  rm -rf /tmp/ab_$v; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ab_$v -o st --output-format csv -- python $R/bench.py --workload akaze61 --batch 64 --steps 4 --warmup 2 --cpu-frames 0 --lib $L > /tmp/ab_$v.json 2>/dev/null
  python - $v /tmp/ab_$v <<'PY'
import csv,glob,sys,json
v,d=sys.argv[1],sys.argv[2]
f=glob.glob(d+"/**/*kernel_stats.csv",recursive=True)[0]
o={}
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "k_akz_dhess" in n: o[n.split("(")[0].replace("void ","")]=round(float(r["AverageNs"])/1e3,1)
print(v,o)
PY
done
