cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ep_trace -o t -- env REPS=10 python $R/tools/gpu_epnp_path.py > /dev/null 2>&1
python - <<'P'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/ep_trace/**/t_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:90]:<90} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
P
