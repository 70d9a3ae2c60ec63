cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MR_PNP_SO=$R/monorun_amd/variants/libmr_k2x.so
for g in 0 2 4; do
  rm -rf /tmp/k2pmc_$g
  MR_K2_G=$g rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/k2pmc_$g -o p -- env WHICH=k2 REPS=12 python $R/tools/gpu_noc_path.py > /dev/null 2>&1
  python - $g <<'P'
import csv, glob, sys, collections
g = sys.argv[1]
f = glob.glob(f'/tmp/k2pmc_{g}/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'noc_decode' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value'])); meta = (r['Kernel_Name'][:50], r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], r['LDS_Block_Size'])
print('G', g, meta, {k: round(sum(v) / len(v)) for k, v in acc.items()})
P
done
