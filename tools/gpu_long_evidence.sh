#!/bin/bash
# The long evidence runs of a round (through gpurun, from the repo root): million-object parity sweeps of both flows, long fuzz
# runs, and the spread of the bench line over repeated runs.  Outputs under gpurun_out/ (copy what is to be kept into profiles/).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${TAG:-r03}
PARTS=${PARTS:-k0 epnp k0_56 fuzz fuzz_epnp repeats}      # which runs; WPO=n (waves per object) is passed through to the K0 sweeps, SUFFIX names their files
cd $R
has() { [[ " $PARTS " == *" $1 "* ]]; }
has k0 && { NSEEDS=${NSEEDS:-1000} python tests/sweeps/gpu_parity_sweep.py > $O/${T}_gpu_parity_sweep_1M${SUFFIX:-}.txt 2>&1; tail -1 $O/${T}_gpu_parity_sweep_1M${SUFFIX:-}.txt; }
has epnp && { NSEEDS=${NSEEDS:-1000} python tests/sweeps/gpu_epnp_parity_sweep.py > $O/${T}_gpu_epnp_parity_sweep_1M.txt 2>&1; tail -2 $O/${T}_gpu_epnp_parity_sweep_1M.txt; }
has k0_56 && { HW=56 B=512 NSEEDS=64 python tests/sweeps/gpu_parity_sweep.py > $O/${T}_gpu_parity_sweep_56x56_long${SUFFIX:-}.txt 2>&1; tail -1 $O/${T}_gpu_parity_sweep_56x56_long${SUFFIX:-}.txt; }
has fuzz && { TRIALS=${TRIALS:-2000} python tests/sweeps/gpu_fuzz.py > $O/${T}_fuzz_long.txt 2>&1; tail -1 $O/${T}_fuzz_long.txt; }
has fuzz_epnp && { TRIALS=${TRIALS:-2000} python tests/sweeps/gpu_epnp_fuzz.py > $O/${T}_epnp_fuzz_long.txt 2>&1; tail -2 $O/${T}_epnp_fuzz_long.txt; }
has repeats || exit 0
: > $O/${T}_bench_repeats.txt
for i in $(seq 1 ${REPEATS:-12}); do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json, sys
b = json.loads(sys.stdin.read())
print('run %2d value %.3f M solves/s  ms_per_step %.4f  single_stream %.3f M  rotation_normalised %.3f M  kernel avg %.1f us' % ($i, b['value'] / 1e6, b['ms_per_step'], b['single_stream']['value'] / 1e6, b['value_rotation_normalised'] / 1e6, b['roofline']['isolated_launch']['kernel_ms_avg'] * 1e3))" >> $O/${T}_bench_repeats.txt
done
python - <<P >> $O/${T}_bench_repeats.txt
import re
v = [float(re.search(r'value ([0-9.]+)', l).group(1)) for l in open('$O/${T}_bench_repeats.txt') if l.startswith('run')]
s = [float(re.search(r'single_stream ([0-9.]+)', l).group(1)) for l in open('$O/${T}_bench_repeats.txt') if l.startswith('run')]
import statistics as st
print('# %d fresh processes of the driver command (python bench.py --gpus 1 --steps 20 --warmup 5): value min %.2f median %.2f max %.2f M solves/s; single_stream min %.2f median %.2f max %.2f' % (len(v), min(v), st.median(v), max(v), min(s), st.median(s), max(s)))
P
tail -1 $O/${T}_bench_repeats.txt
