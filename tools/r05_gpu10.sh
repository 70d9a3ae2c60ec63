for v in cons1 cons2 cons3 cons4; do
echo "== $v"
MR_PNP_SO=$GRAFT_REPO_ROOT/monorun_amd/variants/libmr_$v.so bash tools/epnp_valu_per_launch.sh 2>&1 | grep "consensus" | head -1 | cut -c1-200
done
