# FCCH workload: the folded sweep (product / profiling build) against the two-kernel form (GMR1_HIP_FCCH_UNFOLDED) and the fallback
# in which every tile gives up at once (GMR1_HIP_FCCH_FOLD_POLLS=0), same box, alternately; kernel durations of one run each
cd $GRAFT_REPO_ROOT
run() { env $2 GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so python3 bench.py --workload fcch --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$1: %.4f ms  %.1f %%  toa identical %s' % (d['ms_per_step'], 100*d['roofline']['frac'], d['checks'].get('toa_identical_to_oracle')))"; }
for i in 1 2; do
run folded X=1
run unfolded GMR1_HIP_FCCH_UNFOLDED=1
run gave_up GMR1_HIP_FCCH_FOLD_POLLS=0
done
