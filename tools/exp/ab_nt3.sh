# configs[4] from samples, baseline library (osmo-gmr_amd/libgmr1_hip_exp.so, see README) against the product library, alternately
cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_exp.so
run() { python3 bench.py --workload nt3 --steps 50 --no-cpu $2 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$1', d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2; do
GMR1_HIP_LIBRARY=$E run base_acc ""
run new_acc ""
GMR1_HIP_LIBRARY=$E run base_generic "--conv-decoder generic"
run new_generic "--conv-decoder generic"
done
