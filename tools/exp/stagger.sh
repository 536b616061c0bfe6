# headline kernel with every other wave of the first resident generation delayed (GMR1_HIP_STAGGER, profiling build): kernel ms
cd $GRAFT_REPO_ROOT
run() { GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so GMR1_HIP_STAGGER=$1 python3 bench.py --no-cpu --no-extras --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('stagger $1: %.4f ms  %.1f %%  crc_pass %.4f' % (d['roofline']['kernel_ms'], 100*d['roofline']['frac'], d['checks']['crc_pass_frac']))"; }
for s in 0 6 1025 1026 1027 775 10 0 6; do run $s; done
