# configs[4] from samples: the round-5 build (libgmr1_hip_base.so) against the current one, alternately
cd $GRAFT_REPO_ROOT
run() { GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/$1 python3 bench.py --workload nt3 --no-cpu --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$1: %.4f ms  %.1f %%' % (d['ms_per_step'], 100*d['roofline']['frac']), d.get('checks',{}).get('speech_class1_recovered_frac'))"; }
for i in 1 2; do run libgmr1_hip_base.so; run libgmr1_hip.so; done
