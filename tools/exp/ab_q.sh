# headline kernel: the round-5 build (libgmr1_hip_base.so) against experimental builds and the current one (product; profiling
# build with the cut-offs: 100 = no re-read of mis-speculated bursts, 101 = every burst takes the re-read route), alternately
cd $GRAFT_REPO_ROOT
run() { GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/$1 GMR1_HIP_DBG_STOP=$2 python3 bench.py --no-cpu --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); p=d.get('roofline_planar',{}); print('$1 stop $2: il %.4f ms %.1f %%  pl %.4f same=%s  crc_pass %.4f  missed %s' % (d['roofline']['kernel_ms'], 100*d['roofline']['frac'], p.get('kernel_ms',0), p.get('outputs_bit_identical_to_interleaved'), d['checks']['crc_pass_frac'], d['checks'].get('mis_speculated_picks_per_launch')))"; }
for i in 1 2; do
for l in ${LIBS:-libgmr1_hip_base.so libgmr1_hip_qu.so libgmr1_hip.so}; do run $l 0; done
for st in ${PROF_STOPS:-0 100}; do run libgmr1_hip_prof.so $st; done
done
for i in 1 2; do for l in ${PROF_LIBS:-}; do for st in ${PROF_STOPS:-0 100}; do run $l $st; done; done; done
