# headline kernel, baseline library (osmo-gmr_amd/libgmr1_hip_base.so, see README) against an experimental build
# (libgmr1_hip_expA.so) or, without one, the product library -- alternately, three rounds
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); o=d['other_decoder']; print('$1 acc il %.4f pl %.4f | generic il %.4f pl %.4f  same=%s' % (d['roofline']['kernel_ms'], d['roofline_planar']['kernel_ms'], o['kernel_ms'], o['planar_kernel_ms'], d['roofline_planar']['outputs_bit_identical_to_interleaved']))"; }
for i in 1 2 3; do
GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_base.so run base
if [ -f osmo-gmr_amd/libgmr1_hip_expA.so ]; then GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_expA.so run expA; else run new; fi
done
