cd $GRAFT_REPO_ROOT
GMR1_HIP_LIBRARY=$GRAFT_REPO_ROOT/osmo-gmr_amd/libgmr1_hip_prof.so GMR1_HIP_RX_TIMING=1 python3 bench.py --workload rx --no-cpu --steps 20 --warmup 3 --preroll-s 0.2 2>&1 | grep "rx_run:" | tail -8
