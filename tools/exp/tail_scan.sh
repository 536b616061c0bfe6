# how the headline kernel's time depends on the number of waves around the multiples of the resident set (1024 SIMDs x 6 waves x 4 bursts)
cd $GRAFT_REPO_ROOT
for n in 49152 73728 96000 98304 100000 102400 122880 147456 200000; do
python3 bench.py --no-cpu --no-extras --steps 100 --bursts $n $@ 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); k=d['roofline']['kernel_ms']; print($n, '%.4f ms  %.3f ns/burst' % (k, k*1e6/$n))"
done
