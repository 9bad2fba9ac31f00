# one variant's figures for A/B runs of the trace fill's chain kernels (scripts/ab_prebuilt.sh): trace parity, then the serial event times of
# the doubling + scan chain, the row kernel and the Fq12 chain at n = 128
python -m pytest tests/test_gpu_trace.py -q -x 2>&1 | tail -1
python3 scripts/perf_trace.py 128 2>/dev/null | python3 -c "
import sys
for l in sys.stdin:
    f=l.split(); d=dict(x.split('=') for x in f[1:])
    print(f[0], ' '.join('%s=%s' % (k, d[k]) for k in ('trace_curve_chain','trace_curve_rows','trace_fq12_chain','trace_gadgets') if k in d))
"
