"""Compares the TRACE lines (libipdm_hip_trace.so: a checksum per convolution) of the first forward of a log with the second's."""
import sys
lines = [ln.split(None, 2) for ln in open(sys.argv[1]) if ln.startswith("TRACE")]
nets = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # forwards of ONE network: 4 per process (dbg_bf16x3_fwd.py)
body = [ln[2].rstrip() for ln in lines]
n = len(body) // (4 * nets)
first, second = body[:n], body[n:2 * n]
bad = [i for i in range(n) if first[i] != second[i]]
print("%d convolutions per forward; %d differ between the first and the second forward" % (n, len(bad)))
for i in bad[:6]:
    print("  #%d first : %s\n  #%d second: %s" % (i, first[i], i, second[i]))
