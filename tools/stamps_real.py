#!/usr/bin/env python3
"""stamps_real.py N < stderr of a -DSECEDO_STAMPS run: per-workgroup real durations (10 ns ticks), per tile."""
import sys
n = int(sys.argv[1])
rows = [l.split() for l in sys.stdin if 'stamps-wg' in l][-n:]
d = []
for r in rows:
    d.append((int(r[1]), int(r[3]), int(r[4]), int(r[r.index('realbegin') + 1]), int(r[r.index('realend') + 1])))
t0 = min(x[3] for x in d)
print('start spread %d ticks, last end %d ticks' % (max(x[3] for x in d) - t0, max(x[4] for x in d) - t0))
durs = sorted(x[4] - x[3] for x in d)
print('duration min %d median %d mean %d max %d' % (durs[0], durs[len(durs) // 2], sum(durs) // len(durs), durs[-1]))
line = []
for k, rb, re, b, e in d:
    if rb == 0 and line:
        print(' '.join(line)); line = []
    line.append('%5d' % ((e - b) // 10))
print(' '.join(line))
