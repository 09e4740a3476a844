#!/bin/bash
# GPU box: would a batch call gain from running its two halves on the library's two internal streams (one fork + one join per call)?
# Emulated with the overlap region API: a region around every PAIR of half-batch calls, against one plain call of the whole batch.
cd $GRAFT_REPO_ROOT
for c in 0 1 2 3; do for b in 8 32; do
  h=$((b / 2))
  python3 tools/bench_config.py --config $c --batch $b --steps 200 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg %d  one call of %2d frames            %8.2f us per %2d frames  %.4f' % (d['config'], $b, d['launch_us'], $b, d['frac_of_8TBps']))"
  python3 tools/bench_config.py --config $c --batch $h --steps 400 --overlap --region-every 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg %d  two calls of %2d in a region each %8.2f us per %2d frames  %.4f' % (d['config'], $h, 2 * d['launch_us'], $b, d['frac_of_8TBps']))"
done; done
