#!/bin/bash
# GPU box: stripe size x depth of the line call's look-ahead (developer build tools/bin/la.so)
cd $GRAFT_REPO_ROOT
export VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$GRAFT_REPO_ROOT/tools/bin/la.so
for depth in 1 2; do for kb in 512 1024 2048 4096 8192; do
  echo "== depth $depth stripe ${kb} KB"
  VFGS_LA_DEPTH=$depth VFGS_LA_STRIPE_KB=$kb python3 tools/line_api_bench.py --sizes 1920x1080,3840x2160,7680x4320 --frames 8 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-10s steady %7.3f ms/frame  second frame %7.2f  exact %s' % (d['size'], d['hip_steady_ms_per_frame'], d['hip_second_frame_ms'], d.get('bit_exact_vs_reference')))"
done; done
