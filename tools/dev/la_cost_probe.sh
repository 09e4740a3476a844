cd $GRAFT_REPO_ROOT
export VFGS_ALLOW_DEV_BUILD=1 VFGS_LIB=$PWD/tools/bin/laprobe.so
for p in 0 1 2 4 3 7; do echo "== probe mask $p (1 = no snapshot copy, 2 = no compare, 4 = no hand-back copy)"; VFGS_LA_PROBE=$p python3 tools/line_api_bench.py --sizes 7680x4320 --frames 8 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   steady %.2f ms per frame' % d['hip_steady_ms_per_frame'])"; done
