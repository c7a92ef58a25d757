R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
line() { BSKGPU_STRIDE_PAD=$1 python3 bench.py --no-cpu-baseline --no-extra --full-line --steps 20000 --warmup 2000 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pad %-5s wall_us %6.3f stamped %6.3f' % ('$1', r['wall_us_per_launch'], r['kernel_us_stamped']))"; }
for rep in 1 2; do for pad in 0 16 32 48 96 160 224 288 1056 2080 8224 32800; do line $pad; done; done
